"""EXPERIMENT, measured slower: bench.py with the concurrent lanes on CU-MASKED streams, so that the lanes really
run side by side (an HBM-bound kernel of one lane beside an MFMA-bound kernel of the other) instead of meeting only in kernel
tails.  PMHIP_LANE_CU_MASK=1: equal contiguous shares; "a:b": lane 0 gets a of every a+b CUs (two lanes).
    PMHIP_LANE_CU_MASK=1 PM_BENCH_LANE_SPLIT=32,32 PMHIP_PERSIST256=128 python tools/cu_mask_lanes.py --no-extra --no-cpu-baseline"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from paintmind_amd.generate import Pipeline


def _masked_stream(device, lane, n_lanes, mode):
    """a HIP stream restricted to a share of the CUs, so that concurrent lanes really run side
    by side (an HBM-bound kernel of one lane beside an MFMA-bound kernel of the other) instead of meeting only in kernel tails.
    CU mask bit i addresses XCD i % 8, local CU i / 8 (KFD's symmetric mapping), so a contiguous bit range gives a lane the same
    CUs in every XCD.  mode "1": equal contiguous shares; mode "a:b": lane 0 gets a of every a+b CUs."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    if ":" in mode and n_lanes == 2:
        a, b = (int(x) for x in mode.split(":"))
        cut = ncu * a // (a + b) // 8 * 8
        lo, hi = (0, cut) if lane == 0 else (cut, ncu)
    else:
        per = ncu // n_lanes // 8 * 8
        lo, hi = lane * per, (ncu if lane == n_lanes - 1 else (lane + 1) * per)
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for bit in range(lo, hi):
        mask[bit // 32] |= 1 << (bit % 32)
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(st.value, device=device)



MODE = os.environ.get("PMHIP_LANE_CU_MASK", "0")
if MODE != "0":
    Pipeline._new_lane_stream = staticmethod(lambda device, lane, n_lanes: _masked_stream(device, lane, n_lanes, MODE))

if __name__ == "__main__":
    import bench
    sys.exit(bench.main())
