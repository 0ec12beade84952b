"""How often does the bf16 attention kernel leave its steady loop on the bench's REAL Q / K?  Needs the
debug build:  PM_EXTRA_FLAGS=-DPM_ATTN_COUNT bash paintmind_amd/csrc/build.sh ; python tools/attn_rescale_count.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from paintmind_amd import _lib

dev = torch.device("cuda:0")
lib = C.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else C.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libpaintmind_hip.so"))
out = (C.c_ulonglong * 4)()
for workload in (bench.DEFAULT_WORKLOAD, "maskgit-text-24L-d768-T8"):
    model = bench.build(workload, dev, torch.bfloat16)
    step = bench.make_step(workload, model, dev, 0)
    step(0, streams=1)
    torch.cuda.synchronize()
    assert lib.pmhip_debug_attention_counters(out, 1) == 0
    for i in range(2):
        step(1 + i, streams=1)
    torch.cuda.synchronize()
    assert lib.pmhip_debug_attention_counters(out, 1) == 0
    resc, steady, slow = out[0], out[1], out[2]
    print(f"{workload}: steady half-tile steps (per wave) {steady}, slow steps {slow}, rescale-branch executions inside the steady loop {resc} "
          f"= {resc / max(steady, 1):.5f} of the steady steps")
    del model, step
    torch.cuda.empty_cache()
