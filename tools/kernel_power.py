"""Which kernels of the decode loop run AT the socket power cap (their speed is then set by energy per flop, and overlap tricks
buy nothing) and which below it (classic latency / bandwidth work can still pay)?  Each bench-shaped kernel is looped for ~3 s
while `rocm-smi` is polled for the package power and the shader clock.   python tools/kernel_power.py"""
import ctypes as C
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paintmind_amd import _lib, ops, packing

dev = torch.device("cuda:0")
lib = _lib.load()
bf = torch.bfloat16


def hwmon():
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if os.path.exists(d + "/power1_input") or os.path.exists(d + "/power1_average"):
            return d
    return None


HW = hwmon()


def read(name):
    try:
        return float(open(f"{HW}/{name}").read())
    except Exception:
        return float("nan")


def measure(label, fn, work, unit, seconds=4.0):
    samples, stop = [], threading.Event()

    def sampler():
        import re
        import subprocess
        while not stop.is_set():
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
            pw = re.search(r"Power \(W\): ([\d.]+)", out)
            ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz", out)
            if pw and ck:
                samples.append((time.time(), float(pw.group(1)), float(ck.group(1))))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / n
    tail = [s for s in samples if s[0] - t0 > seconds * 0.4]
    pw = sum(s[1] for s in tail) / max(len(tail), 1)
    clk = sum(s[2] for s in tail) / max(len(tail), 1)
    rate = work / ms / (1e9 if unit == "TF/s" else 1e6)
    print(f"{label:58s} {ms*1e3:8.1f} us  {rate:8.0f} {unit}  power {pw:6.0f} W  sclk {clk:5.0f} MHz", flush=True)


M, D = 65536, 512
a = (torch.randn(M, D, device=dev) * 0.7).to(bf)
hi, lo = ops.split_hilo(torch.randn(M, D, device=dev))
coef = ops.ln_coef(hi)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)


def fold(wt):
    return packing.ln_fold(wt, gamma, beta)


wqkv = torch.randn(1536, D, device=dev) * D ** -0.5
wg, c, d = fold(wqkv)
measure("QKV head-split GEMM, LN folded (65536x1536x512)", lambda: ops.gemm_heads_ln(hi, wg, 8, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.18, coef, c, d),
        2 * M * 1536 * D, "TF/s")
lin = torch.nn.Linear(D, 2 * 1368).to(dev)
w12p32, b12p, hp = packing.pack_w12(lin, torch.float32)
wg2, c2, d2 = fold(w12p32)
measure("SwiGLU w12 GEMM, LN folded (65536x2816x512)", lambda: ops.gemm_swiglu_ln(hi, wg2, b12p, coef, c2, d2), 2 * M * 2816 * D, "TF/s")
q = (torch.randn(64, 8, 1024, 64, device=dev) * 0.5).to(bf); k = torch.randn(64, 8, 1024, 64, device=dev).to(bf); vt = torch.randn(64, 8, 64, 1024, device=dev).to(bf)
measure("attention B64 H8 N1024", lambda: ops.attention(q, k, vt, 1024, use_exp2=True), 4 * 64 * 8 * 1024 * 1024 * 64, "TF/s")
wo = (torch.randn(D, D, device=dev) * D ** -0.5).to(bf); bo = torch.randn(D, device=dev)
parts = torch.empty(M, D // 64, 2, device=dev)
s = ops.stream_ptr(dev)
measure("out-projection producer, in place + stats (65536x512x512)",
        lambda: lib.pmhip_gemm_hilo_stats(a.data_ptr(), D, wo.data_ptr(), D, bo.data_ptr(), hi.data_ptr(), lo.data_ptr(), D, 0, hi.data_ptr(), lo.data_ptr(), D, M, D, D,
                                          parts.data_ptr(), s), M * (D * 2 + D * 8), "GB/s")
hid = (torch.randn(M, 1408, device=dev) * 0.5).to(bf); w3 = (torch.randn(D, 1408, device=dev) * 1408 ** -0.5).to(bf)
hi, lo = ops.split_hilo(torch.randn(M, D, device=dev))
measure("FFN w3 producer, in place + stats (65536x512x1408)",
        lambda: lib.pmhip_gemm_hilo_stats(hid.data_ptr(), 1408, w3.data_ptr(), 1408, bo.data_ptr(), hi.data_ptr(), lo.data_ptr(), D, 0, hi.data_ptr(), lo.data_ptr(), D, M, D,
                                          1408, parts.data_ptr(), s), 2 * M * D * 1408, "TF/s")
wl = torch.randn(8192, D, device=dev) * D ** -0.5
wg3, c3, d3 = fold(wl)
bl = torch.randn(8192, device=dev)
measure("logits GEMM, LN folded, f32 out (65536x8192x512)", lambda: ops.gemm_ln(hi, wg3, coef, c3, d3, bias=bl, out_dtype=torch.float32), 2 * M * 8192 * D, "TF/s")
logits = torch.randn(16384, 8192, device=dev)
ids = torch.full((16384,), 8192, dtype=torch.long, device=dev)
measure("sample_rows (16384 rows x 8192 f32)", lambda: ops.sample_rows(logits, ids, 8192, 5, 0.7, seed=3, step=1), 16384 * 8192 * 4, "GB/s")
x = torch.randn(M, D, device=dev)
measure("layernorm f32 -> bf16 (HBM stream)", lambda: ops.layernorm(x, gamma, beta, out_dtype=bf), M * D * 6, "GB/s")
A = (torch.rand(8192, 8192, device=dev) * 2 - 1).to(bf); W = (torch.rand(8192, 8192, device=dev) * 2 - 1).to(bf)
measure("GEMM 8192^3 (long K)", lambda: ops.gemm(A, W, out_dtype=bf), 2 * 8192 ** 3, "TF/s")
