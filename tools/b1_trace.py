"""B = 1 latency anatomy: one 8-step generate of a single image (graph replay) under rocprofv3 --kernel-trace.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b1 -o t -- python3 tools/b1_trace.py ; python3 tools/b1_trace.py report gpurun_out/b1"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "report":
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last replay: the final ~1800 kernels; find the last long gap (> 2 ms) as the call boundary
    starts = [int(r["Start_Timestamp"]) for r in rows]
    ends = [int(r["End_Timestamp"]) for r in rows]
    cut = 0
    for i in range(len(rows) - 1, 0, -1):
        if starts[i] - ends[i - 1] > 2_000_000:
            cut = i
            break
    rows, starts, ends = rows[cut:], starts[cut:], ends[cut:]
    busy = sum(e - s for s, e in zip(starts, ends))
    span = ends[-1] - starts[0]
    gaps = [starts[i + 1] - ends[i] for i in range(len(rows) - 1)]
    print(f"{len(rows)} kernels, span {span/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, gaps {sum(g for g in gaps if g > 0)/1e6:.2f} ms "
          f"(median gap {sorted(gaps)[len(gaps)//2]/1e3:.1f} us)")
    agg = {}
    for r, s, e in zip(rows, starts, ends):
        k = r["Kernel_Name"].split("(")[0][-60:]
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {t/1e6:7.2f} ms  {n:5d} x {t/n/1e3:7.1f} us  {k}")
    sys.exit(0)

import torch

import bench

dev = torch.device("cuda:0")
model = bench.build(bench.DEFAULT_WORKLOAD, dev, torch.bfloat16)
flags = [True] * 8
for i in range(4):
    model.generate_ids(None, 1, 8, 1.0, 5, flags, seed=1 + i, use_graph=True, streams=1)
    torch.cuda.synchronize()
import time
time.sleep(0.01)
model.generate_ids(None, 1, 8, 1.0, 5, flags, seed=9, use_graph=True, streams=1)
torch.cuda.synchronize()
