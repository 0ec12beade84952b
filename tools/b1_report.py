"""per-kernel summary of the LAST generate call in a rocprofv3 kernel trace of tools/b1_trace.py:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b1 -o t -- python3 tools/b1_trace.py ; python3 tools/b1_report.py gpurun_out/b1"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows]
ends = [int(r["End_Timestamp"]) for r in rows]
cut = 0
for i in range(len(rows) - 1, 0, -1):
    if starts[i] - ends[i - 1] > 2_000_000:
        cut = i
        break
rows, starts, ends = rows[cut:], starts[cut:], ends[cut:]
agg = collections.OrderedDict()
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:90]
    a = agg.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(a[1] for a in agg.values())
print(f"{len(rows)} kernels, span {(ends[-1] - starts[0]) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms")
for n, a in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{a[1] / 1e6:7.2f} ms {a[0]:5d} x {a[1] / a[0] / 1e3:7.1f} us  {n}")
