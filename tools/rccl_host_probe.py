"""Which thread burns a core once RCCL is in the process?  One rank (world_size 1 over nccl = RCCL), AMD_DIRECT_DISPATCH as
given in the environment; per-thread CPU seconds over 3 s windows of plain GPU work: (a) before the process group exists, (b) after
init_process_group, (c) after the first collective, (d) with a gather per iteration.
    AMD_DIRECT_DISPATCH=0 python tools/rccl_host_probe.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def threads():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read().rsplit(")", 1)[1].split()
            out[t] = (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")
        except OSError:
            pass
    return out


def window(label, extra=None, seconds=3.0):
    torch.cuda.synchronize()
    t0, c0 = time.time(), threads()
    while time.time() - t0 < seconds:
        for _ in range(20):
            y = x @ x
        if extra:
            extra(y)
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():                            # the main thread sleeps: whatever burns a core is not this loop
            time.sleep(0.002)
    dt = time.time() - t0
    c1 = threads()
    busy = sorted(((c1[t] - c0.get(t, 0.0)) / dt, t) for t in c1)
    print(f"{label:40s} " + ", ".join(f"{b:.2f}" for b, _ in busy[::-1][:5]) + f"   ({len(c1)} threads; cores of the five busiest)", flush=True)


print("AMD_DIRECT_DISPATCH =", os.environ.get("AMD_DIRECT_DISPATCH", "unset"))
window("(a) no process group")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
window("(b) after init_process_group")
t = torch.ones(4, device=dev)
dist.all_reduce(t)
window("(c) after the first collective")
buf = [torch.empty(64, 3, 256, 256, device=dev)]
img = torch.randn(64, 3, 256, 256, device=dev)
window("(d) a synchronous gather per iteration", lambda y: dist.gather(img, buf, dst=0))
hs = []
window("(e) an async gather per iteration", lambda y: hs.append(dist.gather(img, buf, dst=0, async_op=True)))
for h in hs:
    h.wait()
hs.clear()
window("(e1) async gather, handle dropped", lambda y: dist.gather(img, buf, dst=0, async_op=True))
side = torch.cuda.Stream()


def waited_on_side(y):
    h = dist.gather(img, buf, dst=0, async_op=True)
    with torch.cuda.stream(side):
        h.wait()


window("(e2) async gather, waited on a side stream", waited_on_side)


def sync_from_side(y):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.gather(img, buf, dst=0)


window("(e3) sync gather issued from a side stream", sync_from_side)
dist.destroy_process_group()
window("(f) after destroy_process_group")
