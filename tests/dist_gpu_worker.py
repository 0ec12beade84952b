"""One rank of the real-pipeline sharding test (started as a child process by test_gpu_dist.py).

usage: dist_gpu_worker.py RANK WORLD PORT BACKEND N_PROMPTS OUT.pt [plain]
Builds the tiny pipeline from the committed golden weights on its GPU, runs generate_sharded() and, on rank 0,
saves the gathered images.  With BACKEND=gloo every rank may share cuda:0 (the 1-GPU box); with nccl rank r uses
cuda:r.

With the trailing `plain` rank 0 ALSO runs the drop-in Pipeline.generate() with its default arguments (graph replay requested)
on the whole prompt list and saves {"sharded", "plain", "switches", "dispatch"}: the parent starts these children under
AMD_DIRECT_DISPATCH=0 -- the runtime mode every rank of a bench.py N > 1 job is in -- and compares with its own graph-replayed
result."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, backend, n_prompts, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), sys.argv[6]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import paintmind_amd as pm
    from paintmind_amd.dist import generate_sharded
    from paintmind_amd.generate import Pipeline
    from util import load_golden, to_torch_sd

    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    kw = {"device_id": dev} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        p, _ = load_golden("tiny_pipeline.npz")
        pipe = Pipeline(pm.Config(pm.ver2cfg["tiny-pipeline"]), stage1_pretrained=False)
        pipe.load_state_dict(to_torch_sd(p), strict=False)
        pipe = pipe.to(dev).eval()
        prompts = [f"p{i}" for i in range(n_prompts)]
        res = generate_sharded(pipe, prompts, seed=7, timesteps=6, save_interval=2, topk=4)
        if rank == 0 and len(sys.argv) > 7 and sys.argv[7] == "plain":
            plain = pipe.generate(prompts, seed=7, timesteps=6, save_interval=2, topk=4)        # use_graph default: requested
            torch.save({"sharded": res, "plain": plain, "switches": pipe.engine().switches,
                        "dispatch": os.environ.get("AMD_DIRECT_DISPATCH", "unset")}, out)
        elif rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
