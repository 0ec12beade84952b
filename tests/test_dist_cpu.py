"""N > 1 path on CPU: world_size-2 gloo process group exercising the sharding + gather logic of
paintmind_amd/dist.py with a stand-in pipeline (no compute: the product has no CPU path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from paintmind_amd.dist import gather_images, generate_sharded, shard_range


def test_shard_range_is_a_partition():
    for n in (0, 1, 5, 64, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class FakePipe:
    """generate() returns images that encode (seed, global image index, step): what the real path guarantees"""

    image_shape = (3, 4, 4)

    class TM:
        base_index = 0

    def __init__(self):
        self.text_model = self.TM()

    def generate(self, text, seed, image_base, keep_on_device, timesteps=4, save_interval=2, **kw):
        assert self.text_model.base_index == image_base
        out = []
        for step in range(0, timesteps, save_interval):
            img = torch.stack([torch.full((3, 4, 4), float(seed * 1000 + (image_base + i) * 10 + step)) for i in range(len(text))])
            out.append(img)
        return out


class FakePipeNoShape(FakePipe):
    image_shape = None          # the fallback: ranks agree on the image shape through one extra all_gather


def _worker(rank, world, port, n_prompts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prompts = [f"p{i}" for i in range(n_prompts)]
        # count the collectives generate_sharded issues: exactly one gather per call
        calls = {"gather": 0, "all_gather": 0}
        real_gather, real_all_gather = dist.gather, dist.all_gather

        def counting_gather(*a, **k):
            calls["gather"] += 1
            return real_gather(*a, **k)

        def counting_all_gather(*a, **k):
            calls["all_gather"] += 1
            return real_all_gather(*a, **k)
        dist.gather, dist.all_gather = counting_gather, counting_all_gather
        try:
            res = generate_sharded(FakePipe(), prompts, seed=7, timesteps=4, save_interval=2)
            assert calls == {"gather": 1, "all_gather": 0}, calls
            res2 = generate_sharded(FakePipeNoShape(), prompts, seed=7, timesteps=4, save_interval=2)
            assert calls["gather"] == 2 and calls["all_gather"] == (1 if n_prompts < world else 0), calls
            if rank == 0:
                assert all(torch.equal(a, b) for a, b in zip(res, res2))
        finally:
            dist.gather, dist.all_gather = real_gather, real_all_gather
        lo, hi = shard_range(n_prompts, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float32).reshape(-1, 1)
        counts = [shard_range(n_prompts, r, world)[1] - shard_range(n_prompts, r, world)[0] for r in range(world)]
        g = gather_images(local, counts)
        if rank == 0:
            q.put((res, g))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("n_prompts", [1, 5, 8])       # 1 prompt on 2 ranks: rank 1 has an empty shard
def test_sharded_generate_matches_single_process(n_prompts):
    world = 2
    ctx = mp.get_context("spawn")
    for attempt in range(2):                    # the port is probed, released and re-bound by the children: retry once on a lost race
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n_prompts, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res, g = q.get(timeout=300)         # a cold `import torch` in the children can take minutes
        except Exception:
            for p in procs:
                if p.is_alive():
                    p.terminate()
            if attempt == 1:
                raise
            continue
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        break
    single = FakePipe().generate([f"p{i}" for i in range(n_prompts)], seed=7, image_base=0, keep_on_device=True, timesteps=4,
                                 save_interval=2)
    assert len(res) == len(single)
    for a, b in zip(res, single):
        assert torch.equal(a, b)
    assert torch.equal(g, torch.arange(n_prompts, dtype=torch.float32).reshape(-1, 1))
