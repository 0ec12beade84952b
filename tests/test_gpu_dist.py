"""N > 1 with the REAL pipeline: two ranks run generate_sharded() on the tiny pipeline and rank 0's gathered result
must equal the single-process Pipeline.generate() of the whole prompt list bit for bit (Philox keyed by the global
image index, synthetic text features keyed by the global prompt index).  Ranks are separate child processes.

On a 1-GPU box both ranks share cuda:0 and exchange through a gloo group; with >= 2 devices the same test also
runs one rank per GPU over RCCL."""
import os
import socket
import subprocess
import sys

import pytest
import torch

import paintmind_amd as pm
from gpu_common import dev
from paintmind_amd.generate import Pipeline
from util import load_golden, to_torch_sd

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(backend, world, n_prompts, out, extra_env=None, extra_args=()):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), str(r), str(world), port, backend,
                               str(n_prompts), out, *extra_args], env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def _single(n_prompts):
    p, _ = load_golden("tiny_pipeline.npz")
    pipe = Pipeline(pm.Config(pm.ver2cfg["tiny-pipeline"]), stage1_pretrained=False)
    pipe.load_state_dict(to_torch_sd(p), strict=False)
    pipe = pipe.to(dev()).eval()
    return pipe.generate([f"p{i}" for i in range(n_prompts)], seed=7, timesteps=6, save_interval=2, topk=4)


@pytest.mark.parametrize("n_prompts", [1, 5])
def test_two_ranks_sharing_one_gpu_match_single_process(tmp_path, n_prompts):
    out = str(tmp_path / "res.pt")
    _run_ranks("gloo", 2, n_prompts, out)
    res = torch.load(out)
    single = _single(n_prompts)
    assert len(res) == len(single) == 3
    for a, b in zip(res, single):
        assert torch.equal(a, b)


def test_rank_mode_children_run_the_eager_loop_and_match_the_graph_replayed_parent(tmp_path):
    """What a rank of a bench.py N > 1 job runs (bench.py top: AMD_DIRECT_DISPATCH=0, set before the HIP runtime loads): fresh
    children in that mode call generate_sharded() over gloo and the drop-in generate() with graph replay REQUESTED; the library
    must downgrade to the eager loop (graph replay is broken in that runtime mode on ROCm 7.2), say so in engine.switches, and
    both results must equal this process's graph-replayed generate() bit for bit."""
    out = str(tmp_path / "res.pt")
    _run_ranks("gloo", 2, 5, out, extra_env={"AMD_DIRECT_DISPATCH": "0"}, extra_args=("plain",))
    res = torch.load(out)
    assert res["dispatch"] == "0" and res["switches"]["graph_replay_off"] is True
    assert os.environ.get("AMD_DIRECT_DISPATCH", "1") != "0"          # the parent is in the default mode: graphs replay here
    p, _ = load_golden("tiny_pipeline.npz")
    pipe = Pipeline(pm.Config(pm.ver2cfg["tiny-pipeline"]), stage1_pretrained=False)
    pipe.load_state_dict(to_torch_sd(p), strict=False)
    pipe = pipe.to(dev()).eval()
    assert pipe.engine().switches["graph_replay_off"] is False
    prompts = [f"p{i}" for i in range(5)]
    single = pipe.generate(prompts, seed=7, timesteps=6, save_interval=2, topk=4, use_graph=True)
    single2 = pipe.generate(prompts, seed=7, timesteps=6, save_interval=2, topk=4, use_graph=True)      # a pure replay
    assert len(res["sharded"]) == len(res["plain"]) == len(single) == 3
    for a, b, c, d in zip(res["sharded"], res["plain"], single, single2):
        assert torch.equal(a, c) and torch.equal(b, c) and torch.equal(c, d)


def test_generate_with_an_odd_token_count_on_the_graph_path():
    """ids of B * tokens int64 with B * tokens odd are 8- but not 16-byte sized: the graph path's stream-ordered ids copies must
    take them (copy16_async refused them).  tiny-pipeline has 16 tokens; a 3 x 3 grid has 9."""
    import copy
    vq = copy.deepcopy(pm.ver2cfg["tiny-vqgan"])
    vq["enc"]["image_size"] = vq["dec"]["image_size"] = 24                    # 3 x 3 tokens of 8 x 8 pixels
    pm.ver2cfg["tiny-vqgan-9tok"] = vq
    cfg = dict(pm.ver2cfg["tiny-pipeline"], stage1="tiny-vqgan-9tok")
    try:
        torch.manual_seed(5)
        pipe = Pipeline(pm.Config(cfg), stage1_pretrained=False).to(dev()).eval()
    finally:
        del pm.ver2cfg["tiny-vqgan-9tok"]
    assert pipe.num_tokens == 9
    a = pipe.generate(["a"], seed=3, timesteps=4, save_interval=1, topk=2, use_graph=True)
    b = pipe.generate(["a"], seed=3, timesteps=4, save_interval=1, topk=2, use_graph=True)
    e = pipe.generate(["a"], seed=3, timesteps=4, save_interval=1, topk=2, use_graph=False)
    for x, y, z in zip(a, b, e):
        assert torch.equal(x, y) and torch.equal(x, z)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("n_prompts", [1, 5])
def test_two_ranks_over_rccl_match_single_process(tmp_path, n_prompts):
    out = str(tmp_path / "res.pt")
    _run_ranks("nccl", 2, n_prompts, out)
    res = torch.load(out)
    single = _single(n_prompts)
    for a, b in zip(res, single):
        assert torch.equal(a, b)
