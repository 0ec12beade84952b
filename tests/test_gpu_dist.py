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


def _run_ranks(backend, world, n_prompts, out):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), str(r), str(world), port, backend,
                               str(n_prompts), out], env=env) for r in range(world)]
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


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("n_prompts", [1, 5])
def test_two_ranks_over_rccl_match_single_process(tmp_path, n_prompts):
    out = str(tmp_path / "res.pt")
    _run_ranks("nccl", 2, n_prompts, out)
    res = torch.load(out)
    single = _single(n_prompts)
    for a, b in zip(res, single):
        assert torch.equal(a, b)
