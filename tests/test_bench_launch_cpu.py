"""bench.py --gpus N must start its own N ranks when no launcher wrapped it (decided before any GPU call), relay rank
0's JSON line and carry the per-rank figures.  Exercised on CPU with the compute-free `launch-selftest` workload
(gloo group, the same shard/gather plumbing as the real run)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "launch-selftest", "--steps", "3", *extra],
                       capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    out = _run("--gpus", "2")
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and len(out["per_rank_images_per_s"]) == 2 and out["value"] > 0


def test_single_rank_runs_in_process_and_forced_dist_spawns_one_child():
    assert _run("--gpus", "1")["n_gpus"] == 1
    assert _run("--gpus", "1", env={"PM_BENCH_FORCE_DIST": "1"})["n_gpus"] == 1


def test_mismatched_world_size_is_an_error():
    e = dict(os.environ, WORLD_SIZE="3", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "launch-selftest", "--gpus", "2"],
                       capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_work_accounting_is_consistent():
    """bench.py's per-kernel-kind flops (roofline_by_kernel) add up to the GEMM total that roofline_gemm_family uses, for every
    workload; the residual producers' algorithmic bytes are the A operand + the pair in + out."""
    import bench
    for wl in ("maskgit-uncond-12L-d512-T8", "maskgit-text-24L-d768-T8", "maskgit-text-24L-d1024-512px-T18", "vit-s-recon"):
        kinds = {}
        gf, af, _ = bench.work_per_step(wl, kinds=kinds)
        by_kind = sum(v for k, v in kinds.items() if k.startswith("gemm_"))
        cfg_name, B, T, L = bench.WORKLOADS[wl]
        if cfg_name is None:
            assert abs(gf - by_kind - B * 2 * 1024 * 32 * 8192) < 1e-6 * gf          # the VQ distance product is not a GEMM launch
        elif L is None:
            assert abs(gf - by_kind) < 1e-9 * gf
        else:
            assert 0 < gf - by_kind < 0.02 * gf      # the context K / V + context_proj: per step in the totals, once per loop in the kinds
        assert af > 0 and kinds["bytes_resid2b"] > 0
    kinds = {}
    bench.layer_flops(512, 8, 2048, 1024, False, None, kinds)
    assert kinds["bytes_resid2b"] == 1024 * (512 * 2 + 512 * 8) and kinds["bytes_resid"] == 1024 * (1368 * 2 + 512 * 8)


def test_rank_processes_run_eagerly_without_the_polling_helper_thread():
    """bench.py sets AMD_DIRECT_DISPATCH=0 and drops hipGraph replay for every rank of an N > 1 job before anything loads the HIP
    runtime (profiles/r05_c_host_polling.txt: 0.12 instead of 1.02 host cores per rank at the same throughput; graph replay is not
    usable in that runtime mode on ROCm 7.2); a single-GPU run keeps graph replay; PM_BENCH_RANK_GRAPH=1 opts out."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = "import os, bench; print(os.environ.get('AMD_DIRECT_DISPATCH', 'unset'), bench.USE_GRAPH)"

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("AMD_DIRECT_DISPATCH", "WORLD_SIZE", "PM_BENCH_RANK_GRAPH", "PM_BENCH_NO_GRAPH",
                                                               "PM_BENCH_FORCE_DIST")}
        e.update(env)
        return subprocess.run([sys.executable, "-c", probe], cwd=root, env=e, capture_output=True, text=True, timeout=120).stdout.split()

    assert run() == ["unset", "True"]
    assert run(WORLD_SIZE="8") == ["0", "False"]
    assert run(WORLD_SIZE="8", PM_BENCH_RANK_GRAPH="1") == ["unset", "True"]
    assert run(AMD_DIRECT_DISPATCH="0") == ["0", "False"]           # whoever sets the mode gets the eager loop
    assert run(PM_BENCH_FORCE_DIST="1") == ["0", "False"]           # one GPU, one RCCL rank: exactly the program a rank runs
