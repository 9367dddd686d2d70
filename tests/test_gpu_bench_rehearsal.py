"""bench.py launched the way the driver launches its scaling runs (torch.distributed.run, one process per rank), on the ONE
GPU of the test box: SSRLCV_BENCH_BACKEND=gloo puts both ranks on cuda:0 and stages the collectives through the host.  Not a
measurement -- it keeps the N > 1 legs of the bench line (pair owners, per-rank stage spread, wire bytes, max over ranks)
running end to end between the rounds in which a multi-GPU node is available."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_runs_with_two_ranks_under_torchrun():
    env = dict(os.environ, SSRLCV_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    env.pop("SSRLCV_HIP_LIB", None)
    env.pop("SSRLCV_DEV_BUILD", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "1024",
           "--nview-size", "1024", "--nview-steps", "1", "--no-matcher", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["images_per_gpu"] == 2 and d["config"]["library"].startswith("release")
    nv = d["nview"]
    assert nv["n_gpus"] == 2 and nv["views"] == 4 and nv["pairs"] == 6 and nv["scaling"] == "strong"
    assert len(nv["pair_cost_share_per_rank"]) == 2 and abs(sum(nv["pair_cost_share_per_rank"]) - 1.0) < 1e-6
    assert set(nv["stage_ms_per_step_over_ranks"]) >= {"sift", "match", "merge", "triangulate"}
    assert "cpu_baseline" not in d and "flow" not in d.get("class_api", {})  # single-GPU-run legs
