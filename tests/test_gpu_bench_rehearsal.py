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


def _bench_line(ranks, port, mode, extra=()):
    env = dict(os.environ, SSRLCV_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", SSRLCV_EXCHANGE=mode)
    env.pop("SSRLCV_HIP_LIB", None)
    env.pop("SSRLCV_DEV_BUILD", None)
    common = ["--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--size", "1024", "--nview-size", "1024", "--nview-steps", "1",
              "--no-matcher", "--no-cpu-baseline", "--pushbroom-size", "1024"] + list(extra)
    if ranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + common
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_sharded_legs_report_equal_checksums():
    """Round 6: every rank's cloud / MatchSet checksum is compared across the ranks and with a world-1 run of the same
    deterministic scene (bench.py `nview.result_check`), for both exchange modes (per-rank exact-size broadcasts and one
    padded all-gather).  Here the world-1 run is made in place; on the driver's 8-GPU run the reference value is the one
    committed under tests/golden/nview_checksums.json."""
    one = _bench_line(1, 0, "bcast", ["--no-class-api", "--no-pushbroom"])["nview"]["result_check"]
    assert one["equal_across_ranks"] and one["bundles"] > 1000
    for ranks, port, mode in ((2, 29541, "bcast"), (2, 29542, "allgather"), (4, 29543, "allgather")):
        d = _bench_line(ranks, port, mode, ["--no-class-api", "--no-pushbroom"])
        rc = d["nview"]["result_check"]
        assert d["nview"]["comm"]["rccl_world"] == ranks and d["nview"]["wire"]["mode"] == mode
        assert rc["equal_across_ranks"] and "FAILED" not in rc, rc
        for k in ("cloud", "multi_matches", "keypoints"):
            assert rc["checksums"][k] == one["checksums"][k], (ranks, mode, k)


@pytest.mark.gpu
def test_bench_runs_with_two_ranks_under_torchrun():
    d = _bench_line(2, 29533, "bcast")
    pb = d["pushbroom8"]   # config[4]'s leg (here at 1024^2: eight strips over two ranks)
    assert pb["n_gpus"] == 2 and pb["points"] > 1000 and set(pb["stage_ms"]) >= {"sift", "match", "merge", "triangulate"}
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["config"]["images_per_gpu"] == 2 and d["config"]["library"].startswith("release")
    nv = d["nview"]
    assert nv["n_gpus"] == 2 and nv["views"] == 4 and nv["pairs"] == 6 and nv["scaling"] == "strong"
    assert len(nv["pair_cost_share_per_rank"]) == 2 and abs(sum(nv["pair_cost_share_per_rank"]) - 1.0) < 1e-6
    assert set(nv["stage_ms_per_step_over_ranks"]) >= {"sift", "match", "merge", "triangulate"}
    assert "cpu_baseline" not in d and "flow" not in d.get("class_api", {})  # single-GPU-run legs
