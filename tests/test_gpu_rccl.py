"""RCCL on the one-GPU box: a process group of world size 1 under backend "nccl" loads librccl, forms a communicator and runs
the path's collectives on device tensors -- everything short of a second GPU.  (CoverAlgorithm.py:166-182 is the reference's
process-level sharding these collectives close.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env(**kw):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def test_one_rank_rccl_group_runs_the_paths_collectives(tmp_path):
    out = os.path.join(str(tmp_path), "rccl.json")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rank_rccl.py"), out, str(_free_port())],
                         env=_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    with open(out) as fh:
        got = json.load(fh)
    assert got["backend"] == "nccl" and got["world"] == 1
    assert got["gather_scores_index_of_rank_equal"] and got["gather_scores_with_positions_equal"] and got["gather_scores_permuted_equal"]
    assert got["all_gather_into_tensor_equal"] and got["all_reduce_max"] == 3.25
    assert got["all_pairwise_through_rccl_equal"]
    assert any("rccl" in name for name in got["rccl_libraries_mapped"]), got["rccl_libraries_mapped"]


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_collectives_under_nccl_at_world_one(mode):
    """bench.py --gpus 1 with ACOSS_BENCH_FORCE_DIST=1: the barrier, the all_gather_into_tensor of the timed scores and the
    all_reduce(MAX) of the elapsed time run through RCCL; the line says so and its parity block still holds."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--songs", "40", "--frames", "300",
           "--pairs-per-step", "256", "--cpu-pairs", "64", "--no-extras"] + (["--strong"] if mode == "strong" else [])
    res = subprocess.run(cmd, env=_env(ACOSS_BENCH_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    coll = out["config"]["collective"]
    assert coll["backend"] == "nccl" and coll["world"] == 1 and "all_gather_into_tensor" in coll["ran"]
    assert out["n_gpus"] == 1 and out["scaling"] == mode
    if mode == "weak":
        assert out["parity"]["identical"] is True
    else:
        assert out["Ds_symmetric"] is True and out["scores_sum"] > 0.0
        # the same job without any process group: identical scores
        res2 = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
        assert res2.returncode == 0, res2.stdout[-2000:] + res2.stderr[-4000:]
        out2 = json.loads([ln for ln in res2.stdout.splitlines() if ln.startswith("{")][0])
        assert out2["config"]["collective"] is None and out2["scores_crc32"] == out["scores_crc32"]


def test_strong_scaling_four_ranks_equal_one_rank():
    """bench.py --strong: the fixed job sharded over 4 ranks (sharing the box's GPU, gloo for the gather) returns the same
    score vector, bit for bit, as the 1-rank run."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--songs", "60", "--frames", "400", "--strong"]
    outs = []
    for n in (1, 4):
        res = subprocess.run(base + ["--gpus", str(n)], env=_env(ACOSS_BENCH_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        outs.append(json.loads(lines[0]))
    assert outs[0]["n_gpus"] == 1 and outs[1]["n_gpus"] == 4 and outs[1]["scaling"] == "strong"
    assert outs[1]["config"]["collective"]["backend"] == "gloo"
    assert outs[0]["scores_crc32"] == outs[1]["scores_crc32"] and outs[0]["scores_sum"] == outs[1]["scores_sum"]
    assert outs[1]["Ds_symmetric"] is True
