"""Multi-rank compute on the GPU: CoverAlgorithm.all_pairwise sharded over two ranks (CoverAlgorithm.py:166-182 is the
reference's own parallel driver: joblib over chunks of the pair list) and bench.py starting its own ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_all_pairwise_equals_reference_scores(golden, tmp_path, world):
    """Fresh child ranks (all on GPU 0, gloo for the gather): every one of the 12 720 config-1 scores of the gathered
    matrices equals the reference's, on every rank.  (Four ranks is what a one-GPU box allows: at most six processes may
    hold the card at once, this one included; world size 8 is rehearsed on the CPU in tests/test_sharding.py.)"""
    import warnings
    warnings.simplefilter("ignore")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_rank_all_pairwise.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    from acoss_amd import synth
    g = golden("config1_scores")
    pairs = synth.all_pairs(synth.config1().n_songs)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "Ds_rank%d.npz" % r))
        assert int(z["world"][0]) == world
        for key in ("chroma_qmax", "chroma_dmax"):
            D = z[key]
            assert np.array_equal(D[pairs[:, 0], pairs[:, 1]], g[key].astype(np.float32)), (r, key)
            assert np.array_equal(D, D.T)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 4` with no launcher around it: the parent starts the ranks, rank 0's line says n_gpus 4,
    counts every rank's pairs and carries its own cpu_baseline + parity check; a --gpus that disagrees with the launcher's
    world size is an error."""
    env = _env()
    env["ACOSS_BENCH_DIST_BACKEND"] = "gloo"            # the ranks share the box's one GPU
    n = 4
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--songs", "60",
           "--frames", "400", "--pairs-per-step", "256", "--cpu-pairs", "64", "--no-extras"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["scaling"] == "weak" and out["config"]["parallelism"].startswith("pair-shard x%d" % n)
    assert abs(out["value"] - n * 2 * 256 / (out["ms_per_step"] * 2 * 1e-3)) / out["value"] < 1e-3
    assert out["cpu_baseline"]["kind"] == "port" and out["parity"]["identical"] is True
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)
