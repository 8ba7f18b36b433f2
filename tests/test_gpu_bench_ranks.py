"""bench.py's N > 1 code path on the one-GPU box (-m gpu): two ranks that share the device (AGX_BENCH_SHARE_GPU=1: the gather then runs over gloo with
CPU tensors — RCCL refuses two ranks on one device), launched exactly as the driver launches it (torch.distributed.run, one rank per "GPU").  What the
job delivers to rank 0 must be, unit by unit, what a one-rank run of the same configuration delivers; the line must carry the same-configuration
single-GPU time measured in the same run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("staged", [False, True], ids=["text", "staged"])
def test_two_ranks_deliver_what_one_rank_delivers(tmp_path, staged, monkeypatch):
    """staged: the read alignments handed over as tmp/_agx_pairs.<u>.bin (AGX_BENCH_STAGED=1) — the way `bench.py --gpus N` runs the whole-human configuration: rank 0
    generates, every rank loads its own units from the staged files and writes their caches."""
    import aligngraph_amd as A
    assert A.device_count() > 0
    if staged:
        monkeypatch.setenv("AGX_BENCH_STAGED", "1")
    common = ["--steps", "2", "--warmup", "1", "--config", "custom", "--chroms", "400000,250000,300000,150000,200000", "--pairs", "180000",
              "--cpu-sample-pairs", "0", "--workdir", str(tmp_path / "work")]
    env = dict(os.environ, AGX_BENCH_SHARE_GPU="1", AGX_BENCH_DIGEST=str(tmp_path / "two.json"))
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-config-steps", "2"] + common, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["units"] == 5
    assert line["single_gpu_ms_same_config"] and line["speedup_vs_1gpu"] > 0
    env1 = dict(os.environ, AGX_BENCH_DIGEST=str(tmp_path / "one.json"))
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env1, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p1.returncode == 0, p1.stderr[-2000:]
    two, one = json.load(open(tmp_path / "two.json")), json.load(open(tmp_path / "one.json"))
    assert sorted(two) == sorted(one) == [str(u) for u in range(5)]
    assert two == one, "the two-rank job delivered other bytes than the one-rank job"
