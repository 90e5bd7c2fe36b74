"""`python bench.py --gpus N` launches its own ranks (the driver's N > 1 command has no launcher in front of it).
CPU only: the plumbing flag stops every rank behind the first collective, before anything touches a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_launches_its_own_ranks_and_relays_one_line():
    p = run(["--gpus", "2", "--backend", "gloo", "--plumbing-only"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout               # ONE line on stdout, everything else on stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_in_allreduce"] == 2


def test_bench_exits_non_zero_when_a_rank_fails():
    # no HIP device here: every rank stops with "needs a HIP device"; the parent must not report success
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    p = run(["--gpus", "2", "--backend", "gloo"])
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith('{"metric"')]


def test_world_size_mismatch_is_refused():
    p = run(["--gpus", "2", "--plumbing-only"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "launcher started 1" in p.stderr
