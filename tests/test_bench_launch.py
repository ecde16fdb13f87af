"""bench.py's own launcher: `--gpus N` must start N ranks itself.  Checked on CPU with
`--check-launch`, which runs everything of a multi-rank bench run except the GPU work (rendezvous over gloo,
the pipelined interval gather with the workload's shapes, the result line)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra=None, workload="tiny"):
    env = dict(os.environ)
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--check-launch", "--workload", workload,
                           "--steps", "3", "--warmup", "1"] + extra, capture_output=True, text=True, env=env, timeout=280)


@pytest.mark.timeout(300)
def test_gpus_2_starts_two_ranks_and_relays_one_line():
    p = run_bench(["--gpus", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_in_group"] == 2
    assert out["steps"] == 3 and out["warmup"] == 1


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_run():
    p = run_bench(["--gpus", "2"], {"FMX_BENCH_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_single_rank_needs_no_launcher():
    p = run_bench(["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip())["n_gpus"] == 1


def test_rank_count_must_match_gpus_flag():
    p = run_bench(["--gpus", "1"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "started 2 ranks" in p.stderr


@pytest.mark.timeout(300)
@pytest.mark.parametrize("workload,ranks,extra", [("c5", 2, []), ("c3", 3, ["--delivery", "all"]),
                                                  ("c3", 2, ["--exchange", "pairs", "--delivery", "all"]),
                                                  ("tiny", 3, ["--exchange", "pairs"]), ("c4", 3, []), ("c4ref", 2, [])])
def test_driver_shapes_dry_run(workload, ranks, extra):
    """What the driver's scaling run launches, without GPUs: the literal workloads' per-rank batch (1M patterns: 8 MB of
    packed intervals per rank -- a wide one through the escape list -- gathered to rank 0 through the pipelined
    exchange; the other form and the other delivery by flag) and the regex workloads' exchange -- result lists of
    unequal lengths with an empty rank, ids made global, sizes then padded payload -- over gloo, three ranks for C4."""
    p = run_bench(["--gpus", str(ranks)] + extra, workload=workload)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == ranks and out["config"]["ranks_in_group"] == ranks and workload in out["config"]["workload"]
