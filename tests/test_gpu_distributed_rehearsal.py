"""The multi-GPU code path on ONE GPU (``-m gpu``): ``bench.py``'s distributed loop with a single rank that still
issues the RCCL collective (``MCF_BENCH_FORCE_DIST=1``) -- sharded sweep -> all-gather -> replicated pivot(s), captured
in a graph and REPLAYED.  A replayed graph calls no library function between two polls, which is exactly the case a
cached control block gets wrong (round 2: the loop never saw the status change and spun forever).  Run as a child
process: torch and the engine library must not share a process with the ctypes-driven tests."""

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rule", ["dantzig", "candidate_list", "devex"])
def test_one_rank_rccl_rehearsal_of_the_sharded_loop(gpu_engine_module, rule):
    env = dict(os.environ, MCF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--rule", rule, "--workload", "netgen_8_14a",
           "--steps", "600", "--warmup", "100", "--no-hbm-point"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 600 and line["pivots_per_sec"] > 1000
    assert line["config"]["pivot_loop"] in ("captured graph", "eager")
    assert line["roofline"]["frac"] <= 1.0 and line["value"] > 0


def test_one_rank_rehearsal_of_the_default_multi_gpu_line(gpu_engine_module):
    """The line the driver's `bench.py --gpus N` produces by default, with one rank: the arc-sharded headline, the larger
    sharded point, and the batch of independent instances sharded across ranks (no collective in its data path)."""
    env = dict(os.environ, MCF_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "200", "--warmup", "20"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["value"] > 0 and "hbm_point" in line
    bp = line["batched_point"]
    assert bp["all_optimal"] and bp["pivots"] > 400_000 and bp["pivots_per_sec"] > 1e6 and bp["scaling"] == "weak"
