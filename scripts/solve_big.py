#!/usr/bin/env python3
"""Time-to-optimal of a BASELINE-size instance with progress lines (a silent GPU command is killed after 7 minutes).
    python scripts/solve_big.py netgen_1m_16m 2 [max_seconds]      (MCF_SOLVE_OPTS='{"resident_rc": false}' passes engine options)"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from network_flow_solver_amd import engine, generators  # noqa: E402

name, rule = sys.argv[1], int(sys.argv[2])
max_seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 900.0
t0 = time.time()
inst = generators.named_instance(name)
print(f"{name}: generated in {time.time() - t0:.1f}s", flush=True)
t0 = time.time()
opts = json.loads(os.environ.get("MCF_SOLVE_OPTS", "{}"))
eng = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, **opts)
print(f"options {opts} -> pricing mode {eng.stats()['pricing_mode']}", flush=True)
print(f"create {time.time() - t0:.1f}s", flush=True)
t0 = time.time()
last = [t0]


def progress(pivots, cap, elapsed):
    now = time.time()
    if now - last[0] > 20:
        last[0] = now
        print(f"  {pivots} pivots, {now - t0:.0f}s, {pivots / (now - t0):.0f} pivots/s", flush=True)
    return now - t0 > max_seconds


eng.solve(max_pivots=200_000_000, progress=progress, progress_interval=250_000)
res = eng.result()
dt = time.time() - t0
print(f"status={res.status} pivots={res.stats['pivots']} seconds={dt:.1f} pivots/s={res.stats['pivots'] / dt:.0f} objective={res.objective} "
      f"artificial_flow={res.stats['artificial_flow']} degenerate={res.stats['degenerate']} tree_blocks={res.stats['tree_blocks']} "
      f"rebuilds={res.stats['tree_rebuilds']} rc_dropped_at={res.stats['rc_dropped_at']} moved/subtree={res.stats['nodes_moved'] / max(res.stats['subtree_nodes'], 1):.2f}", flush=True)
if res.status == "optimal":
    from conftest import check_optimality
    check_optimality(inst, res.flow, res.potential)
    print("certified optimal", flush=True)
