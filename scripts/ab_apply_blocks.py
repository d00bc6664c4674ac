#!/usr/bin/env python3
"""Workgroups of the tree permutation (MCF_APPLY_BLOCKS) at 1 M nodes: pivots/s, candidate list and Devex.  python scripts/ab_apply_blocks.py"""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from network_flow_solver_amd import engine, generators
inst = generators.named_instance("netgen_1m_16m")
for ab in ("256", "512", "1024", "2048"):
    os.environ["MCF_APPLY_BLOCKS"] = ab
    for rule in (2, 1):
        with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule) as eng:
            eng.solve(max_pivots=2000)
            t0 = time.time(); eng.solve(max_pivots=40000); dt = time.time() - t0
        print(f"apply_blocks<={ab} rule={rule}: {40000 / dt / 1e3:.1f} K pivots/s", flush=True)
