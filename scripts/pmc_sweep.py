"""Workload for the rocprofv3 --pmc passes: a handful of pivots on the 1M/16M instance, then
`reps` back-to-back pricing sweeps (read-only), so the per-dispatch counters of k_price can be
averaged.  Usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 scripts/pmc_sweep.py"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators
inst = generators.named_instance(os.environ.get("WL", "netgen_1m_16m"))
# full_sweeps = 1 like bench.py's roofline points: every sweep reads its whole shard (from 4 M arcs on: the key-code kernel)
with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=int(os.environ.get("RULE", "0")),
                      full_sweeps=int(os.environ.get("FULL_SWEEPS", "1"))) as eng:
    eng.solve(max_pivots=int(os.environ.get("PIVOTS", "100")))
    ms = eng.time_pricing(reps=int(os.environ.get("REPS", "10")))
    print("ms_per_sweep", ms, "algorithmic_bytes", eng.stats()["price_bytes"])
