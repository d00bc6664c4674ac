"""One (partial) solve of a named instance, for rocprofv3 --kernel-trace --stats.
usage: prof_solve.py INSTANCE [RULE] [MAX_PIVOTS] [CYCLE_SCAN]"""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators

name = sys.argv[1]
rule = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000_000
cs = int(sys.argv[4]) if len(sys.argv) > 4 else 0
inst = generators.named_instance(name)
with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, cycle_scan=cs,
                      use_graph=os.environ.get("MCF_USE_GRAPH", "1") != "0") as eng:  # (profiles are taken with MCF_USE_GRAPH=0: eager launches, per-kernel durations unchanged)
    t0 = time.perf_counter()
    eng.solve(max_pivots=cap)
    dt = time.perf_counter() - t0
    st = eng.stats()
print(json.dumps({"instance": name, "rule": rule, "cycle_scan": cs, "seconds": round(dt, 4), "pivots": st["pivots"],
                  "us_per_pivot": round(1e6 * dt / max(st["pivots"], 1), 2), "status": st["status"],
                  "avg_cycle": round(st["cycle_arcs"] / max(st["pivots"], 1), 1),
                  "avg_moved": round(st["nodes_moved"] / max(st["pivots"], 1), 1),
                  "avg_subtree": round(st["subtree_nodes"] / max(st["pivots"], 1), 1),
                  "rounds_per_scan": round(st["scan_rounds"] / max(st["cycle_scans"], 1), 2)}))
