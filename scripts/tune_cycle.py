"""Climb budget sweep: time-to-optimal with the cycle found by climbing (cycle_scan = -1), by the
position-space scan alone (1) and by hybrids (k = climb k - 1 round trips, then scan)."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators

OUT = ROOT / "gpurun_out" / "tune_cycle.log"
OUT.parent.mkdir(exist_ok=True)
names = sys.argv[1:] or ["netgen_8_08a", "netgen_8_10a", "netgen_8_12a", "gridgen_8_14a", "netgen_8_14a", "goto_8_12a", "goto_8_14a", "goto_8_16a"]
rows = []
for name in names:
    inst = generators.named_instance(name)
    for rule in (0, 2):
        base = None
        for cs in (1, 3, -1):
            if name == "goto_8_16a" and cs == -1:
                continue
            with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, cycle_scan=cs) as eng:
                t0 = time.perf_counter()
                eng.solve(max_pivots=50_000_000)
                dt = time.perf_counter() - t0
                r = eng.result()
            assert r.status == "optimal"
            if base is None:
                base = r.objective
            assert r.objective == base
            st = r.stats
            row = {"instance": name, "n": inst.n, "m": inst.m, "rule": rule, "cycle_scan": cs, "solve_s": round(dt, 4),
                   "pivots": st["pivots"], "pivots_per_s": round(st["pivots"] / dt), "avg_cycle": round(st["cycle_arcs"] / max(st["pivots"], 1), 1),
                   "scans": st["cycle_scans"], "rounds_per_scan": round(st["scan_rounds"] / max(st["cycle_scans"], 1), 2), "mode": st["pricing_mode"]}
            rows.append(row)
            with OUT.open("a") as fh:
                fh.write(json.dumps(row) + "\n")
            print(json.dumps(row), flush=True)
(ROOT / "gpurun_out" / "tune_cycle.json").write_text(json.dumps(rows, indent=1))
