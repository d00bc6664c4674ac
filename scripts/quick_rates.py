"""GPU-only time-to-optimal for the engine rules on named instances (no oracle): quick A/B between builds.
usage: quick_rates.py [instance ...]   env: MID=-1/0/1, CS=cycle_scan, FUSED=0/1"""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators
names = sys.argv[1:] or ["netgen_8_08a", "netgen_8_10a", "netgen_8_12a", "gridgen_8_14a", "netgen_8_14a", "goto_8_14a", "goto_8_16a", "netgen_8_16a"]
kw = dict(mid_loop=int(os.environ.get("MID", "0")), cycle_scan=int(os.environ.get("CS", "0")), fused=os.environ.get("FUSED", "1") == "1")
for name in names:
    inst = generators.named_instance(name)
    row = {"instance": name, "n": inst.n, "m": inst.m}
    for rule, label in ((0, "dantzig"), (1, "devex"), (2, "cand")):
        if rule == 1 and inst.m > 200_000:
            continue
        best = None
        for rep in range(2 if inst.m <= 200_000 else 1):
            with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, **kw) as eng:
                t0 = time.perf_counter()
                eng.solve(max_pivots=50_000_000)
                dt = time.perf_counter() - t0
                st = eng.stats()
            assert st["status"] == "optimal"
            if best is None or dt < best[0]:
                best = (dt, st)
        dt, st = best
        row[label] = {"kpiv_s": round(st["pivots"] / dt / 1e3, 1), "us_piv": round(1e6 * dt / st["pivots"], 2), "pivots": st["pivots"], "mode": st["pricing_mode"]}
    print(json.dumps(row), flush=True)
