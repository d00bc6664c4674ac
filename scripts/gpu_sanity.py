"""First-contact GPU check: every synthetic golden through the HIP engine (both rules),
compared with the goldens and with the CPU emulation's pivot count."""
import json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import oracle
from network_flow_solver_amd import engine, generators

syn = json.load(open(ROOT / "tests/golden/synthetic.json"))
bad = 0
for s in syn:
    z = np.load(ROOT / "tests/golden" / s["file"])
    inst = generators.ArcSoA(int(z["n"]), z["tail"], z["head"], z["cost"], z["cap"], z["supply"], s["name"])
    e = list(s["expected"].values())[0]
    for rule in (0, 1):
        for graph in (False, True):
            t0 = time.time()
            with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, use_graph=graph) as eng:
                eng.solve()
                r = eng.result()
            dt = time.time() - t0
            em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule)
            ok = r.status == "optimal" and r.objective == round(e["objective"]) and r.stats["pivots"] == em["pivots"] and np.array_equal(r.flow, em["flow"])
            bad += not ok
            print(s["name"], "rule", rule, "graph", graph, r.status, r.objective, round(e["objective"]), "pivots", r.stats["pivots"], em["pivots"], "OK" if ok else "MISMATCH", f"{dt*1e3:.1f} ms", f"{r.stats['pivots']/max(r.stats['solve_seconds'],1e-9):.0f} piv/s", flush=True)
print("bad =", bad)
sys.exit(1 if bad else 0)
