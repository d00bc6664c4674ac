"""Candidate cache (McfView::candx) and the mid-solve drop of the resident reduced costs: candidate-list solves on the grid path
under every combination that changes who writes / reads the records, against the CPU emulation (same pivots, flows,
potentials, logical tree).  usage: check_candx.py [quick]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle
from network_flow_solver_amd import engine, generators

quick = len(sys.argv) > 1
insts = [generators.netgen_style(3000, 24000, seed=11), generators.goto_style(40, 40, seed=12), generators.gridgen_style(50, 50, seed=13)]
if not quick:
    insts += [generators.netgen_style(20000, 160000, seed=14)]
bad = runs = 0
for inst in insts:
    em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2)
    emf = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2 | (1 << 8))
    for layout in (-1, 6, 4):
        for opts in (dict(), dict(resident_rc=False), dict(full_sweeps=-1), dict(rc_drop=1), dict(rc_drop=1, full_sweeps=-1), dict(use_graph=False, rc_drop=2),
                     dict(forward_first=True), dict(compressed_keys=1, full_sweeps=-1), dict(batch_pivots=7, rc_drop=1), dict(cycle_scan=-1)):
            ref = emf if opts.get("forward_first") else em
            runs += 1
            try:
                with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2, tree_blocks=layout, fused=False, mid_loop=-1, **opts) as eng:
                    for budget in (1, 50, 4999):     # budgeted prefixes, then to the end
                        eng.solve(max_pivots=budget)
                    eng.solve()
                    r, t = eng.result(), eng.tree()
                ok = (r.status == ref["status"] and r.objective == ref["objective"] and r.stats["pivots"] == ref["pivots"]
                      and np.array_equal(r.flow, ref["flow"]) and np.array_equal(r.potential, ref["potential"])
                      and np.array_equal(t["order"], ref["order"]) and np.array_equal(t["parent"], ref["parent"]))
                if "rc_drop" in opts and r.stats["pivots"] > 9000 and not r.stats["rc_dropped_at"]:
                    ok = False
                info = {"pivots": [int(r.stats["pivots"]), ref["pivots"]], "dropped_at": int(r.stats["rc_dropped_at"]), "mode": int(r.stats["pricing_mode"])}
            except Exception as exc:  # noqa: BLE001
                ok, info = False, {"exc": str(exc)}
            if not ok:
                bad += 1
                print("MISMATCH", json.dumps({"inst": inst.name, "layout": layout, "opts": opts, **info}), flush=True)
    print(inst.name, "runs", runs, "bad", bad, flush=True)
print(json.dumps({"runs": runs, "bad": bad}))
sys.exit(1 if bad else 0)
