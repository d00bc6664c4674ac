"""Blocked preorder list on the GPU against the CPU emulation of the DENSE preorder array: pivots, flows, potentials and the
logical tree (order, positions, sizes, depths) must be identical for every block size and pool size.
usage: check_tree_layout.py [quick]"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle
from conftest import check_tree_invariants
from network_flow_solver_amd import engine, generators

quick = len(sys.argv) > 1
insts = [generators.netgen_style(64, 512, seed=1), generators.netgen_style(256, 2048, seed=2), generators.gridgen_style(16, 16, seed=3),
         generators.goto_style(12, 12, seed=4), generators.netgen_style(1024, 8192, seed=5)]
if not quick:
    insts += [generators.netgen_style(6000, 48000, seed=6), generators.goto_style(70, 70, seed=7)]
bad = runs = 0
for inst in insts:
    for rule in (0, 1, 2):
        em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule)
        for shift, pool in ((2, 0), (2, -1), (3, 5), (4, 0), (6, 0), (6, -1), (7, 0), (8, 3)):
            for opts in (dict(), dict(cycle_scan=-1), dict(full_sweeps=-1, compressed_keys=1), dict(resident_rc=False), dict(use_graph=False, climb_depth=-1)):
                if opts.get("compressed_keys") and rule == 1:
                    continue
                t0 = time.time()
                runs += 1
                try:
                    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, tree_blocks=shift, tree_pool=pool, **opts) as eng:
                        eng.solve()
                        r, t = eng.result(), eng.tree()
                except Exception as exc:  # noqa: BLE001
                    bad += 1
                    print("EXC", json.dumps({"inst": inst.name, "rule": rule, "shift": shift, "pool": pool, "opts": opts}), exc, flush=True)
                    continue
                ok = (r.status == em["status"] and r.objective == em["objective"] and r.stats["pivots"] == em["pivots"] and r.stats["tree_blocks"] == shift
                      and np.array_equal(r.flow, em["flow"]) and np.array_equal(r.potential, em["potential"])
                      and all(np.array_equal(t[k], em[k]) for k in ("order", "pos", "psize", "parent", "depth", "size")))
                if ok:
                    check_tree_invariants(inst.n, t["parent"], t["size"], t["pos"], t["order"], t["depth"], t["psize"])
                else:
                    bad += 1
                    print("MISMATCH", json.dumps({"inst": inst.name, "rule": rule, "shift": shift, "pool": pool, "opts": opts, "status": [r.status, em["status"]],
                                                  "pivots": [int(r.stats["pivots"]), em["pivots"]], "objective": [r.objective, em["objective"]],
                                                  "order_diffs": int((t["order"] != em["order"]).sum())}), flush=True)
        print(inst.name, "rule", rule, "runs", runs, "bad", bad, flush=True)
print(json.dumps({"runs": runs, "bad": bad}))
sys.exit(1 if bad else 0)
