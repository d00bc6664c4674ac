"""Whole solves (time to optimal) of named instances x rules, engine defaults; prints pivots/s, objective and what the handle did.
usage: whole_solves.py name:rule ...   (run from the tree whose library is to be measured)"""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path.cwd())); sys.path.insert(0, str(Path.cwd() / "tests"))
from network_flow_solver_amd import engine, generators
from conftest import check_optimality
for a in sys.argv[1:]:
    name, rule = a.split(":")[0], int(a.split(":")[1])
    inst = generators.named_instance(name)
    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule) as eng:
        t0 = time.perf_counter()
        eng.solve(max_pivots=200_000_000)
        dt = time.perf_counter() - t0
        res = eng.result()
    if res.status == "optimal":
        check_optimality(inst, res.flow, res.potential)
    st = res.stats
    print(json.dumps({"instance": name, "rule": rule, "status": res.status, "certified": res.status == "optimal", "objective": res.objective, "pivots": st["pivots"],
                      "seconds": round(dt, 3), "kpivots_s": round(st["pivots"] / dt / 1e3, 2), "subtree_per_pivot": round(st["subtree_nodes"] / max(st["pivots"], 1), 1),
                      "moved_per_pivot": round(st["nodes_moved"] / max(st["pivots"], 1), 1), "cycle_per_pivot": round(st["cycle_arcs"] / max(st["pivots"], 1), 1),
                      "tree_blocks": st.get("tree_blocks", 0), "rc_dropped_at": st.get("rc_dropped_at", 0), "mode": st["pricing_mode"]}), flush=True)
