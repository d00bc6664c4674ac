"""A/B of the tree layouts (dense preorder array vs blocked preorder list) on whole solves / budgets.
usage: ab_tree_layout.py instance rule max_pivots layout[,layout...]      layout = -1 (dense) or log2 block size (0 = auto)"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from network_flow_solver_amd import engine, generators
name, rule, cap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
layouts = [int(x) for x in sys.argv[4].split(",")]
inst = generators.named_instance(name)
for lay in layouts:
    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, tree_blocks=lay) as eng:
        t0 = time.perf_counter()
        eng.solve(max_pivots=cap)
        dt = time.perf_counter() - t0
        st = eng.stats()
    print(json.dumps({"instance": name, "rule": rule, "layout": lay, "tree_blocks": st["tree_blocks"], "status": st["status"], "pivots": st["pivots"], "seconds": round(dt, 3),
                      "kpivots_s": round(st["pivots"] / dt / 1e3, 2), "nodes_moved_per_pivot": round(st["nodes_moved"] / max(st["pivots"], 1), 1),
                      "subtree_per_pivot": round(st["subtree_nodes"] / max(st["pivots"], 1), 1), "rebuilds": st["tree_rebuilds"]}), flush=True)
