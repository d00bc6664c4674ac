"""Solve INSTANCE with RULE for PIVOTS pivots and save the basis (basic arcs, non-basic arcs at their capacity) bit-packed:
a warm start from it puts later profiling runs straight into the late phase of a long solve.
usage: save_basis.py instance rule pivots out.npz"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from network_flow_solver_amd import engine, generators
name, rule, pivots, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
inst = generators.named_instance(name)
with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule) as eng:
    t0 = time.time()
    last = [t0]
    def progress(p, cap, el):
        if time.time() - last[0] > 20:
            last[0] = time.time(); print(f"  {p} pivots, {time.time() - t0:.0f}s", flush=True)
        return False
    eng.solve(max_pivots=pivots, progress=progress, progress_interval=250_000)
    res = eng.result()
at_upper = (~res.in_tree) & (inst.cap > 0) & (res.flow == inst.cap)
Path(out).parent.mkdir(parents=True, exist_ok=True)
np.savez_compressed(out, in_tree=np.packbits(res.in_tree), at_upper=np.packbits(at_upper), m=inst.m, pivots=res.stats["pivots"])
print(f"saved {out}: {res.stats['pivots']} pivots, {int(res.in_tree.sum())} basic arcs, {int(at_upper.sum())} at capacity, status {res.status}", flush=True)
