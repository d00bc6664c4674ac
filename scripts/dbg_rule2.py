import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle
from conftest import load_synthetic
from network_flow_solver_amd import engine
_, inst = load_synthetic()[7]
for fused_inst in (False,):
    eng = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2, use_graph=False, batch_pivots=1)
    for k in range(1, 40):
        eng.solve(max_pivots=1)
        r = eng.result()
        em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=2, max_pivots=k)
        same = np.array_equal(r.flow, em["flow"]) and np.array_equal(r.potential, em["potential"])
        print(k, "gpu pivots", r.stats["pivots"], "arcs", r.stats["arcs_priced"], "| emul", em["pivots"], em["arcs_priced"], em["minor_pivots"], em["major_sweeps"], "same" if same else "DIFF")
        if not same:
            break
