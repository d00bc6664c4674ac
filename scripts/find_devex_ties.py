#!/usr/bin/env python3
"""Which seeds of tests/test_gpu_parity.py:_tie_rich_instance meet same-direction and forward/backward merit ties at
the maximum during a Devex solve (run on the GPU box; the parity test pins seeds found here)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from network_flow_solver_amd import engine  # noqa: E402
from test_gpu_parity import _tie_rich_instance  # noqa: E402

for seed in range(1, 16):
    inst = _tie_rich_instance(seed)
    same = cross = 0
    with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=1) as eng:
        for budget in (0, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 10 ** 9):
            if budget:
                eng.solve(max_pivots=budget)
            res, tree, w = eng.result(), eng.tree(), eng.weights()
            rc = inst.cost + tree["pi"][inst.tail] - tree["pi"][inst.head]
            viol = (-(tree["state"].astype(np.int64)) * rc).astype(np.float64)
            merit = np.where(viol > 0, viol * viol / w.astype(np.float64), 0.0)
            if merit.max() > 0:
                top = np.nonzero(merit == merit.max())[0]
                if len(top) > 1:
                    if len(set(int(tree["state"][i]) for i in top)) == 2:
                        cross += 1
                    else:
                        same += 1
            if res.status == "optimal":
                break
    print(f"seed {seed}: same-direction ties at the max {same}, forward/backward ties {cross}, pivots {res.stats['pivots']}", flush=True)
