"""Debug aid: step the GPU engine pivot by pivot against the CPU emulation and report the first divergence."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import oracle
from network_flow_solver_amd import engine, generators
name = sys.argv[1] if len(sys.argv) > 1 else "netgen_8_10a"
rule = int(sys.argv[2]) if len(sys.argv) > 2 else 2
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
graph = (sys.argv[4] != '0') if len(sys.argv) > 4 else True
cs = int(sys.argv[5]) if len(sys.argv) > 5 else 0
print('---', name, 'rule', rule, 'step', step, 'graph', graph, 'cycle_scan', cs)
inst = generators.named_instance(name)
eng = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, use_graph=graph, cycle_scan=cs)
k = 0
while True:
    k += step
    try:
        eng.solve(max_pivots=step)
    except Exception as exc:
        print("engine error after", k, "pivots:", exc)
        st = eng.stats(); print(st)
        break
    t = eng.tree(); st = eng.stats()
    em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, max_pivots=k, climb_budget=0)
    bad = [key for key in ("order", "parent", "size", "pos", "depth", "psize") if not np.array_equal(t[key], em[key])]
    if bad or st["pivots"] != em["pivots"]:
        print("diverged at", k, "pivots: fields", bad, "gpu pivots", st["pivots"], "emul", em["pivots"])
        for key in bad:
            idx = np.nonzero(t[key] != em[key])[0]
            print("  ", key, "first diffs at", idx[:8], "gpu", t[key][idx[:8]], "emul", em[key][idx[:8]])
        break
    if st["status"] != "iteration_limit":
        print("agree to the end:", st["pivots"], st["status"])
        break
