import sys, time
sys.path.insert(0, "/root/repo")
from network_flow_solver_amd import engine, generators
insts = [generators.netgen_style(256, 2048, seed=1 + k) for k in range(512)]
t0 = time.time()
engines = [engine.McfEngine(i.n, i.tail, i.head, i.cost, i.cap, i.supply, rule=0) for i in insts]
t1 = time.time()
print(f"create: {1e3 * (t1 - t0) / len(engines):.3f} ms per handle")
ms = engine.solve_batch(engines)
t2 = time.time()
res = [e.result() for e in engines]
t3 = time.time()
print(f"solve_batch {1e3 * (t2 - t1):.1f} ms (kernel {ms:.1f}); result(): {1e3 * (t3 - t2) / len(engines):.3f} ms per handle")
for e in engines: e.close()
t4 = time.time()
print(f"close: {1e3 * (t4 - t3) / len(engines):.3f} ms per handle")
import cProfile, pstats
