"""Diagnostic: phase shares of the fused small-instance kernel (build with -DMCF_STAMPS)."""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
os.environ["MCF_HIP_LIB"] = str(ROOT / "scripts" / "libmcf_stamps.so")
sys.path.insert(0, str(ROOT))
import numpy as np
from network_flow_solver_amd import engine, generators
inst = generators.named_instance("netgen_8_08a")
for rule in (0, 1):
    eng = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule)
    eng.solve()
    st = eng.stats()
    out = (ctypes.c_ulonglong * 8)()
    eng._lib.mcf_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    eng._lib.mcf_debug_stamps(eng._h, out)
    v = np.array(list(out), dtype=np.float64)
    names = ["copy_in", "price", "argmax", "walk", "finish", "apply", "copy_out", "-"]
    piv = max(st["pivots"], 1)
    print("rule", rule, "pivots", piv, "solve_s", st["solve_seconds"], "ticks/pivot in loop", v[1:6].sum() / piv)
    for n, x in zip(names, v):
        print(f"   {n:9s} {x:14.0f}  per pivot {x / piv:9.1f}  share {100 * x / v.sum():5.1f}%")
    out2 = (ctypes.c_ulonglong * 24)()
    eng._lib.mcf_debug_pivot_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    eng._lib.mcf_debug_pivot_stamps(eng._h, out2, 1)
    w = np.array(list(out2), dtype=np.float64)
    for n, x in zip(["(to walk start)", "begin", "cycle_init", "climb", "decide"], w[12:17]):
        print(f"      walk/{n:16s} per pivot {x / piv:9.1f}")
    eng.close()
