"""Late phase of a long solve without the long solve: warm start from a saved basis (scripts/save_basis.py), then PIVOTS more
pivots -- under rocprofv3, with the stamps build, or plain for a rate.  usage: late_phase.py instance rule basis.npz pivots [warm]
env: MCF_TREE_BLOCKS, RC (0 = start without resident reduced costs, as after the drop)"""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
from network_flow_solver_amd import engine, generators
name, rule, basis, pivots = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
warm = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
inst = generators.named_instance(name)
z = np.load(basis)
m = int(z["m"])
in_tree = np.unpackbits(z["in_tree"])[:m].astype(np.int8)
at_upper = np.unpackbits(z["at_upper"])[:m].astype(np.int8)
kw = {}
if os.environ.get("RC", "1") == "0":
    kw["resident_rc"] = False
with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, **kw) as eng:
    assert eng.set_basis(in_tree, at_upper), eng.last_error()
    eng.solve(max_pivots=warm)
    s0 = eng.stats()
    t0 = time.perf_counter()
    eng.solve(max_pivots=pivots)
    dt = time.perf_counter() - t0
    s1 = eng.stats()
    if hasattr(eng._lib, "mcf_debug_pivot_stamps"):
        import ctypes
        out = (ctypes.c_ulonglong * 24)()
        eng._lib.mcf_debug_pivot_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        eng._lib.mcf_debug_pivot_stamps(eng._h, out, 1)
        v = np.array(list(out), dtype=np.float64)
        names = ["stage ctx+cand", "minor key + argmax", "accounting+begin", "cycle_init + barrier", "scan: setup", "scan: rounds", "scan: hit pass",
                 "scan: reduce+merge", "decide", "barrier+finish", "publish ctx"]
        for n, x in zip(names, v[:11]):
            print(f"   {n:22s} per launch {x / max(v[23], 1):9.1f}  share {100 * x / v[:11].sum():5.1f}%")
    p = s1["pivots"] - s0["pivots"]
    print(json.dumps({"instance": name, "rule": rule, "pivots": p, "us_per_pivot": round(1e6 * dt / max(p, 1), 2), "kpivots_s": round(p / dt / 1e3, 2),
                      "cycle_arcs": round((s1["cycle_arcs"] - s0["cycle_arcs"]) / max(p, 1), 1), "subtree": round((s1["subtree_nodes"] - s0["subtree_nodes"]) / max(p, 1), 1),
                      "moved": round((s1["nodes_moved"] - s0["nodes_moved"]) / max(p, 1), 1), "tree_blocks": s1["tree_blocks"], "mode": s1["pricing_mode"],
                      "dropped_at": s1["rc_dropped_at"], "status": s1["status"]}), flush=True)
