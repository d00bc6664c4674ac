"""Diagnostic: phase shares inside k_pivot (build: hipcc -DMCF_STAMPS ... -o scripts/libmcf_stamps.so)."""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
os.environ["MCF_HIP_LIB"] = str(ROOT / "scripts" / "libmcf_stamps.so")
sys.path.insert(0, str(ROOT))
import numpy as np
from network_flow_solver_amd import engine, generators
MID = os.environ.get("MID", "0") == "1"
names = ["stage ctx+cand" if not MID else "loop top", "minor key + argmax", "accounting+begin", "cycle_init + barrier", "scan: setup", "scan: rounds", "scan: hit pass",
         "scan: reduce+merge", "decide", "barrier+finish", "publish ctx" if not MID else "update pass (apply+rcupd)"]
CASES = (("netgen_8_10a", 1, 10**9), ("netgen_8_12a", 1, 10**9), ("netgen_8_14a", 1, 10**9), ("netgen_8_14a", 2, 10**9)) if MID else (("netgen_8_10a", 0, 10**9), ("netgen_8_14a", 0, 10**9), ("netgen_8_14a", 2, 10**9), ("gridgen_8_14a", 1, 10**9), ("goto_8_16a", 0, 60000))
if len(sys.argv) > 1:   # name:rule:cap ...
    CASES = tuple((a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2])) for a in sys.argv[1:])
for case in CASES:
    name, rule, cap = case[:3]
    inst = generators.named_instance(name)
    eng = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, mid_loop=1 if MID else -1)
    out = (ctypes.c_ulonglong * 24)()
    eng._lib.mcf_debug_pivot_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    eng._lib.mcf_debug_pivot_stamps(eng._h, out, 1)
    window = int(os.environ.get("WINDOW", "0")) or cap
    done = 0
    while done < cap:
        eng.solve(max_pivots=min(window, cap - done))
        st = eng.stats()
        eng._lib.mcf_debug_pivot_stamps(eng._h, out, 1)
        v = np.array(list(out), dtype=np.float64)
        launches = max(v[23], 1)
        print(f"{name} rule {rule}: pivots {st['pivots']} launches {int(launches)} ticks/launch {v[:11].sum() / launches:.0f} "
              f"(solve {1e6 * st['solve_seconds'] / max(st['pivots'], 1):.2f} us/pivot; tree_blocks {st['tree_blocks']} rebuilds {st['tree_rebuilds']} rc dropped at {st['rc_dropped_at']})")
        for n, x in zip(names, v[:11]):
            print(f"   {n:22s} per launch {x / launches:9.1f}  share {100 * x / v[:11].sum():5.1f}%", flush=True)
        if v[20] > 0:
            print(f"   scan: coarse pass         per launch {v[21] / launches:9.1f}   (not in the shares above; 'scan: rounds' is the fine pass alone when this is non-zero)")
            print(f"   scans {int(v[20])} ({100 * v[20] / launches:.0f}% of launches): blocks in the coarse pass {v[17] / v[20]:.0f}, flagged {v[18] / v[20]:.1f}, "
                  f"one-sided ancestors found {v[19] / v[20]:.1f} per scan; scans with > 128 flagged blocks {100 * v[22] / v[20]:.1f}%, > 512: {100 * v[11] / v[20]:.1f}%", flush=True)
        done = st["pivots"]
        if st["status"] != "iteration_limit":
            break
    eng.close()
