"""Differential fuzzing on the GPU box: random seeded instances x random engine options against the CPU emulation
(pivot count, flows, potentials, tree must be identical).  usage: fuzz_gpu.py [seconds] [first_seed]"""
import json, random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import oracle
from network_flow_solver_amd import engine, generators

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t_end = time.time() + budget
max_runs = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
runs = fails = 0
prev_cfg = None
log = ROOT / "gpurun_out" / "fuzz.log"
log.parent.mkdir(exist_ok=True)
while time.time() < t_end and runs < max_runs:
    rng = random.Random(seed)
    fam = rng.choice(["netgen", "gridgen", "goto"])
    if fam == "netgen":
        n = rng.choice([40, 130, 300, 700, 1500, 3000, 6000, 12000, 20000, 45000, 90000])   # (> 32 768 positions: coarse index)
        inst = generators.netgen_style(n, n * rng.choice([4, 8, 12]), seed=seed)
    elif fam == "gridgen":
        w = rng.choice([6, 12, 20, 33, 50, 70, 110]); inst = generators.gridgen_style(w, rng.choice([w, w + 3]), seed=seed)
    else:
        w = rng.choice([6, 12, 20, 33, 50, 70, 110, 200]); inst = generators.goto_style(w, w, seed=seed)
    rule = rng.choice([0, 1, 2])
    opts = dict(fused=rng.random() < 0.5, mid_loop=rng.choice([-1, 0, 1]), cycle_scan=rng.choice([-1, 0, 0, 1, 3]),
                full_sweeps=rng.choice([-1, 0, 1]), use_graph=rng.random() < 0.7, batch_pivots=rng.choice([7, 32, 64]),
                climb_depth=rng.choice([-1, 0, 0, 2, 9]),                                   # depth gate of the cycle search
                compressed_keys=rng.choice([-1, 1, 1]), vkey_half_log2=rng.choice([0, 0, 9, 14]),   # key codes, narrow levels
                key_mode=rng.choice([0, 0, 0, 1, 2, 3]) if rule != 1 else 0,                 # the specialised entering rules' keys
                overlap_update=rng.choice([0, 0, 1]),                                        # two-stream graphs
                tree_blocks=rng.choice([0, 0, -1, 6, 7]), tree_pool=rng.choice([0, 0, 2, 40]),  # [r3] blocked preorder list, small pools (rebuilds)
                rc_drop=rng.choice([0, 0, -1, 1, 3]), pivot_run=rng.choice([0, 0, 0, 3]))       # [r3] reduced costs given up mid-solve, run shape
    prio = np.random.default_rng(seed).integers(0, 4, size=len(inst.tail)).astype(np.int8) if opts["key_mode"] == 2 else None
    opts["arc_priority"] = prio
    cost = inst.cost * rng.choice([1, 1, 300])                                               # x300: big-M >= 2^29 (level coding)
    if int(np.abs(cost).max()) * (inst.n + 2) >= 2 ** 43:
        cost = inst.cost
    inst.cost = cost
    cap = rng.choice([10 ** 9, 10 ** 9, 137, 2500])
    if inst.n > 30000:
        cap = rng.choice([2500, 12000, 40000])   # (a full emulated solve takes minutes at this size)
    em = oracle.emul_solve(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule | (opts["key_mode"] << 8),
                           max_pivots=cap, arc_priority=prio)
    try:
        with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, **opts) as eng:
            if rng.random() < 0.3 and cap > 200:          # budgeted prefix + resume
                eng.solve(max_pivots=rng.choice([1, 50, 199]))
                done = eng.stats()["pivots"]
                eng.solve(max_pivots=cap - done if cap < 10 ** 9 else cap)
            else:
                eng.solve(max_pivots=cap)
            r, t = eng.result(), eng.tree()
        ok = (r.stats["pivots"] == em["pivots"] and np.array_equal(r.flow, em["flow"]) and np.array_equal(r.potential, em["potential"])
              and np.array_equal(t["order"], em["order"]) and np.array_equal(t["parent"], em["parent"]) and np.array_equal(t["depth"], em["depth"])
              # (at exactly the budget the engine prices once more and may say "optimal" where the emulation, which
              #  does not, says "iteration_limit": simplex.py:1678-1699)
              and (r.status == em["status"] or (r.stats["pivots"] == cap and {r.status, em["status"]} == {"optimal", "iteration_limit"})))
    except Exception as exc:  # noqa: BLE001
        ok = False
        r = None
        print("EXC", exc)
    runs += 1
    if not ok:
        fails += 1
        detail = {}
        if r is not None:
            detail = {"flow_diffs": int((r.flow != em["flow"]).sum()), "pot_diffs": int((r.potential != em["potential"]).sum()),
                      "order_diffs": int((t["order"] != em["order"]).sum()), "deg": [int(r.stats["degenerate"]), em["degenerate"]],
                      "status": [r.status, em["status"]], "minor": [int(r.stats.get("cycle_scans", -1))]}
        msg = json.dumps({"seed": seed, "family": fam, "n": inst.n, "m": inst.m, "rule": rule, "opts": {k: v for k, v in opts.items() if k != "arc_priority"}, "cap": cap,
                          "gpu_pivots": None if r is None else r.stats["pivots"], "emul_pivots": em["pivots"], "detail": detail,
                          "previous_run": prev_cfg})
        print("MISMATCH", msg, flush=True)
        with log.open("a") as fh:
            fh.write(msg + "\n")
    prev_cfg = {"seed": seed, "family": fam, "n": inst.n, "rule": rule, "opts": {k: v for k, v in opts.items() if k != "arc_priority"}, "cap": cap}
    seed += 1
    if runs % 10 == 0:
        print(f"  ... {runs} runs, {fails} mismatches, {t_end - time.time():.0f} s left", flush=True)   # (a silent GPU command is killed)
print(json.dumps({"runs": runs, "fails": fails, "last_seed": seed - 1}))
sys.exit(1 if fails else 0)
