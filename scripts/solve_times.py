"""Time-to-optimal table: every engine rule on the GPU vs the C oracle (reference algorithm, 1 host
thread) on the same seeded instances.  Objectives must agree exactly."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import oracle
from network_flow_solver_amd import engine, generators

rows = []
LOG = ROOT / "gpurun_out" / "solve_times.log"
LOG.parent.mkdir(exist_ok=True)


def log(msg):
    with LOG.open("a") as fh:
        fh.write(msg + "\n")
    print(msg, flush=True)

for name, with_oracle in (("netgen_8_08a", True), ("netgen_8_10a", True), ("netgen_8_12a", True), ("gridgen_8_14a", False),
                          ("netgen_8_14a", False), ("goto_8_16a", False), ("netgen_8_16a", False)):
    inst = generators.named_instance(name)
    row = {"instance": name, "n": inst.n, "m": inst.m}
    objs = set()
    for rule, label in ((0, "dantzig"), (1, "devex_block"), (2, "candidate_list")):
        log(f"  {name} {label} ...")
        t0 = time.perf_counter()
        with engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule) as eng:
            t1 = time.perf_counter()
            eng.solve(max_pivots=50_000_000)
            t2 = time.perf_counter()
            r = eng.result()
        assert r.status == "optimal", (name, label, r.status)
        objs.add(r.objective)
        row[label] = {"solve_s": round(t2 - t1, 4), "setup_s": round(t1 - t0, 4), "pivots": r.stats["pivots"],
                      "pivots_per_s": round(r.stats["pivots"] / (t2 - t1)),
                      "cycle_arcs_per_pivot": round(r.stats["cycle_arcs"] / max(r.stats["pivots"], 1), 1),
                      "nodes_moved_per_pivot": round(r.stats["nodes_moved"] / max(r.stats["pivots"], 1), 1),
                      "scan_rounds_per_scan": round(r.stats["scan_rounds"] / max(r.stats["cycle_scans"], 1), 2)}
    if with_oracle:
        log(f"  {name} oracle ...")
        t0 = time.perf_counter()
        o = oracle.solve_soa(inst, "dantzig", reference_order=False)
        row["oracle_dantzig_cpu"] = {"solve_s": round(o["seconds"], 3), "pivots": o["iterations"]}
        objs.add(int(round(o["objective"])))
    fix = json.loads((ROOT / "tests" / "golden" / "baseline_objectives.json").read_text()).get(name)
    if fix:
        objs.add(fix["objective"])   # the oracle's objective, computed in the build container (make_baseline_objectives.py)
    assert len(objs) == 1, (name, objs)
    row["objective"] = objs.pop()
    rows.append(row)
    log(json.dumps(row))
(ROOT / "gpurun_out" / "solve_times.json").write_text(json.dumps(rows, indent=1))
