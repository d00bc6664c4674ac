#!/usr/bin/env python3
"""bench.py -- pivots/sec + arcs-priced/sec of the MI355X network-simplex engine.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--rule R] [--scaling weak|strong]

A *step* is one pivot: one pass of the hot path (pricing sweep -> ratio test / flow update
-> tree + potential update) over the device-resident instance.  W untimed pivots, then
EXACTLY K timed pivots between barrier + device synchronisation on both sides; rank 0 prints
ONE JSON line.  Inputs are resident in HBM before the timed region starts.

Workload at N=1 (default): BASELINE.json configs[1] -- "netgen_8_08a, 1xMI355X, full-scan
Dantzig pricing kernel" -- as a seeded synthetic stand-in (the LEMON file is not obtainable
offline).  That instance is 27 KB per sweep: it runs as ONE persistent LDS-resident workgroup
(k_solve_small) and is latency-bound by construction, which is what its `roofline` object says.
The same line therefore also carries
  * ``hbm_point``: configs[4]'s shape (1 M nodes / 16 M arcs), where the pricing sweep (k_price_v) is the dominant,
    bandwidth-bound kernel -- its 67 MB working set fits the 256 MiB Infinity Cache;
  * ``hbm_point_beyond_infinity_cache``: 6 M nodes / 96 M arcs (403 MB of key codes per sweep), where the same
    kernel really streams from HBM;
  * ``config_points``: the other single-GPU-runnable BASELINE.json configs with the engine's defaults.

``roofline`` always describes the kernel the timed region ran.  ``achieved`` = the kernel's own compulsory bytes per
launch (4 B per arc for the key-code sweep, 9 B for the resident-reduced-cost sweep, 13 B with
Devex weights) / its launch
duration measured with HIP events on the engine's stream; ``traffic`` = HBM-side bytes per launch from the committed
rocprofv3 PMC passes (profiles/pmc_traffic.json).  SURVEY.md section 8d's per-arc figure for the gather formulation
(13 B/arc + 8 B/node) is kept as ``survey_8d`` for comparison; it is not a physical rate of this kernel.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); ``python bench.py --gpus N`` without a
launcher spawns the N ranks itself.  Every rank holds the replicated instance, prices its own arc shard, and one small
all-gather per sweep picks the entering arc(s).
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E datasheet peak (/opt/skills/guides/MI355X_MICROARCH.md; ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (family builder args) -- per-GPU shape under weak scaling
    "netgen_8_08a": ("netgen", 256, 2048),
    "gridgen_8_14a": ("gridgen", 128, 128),      # BASELINE.json configs[2]: 16 385 nodes / 131 080 arcs
    "goto_8_16a": ("goto", 256, 256),            # BASELINE.json configs[3]: 65 536 nodes / 524 288 arcs
    "netgen_8_14a": ("netgen", 16384, 131072),
    "netgen_8_16a": ("netgen", 65536, 524288),
    "netgen_8_18a": ("netgen", 262144, 2097152),
    "netgen_8_20a": ("netgen", 1 << 20, 8 << 20),
    "netgen_1m_16m": ("netgen", 1 << 20, 16 << 20),
    "netgen_4m_64m": ("netgen", 4 << 20, 64 << 20),
    "netgen_6m_96m": ("netgen", 6 << 20, 96 << 20),   # 403 MB of key codes per sweep: beyond the 256 MiB Infinity Cache
}
RULE_NAMES = {0: "full-scan Dantzig", 1: "block-search Devex", 2: "candidate list"}
MODE_NAMES = {0: "gather sweep", 1: "resident reduced costs", 2: "fused LDS loop", 3: "persistent single-workgroup loop"}

_instances: dict = {}


def make_instance(workload: str, scale: int = 1):
    from network_flow_solver_amd import generators

    key = (workload, scale)
    if key not in _instances:
        fam, n, m = WORKLOADS[workload]
        name = f"{workload}(synthetic,x{scale})"
        if fam == "gridgen":
            inst = generators.gridgen_style(n * scale, m, seed=1, name=name)
        elif fam == "goto":
            inst = generators.goto_style(n * scale, m, seed=1, name=name)
        else:
            inst = generators.netgen_style(n * scale, m * scale, seed=1, name=name)
        if len(_instances) >= 2:                 # keep the process small: at most two instances stay cached
            _instances.pop(next(iter(_instances)))
        _instances[key] = inst
    return _instances[key]


def run_pivots(eng, count: int) -> int:
    """Exactly `count` more pivots; restarts from the start basis when the optimum is reached.  One mcf_solve call per
    leg: the control block the solve leaves on the host answers stats() without another device round trip."""
    done = 0
    restarts = 0
    _, before = eng.poll()                       # (the host copy of the control block: no device round trip)
    while done < count:
        eng.solve(max_pivots=count - done)
        status, now = eng.poll()
        done += now - before
        before = now
        if status != 2 and done < count:         # not "iteration_limit": optimal (or infeasible) before the count was reached
            eng.reset()
            restarts += 1
            before = 0
    return restarts


def kernel_name(mode: int, rule: int, incremental: bool, bytes_per_arc: float = 9.0, variant: int = 0) -> str:
    """The pricing kernel as rocprofv3 names it (profiles/*kernel_stats*.csv)."""
    if mode == 2:
        return "k_solve_small"
    if mode == 3:
        return "k_solve_mid"
    if mode == 1 and rule != 1 and bytes_per_arc < 5:                      # 4-byte key codes; second argument: non-temporal loads
        return f"k_price_v<{'true' if incremental else 'false'}, {'true' if variant & 2 else 'false'}>"
    if mode == 1:
        return f"k_price_rc<{1 if rule == 1 else 0},false,{'true' if incremental else 'false'}>"
    return f"k_price<{1 if rule == 1 else 0},false>"


def pmc_traffic(workload: str, kernel: str):
    """HBM-side bytes per launch from the committed rocprofv3 --pmc passes (rocprofv3 cannot run inside bench.py)."""
    try:
        table = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        hit = table.get(f"{workload}|{kernel}")
        if hit:
            return hit["bytes_per_launch"], hit["source"]
    except Exception:
        pass
    return None, None


def measure_single(workload: str, steps: int, warmup: int, rule: int, profile_pass: bool = True, full_sweeps: int = 1) -> dict:
    """full_sweeps=1: every Dantzig / candidate-list sweep prices every arc (BASELINE.json's "full-scan" configs and
    the roofline measurement); 0: the engine's default, which from 4 M arcs on skips pricing workgroups whose arcs have
    not changed since they last swept them (same entering arcs, fewer bytes)."""
    import torch

    from network_flow_solver_amd import engine

    inst = make_instance(workload)
    eng = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, device=0,
                           full_sweeps=full_sweeps)
    run_pivots(eng, warmup)
    s0 = eng.stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    restarts = run_pivots(eng, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = eng.stats()
    mode = int(s1.get("pricing_mode", 0))
    pivots = s1["pivots"] - s0["pivots"] if restarts == 0 else steps
    arcs = s1["arcs_priced"] - s0["arcs_priced"] if restarts == 0 else steps * inst.m
    # arcs whose reduced cost was actually read: the grid sweeps count them per workgroup (kernel-per-phase path);
    # the single-workgroup loops read every arc they cover
    swept = s1["arcs_swept"] - s0["arcs_swept"] if (restarts == 0 and mode == 1 and rule != 1) else arcs
    incremental = bool(swept < 0.98 * arcs)
    out = {
        "workload": f"{inst.name}: {inst.n} nodes / {inst.m} arcs", "n": inst.n, "m": inst.m,
        "pivots": int(pivots), "seconds": dt, "pivots_per_sec": pivots / dt, "arcs_priced_per_sec": swept / dt,
        "arcs_covered_per_sec": arcs / dt, "incremental_sweeps": incremental,
        "ms_per_step": 1e3 * dt / max(pivots, 1), "restarts": restarts, "pricing_mode": MODE_NAMES.get(mode, str(mode)),
    }
    kname = kernel_name(mode, rule, incremental)
    if mode in (2, 3) and s1["loop_launches"] > s0["loop_launches"]:
        # one persistent workgroup ran the timed pivots: its launches were bracketed by HIP events on the engine's
        # stream INSIDE the timed region.  Latency-bound: the bytes are SURVEY 8d's per-pivot figures.
        launches = s1["loop_launches"] - s0["loop_launches"]
        ms = (s1["loop_ms"] - s0["loop_ms"]) / launches
        sweeps = pivots  # Dantzig / Devex: one pass per pivot (candidate list: fewer; arcs_priced already says so)
        nbytes = ((17 if rule == 1 else 13) * arcs + 8 * (inst.n + 1) * sweeps
                  + 20 * (s1["nodes_moved"] - s0["nodes_moved"]) + 28 * (s1["cycle_arcs"] - s0["cycle_arcs"])) / launches
        achieved = nbytes / (ms * 1e-3) / 1e9
        traffic, src = pmc_traffic(workload, kname)
        out["roofline"] = {
            "kernel": kname, "bound": "hbm", "workload": out["workload"], "achieved": achieved, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src,
            "bytes_per_launch": int(nbytes), "ms_per_launch": ms, "launches_in_timed_region": int(launches),
            "pivots_per_launch": pivots / launches, "us_per_pivot_in_kernel": 1e3 * ms * launches / max(pivots, 1),
            "note": "the timed region ran this ONE persistent workgroup (instance resident in LDS / one CU's caches): "
                    "latency-bound by construction -- per-pivot latency is the figure of merit, the HBM fraction is ~0; "
                    "bytes = SURVEY 8d per-pivot figures (13 B/arc priced + 8 B/node per sweep, 20 B/node moved, "
                    "28 B/cycle arc)",
        }
    elif profile_pass and mode in (0, 1):
        # dominant-kernel timing: HIP events on the engine's stream around every kernel of the same K pivots
        # (second pass from the same start state, deterministic -> same pivots, same kernels: eager instead of graph)
        eng.close()
        eng = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule, device=0,
                               profile=True, full_sweeps=full_sweeps)
        run_pivots(eng, warmup)
        p0 = eng.stats()
        run_pivots(eng, steps)
        p1 = eng.stats()
        launches = max(p1["price_launches"] - p0["price_launches"], 1)
        price_ms = (p1["price_ms"] - p0["price_ms"]) / launches
        pivot_ms = (p1["pivot_ms"] - p0["pivot_ms"]) / launches
        apply_ms = (p1["apply_ms"] - p0["apply_ms"]) / launches
        sweep_ms = eng.time_pricing(reps=50)            # back-to-back launches, no events in between
        out["kernel_ms"] = {"price": price_ms, "pivot": pivot_ms, "apply": apply_ms, "price_back_to_back": sweep_ms}
        per_pass = (p1["arcs_priced"] - p0["arcs_priced"]) / launches     # arcs one launch prices
        # the engine states the compulsory bytes of one launch of ITS sweep kernel (mcf_stats.price_bytes): 4 B per arc
        # for the key-code sweep, 9 B (13 Devex) for the resident-reduced-cost sweep, SURVEY 8d's figure for the gather
        nbytes = float(p1["price_bytes"])
        kname = kernel_name(mode, rule, incremental, nbytes / max(per_pass, 1), int(p1.get("sweep_variant", 0)))
        # The kernel's duration: 50 back-to-back launches of the sweep between two HIP events on the engine's stream, right
        # after the timed pivots, on the same device state.  (An event PAIR around every single eager launch of the profiled
        # pass also brackets that launch's dispatch latency, ~3 us: kept as *_event_pairs_in_loop.  rocprofv3's per-kernel
        # timestamps of the same loop agree with the back-to-back figure: profiles/r02_kernel_stats_bench_*.csv.)
        achieved = nbytes / (sweep_ms * 1e-3) / 1e9
        traffic, src = pmc_traffic(workload, kname)
        survey = (17 if rule == 1 else 13) * per_pass + 8 * (inst.n + 1)
        out["roofline"] = {
            "kernel": kname, "bound": "hbm", "workload": out["workload"],
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "frac_over_1": bool(achieved > HBM_PEAK_GBPS),   # never clamped: a fraction above 1 means the byte count or the timing is wrong
            "traffic": traffic, "traffic_source": src,
            "traffic_GBps": (traffic / (sweep_ms * 1e-3) / 1e9) if traffic else None,
            "traffic_frac_of_peak": (traffic / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
            "bytes_per_launch": int(nbytes), "ms_per_launch": sweep_ms,
            "ms_per_launch_event_pairs_in_loop": price_ms, "achieved_event_pairs_in_loop": nbytes / (price_ms * 1e-3) / 1e9,
            "working_set_fits_infinity_cache": bool(nbytes < 256 * 2 ** 20),
            "survey_8d": {"bytes_per_launch": int(survey), "ratio_to_kernel_bytes": survey / max(nbytes, 1),
                          "note": "SURVEY 8d prices the gather formulation (13 B/arc + 8 B/node); the resident sweeps move "
                                  "4 B/arc (key codes) or 9 B/arc (reduced cost + state) and leave the rest to k_update -- "
                                  "not a physical rate of this kernel"},
            "note": "ms_per_launch: 50 launches of the sweep between two HIP events on the engine's stream (kernel duration; agrees "
                    "with rocprofv3's average for the kernel in the same loop); *_event_pairs_in_loop: one event pair around every "
                    "eager launch of the profiled pass of the same K pivots, which includes ~3 us of dispatch latency per launch",
        }
        # the tree-update kernel of the same pass (k_update / k_update_bpl): SURVEY 8d's algorithmic bytes -- 20 B per node of the
        # re-hung subtrees + 44 B per arc incident to them (adjacency entry, other end's position, reduced-cost read-modify-write,
        # state, key code) -- against its duration (event pairs: includes ~3 us of dispatch per launch) and the PMC traffic of
        # the committed rocprofv3 pass.  A latency chain, not a stream: the fraction of the HBM peak says how far from bandwidth-
        # bound it is, `bytes_moved_over_needed` how much of what it rewrites is waste (the dense array shifts every position
        # between the subtree's old and new place; the blocked list moves the subtree and at most two cut-off runs of a block).
        d_piv = max(p1["pivots"] - p0["pivots"], 1)
        sub = (p1["subtree_nodes"] - p0["subtree_nodes"]) / d_piv
        moved = (p1["nodes_moved"] - p0["nodes_moved"]) / d_piv
        deg = 2.0 * inst.m / max(inst.n, 1)
        need = 20.0 * sub + (44.0 * deg * sub if mode == 1 else 0.0)
        uname = ("k_update_bpl" if int(p1.get("tree_blocks", 0)) else "k_update") if mode == 1 else ("k_update_bpl" if int(p1.get("tree_blocks", 0)) else "k_apply")
        utraffic, usrc = pmc_traffic(workload, uname)
        out["update_roofline"] = {
            "kernel": uname, "bound": "hbm", "tree_layout": f"blocked preorder list, blocks of {1 << int(p1['tree_blocks'])} slots" if int(p1.get("tree_blocks", 0)) else "dense preorder array",
            "subtree_nodes_per_pivot": sub, "positions_rewritten_per_pivot": moved, "bytes_moved_over_needed": (20.0 * moved) / max(20.0 * sub, 1e-9),
            "algorithmic_bytes_per_launch": need, "ms_per_launch_event_pairs": apply_ms,
            "achieved": need / max(apply_ms * 1e-3, 1e-12) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": need / max(apply_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBPS, "traffic": utraffic, "traffic_source": usrc,
            "note": "per launch = per basis swap; 44 B x (2m/n) arcs per re-hung node is the average adjacency, the patched arcs are not counted one by one"}
    eng.close()
    return out


def host_cores() -> int:
    """CPU threads this process may really use (affinity mask, cgroup quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(workload: str, budget_seconds: float = 10.0, max_threads: int = 64) -> dict:
    """The oracle -- a C restatement of the reference's algorithm -- on the same workload, bounded to roughly
    `budget_seconds` of wall time per leg: one thread, then every host core (independent solves side by side: the
    reference has no parallel path, so throughput over cores is all a CPU can add)."""
    import concurrent.futures as cf

    import oracle

    inst = make_instance(workload)
    small = inst.m <= 1 << 17

    def one_thread(budget: float):
        if small:
            t0 = time.perf_counter()
            pivots = arcs = solves = 0
            while time.perf_counter() - t0 < budget:
                r = oracle.solve_soa(inst, "dantzig", reference_order=False)
                pivots += r["iterations"]
                arcs += r["arcs_priced"]
                solves += 1
            return pivots, arcs, time.perf_counter() - t0, f"{solves} complete Dantzig solves of {inst.name}"
        r = oracle.solve_soa(inst, "dantzig", reference_order=False, pivot_budget=6)
        per = r["seconds"] / max(r["iterations"], 1)
        budget_p = int(max(6, min(400, budget / max(per, 1e-9))))
        r = oracle.solve_soa(inst, "dantzig", reference_order=False, pivot_budget=budget_p)
        return r["iterations"], r["arcs_priced"], r["seconds"], f"first {r['iterations']} Dantzig pivots of {inst.name}"

    pivots, arcs, dt, sample = one_thread(budget_seconds)
    out = {"value": pivots / dt, "unit": "pivots/s", "arcs_priced_per_sec": arcs / dt, "cores": 1, "kind": "port",
           "sample": sample, "host_cores_available": host_cores(),
           "note": "oracle/ref_simplex.c: the reference's float64 two-phase algorithm (per-pivot BFS rebuild + full "
                   "Dantzig scan) restated in C; the pure-Python reference itself cannot travel to the GPU box"}
    threads = min(host_cores(), max_threads if small else 8)   # (a 16 M-arc oracle instance holds ~2 GB per thread)
    if threads > 1:
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(max_workers=threads) as pool:   # ctypes releases the GIL inside ref_solve
            res = list(pool.map(lambda _: one_thread(budget_seconds * 0.6), range(threads)))
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": sum(r[0] for r in res) / wall, "unit": "pivots/s",
                            "arcs_priced_per_sec": sum(r[1] for r in res) / wall, "cores": threads,
                            "sample": f"{threads} threads, each: {res[0][3]}"}
    return out


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (nothing in this
    process has touched the GPU) and relay rank 0's JSON line."""
    import socket

    import __graft_entry__ as ge

    ge.build_hip()   # once, before the ranks start
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies takes the others' collectives with it: poll, and end everybody as soon as one has failed
    import threading

    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=30))
        except subprocess.TimeoutExpired:
            p.kill()
            rcs.append(p.wait())
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in out_chunks if c).decode())
    return 1 if failed else max(abs(rc) for rc in rcs)


def batched_point(rule: int, instances: int = 1024, nodes: int = 256, arcs: int = 2048, label: str = "netgen_8_08a") -> dict:
    """A BATCH of independent instances of one BASELINE size (seeds 1 .. instances), whole solves, one persistent workgroup
    (one CU) per instance in one launch (mcf_solve_batch): LDS-resident at netgen_8_08a size, state in global memory above.
    The headline `value` is one instance on one CU of 256 -- latency-bound by construction; this is the same per-pivot code
    filling the chip, and the GPU-side counterpart of cpu_baseline.all_cores (independent solves on every host core)."""
    from network_flow_solver_amd.batching import measure_batch

    b = measure_batch(rule, instances, nodes, arcs)
    return {"workload": f"{instances} independent {label}-sized instances ({nodes} nodes / {arcs} arcs, seeds 1..{instances}), whole solves",
            "pricing": RULE_NAMES[rule], "engine_path": MODE_NAMES.get(b["pricing_mode"], "?") + ", one persistent workgroup per instance",
            "launches": 1, "workgroups": instances, "all_optimal": b["all_optimal"],
            "pivots": b["pivots"], "kernel_ms": b["kernel_ms"], "wall_ms": 1e3 * b["wall_s"],
            "pivots_per_sec": b["pivots"] / b["wall_s"], "pivots_per_sec_in_kernel": b["pivots"] / (b["kernel_ms"] * 1e-3),
            "value": b["arcs_priced"] / b["wall_s"], "unit": "arcs/s", "solves_per_sec": instances / b["wall_s"],
            "note": "wall = host call incl. job upload and control-block read-back; handles created beforehand (instances resident in HBM)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--rule", default="dantzig", choices=["dantzig", "devex", "candidate_list"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = arcs per GPU fixed (the instance grows with N); strong = the BASELINE instance fixed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-point", action="store_true")
    args = ap.parse_args()
    rule = {"dantzig": 0, "devex": 1, "candidate_list": 2}[args.rule]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.gpus > 1 or world > 1 or os.environ.get("MCF_BENCH_FORCE_DIST") == "1":  # env: rehearse the sharded path on 1 GPU
        from network_flow_solver_amd import distributed

        distributed.bench_main(args, WORKLOADS, HBM_PEAK_GBPS)
        return

    import __graft_entry__ as ge

    ge.build_hip()
    workload = args.workload or "netgen_8_08a"
    head = measure_single(workload, args.steps, args.warmup, rule)
    line = {
        "metric": "pivots/sec + arcs-priced/sec (value = arcs-priced/sec; pivots_per_sec alongside) on netgen_8-style DIMACS",
        "value": head["arcs_priced_per_sec"], "unit": "arcs/s", "pivots_per_sec": head["pivots_per_sec"],
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": head["workload"], "pricing": RULE_NAMES[rule],
                   "step": "one pivot (price + ratio test + tree/potential update)", "parallelism": "1 GPU",
                   "engine_path": head["pricing_mode"],
                   "sweeps": "full (every arc priced every pivot)",
                   "restarts_in_timed_region": head["restarts"]},
        "roofline": head.get("roofline"), "kernel_ms": head.get("kernel_ms"),
        "reference_published": {"pivots_per_sec": 52, "instance": "netgen_8_08a (real file)", "hardware": "unstated CPU",
                                "source": "benchmarks/results/after-iteration-fix.json:89-98"},
    }
    default_run = not args.no_hbm_point and workload == "netgen_8_08a"
    if default_run:
        big = measure_single("netgen_1m_16m", min(args.steps, 200), min(args.warmup, 20), rule)
        line["hbm_point"] = {
            "workload": big["workload"], "value": big["arcs_priced_per_sec"], "unit": "arcs/s",
            "pivots_per_sec": big["pivots_per_sec"], "ms_per_step": big["ms_per_step"], "steps": big["pivots"],
            "roofline": big.get("roofline"), "kernel_ms": big.get("kernel_ms"), "update_roofline": big.get("update_roofline"),
            "note": "a budget shorter than the captured graph runs as eager launches of just the slots it needs (no idle graph slots)",
        }
        try:
            from network_flow_solver_amd import engine

            ms = engine.time_copy(1 << 30, reps=10)
            line["hbm_point"]["measured_copy_GBps"] = 2 * (1 << 30) / (ms * 1e-3) / 1e9  # read + write
        except Exception as exc:  # measurement aid only
            line["hbm_point"]["measured_copy_GBps"] = None
            line["hbm_point"]["copy_error"] = str(exc)
        # the engine's default at this size: incremental sweeps (same entering arcs; fewer arcs read per pivot)
        inc = measure_single("netgen_1m_16m", 2000, 200, rule, profile_pass=False, full_sweeps=0)
        line["hbm_point"]["incremental_sweeps"] = {
            "pivots_per_sec": inc["pivots_per_sec"], "ms_per_step": inc["ms_per_step"], "steps": inc["pivots"],
            "arcs_read_per_sec": inc["arcs_priced_per_sec"], "arcs_covered_per_sec": inc["arcs_covered_per_sec"],
            "note": "a pricing workgroup whose arcs did not change since it last swept them keeps its candidate; the entering "
                    "arc is still the arg-max over all arcs (pivot sequence identical, asserted by the GPU tests)"}
        # the other single-GPU-runnable BASELINE.json configs, measured the same way (no profiled pass)
        pts = []
        for wl, r, label in (("gridgen_8_14a", 1, "configs[2]: gridgen_8_14a, block-search Devex"),
                             ("goto_8_16a", 0, "configs[3] shape on 1 GPU: goto_8_16a, full-scan Dantzig"),
                             ("netgen_1m_16m", 2, "configs[4] shape on 1 GPU, candidate-list rule (the reference's default family)")):
            m_ = measure_single(wl, 2000, 200, r, profile_pass=False, full_sweeps=0)  # engine defaults
            pts.append({"config": label, "workload": m_["workload"], "pivots_per_sec": m_["pivots_per_sec"],
                        "arcs_priced_per_sec": m_["arcs_priced_per_sec"], "arcs_covered_per_sec": m_["arcs_covered_per_sec"],
                        "ms_per_step": m_["ms_per_step"], "steps": m_["pivots"], "engine_path": m_["pricing_mode"]})
        line["config_points"] = pts
        # beyond the Infinity Cache: 96 M arcs = 403 MB of key codes per sweep (906 MB of reduced costs + states)
        _instances.clear()
        far = measure_single("netgen_6m_96m", 40, 8, rule)
        line["hbm_point_beyond_infinity_cache"] = {
            "workload": far["workload"], "value": far["arcs_priced_per_sec"], "unit": "arcs/s",
            "pivots_per_sec": far["pivots_per_sec"], "ms_per_step": far["ms_per_step"], "steps": far["pivots"],
            "roofline": far.get("roofline"), "kernel_ms": far.get("kernel_ms"), "update_roofline": far.get("update_roofline")}
        _instances.clear()
    if default_run:
        try:   # secondary points: a failure here must not cost the headline line
            line["batched_point"] = batched_point(rule)
            # the largest size the reference publishes numbers for, under its default pricing strategy
            line["batched_point_netgen_8_12a"] = batched_point(1, instances=256, nodes=4096, arcs=32768, label="netgen_8_12a")
        except Exception as exc:  # noqa: BLE001
            line["batched_point_error"] = f"{type(exc).__name__}: {exc}"
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(workload)
        if "hbm_point" in line:
            line["hbm_point"]["cpu_baseline"] = cpu_baseline("netgen_1m_16m", budget_seconds=10.0)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
