"""Arc-sharded multi-GPU pivoting: one process per GPU over ``torch.distributed``.

Scheme: the arc list is cut into shards, one per rank (a shard is the
rank's contiguous 1/world share of each of the eight XCD head buckets).
Every rank keeps the WHOLE device-resident state (arc SoA, flows, potentials, preorder
tree -- memory is not the constraint at 288 GB per GPU) but prices only its own shard.
Per pivot:

    1. each rank: pricing sweep over its shard -> one 16-byte candidate (key, arc)
    2. ONE collective: all-gather of the candidates (RCCL has no MIN-LOC; 16 B per rank is
       latency-bound on xGMI, the ring's per-link bandwidth is irrelevant)
    3. each rank: the same deterministic pivot + tree/potential update on its replica

Step 3 is replicated rather than broadcast: the replicas stay bit-identical because every
input of the pivot (the gathered candidate list, integer state) is identical, so no second
message per pivot is needed.  The collective is issued on torch's current stream and the
engine kernels on the same stream, so a batch of pivots is enqueued without any host
synchronisation; the host polls the 16-byte status once per batch.

The pivot loop is written against a tiny engine protocol (``price_local``, ``pivot``,
``poll``, ``set_max_pivots``) so the CPU test-suite can run it with world_size 2 over gloo
on a CPU stand-in; the product adapter is ``HipShardEngine``.
"""

from __future__ import annotations

import json
import os
import time

import numpy as np


def shard_slices(bucket_off, world: int, rank: int) -> list[tuple[int, int]]:
    """Engine-order arc ranges of `rank` for a full (Dantzig) sweep: its 1/world share of every
    XCD head bucket (mcf_bucket_slice in csrc/mcf_core.h with one block), so each GPU keeps all
    eight XCDs busy."""
    out = []
    for x in range(len(bucket_off) - 1):
        s, length = bucket_off[x], bucket_off[x + 1] - bucket_off[x]
        out.append((s + length * rank // world, s + length * (rank + 1) // world))
    return out


class HipShardEngine:
    """Adapter: the HIP engine (C ABI) driven on torch's current stream."""

    def __init__(self, inst, rule: int, rank: int, world: int, device: int, block_size: int = 0, full_sweeps: int = 0):
        import torch

        from . import engine

        self.torch = torch
        self.eng = engine.McfEngine(inst.n, inst.tail, inst.head, inst.cost, inst.cap, inst.supply, rule=rule,
                                    block_size=block_size, device=device, shard=(rank, world), full_sweeps=full_sweeps)
        self.device = torch.device("cuda", device)

    def new_candidate_buffers(self, world: int):
        t = self.torch
        return (t.zeros(2, dtype=t.int64, device=self.device), t.zeros(2 * world, dtype=t.int64, device=self.device))

    def _stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def price_local(self, out) -> None:
        self.eng.enqueue_price(self._stream(), out.data_ptr())

    def pivot(self, cands, ncand: int) -> None:
        self.eng.enqueue_pivot(self._stream(), cands.data_ptr(), ncand)

    def poll(self):
        return self.eng.poll(self._stream())

    def set_max_pivots(self, total: int) -> None:
        self.eng.set_max_pivots(total)

    # candidate-list rule: one sweep -> a LIST of candidates per rank -> one all-gather -> minor_cap + 1 pivots
    def list_info(self) -> tuple[int, int]:
        return self.eng.shard_info()

    def new_list_buffers(self, world: int, list_len: int):
        t = self.torch
        return (t.full((2 * list_len,), -1, dtype=t.int64, device=self.device),
                t.full((2 * list_len * world,), -1, dtype=t.int64, device=self.device))

    def price_list(self, out) -> None:
        self.eng.enqueue_price_list(self._stream(), out.data_ptr())

    def pivots(self, cands, ncand: int, count: int) -> None:
        self.eng.enqueue_pivots(self._stream(), cands.data_ptr(), ncand, count)

    def close(self) -> None:
        self.eng.close()


def _enqueue_batch(eng, dist, world, local, gathered, batch, group, gather, listing=None):
    """`batch` pivot slots.  listing = (list_len, minor_cap) selects the candidate-list protocol: one sweep, ONE
    all-gather of the ranks' candidate lists, then minor_cap + 1 replicated pivots that re-price the gathered list --
    the collective is amortised over minor_cap + 1 pivots ("amortisation lever")."""
    if listing is not None:
        list_len, minor_cap = listing
        per = minor_cap + 1
        for _ in range(max(1, (batch + per - 1) // per)):
            eng.price_list(local)
            if gather:
                dist.all_gather_into_tensor(gathered, local, group=group)
                eng.pivots(gathered, list_len * world, per)
            else:
                eng.pivots(local, list_len, per)
        return
    for _ in range(batch):
        eng.price_local(local)
        if gather:
            dist.all_gather_into_tensor(gathered, local, group=group)
            eng.pivot(gathered, world)
        else:
            eng.pivot(local, 1)


class PivotLoop:
    """`batch` pivots per host round trip; optionally replayed from a captured graph.

    Graph mode (GPU engines only) captures price -> all-gather -> pivot x batch once on a side
    stream (RCCL collectives are capturable) and replays it, which removes the per-pivot host
    cost of the eager loop (two kernel enqueues + one torch.distributed call, ~40 us).  Any
    failure while capturing falls back to the eager loop."""

    def __init__(self, eng, dist, world: int, batch: int = 32, group=None, always_gather: bool = False,
                 use_graph: bool = False, listing: bool = False):
        self.eng, self.dist, self.world, self.batch, self.group = eng, dist, world, batch, group
        self.gather = world > 1 or always_gather
        self.listing = eng.list_info() if listing else None       # candidate-list rule: (list length per rank, minor cap)
        if self.listing is not None:
            self.local, self.gathered = eng.new_list_buffers(world, self.listing[0])
        else:
            self.local, self.gathered = eng.new_candidate_buffers(world)
        self.graph = None
        self.graph_error = None
        if use_graph and hasattr(eng, "torch"):
            self._try_capture()

    def _try_capture(self):
        torch = self.eng.torch
        try:
            # one eager batch first: communicator set-up and lazy allocations must not be captured
            _enqueue_batch(self.eng, self.dist, self.world, self.local, self.gathered, 1, self.group, self.gather, self.listing)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                _enqueue_batch(self.eng, self.dist, self.world, self.local, self.gathered, self.batch, self.group,
                               self.gather, self.listing)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = g
        except Exception as exc:  # capture unsupported here: stay eager
            self.graph = None
            self.graph_error = f"{type(exc).__name__}: {exc}"
            torch.cuda.synchronize()

    def run(self, max_total_pivots: int):
        """Pivot until a final status or `max_total_pivots`; returns (status, pivots) with status in
        the MCF_ST_* numbering (0 optimal-or-infeasible, 2 iteration limit, 3 unbounded)."""
        self.eng.set_max_pivots(max_total_pivots)
        while True:
            if self.graph is not None:
                self.graph.replay()
            else:
                _enqueue_batch(self.eng, self.dist, self.world, self.local, self.gathered, self.batch, self.group,
                               self.gather, self.listing)
            status, pivots = self.eng.poll()
            if status is not None:
                return status, pivots


def run_pivots(eng, dist, world: int, max_total_pivots: int, batch: int = 32, group=None, always_gather: bool = False,
               listing: bool = False):
    """Eager convenience wrapper (used by the gloo tests)."""
    return PivotLoop(eng, dist, world, batch, group, always_gather, listing=listing).run(max_total_pivots)


# ---------------------------------------------------------------------------------------------
# bench.py --gpus N entry (launched by torch.distributed.run, one rank per GPU)
# ---------------------------------------------------------------------------------------------
def bench_main(args, workloads, hbm_peak_gbps: float) -> None:
    import torch
    import torch.distributed as dist

    from . import generators

    # RCCL prints a version banner on stdout when the communicator comes up; the contract is ONE
    # JSON line on stdout, so everything else that lands on fd 1 is sent to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    rule = {"dantzig": 0, "devex": 1, "candidate_list": 2}[args.rule]
    workload = args.workload or "netgen_8_08a"

    strong = getattr(args, "scaling", "weak") == "strong"
    verbose = os.environ.get("MCF_DIST_VERBOSE") == "1"

    def note(msg: str) -> None:
        if verbose:
            os.write(2, f"[rank {rank}] {msg}\n".encode())

    def measure(wl: str, steps: int, warmup: int) -> dict:
        fam, n1, m1 = workloads[wl]
        # weak scaling: per-GPU arcs fixed, the instance grows with the number of GPUs;
        # strong scaling: the BASELINE instance itself, cut into `world` shards
        scale = 1 if strong else world
        tag = f"{wl}(synthetic,x{scale})"
        if fam == "gridgen":
            inst = generators.gridgen_style(n1 * scale, m1, seed=1, name=tag)
        elif fam == "goto":
            inst = generators.goto_style(n1 * scale, m1, seed=1, name=tag)
        else:
            inst = generators.netgen_style(n1 * scale, m1 * scale, seed=1, name=tag)
        note(f"instance {inst.name} built")
        eng = HipShardEngine(inst, rule, rank, world, local_rank, full_sweeps=1)  # value counts every arc of every sweep: so price them all
        note("engine created")
        force = os.environ.get("MCF_BENCH_FORCE_DIST") == "1"  # 1-GPU rehearsal: still issue the collective
        loop = PivotLoop(eng, dist, world, batch=32, always_gather=force,
                         use_graph=os.environ.get("MCF_DIST_GRAPH", "1") == "1",  # MCF_DIST_GRAPH=0: eager loop
                         listing=rule == 2)   # candidate list: one all-gather of the ranks' lists per minor_cap + 1 pivots
        note(f"pivot loop ready ({'captured graph' if loop.graph is not None else 'eager'}; {loop.graph_error})")
        loop.run(warmup)
        note("warm-up done")
        _, p0 = eng.poll()
        a0 = eng.eng.stats()["arcs_priced"]
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status, p1 = loop.run(p0 + steps)
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        note("timed region done")
        pivots = p1 - p0
        arcs = eng.eng.stats()["arcs_priced"] - a0          # whole-job accounting: every pass counts the arcs of ALL shards
        sweep_ms = eng.eng.time_pricing(reps=20)
        shard_arcs = inst.m // world
        # The sweep kernel this rank's handle launches and its compulsory bytes come from the engine itself
        # (mcf_stats.price_bytes / sweep_variant / pricing_mode), exactly as in bench.py's single-GPU leg: a sharded
        # full-sweep Dantzig handle from 4 M arcs on sweeps 4-byte key codes (k_price_v), not 9-byte reduced costs.
        st = eng.eng.stats()
        variant, mode = int(st.get("sweep_variant", 0)), int(st.get("pricing_mode", 1))
        bytes_per_launch = float(st["price_bytes"])
        per_arc = bytes_per_launch / max(shard_arcs, 1)
        if mode == 1 and rule != 1 and (variant & 1):
            kname = f"k_price_v<{'true' if variant & 4 else 'false'}, {'true' if variant & 2 else 'false'}> (per-rank shard)"
        elif mode == 1:
            kname = f"k_price_rc<{1 if rule == 1 else 0},false,{'true' if variant & 4 else 'false'}> (per-rank shard)"
        else:
            kname = f"k_price<{1 if rule == 1 else 0},false> (per-rank shard)"
        achieved = bytes_per_launch / (sweep_ms * 1e-3) / 1e9
        frac = achieved / hbm_peak_gbps
        # where a pivot's time goes on this rank: the sweep (kernel duration above), the collective (a timed loop of the very
        # all-gather the pivot loop issues), and what is left of the measured time per pivot = pivot + update kernels + gaps
        coll_ms = None
        try:
            list_len, minor_cap = loop.listing if loop.listing is not None else (1, 0)
            buf = torch.zeros(2 * list_len, dtype=torch.int64, device="cuda")
            out = torch.zeros(world * buf.numel(), dtype=torch.int64, device="cuda")
            dist.all_gather_into_tensor(out, buf)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for _ in range(50):
                dist.all_gather_into_tensor(out, buf)
            torch.cuda.synchronize()
            coll_ms = 1e3 * (time.perf_counter() - tc) / 50
        except Exception:  # noqa: BLE001  (gloo rehearsals have no CUDA tensors)
            coll_ms = None
        per_pivot_ms = 1e3 * dt / max(pivots, 1)
        sweeps_per_pivot = 1.0 if loop.listing is None else 1.0 / (loop.listing[1] + 1)
        phases = {"sweep_ms_per_launch": sweep_ms, "sweeps_per_pivot": sweeps_per_pivot,
                  "collective_ms_per_call": coll_ms, "collectives_per_pivot": sweeps_per_pivot if rule == 2 else 1.0,
                  "ms_per_pivot": per_pivot_ms,
                  "pivot_and_update_ms_per_pivot": per_pivot_ms - sweeps_per_pivot * sweep_ms - (coll_ms or 0.0) * (sweeps_per_pivot if rule == 2 else 1.0),
                  "note": "replicated on every rank: the pivot kernel and the tree update; divided by the ranks: the sweep and, for the "
                          "Dantzig / candidate-list rules, the reduced-cost patch"}
        eng.close()
        return {"workload": f"{inst.name}: {inst.n} nodes / {inst.m} arcs, {shard_arcs} arcs per GPU", "pivots": pivots,
                "seconds": dt, "pivots_per_sec": pivots / dt,
                "arcs_priced_per_sec": arcs / dt,
                "ms_per_step": per_pivot_ms, "completed": status == 2,
                "pivot_loop": "captured graph" if loop.graph is not None else "eager", "graph_error": loop.graph_error,
                "phases": phases,
                "roofline": {"kernel": kname, "bound": "hbm",
                             "achieved": achieved, "peak": hbm_peak_gbps, "unit": "GB/s",
                             "frac": frac, "frac_over_1": bool(frac > 1.0), "traffic": None,
                             "bytes_per_launch": int(bytes_per_launch), "bytes_per_arc": per_arc, "ms_per_launch": sweep_ms,
                             "working_set_fits_infinity_cache": bool(bytes_per_launch < 256 * 2 ** 20),
                             "survey_8d": {"bytes_per_launch": int((17 if rule == 1 else 13) * shard_arcs + 8 * (inst.n + 1))},
                             "note": "back-to-back launches between two HIP events on the engine's stream; kernel and bytes from "
                                     "mcf_stats (price_bytes, sweep_variant); traffic: no PMC pass exists for the sharded sweep "
                                     "(the 1-GPU passes are in profiles/pmc_traffic.json)"}}

    head = measure(workload, args.steps, args.warmup)
    line = {
        "metric": "pivots/sec + arcs-priced/sec (value = arcs-priced/sec; pivots_per_sec alongside) on netgen_8-style DIMACS",
        "value": head["arcs_priced_per_sec"], "unit": "arcs/s", "pivots_per_sec": head["pivots_per_sec"],
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
        "config": {"workload": head["workload"], "pricing": {0: "full-scan Dantzig", 1: "block-search Devex", 2: "candidate list"}[rule],
                   "step": "one pivot (sharded price + 16 B all-gather + replicated tree/potential update)",
                   "parallelism": f"arc-sharded x{world}, replicated tree, 1 RCCL all-gather per pivot",
                   "pivot_loop": head["pivot_loop"], "graph_error": head["graph_error"]},
        "roofline": head["roofline"], "phases": head["phases"],
    }
    if not args.no_hbm_point and workload == "netgen_8_08a" and not strong:
        big = measure("netgen_8_18a" if "netgen_8_18a" in workloads else "netgen_8_16a", min(args.steps, 200), min(args.warmup, 20))
        line["hbm_point"] = {k: big[k] for k in ("workload", "pivots_per_sec", "ms_per_step", "roofline", "pivot_loop")}
        line["hbm_point"]["value"] = big["arcs_priced_per_sec"]
        line["hbm_point"]["unit"] = "arcs/s"
    if not args.no_hbm_point and workload == "netgen_8_08a" and not strong:
        # The other way this work spreads over GPUs: INDEPENDENT instances sharded across the ranks -- no data-path
        # collective at all; each rank solves its share as one batched launch (one persistent workgroup per instance).
        from .batching import BatchRun

        per_rank = 1024

        def fence():
            torch.cuda.synchronize()
            dist.barrier()

        run, err = None, ""
        try:
            run = BatchRun(rule, per_rank, 256, 2048, first_seed=1 + rank * per_rank)
        except Exception as exc:  # noqa: BLE001  (every rank must reach the vote below)
            err = f"{type(exc).__name__}: {exc}"
        ready = torch.tensor([1.0 if run is not None else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if ready.item() == 1.0:
            # a rank whose batch fails must still reach both fences and the vote, or the others wait for it forever
            b, fenced = None, [0]

            def counted_fence():
                fenced[0] += 1
                fence()

            try:
                b = run.run(before=counted_fence, after=counted_fence)
            except Exception as exc:  # noqa: BLE001
                err = f"{type(exc).__name__}: {exc}"
                while fenced[0] < 2:
                    counted_fence()
            good = torch.tensor([1.0 if b is not None else 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(good, op=dist.ReduceOp.MIN)
            if good.item() != 1.0:
                b = None
                line["batched_point_error"] = err or "another rank's batched solve failed"
        if ready.item() == 1.0 and b is not None:
            agg = torch.tensor([b["wall_s"], float(b["pivots"]), float(b["arcs_priced"]), 1.0 if b["all_optimal"] else 0.0],
                               dtype=torch.float64, device="cuda")
            tmax = agg[:1].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
            secs = float(tmax.item())
            line["batched_point"] = {
                "workload": f"{per_rank * world} independent netgen_8_08a-sized instances (256 nodes / 2048 arcs), {per_rank} per GPU, whole solves",
                "parallelism": f"instances sharded x{world}, no collective in the data path", "scaling": "weak",
                "pivots": int(agg[1].item()), "seconds": secs, "pivots_per_sec": float(agg[1].item()) / secs,
                "value": float(agg[2].item()) / secs, "unit": "arcs/s", "solves_per_sec": per_rank * world / secs,
                "all_optimal": bool(agg[3].item() == world)}
        elif ready.item() != 1.0:
            line["batched_point_error"] = err or "another rank could not create its handles"
        if run is not None:
            run.close()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()
