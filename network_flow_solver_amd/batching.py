"""Batches of independent instances: one persistent workgroup (one CU) per instance in one launch (``mcf_solve_batch``).

The measurement helper shared by ``bench.py`` (one GPU) and ``distributed.bench_main`` (instances sharded across ranks:
independent objects, so no data-path collective -- each rank solves its own share)."""

from __future__ import annotations

import time

from . import engine, generators


def measure_batch(rule: int, instances: int, nodes: int, arcs: int, first_seed: int = 1, before=None, after=None) -> dict:
    """Create `instances` netgen-style instances (seeds first_seed ...), solve them in one batched launch, return counts and
    times.  `before` / `after`: optional callables run right before / after the timed call (rank barriers)."""
    insts = [generators.netgen_style(nodes, arcs, seed=first_seed + k) for k in range(instances)]
    engines = [engine.McfEngine(i.n, i.tail, i.head, i.cost, i.cap, i.supply, rule=rule, mid_loop=1) for i in insts]
    try:
        engine.solve_batch(engines[:4], max_pivots=5)          # warm-up launch
        for eng in engines[:4]:
            eng.reset()
        if before is not None:
            before()
        t0 = time.perf_counter()
        ms = engine.solve_batch(engines)
        if after is not None:
            after()
        wall = time.perf_counter() - t0
        stats = [eng.stats() for eng in engines]
        return {"instances": instances, "pivots": sum(st["pivots"] for st in stats), "arcs_priced": sum(st["arcs_priced"] for st in stats),
                "kernel_ms": ms, "wall_s": wall, "all_optimal": all(st["status"] == "optimal" for st in stats),
                "pricing_mode": int(stats[0]["pricing_mode"])}
    finally:
        for eng in engines:
            eng.close()
