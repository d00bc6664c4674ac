"""Batches of independent instances: one persistent workgroup (one CU) per instance in one launch (``mcf_solve_batch``).

The measurement helper shared by ``bench.py`` (one GPU) and ``distributed.bench_main`` (instances sharded across ranks:
independent objects, so no data-path collective -- each rank solves its own share)."""

from __future__ import annotations

import time

from . import engine, generators


class BatchRun:
    """`instances` netgen-style instances (seeds first_seed ...) with their handles; ``run`` solves them in one batched
    launch.  Creation and run are separate so that ranks can agree that everybody got this far before the timed part."""

    def __init__(self, rule: int, instances: int, nodes: int, arcs: int, first_seed: int = 1):
        self.instances = instances
        self.engines: list = []
        try:
            for k in range(instances):
                i = generators.netgen_style(nodes, arcs, seed=first_seed + k)
                self.engines.append(engine.McfEngine(i.n, i.tail, i.head, i.cost, i.cap, i.supply, rule=rule, mid_loop=1))
            engine.solve_batch(self.engines[:4], max_pivots=5)          # warm-up launch
            for eng in self.engines[:4]:
                eng.reset()
        except BaseException:
            self.close()
            raise

    def run(self, before=None, after=None) -> dict:
        """`before` / `after`: optional callables run right before / after the timed call (rank barriers)."""
        if before is not None:
            before()
        t0 = time.perf_counter()
        ms = engine.solve_batch(self.engines)
        if after is not None:
            after()
        wall = time.perf_counter() - t0
        stats = [eng.stats() for eng in self.engines]
        return {"instances": self.instances, "pivots": sum(st["pivots"] for st in stats),
                "arcs_priced": sum(st["arcs_priced"] for st in stats), "kernel_ms": ms, "wall_s": wall,
                "all_optimal": all(st["status"] == "optimal" for st in stats), "pricing_mode": int(stats[0]["pricing_mode"])}

    def close(self) -> None:
        for eng in self.engines:
            eng.close()
        self.engines = []


def measure_batch(rule: int, instances: int, nodes: int, arcs: int, first_seed: int = 1) -> dict:
    run = BatchRun(rule, instances, nodes, arcs, first_seed)
    try:
        return run.run()
    finally:
        run.close()
