"""Seeded random problem descriptions with the awkward features the reference's API allows:
negative costs, zero and unlimited capacities, lower bounds, parallel arcs, fractional data,
undirected edges, infeasible demand patterns.  Unlimited capacities only point from a lower to a
higher node index, so no uncapacitated cycle exists (the reference's phase-1 costs `c - 1 - eps*idx`
(simplex.py:1165) can make it report such a cycle as unbounded even when the true costs are
non-negative; that quirk is not a parity target)."""

import random


def make(seed: int):
    rng = random.Random(9000 + seed)
    n = rng.randint(2, 9)
    ids = [f"v{i}" for i in range(n)]
    directed = rng.random() < 0.85
    frac = rng.random() < 0.2
    unit = 0.5 if frac else 1.0
    supplies = [0.0] * n
    for _ in range(rng.randint(1, 3)):
        a, b = rng.sample(range(n), 2)
        q = rng.randint(1, 9) * unit
        supplies[a] += q
        supplies[b] -= q
    nodes = [{"id": ids[i], "supply": supplies[i]} for i in range(n)]
    arcs = []
    for _ in range(rng.randint(0, 3 * n)):
        a, b = rng.sample(range(n), 2)
        cost = rng.randint(-4 if directed else 0, 12) * (0.25 if frac and rng.random() < 0.5 else 1.0)
        r = rng.random()
        if directed and r < 0.15 and a < b:
            cap = None
        elif r < 0.25:
            cap = 0.0
        else:
            cap = rng.randint(1, 12) * unit
        lower = 0.0
        if directed and cap not in (None, 0.0) and rng.random() < 0.2:
            lower = min(cap, rng.randint(1, 3) * unit)
        if not directed and cap is None:
            cap = 5.0
        arcs.append({"tail": ids[a], "head": ids[b], "capacity": cap, "cost": float(cost), "lower": lower})
    return nodes, arcs, directed
