"""Seeded synthetic DIMACS-style min-cost-flow instance generators.

The DIMACS/LEMON instance files the reference benchmarks on (netgen_8_08a,
gridgen_8_14a, goto_8_16a, ...) are downloaded by the reference at run time
(/root/reference/benchmarks/scripts/download_dimacs.py:51-161) and are not
available offline, so the build carries its own generators with the same
shape parameters:

* ``netgen_style``  -- sqrt(n) sources and sinks, integer costs 1..10^4,
  capacities 1..10^3, plus a feasibility skeleton whose arcs carry the whole
  supply.  Arcs are emitted tail-major, as NETGEN does.
* ``gridgen_style`` -- W x H grid plus one super node, m ~= 8 n.
* ``goto_style``    -- grid-on-torus with one source and one sink, m = 8 n,
  the family the reference's GOTO heuristic
  (/root/reference/src/network_solver/simplex.py:358-363) switches to Dantzig for.

Everything is integer and deterministic for a given (family, size, seed).
Instances are returned in structure-of-arrays form (``ArcSoA``) because that
is the layout the HIP engine consumes; ``to_network_problem`` / ``write_dimacs``
convert to the reference-style object model and DIMACS text.
"""

from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass
from pathlib import Path

import numpy as np


@dataclass
class ArcSoA:
    """A min-cost-flow instance in flat arrays (node ids are 0-based ints).

    DIMACS node id = index + 1.  ``cap`` uses -1 for "uncapacitated".
    """

    n: int
    tail: np.ndarray  # int32[m]
    head: np.ndarray  # int32[m]
    cost: np.ndarray  # int64[m]
    cap: np.ndarray  # int64[m]
    supply: np.ndarray  # int64[n]
    name: str = "instance"

    @property
    def m(self) -> int:
        return int(self.tail.shape[0])

    def sha256(self) -> str:
        h = hashlib.sha256()
        h.update(np.int64(self.n).tobytes())
        for a in (self.tail, self.head, self.cost, self.cap, self.supply):
            h.update(np.ascontiguousarray(a).tobytes())
        return h.hexdigest()


def _split_total(rng: np.random.Generator, total: int, parts: int) -> np.ndarray:
    """Split ``total`` into ``parts`` positive integers."""
    if parts == 1:
        return np.array([total], dtype=np.int64)
    base = np.ones(parts, dtype=np.int64)
    rest = total - parts
    cuts = np.sort(rng.integers(0, rest + 1, size=parts - 1))
    pieces = np.diff(np.concatenate(([0], cuts, [rest])))
    return base + pieces.astype(np.int64)


def _dedupe_and_fill(
    rng: np.random.Generator,
    n: int,
    skel_tail: np.ndarray,
    skel_head: np.ndarray,
    m: int,
) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Return m distinct (tail, head) pairs that contain the skeleton.

    The third array flags skeleton arcs.
    """
    skel_key = skel_tail.astype(np.int64) * n + skel_head.astype(np.int64)
    skel_key = np.unique(skel_key)
    want = m - skel_key.shape[0]
    if want < 0:
        raise ValueError("m is smaller than the feasibility skeleton")
    extra = np.empty(0, dtype=np.int64)
    while extra.shape[0] < want:
        need = want - extra.shape[0]
        draw = int(need * 1.1) + 16
        t = rng.integers(0, n, size=draw, dtype=np.int64)
        h = rng.integers(0, n, size=draw, dtype=np.int64)
        ok = t != h
        k = t[ok] * n + h[ok]
        k = k[~np.isin(k, skel_key)]
        extra = np.unique(np.concatenate((extra, k)))
        if extra.shape[0] > want:
            # np.unique sorted the keys; drop a random subset, not the tail end
            keep = rng.permutation(extra.shape[0])[:want]
            extra = extra[np.sort(keep)]
    keys = np.concatenate((skel_key, extra))
    is_skel = np.concatenate(
        (np.ones(skel_key.shape[0], dtype=bool), np.zeros(extra.shape[0], dtype=bool))
    )
    order = np.argsort(keys, kind="stable")  # tail-major, then head
    keys = keys[order]
    is_skel = is_skel[order]
    return (keys // n).astype(np.int32), (keys % n).astype(np.int32), is_skel


def netgen_style(n: int, m: int, seed: int = 0, name: str | None = None) -> ArcSoA:
    """NETGEN-shaped transshipment instance (input 2 and 5)."""
    rng = np.random.default_rng([0x6E657467, n, m, seed])
    k = max(1, int(math.isqrt(n)))
    k = min(k, n // 2) if n >= 2 else 1
    total = 1000 * k
    supply = np.zeros(n, dtype=np.int64)
    perm = rng.permutation(n)
    sources = perm[:k]
    sinks = perm[k : 2 * k]
    supply[sources] = _split_total(rng, total, k)
    supply[sinks] -= _split_total(rng, total, k)
    # feasibility skeleton: one directed Hamiltonian cycle that can carry everything
    cyc = rng.permutation(n).astype(np.int64)
    tail, head, is_skel = _dedupe_and_fill(rng, n, cyc, np.roll(cyc, -1), m)
    cost = rng.integers(1, 10_001, size=m, dtype=np.int64)
    cap = rng.integers(1, 1_001, size=m, dtype=np.int64)
    cap[is_skel] = total
    return ArcSoA(n, tail, head, cost, cap, supply, name or f"netgen_style_n{n}_m{m}_s{seed}")


def gridgen_style(width: int, height: int, seed: int = 0, name: str | None = None) -> ArcSoA:
    """GRIDGEN-shaped instance: W x H grid + one super node, m = 8 n."""
    rng = np.random.default_rng([0x67726964, width, height, seed])
    g = width * height
    n = g + 1
    m = 8 * n
    k = max(1, int(math.isqrt(n)))
    total = 1000 * k
    supply = np.zeros(n, dtype=np.int64)
    perm = rng.permutation(g)
    supply[perm[:k]] = _split_total(rng, total, k)
    supply[perm[k : 2 * k]] -= _split_total(rng, total, k)
    idx = np.arange(g, dtype=np.int64).reshape(height, width)
    st, sh = [], []
    # grid skeleton, both directions, so every source reaches every sink
    st += [idx[:, :-1].ravel(), idx[:, 1:].ravel(), idx[:-1, :].ravel(), idx[1:, :].ravel()]
    sh += [idx[:, 1:].ravel(), idx[:, :-1].ravel(), idx[1:, :].ravel(), idx[:-1, :].ravel()]
    # super node touches one grid column in both directions
    col = idx[:, 0].ravel()
    st += [np.full(col.shape, g, dtype=np.int64), col]
    sh += [col, np.full(col.shape, g, dtype=np.int64)]
    tail, head, is_skel = _dedupe_and_fill(rng, n, np.concatenate(st), np.concatenate(sh), m)
    cost = rng.integers(1, 10_001, size=m, dtype=np.int64)
    cap = rng.integers(1, 1_001, size=m, dtype=np.int64)
    cap[is_skel] = total
    return ArcSoA(n, tail, head, cost, cap, supply, name or f"gridgen_style_{width}x{height}_s{seed}")


def goto_style(width: int, height: int, seed: int = 0, name: str | None = None) -> ArcSoA:
    """GOTO-shaped instance: grid on a torus, one source, one sink, m = 8 n."""
    rng = np.random.default_rng([0x676F746F, width, height, seed])
    n = width * height
    m = 8 * n
    total = 1000 * max(1, int(math.isqrt(n)))
    supply = np.zeros(n, dtype=np.int64)
    s, t = rng.choice(n, size=2, replace=False)
    supply[s] = total
    supply[t] = -total
    idx = np.arange(n, dtype=np.int64).reshape(height, width)
    right = np.roll(idx, -1, axis=1)
    down = np.roll(idx, -1, axis=0)
    st = [idx.ravel(), idx.ravel()]
    sh = [right.ravel(), down.ravel()]
    # longer torus jumps give the characteristic many-alternative-paths structure
    for jump in (2, 3, 5):
        st += [idx.ravel(), idx.ravel()]
        sh += [np.roll(idx, -jump, axis=1).ravel(), np.roll(idx, -jump, axis=0).ravel()]
    stc, shc = np.concatenate(st), np.concatenate(sh)
    ok = stc != shc
    stc, shc = stc[ok], shc[ok]
    keys = np.unique(stc * n + shc)
    if keys.shape[0] > m:
        # keep the unit-step skeleton, drop surplus jump arcs
        unit = np.unique(np.concatenate((idx.ravel() * n + right.ravel(), idx.ravel() * n + down.ravel())))
        unit = unit[unit // n != unit % n]
        others = keys[~np.isin(keys, unit)]
        others = others[np.sort(rng.permutation(others.shape[0])[: m - unit.shape[0]])]
        keys = np.concatenate((unit, others))
    tail, head, _ = _dedupe_and_fill(rng, n, keys // n, keys % n, m)
    skel = ((head.astype(np.int64) == right.ravel()[tail]) | (head.astype(np.int64) == down.ravel()[tail]))
    cost = rng.integers(1, 10_001, size=m, dtype=np.int64)
    cap = rng.integers(1, 1_001, size=m, dtype=np.int64)
    cap[skel] = total
    return ArcSoA(n, tail, head, cost, cap, supply, name or f"goto_style_{width}x{height}_s{seed}")


# Named stand-ins for the BASELINE.json configs (the real files are unobtainable offline).
def named_instance(name: str) -> ArcSoA:
    """Build the synthetic stand-in for one of the BASELINE.json config names."""
    table = {
        "netgen_8_08a": lambda: netgen_style(256, 2048, seed=1, name="netgen_8_08a(synthetic)"),
        "netgen_8_08b": lambda: netgen_style(256, 2048, seed=2, name="netgen_8_08b(synthetic)"),
        "netgen_8_10a": lambda: netgen_style(1024, 8192, seed=1, name="netgen_8_10a(synthetic)"),
        "netgen_8_12a": lambda: netgen_style(4096, 32768, seed=1, name="netgen_8_12a(synthetic)"),
        "netgen_8_14a": lambda: netgen_style(16384, 131072, seed=1, name="netgen_8_14a(synthetic)"),
        "netgen_8_16a": lambda: netgen_style(65536, 524288, seed=1, name="netgen_8_16a(synthetic)"),
        "netgen_8_18a": lambda: netgen_style(262144, 2097152, seed=1, name="netgen_8_18a(synthetic)"),
        "netgen_8_20a": lambda: netgen_style(1 << 20, 8 << 20, seed=1, name="netgen_8_20a(synthetic)"),
        "gridgen_8_08a": lambda: gridgen_style(16, 16, seed=1, name="gridgen_8_08a(synthetic)"),
        "gridgen_8_14a": lambda: gridgen_style(128, 128, seed=1, name="gridgen_8_14a(synthetic)"),
        "goto_8_08a": lambda: goto_style(16, 16, seed=1, name="goto_8_08a(synthetic)"),
        "goto_8_12a": lambda: goto_style(64, 64, seed=1, name="goto_8_12a(synthetic)"),
        "goto_8_14a": lambda: goto_style(128, 128, seed=1, name="goto_8_14a(synthetic)"),
        "goto_8_16a": lambda: goto_style(256, 256, seed=1, name="goto_8_16a(synthetic)"),
        "netgen_1m_16m": lambda: netgen_style(1 << 20, 16 << 20, seed=1, name="netgen_1M_16M(synthetic)"),
        "netgen_4m_64m": lambda: netgen_style(4 << 20, 64 << 20, seed=1, name="netgen_4M_64M(synthetic)"),
        "netgen_6m_96m": lambda: netgen_style(6 << 20, 96 << 20, seed=1, name="netgen_6M_96M(synthetic)"),
    }
    if name not in table:
        raise KeyError(f"unknown instance '{name}'; known: {sorted(table)}")
    return table[name]()


def write_dimacs(inst: ArcSoA, path: str | Path) -> None:
    """Write the instance as DIMACS ``p min`` text (1-based node ids, lower bound 0)."""
    lines = [f"c {inst.name}", f"p min {inst.n} {inst.m}"]
    for v in np.nonzero(inst.supply)[0]:
        lines.append(f"n {int(v) + 1} {int(inst.supply[v])}")
    t1 = inst.tail.astype(np.int64) + 1
    h1 = inst.head.astype(np.int64) + 1
    for i in range(inst.m):
        cap = int(inst.cap[i])
        lines.append(f"a {int(t1[i])} {int(h1[i])} 0 {cap} {int(inst.cost[i])}")
    Path(path).write_text("\n".join(lines) + "\n", encoding="utf-8")


def to_node_arc_dicts(inst: ArcSoA) -> tuple[list[dict], list[dict]]:
    """Reference-style ``build_problem`` inputs (DIMACS 1-based string ids)."""
    nodes = [{"id": str(v + 1), "supply": float(inst.supply[v])} for v in range(inst.n)]
    arcs = []
    for i in range(inst.m):
        cap = int(inst.cap[i])
        arcs.append(
            {
                "tail": str(int(inst.tail[i]) + 1),
                "head": str(int(inst.head[i]) + 1),
                "capacity": None if cap < 0 else float(cap),
                "cost": float(inst.cost[i]),
                "lower": 0.0,
            }
        )
    return nodes, arcs
