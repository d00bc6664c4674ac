"""Shared fixtures.  ``-m "not gpu"`` runs everything that needs no device (oracle vs
goldens, host logic, ABI exports); ``-m gpu`` runs the parity tests proper, which call the
HIP engine through the C ABI."""

from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds")


@pytest.fixture(scope="session", autouse=True)
def _native_builds():
    """Compile the oracle (gcc), the CPU emulation (g++) and, when hipcc is around, the HIP
    library.  On the GPU box the prebuilt .so files travel with the snapshot."""
    import __graft_entry__ as ge
    import oracle

    oracle.build()
    oracle.build_emul()
    try:
        ge.build_hip()
    except Exception as exc:  # no hipcc on this machine: the prebuilt library must exist
        if not ge.LIB.exists():
            raise RuntimeError(f"libmcf_hip.so is missing and could not be built: {exc}")


def load_cases():
    return json.loads((GOLDEN / "cases.json").read_text())


def load_synthetic():
    from network_flow_solver_amd.generators import ArcSoA

    out = []
    for s in json.loads((GOLDEN / "synthetic.json").read_text()):
        z = np.load(GOLDEN / s["file"], allow_pickle=False)
        inst = ArcSoA(int(z["n"]), z["tail"], z["head"], z["cost"], z["cap"], z["supply"], s["name"])
        assert inst.sha256() == s["sha256"], f"fixture {s['file']} is corrupt"
        out.append((s, inst))
    return out


CASES = load_cases()
CASE_IDS = [c["name"] for c in CASES]


def golden_flows(expected: dict) -> dict:
    return {(t, h): f for t, h, f in expected["flows"]}


def strategies_agree(case: dict) -> bool:
    """Both reference strategies returned the same flows (necessary for a unique optimum)."""
    e = case["expected"]
    return all("flows" in v for v in e.values()) and len({json.dumps(v["flows"]) for v in e.values()}) == 1


def check_tree_invariants(n: int, parent, size, pos, order, depth=None, psize=None):
    """Preorder-array spanning tree: order is a permutation, pos inverts it, every subtree is
    the contiguous block [pos, pos+size) nested in its parent's, sizes add up."""
    N = n + 1
    assert sorted(np.asarray(order).tolist()) == list(range(N))
    assert all(order[pos[v]] == v for v in range(N))
    assert parent[n] == -1 and pos[n] == 0 and size[n] == N
    child_sum = np.zeros(N, dtype=np.int64)
    for v in range(n):
        p = parent[v]
        assert pos[p] < pos[v] and pos[v] + size[v] <= pos[p] + size[p], f"block of {v} not nested in {p}"
        child_sum[p] += size[v]
    assert np.array_equal(child_sum + 1, np.asarray(size, dtype=np.int64))
    if psize is not None and not (np.asarray(psize) == -1).all():  # position-space sizes: what the cycle scan tests
        # ancestry with (-1 everywhere: the handle keeps none, e.g. the LDS-resident loop always climbs)
        assert np.array_equal(np.asarray(psize)[np.asarray(pos)], np.asarray(size))
    if depth is not None:  # the depth-balanced cycle walk relies on these
        assert depth[n] == 0 and all(depth[v] == depth[parent[v]] + 1 for v in range(n))


def check_optimality(inst, flow, potential):
    """Primal feasibility + complementary slackness: a certificate of optimality that needs
    no oracle (size-independent property used at BASELINE sizes)."""
    flow = np.asarray(flow, dtype=np.int64)
    bal = inst.supply.astype(np.int64).copy()
    np.subtract.at(bal, inst.tail, flow)
    np.add.at(bal, inst.head, flow)
    assert not bal.any(), "flow conservation violated"
    capped = inst.cap >= 0
    assert (flow >= 0).all() and (flow[capped] <= inst.cap[capped]).all(), "capacity violated"
    rc = inst.cost + potential[inst.tail] - potential[inst.head]
    interior = (flow > 0) & (~capped | (flow < inst.cap))
    assert (rc[interior] == 0).all(), "basic/interior arc with non-zero reduced cost"
    at_lower = (flow == 0) & (inst.cap != 0)
    assert (rc[at_lower] >= 0).all(), "arc at lower bound with negative reduced cost"
    at_upper = capped & (flow == inst.cap) & (inst.cap > 0)
    assert (rc[at_upper] <= 0).all(), "arc at upper bound with positive reduced cost"
    return rc


def optimum_is_unique(inst, flow, in_tree, rc) -> bool:
    """Dual non-degeneracy: every non-basic arc has rc != 0  =>  the optimal flow is unique."""
    nonbasic = ~np.asarray(in_tree, dtype=bool)
    return bool((rc[nonbasic] != 0).all())


@pytest.fixture(scope="session")
def gpu_engine_module():
    from network_flow_solver_amd import engine

    if engine.device_count() <= 0:
        pytest.fail("gpu-marked test running without a HIP device: the engine has no CPU fallback")
    return engine
