"""ctypes binding of ``libmcf_hip.so`` (C ABI: ``include/mcf.h``).

This is the only door between Python and the HIP engine.  There is no CPU
fallback: if the shared library is missing, or no HIP device is usable,
creating an engine raises ``EngineUnavailableError``.
"""

from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from pathlib import Path

import numpy as np

from .exceptions import NetworkSolverError

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MCF_HIP_LIB", _PKG / "libmcf_hip.so"))  # override: A/B builds of the same ABI

ABI_VERSION = 3
RULE_DANTZIG = 0
RULE_DEVEX_BLOCK = 1
RULE_CANDIDATE_LIST = 2
STATUS_NAMES = {0: "optimal", 1: "infeasible", 2: "iteration_limit", 3: "unbounded"}
CAP_INF = -1
KEY_PLAIN, KEY_FORWARD_FIRST, KEY_PRIORITY, KEY_CAPACITY = 0, 1, 2, 3   # mcf_options.key_mode

# every symbol include/mcf.h declares (tests check that the library exports each one)
ABI_SYMBOLS = (
    "mcf_default_options", "mcf_create", "mcf_solve", "mcf_solve_batch", "mcf_get_result", "mcf_price_once", "mcf_reset", "mcf_set_basis",
    "mcf_enqueue_price", "mcf_enqueue_pivot", "mcf_shard_info", "mcf_enqueue_price_list", "mcf_enqueue_pivots", "mcf_poll", "mcf_set_max_pivots", "mcf_time_pricing",
    "mcf_time_copy", "mcf_get_tree", "mcf_get_reduced_costs", "mcf_get_pricing_keys", "mcf_get_weights", "mcf_dimacs_scan", "mcf_dimacs_load", "mcf_last_error", "mcf_destroy", "mcf_abi_version", "mcf_device_count",
)


class EngineUnavailableError(NetworkSolverError):
    """The HIP engine cannot run here (library not built, or no MI355X visible)."""


class EngineError(NetworkSolverError):
    """The HIP engine returned an error code."""

    def __init__(self, code: int, message: str):
        super().__init__(f"mcf error {code}: {message}")
        self.code = code


class McfOptions(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32), ("device", ctypes.c_int32), ("rule", ctypes.c_int32),
        ("batch_pivots", ctypes.c_int32), ("use_graph", ctypes.c_int32), ("profile", ctypes.c_int32),
        ("block_size", ctypes.c_int64), ("shard_rank", ctypes.c_int64), ("shard_count", ctypes.c_int64),
        ("price_blocks", ctypes.c_int32), ("no_fused", ctypes.c_int32), ("no_rcache", ctypes.c_int32),
        ("cycle_scan", ctypes.c_int32), ("mid_loop", ctypes.c_int32), ("full_sweeps", ctypes.c_int32),
        ("devex_tuner", ctypes.c_int32), ("devex_stay", ctypes.c_int32), ("forward_first", ctypes.c_int32),
        ("compressed_keys", ctypes.c_int32), ("vkey_half_log2", ctypes.c_int32), ("climb_depth", ctypes.c_int32),
        ("overlap_update", ctypes.c_int32), ("key_mode", ctypes.c_int32), ("arc_priority", ctypes.POINTER(ctypes.c_int8)),
        ("tree_blocks", ctypes.c_int32), ("tree_pool", ctypes.c_int32), ("rc_drop", ctypes.c_int32), ("pivot_run", ctypes.c_int32),
    ]


class McfStats(ctypes.Structure):
    _fields_ = [
        ("pivots", ctypes.c_int64), ("degenerate", ctypes.c_int64), ("bound_flips", ctypes.c_int64),
        ("arcs_priced", ctypes.c_int64), ("nodes_moved", ctypes.c_int64), ("subtree_nodes", ctypes.c_int64),
        ("cycle_arcs", ctypes.c_int64), ("batches", ctypes.c_int64), ("unbounded_arc", ctypes.c_int64),
        ("unbounded_rc", ctypes.c_int64), ("solve_seconds", ctypes.c_double), ("price_ms", ctypes.c_double),
        ("pivot_ms", ctypes.c_double), ("apply_ms", ctypes.c_double), ("price_launches", ctypes.c_int64),
        ("pivot_launches", ctypes.c_int64), ("apply_launches", ctypes.c_int64), ("price_bytes", ctypes.c_int64),
        ("artificial_flow", ctypes.c_int64), ("pricing_mode", ctypes.c_int64),
        ("cycle_scans", ctypes.c_int64), ("scan_rounds", ctypes.c_int64), ("arcs_swept", ctypes.c_int64),
        ("loop_ms", ctypes.c_double), ("loop_launches", ctypes.c_int64), ("sweep_variant", ctypes.c_int64),
        ("tree_blocks", ctypes.c_int64), ("tree_rebuilds", ctypes.c_int64), ("rc_dropped_at", ctypes.c_int64),
        ("run_pairs", ctypes.c_int64), ("run_left_at", ctypes.c_int64),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


PROGRESS_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double)

_lib = None


def load_library():
    """dlopen libmcf_hip.so and declare the prototypes.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise EngineUnavailableError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The engine has no CPU fallback.")
    lib = ctypes.CDLL(str(LIB_PATH))
    vp = ctypes.c_void_p
    i32p, i64p, i8p = (ctypes.POINTER(t) for t in (ctypes.c_int32, ctypes.c_int64, ctypes.c_int8))
    lib.mcf_abi_version.restype = ctypes.c_int
    lib.mcf_device_count.restype = ctypes.c_int
    lib.mcf_default_options.argtypes = [ctypes.POINTER(McfOptions)]
    lib.mcf_default_options.restype = None
    lib.mcf_create.argtypes = [ctypes.c_int32, ctypes.c_int64, i32p, i32p, i64p, i64p, i64p,
                               ctypes.POINTER(McfOptions), ctypes.POINTER(vp)]
    lib.mcf_solve.argtypes = [vp, ctypes.c_int64, PROGRESS_CB, vp, ctypes.c_int64]
    lib.mcf_get_result.argtypes = [vp, i32p, i64p, i64p, i64p, i8p, ctypes.POINTER(McfStats)]
    lib.mcf_price_once.argtypes = [vp, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, i64p, i32p, i64p]
    lib.mcf_solve_batch.argtypes = [ctypes.POINTER(vp), ctypes.c_int32, i64p, ctypes.POINTER(ctypes.c_double)]
    lib.mcf_reset.argtypes = [vp]
    lib.mcf_set_basis.argtypes = [vp, i8p, i8p]
    lib.mcf_enqueue_price.argtypes = [vp, vp, vp]
    lib.mcf_enqueue_pivot.argtypes = [vp, vp, vp, ctypes.c_int32]
    lib.mcf_shard_info.argtypes = [vp, i32p, i32p]
    lib.mcf_enqueue_price_list.argtypes = [vp, vp, vp]
    lib.mcf_enqueue_pivots.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int32]
    lib.mcf_poll.argtypes = [vp, vp, i32p, i64p]
    lib.mcf_set_max_pivots.argtypes = [vp, ctypes.c_int64]
    lib.mcf_time_pricing.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
    lib.mcf_time_copy.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
    lib.mcf_get_tree.argtypes = [vp, i32p, i32p, i32p, i32p, i32p, i8p, i64p, i32p, i32p]
    lib.mcf_get_reduced_costs.argtypes = [vp, i64p, i32p]
    lib.mcf_get_weights.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    lib.mcf_get_pricing_keys.argtypes = [vp, i32p, i32p]
    lib.mcf_dimacs_scan.argtypes = [ctypes.c_char_p, i64p, i64p, ctypes.c_char_p, ctypes.c_int32]
    lib.mcf_dimacs_load.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, i32p, i32p, i64p, i64p, i64p, i64p,
                                    ctypes.c_char_p, ctypes.c_int32]
    lib.mcf_last_error.argtypes = [vp]
    lib.mcf_last_error.restype = ctypes.c_char_p
    lib.mcf_destroy.argtypes = [vp]
    lib.mcf_destroy.restype = None
    for name in ("mcf_create", "mcf_solve", "mcf_get_result", "mcf_price_once", "mcf_reset", "mcf_set_basis", "mcf_enqueue_price",
                 "mcf_enqueue_pivot", "mcf_shard_info", "mcf_enqueue_price_list", "mcf_enqueue_pivots", "mcf_poll", "mcf_set_max_pivots", "mcf_time_pricing", "mcf_time_copy",
                 "mcf_get_tree", "mcf_get_reduced_costs", "mcf_get_pricing_keys", "mcf_get_weights", "mcf_dimacs_scan", "mcf_dimacs_load"):
        getattr(lib, name).restype = ctypes.c_int
    if lib.mcf_abi_version() != ABI_VERSION:
        raise EngineUnavailableError("libmcf_hip.so ABI version mismatch")
    _lib = lib
    return lib


def device_count() -> int:
    return int(load_library().mcf_device_count())


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


@dataclass
class EngineResult:
    status: str
    objective: int
    flow: np.ndarray        # int64[m]
    potential: np.ndarray   # int64[n]
    in_tree: np.ndarray     # bool[m]
    stats: dict


class McfEngine:
    """One device-resident min-cost-flow instance (integer data, 0-based node ids)."""

    def __init__(self, n: int, tail, head, cost, cap, supply, rule: int = RULE_DANTZIG, block_size: int = 0,
                 batch_pivots: int = 64, use_graph: bool = True, profile: bool = False, device: int = -1,
                 shard: tuple[int, int] | None = None, price_blocks: int = 0, fused: bool = True,
                 resident_rc: bool = True, cycle_scan: int = 0, mid_loop: int = 0, full_sweeps: int = 0,
                 devex_tuner: int = 0, devex_stay: bool = False, forward_first: bool = False, compressed_keys: int = 0,
                 vkey_half_log2: int = 0, climb_depth: int = 0, overlap_update: int = 0, key_mode: int = 0, arc_priority=None,
                 tree_blocks: int = 0, tree_pool: int = 0, rc_drop: int = 0, pivot_run: int = 0):
        self._h = None
        lib = load_library()
        if lib.mcf_device_count() <= 0:
            raise EngineUnavailableError("no HIP device visible; the network-simplex engine has no CPU fallback")
        self._lib = lib
        self.n = int(n)
        self.tail = np.ascontiguousarray(tail, dtype=np.int32)
        self.head = np.ascontiguousarray(head, dtype=np.int32)
        self.cost = np.ascontiguousarray(cost, dtype=np.int64)
        self.cap = np.ascontiguousarray(cap, dtype=np.int64)
        self.supply = np.ascontiguousarray(supply, dtype=np.int64)
        self.m = int(self.tail.shape[0])
        if not (self.head.shape[0] == self.cost.shape[0] == self.cap.shape[0] == self.m):
            raise ValueError("arc arrays differ in length")
        if self.supply.shape[0] != self.n:
            raise ValueError("supply must have n entries")
        opt = McfOptions()
        lib.mcf_default_options(ctypes.byref(opt))
        opt.device = device
        opt.rule = rule
        opt.block_size = int(block_size or 0)
        opt.batch_pivots = int(batch_pivots)
        # MCF_USE_GRAPH=0: eager launches instead of captured graphs (the configuration the rocprofv3 summaries are taken in)
        opt.use_graph = 1 if use_graph and os.environ.get("MCF_USE_GRAPH", "1") != "0" else 0
        opt.profile = 1 if profile else 0
        opt.price_blocks = int(price_blocks)
        opt.no_fused = 0 if fused else 1
        opt.no_rcache = 0 if resident_rc else 1
        opt.cycle_scan = int(cycle_scan)
        opt.mid_loop = int(mid_loop)
        opt.full_sweeps = int(full_sweeps)   # 0 auto, 1 never incremental, -1 always incremental
        opt.devex_tuner = int(devex_tuner)   # 0 auto (on unless block_size is given), 1 on, -1 off
        opt.devex_stay = 1 if devex_stay else 0
        opt.forward_first = 1 if forward_first else 0
        opt.compressed_keys = int(compressed_keys)   # 0 auto (on for the Dantzig-key grid sweeps), -1 off
        opt.vkey_half_log2 = int(vkey_half_log2)
        # key variant of the Dantzig / candidate-list sweep (the reference's specialised entering rules): KEY_PLAIN,
        # KEY_FORWARD_FIRST (= forward_first), KEY_PRIORITY (arc_priority: bit 0 forward, bit 1 backward), KEY_CAPACITY
        self.key_mode = int(key_mode) if key_mode else (KEY_FORWARD_FIRST if forward_first else KEY_PLAIN)
        opt.key_mode = self.key_mode
        self._arc_priority = None
        if int(key_mode) == KEY_PRIORITY:
            self._arc_priority = np.ascontiguousarray(arc_priority, dtype=np.int8)
            if self._arc_priority.shape[0] != len(tail):
                raise ValueError("arc_priority needs one byte per arc")
            opt.arc_priority = _p(self._arc_priority, ctypes.c_int8)
        opt.overlap_update = int(overlap_update)     # 1 = pricing of pivot t+1 beside the permutation of pivot t (A/B switch: measured slower)
        opt.climb_depth = int(climb_depth)           # 0 auto, -1 never, k: end points of depth <= k are climbed outright
        # layout of the tree's preorder: 0 auto (blocked list from 32 768 nodes on), -1 dense array, k = blocks of 2^k slots;
        # MCF_TREE_BLOCKS / MCF_TREE_POOL override the default (A/B runs of whole scripts)
        opt.tree_blocks = int(tree_blocks) if tree_blocks else int(os.environ.get("MCF_TREE_BLOCKS", "0"))
        opt.tree_pool = int(tree_pool) if tree_pool else int(os.environ.get("MCF_TREE_POOL", "0"))
        opt.rc_drop = int(rc_drop)   # resident reduced costs given up from this average re-hung subtree size on (0 auto, -1 never)
        opt.pivot_run = int(pivot_run)   # candidate list on the blocked list: pivots back to back in one workgroup (0 off, k pairs per period)
        if shard is not None:
            opt.shard_rank, opt.shard_count = int(shard[0]), int(shard[1])
        self.rule = rule
        h = ctypes.c_void_p()
        rc = lib.mcf_create(self.n, self.m, _p(self.tail, ctypes.c_int32), _p(self.head, ctypes.c_int32),
                            _p(self.cost, ctypes.c_int64), _p(self.cap, ctypes.c_int64),
                            _p(self.supply, ctypes.c_int64), ctypes.byref(opt), ctypes.byref(h))
        if rc != 0:
            msg = (lib.mcf_last_error(None) or b"").decode()
            if rc == -2:
                raise EngineUnavailableError(msg)
            raise EngineError(rc, msg)
        self._h = h

    # -- helpers
    def _check(self, rc: int):
        if rc != 0:
            raise EngineError(rc, (self._lib.mcf_last_error(self._h) or b"").decode())

    def close(self):
        if self._h is not None:
            self._lib.mcf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- solve path
    def solve(self, max_pivots: int = -1, progress=None, progress_interval: int = 100) -> None:
        """Pivot until optimal / unbounded / ``max_pivots`` more pivots.

        ``progress(pivots, max_pivots, elapsed_seconds)`` is called every
        ``progress_interval`` pivots; a truthy return value stops the solve.
        """
        if progress is not None:
            def _cb(_user, pivots, cap, elapsed):
                return 1 if progress(int(pivots), int(cap), float(elapsed)) else 0
            cb = PROGRESS_CB(_cb)
        else:
            cb = PROGRESS_CB(0)
        self._check(self._lib.mcf_solve(self._h, int(max_pivots), cb, None, int(progress_interval)))

    def result(self) -> EngineResult:
        status = ctypes.c_int32(0)
        obj = np.zeros(2, dtype=np.int64)
        flow = np.zeros(max(self.m, 1), dtype=np.int64)
        pot = np.zeros(self.n, dtype=np.int64)
        in_tree = np.zeros(max(self.m, 1), dtype=np.int8)
        stats = McfStats()
        self._check(self._lib.mcf_get_result(self._h, ctypes.byref(status), _p(obj, ctypes.c_int64),
                                             _p(flow, ctypes.c_int64), _p(pot, ctypes.c_int64),
                                             _p(in_tree, ctypes.c_int8), ctypes.byref(stats)))
        objective = (int(obj[0]) << 64) + (int(obj[1]) & ((1 << 64) - 1))
        return EngineResult(STATUS_NAMES[status.value], objective, flow[: self.m], pot,
                            in_tree[: self.m].astype(bool), stats.as_dict())

    def stats(self) -> dict:
        """Counters + coarse status without copying the flows back (status is one of
        running / optimal / iteration_limit / unbounded; infeasibility needs ``result()``)."""
        stats = McfStats()
        self._check(self._lib.mcf_get_result(self._h, None, None, None, None, None, ctypes.byref(stats)))
        d = stats.as_dict()
        st, _ = self.poll()
        d["status"] = "running" if st is None else STATUS_NAMES[st]
        return d

    def reset(self) -> None:
        self._check(self._lib.mcf_reset(self._h))

    def set_basis(self, in_tree, at_upper=None) -> bool:
        """Warm start from a basis (bool/int8 per arc, caller's order).  False when the engine rejected it
        (cycle, empty, flows outside the bounds) -- the handle is then at the cold start, as in the reference."""
        it = np.ascontiguousarray(in_tree, dtype=np.int8)
        au = None if at_upper is None else np.ascontiguousarray(at_upper, dtype=np.int8)
        if it.shape[0] != self.m or (au is not None and au.shape[0] != self.m):
            raise ValueError("basis arrays must have one entry per arc")
        if self.m == 0:
            return False
        rc = self._lib.mcf_set_basis(self._h, _p(it, ctypes.c_int8), None if au is None else _p(au, ctypes.c_int8))
        if rc == -6:
            return False
        self._check(rc)
        return True

    def last_error(self) -> str:
        return (self._lib.mcf_last_error(self._h) or b"").decode()

    def price_once(self, rule: int | None = None, start: int = 0, end: int | None = None):
        """One pricing pass. Returns (arc, dir, key) or None when no arc is eligible."""
        arc, key, d = ctypes.c_int64(-1), ctypes.c_int64(0), ctypes.c_int32(0)
        self._check(self._lib.mcf_price_once(self._h, self.rule if rule is None else rule, int(start),
                                             self.m if end is None else int(end), ctypes.byref(arc),
                                             ctypes.byref(d), ctypes.byref(key)))
        return None if arc.value < 0 else (int(arc.value), int(d.value), int(key.value))

    def tree(self) -> dict:
        N = self.n + 1
        parent, pred, size, pos, order, depth, psize = (np.zeros(N, dtype=np.int32) for _ in range(7))
        state = np.zeros(max(self.m, 1), dtype=np.int8)
        pi = np.zeros(N, dtype=np.int64)
        i32 = ctypes.c_int32
        self._check(self._lib.mcf_get_tree(self._h, _p(parent, i32), _p(pred, i32), _p(size, i32), _p(pos, i32),
                                           _p(order, i32), _p(state, ctypes.c_int8), _p(pi, ctypes.c_int64), _p(depth, i32),
                                           _p(psize, i32)))
        return {"parent": parent, "pred_arc": pred, "size": size, "pos": pos, "order": order, "depth": depth, "psize": psize,
                "state": state[: self.m], "pi": pi}

    def reduced_costs(self):
        """(rc[m] in caller order, resident?) -- what the pricing kernel reads."""
        rc = np.zeros(max(self.m, 1), dtype=np.int64)
        res = ctypes.c_int32(0)
        self._check(self._lib.mcf_get_reduced_costs(self._h, _p(rc, ctypes.c_int64), ctypes.byref(res)))
        return rc[: self.m], bool(res.value)

    def pricing_keys(self):
        """(compressed Dantzig key per arc in caller order, present?) -- what k_price_v reads."""
        k = np.zeros(max(self.m, 1), dtype=np.int32)
        present = ctypes.c_int32(0)
        self._check(self._lib.mcf_get_pricing_keys(self._h, _p(k, ctypes.c_int32), ctypes.byref(present)))
        return k[: self.m], bool(present.value)

    def weights(self) -> np.ndarray:
        """Devex reference weights per arc (caller's order)."""
        w = np.ones(max(self.m, 1), dtype=np.float32)
        self._check(self._lib.mcf_get_weights(self._h, _p(w, ctypes.c_float)))
        return w[: self.m]

    # -- measurement
    def time_pricing(self, reps: int = 20, rule: int | None = None) -> float:
        ms = ctypes.c_double(0.0)
        self._check(self._lib.mcf_time_pricing(self._h, self.rule if rule is None else rule, int(reps), ctypes.byref(ms)))
        return float(ms.value)

    # -- multi-GPU (arc-sharded) building blocks
    def enqueue_price(self, stream: int, cand_out_ptr: int) -> None:
        self._check(self._lib.mcf_enqueue_price(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(cand_out_ptr)))

    def enqueue_pivot(self, stream: int, cands_ptr: int, ncand: int) -> None:
        self._check(self._lib.mcf_enqueue_pivot(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(cands_ptr), int(ncand)))

    def shard_info(self) -> tuple[int, int]:
        """(candidates one sweep leaves, minor pivots allowed per sweep)."""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        self._check(self._lib.mcf_shard_info(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def enqueue_price_list(self, stream: int, cands_out_ptr: int) -> None:
        self._check(self._lib.mcf_enqueue_price_list(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(cands_out_ptr)))

    def enqueue_pivots(self, stream: int, cands_ptr: int, ncand: int, count: int) -> None:
        self._check(self._lib.mcf_enqueue_pivots(self._h, ctypes.c_void_p(stream), ctypes.c_void_p(cands_ptr), int(ncand), int(count)))

    def poll(self, stream: int = 0):
        st, pv = ctypes.c_int32(0), ctypes.c_int64(0)
        self._check(self._lib.mcf_poll(self._h, ctypes.c_void_p(stream), ctypes.byref(st), ctypes.byref(pv)))
        return (None if st.value < 0 else int(st.value)), int(pv.value)

    def set_max_pivots(self, total: int) -> None:
        self._check(self._lib.mcf_set_max_pivots(self._h, int(total)))


def time_copy(nbytes: int, reps: int = 10, device: int = -1) -> float:
    """Milliseconds per device-to-device copy of ``nbytes`` (measured HBM copy ceiling)."""
    lib = load_library()
    ms = ctypes.c_double(0.0)
    rc = lib.mcf_time_copy(device, int(nbytes), int(reps), ctypes.byref(ms))
    if rc == -2:
        raise EngineUnavailableError("no HIP device visible")
    if rc != 0:
        raise EngineError(rc, "mcf_time_copy failed")
    return float(ms.value)


def dimacs_load(path: str):
    """Native DIMACS reader (host code in libmcf_hip.so; no GPU needed).
    Returns (n, tail, head, lower, cap, cost, supply) with 0-based int arrays."""
    from .exceptions import InvalidProblemError

    lib = load_library()
    n, m = ctypes.c_int64(0), ctypes.c_int64(0)
    err = ctypes.create_string_buffer(512)
    bpath = str(path).encode()
    if lib.mcf_dimacs_scan(bpath, ctypes.byref(n), ctypes.byref(m), err, 512) != 0:
        raise InvalidProblemError(err.value.decode())
    tail = np.zeros(max(m.value, 1), np.int32)
    head = np.zeros(max(m.value, 1), np.int32)
    lower, cap, cost = (np.zeros(max(m.value, 1), np.int64) for _ in range(3))
    supply = np.zeros(n.value, np.int64)
    if lib.mcf_dimacs_load(bpath, n.value, m.value, _p(tail, ctypes.c_int32), _p(head, ctypes.c_int32),
                           _p(lower, ctypes.c_int64), _p(cap, ctypes.c_int64), _p(cost, ctypes.c_int64),
                           _p(supply, ctypes.c_int64), err, 512) != 0:
        raise InvalidProblemError(err.value.decode())
    k = m.value
    return n.value, tail[:k], head[:k], lower[:k], cap[:k], cost[:k], supply


def solve_batch(engines, max_pivots=None) -> float:
    """Solve independent instances side by side: one persistent workgroup per engine, one launch per engine path
    (``mcf_solve_batch``).  Every engine must be on the fused LDS path (``stats()["pricing_mode"] == 2``) or on the persistent
    loop (``pricing_mode == 3``; ``mid_loop=1`` asks for it at any size).  ``max_pivots``: None
    (the reference's default budget everywhere), one int, or one int per engine.  Returns the launches' duration in
    milliseconds; results through each engine's ``result()`` as after ``solve()``."""
    engines = list(engines)
    if not engines:
        return 0.0
    lib = engines[0]._lib
    hs = (ctypes.c_void_p * len(engines))(*[e._h.value for e in engines])
    caps = None
    if max_pivots is not None:
        caps = np.ascontiguousarray(np.broadcast_to(np.asarray(max_pivots, dtype=np.int64), (len(engines),)))
    ms = ctypes.c_double(0.0)
    rc = lib.mcf_solve_batch(hs, len(engines), None if caps is None else _p(caps, ctypes.c_int64), ctypes.byref(ms))
    if rc != 0:
        engines[0]._check(rc)
    return float(ms.value)
