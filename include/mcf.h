/*
 * mcf.h -- C ABI of libmcf_hip.so, the MI355X (gfx950) network-simplex pivot engine.
 *
 * The reference (jeffreyhorn/network_flow_solver) is pure Python and has no FFI; its
 * seams are Python-level.  This library is what a ctypes stub
 * behind those seams binds:
 *
 *   reference seam (file:line under /root/reference/src/network_solver/)      entry point here
 *   -----------------------------------------------------------------------   ----------------
 *   NetworkSimplex.__init__: arc build, SoA mirror, initial tree
 *       simplex.py:99-265, 392-456, 619-730 ..................................  mcf_create
 *   NetworkSimplex.solve / _run_simplex_iterations / _pivot
 *       simplex.py:1109-1160, 1176-1425, 1446-1701 ...........................  mcf_solve
 *   result extraction  simplex.py:1703-1765 ..................................  mcf_get_result
 *   PricingStrategy.select_entering_arc  simplex_pricing.py:57-86, 97-137,
 *       310-357 and NetworkSimplex._select_entering_arc_vectorized
 *       simplex.py:528-617 ...................................................  mcf_price_once
 *   ProgressCallback cadence  simplex.py:1143-1154 ...........................  mcf_progress_cb
 *   UnboundedProblemError(entering_arc, reduced_cost)  exceptions.py:65-93 ...  MCF_ST_UNBOUNDED + stats.unbounded_arc
 *   warm start  simplex.py:740-1010, 1491-1532 ..............................  mcf_set_basis
 *   AdaptiveTuner.adapt_block_size  simplex_adaptive.py:98-151 and the
 *       periodic Devex reset  simplex.py:1370-1400 ..........................  inside mcf_solve (MCF_RULE_DEVEX_BLOCK)
 *   specialised pivot strategies  specialized_pivots.py:69-424, 452-527 .....  mcf_options.key_mode (+ arc_priority): row scan,
 *                                                                               min-cost scan, shortest-path / matching
 *                                                                               preference classes, max-flow capacity merit
 *   the benchmark runner's loop over instances
 *       benchmarks/runners/run_benchmark.py (one solve after the other) ......  mcf_solve_batch (one launch, one CU per instance)
 *   parse_dimacs_file  benchmarks/parsers/dimacs.py:77-286 ..................  mcf_dimacs_scan / mcf_dimacs_load
 *
 * Conventions: plain pointers and sizes only; integer return codes (0 = ok, < 0 =
 * MCF_E_*), never exceptions; the caller owns every buffer it passes; the library owns
 * device memory until mcf_destroy; one handle is not thread-safe, distinct handles are
 * independent (handles of small instances share a few pooled streams: their work is
 * serialised on the device, never mixed up).  All problem data are integers: the Python shim scales decimal input
 * (and applies the reference's lower-bound shift, simplex.py:413-428) before the call.
 *
 * There is NO CPU fallback: every compute entry point fails with MCF_E_NO_DEVICE when no
 * HIP device is usable.
 */
#ifndef MCF_H
#define MCF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCF_ABI_VERSION 3

/* return codes */
#define MCF_OK 0
#define MCF_E_BAD_ARG (-1)
#define MCF_E_NO_DEVICE (-2)
#define MCF_E_HIP (-3)
#define MCF_E_ALLOC (-4)
#define MCF_E_RANGE (-5)     /* value outside what the integer engine represents */
#define MCF_E_STATE (-6)     /* call not valid in the handle's current state */
#define MCF_E_INTERNAL (-7)

/* solve status (mcf_get_result) -- mapped by the shim onto the reference's strings
 * (data.py:269-322): optimal / infeasible / iteration_limit; unbounded becomes
 * UnboundedProblemError. */
#define MCF_ST_OPTIMAL 0
#define MCF_ST_INFEASIBLE 1
#define MCF_ST_ITERATION_LIMIT 2
#define MCF_ST_UNBOUNDED 3

/* pricing rules */
#define MCF_RULE_DANTZIG_FULL 0   /* full-scan most-violating arc (simplex_pricing.py:97-137) */
#define MCF_RULE_DEVEX_BLOCK 1    /* round-robin block search, merit rc^2/w, deferred weight update
                                     (simplex_pricing.py:310-357, 271-292), block-size tuner (simplex_adaptive.py:98-151),
                                     weights reset every 64 basis swaps (simplex.py:1370-1400) */
#define MCF_RULE_CANDIDATE_LIST 2 /* full Dantzig sweep keeps one candidate per pricing workgroup; the following
                                     pivots re-price only that list (simplex_pricing.py:375-542, 419-456) */

/* "uncapacitated" marker accepted in cap[] (besides any value >= 2^60) */
#define MCF_CAP_INF (-1)

typedef struct mcf_handle mcf_handle;

typedef struct mcf_options {
    int32_t abi_version;     /* MCF_ABI_VERSION */
    int32_t device;          /* HIP device ordinal; -1 = current device */
    int32_t rule;            /* MCF_RULE_* */
    int32_t batch_pivots;    /* pivots enqueued per host round trip (0 = default 64) */
    int32_t use_graph;       /* 1 = replay a captured hipGraph of batch_pivots pivots */
    int32_t profile;         /* 1 = bracket every kernel with HIP events (no graph), fills *_ms */
    int64_t block_size;      /* Devex block size; 0 = auto (simplex_adaptive.py:89-96) */
    int64_t shard_rank;      /* multi-GPU: this handle prices shard shard_rank of shard_count */
    int64_t shard_count;     /*   (0 or 1 = all arcs); a shard is 1/shard_count of every XCD head bucket */
    int32_t price_blocks;    /* pricing grid size; 0 = auto */
    int32_t no_fused;        /* 1 = never use the fused LDS-resident kernel for small instances */
    int32_t no_rcache;       /* 1 = never keep reduced costs resident: always price by gathering potentials */
    int32_t cycle_scan;      /* cycle search: 0 = auto, -1 = always climb parent pointers, k >= 1 = climb k - 1 round
                                trips, then finish by the position-space scan (1 = scan only) */
    int32_t mid_loop;        /* persistent single-workgroup pivot loop for mid-size instances (k_solve_mid):
                                0 = auto (by size and rule), -1 = never, 1 = whenever the handle allows it */
    int32_t full_sweeps;     /* incremental Dantzig / candidate-list sweeps (a pricing workgroup whose arcs have not changed since
                                it last swept them keeps its candidate): 0 = auto (from 4 M arcs), 1 = never, -1 = always */
    int32_t devex_tuner;     /* Devex block-size tuner (simplex_adaptive.py:98-151): 0 = auto (on when block_size is 0, as in the
                                reference; off for a caller-given block size), 1 = on, -1 = off */
    int32_t devex_stay;      /* Devex block advance: 0 = cyclic (next block after every pivot), 1 = stay on a block until it holds
                                no eligible arc (the reference's loop, simplex_pricing.py:325-355; needs the tuner to converge) */
    int32_t forward_first;   /* 1 = Dantzig / candidate-list keys rank every forward candidate above every backward one: the
                                reference's min-cost entering rule for assignment problems (specialized_pivots.py:191-223) followed
                                by its general pricing for what is left (simplex.py:1061-1064).  The row-scan rule it uses for
                                transportation problems (specialized_pivots.py:69-117) IS MCF_RULE_DANTZIG_FULL. */
    int32_t compressed_keys; /* Dantzig / candidate-list grid sweeps over 4-byte key codes (one per arc, kept exact next to the resident
                                reduced costs) instead of 8 B reduced cost + 1 B state: 0 = auto (full-sweep Dantzig handles from 4 M arcs on, where the
                                sweep is bandwidth-bound and read whole), 1 = on, -1 = off */
    int32_t vkey_half_log2;  /* test hook: log2 of the half width of a code level (0 = 28); small values force the exact-compare path */
    int32_t climb_depth;     /* cycle search: end points no deeper than this are climbed outright whatever cycle_scan says
                                (0 = auto: 3 up to 32 768 nodes, 8 above; -1 = never) */
    int32_t overlap_update;  /* captured graphs in which the pricing of pivot t+1 runs beside the tree permutation of pivot t (it needs
                                only the reduced-cost half of the update): 1 = on, 0 / -1 = off.  Same pivot sequence.  Measured
                                slower at every size (two cross-queue edges per pivot cost ~12 us on this stack;
                                profiles/r02_ab_overlapped_graph.txt), so auto never picks it: kept as an A/B switch. */
    int32_t key_mode;        /* key variant of the Dantzig / candidate-list sweep = the reference's specialised entering rules
                                (specialized_pivots.py:69-424, dispatch :452-527) as variants of the one kernel:
                                0 = plain violation (also the transportation row scan :69-117; forward_first = 1 selects 1),
                                1 = forward candidates first (assignment, :191-223),
                                2 = candidates flagged in arc_priority first (shortest path :338-424, bipartite matching :233-281),
                                3 = capacity x violation (max flow, :284-335).  Not with MCF_RULE_DEVEX_BLOCK. */
    const int8_t* arc_priority; /* key_mode 2: one byte per arc, caller's order; bit 0 = preferred as a forward candidate (flow rises
                                from the lower bound), bit 1 = preferred as a backward candidate.  Read during mcf_create only. */
    int32_t tree_blocks;     /* layout of the spanning tree's preorder: 0 = auto (blocked preorder list from 32 768 nodes on), -1 = dense
                                array, k in 2..10 = blocked list with blocks of 2^k slots.  The blocked list re-hangs a subtree in
                                O(subtree + block) element moves instead of shifting every position between its old and its new
                                place (replaces the per-pivot BFS rebuild basis.py:82-125 and _update_tree_sets simplex.py:1103-1107);
                                same logical preorder, same pivots.  Handles of the persistent loops (pricing_mode 2 / 3, mid_loop = 1)
                                keep the dense array. */
    int32_t tree_pool;       /* blocked list: spare blocks per arena (0 = auto: 1.5 x the dense count; -1 = none, so that every pivot
                                rewrites the whole list -- a test hook) */
    int32_t rc_drop;         /* resident reduced costs are given up in mid-solve -- pricing then gathers the potentials, as with no_rcache --
                                once the re-hung subtrees average more than this many nodes over a batch of pivots: from there on the
                                patch of the incident arcs' reduced costs costs more per pivot than the dearer sweeps.  0 = auto
                                (from 100 000 nodes on: candidate list 384, Devex 1 500; else never), -1 = never, k > 0 = that threshold (candidate
                                list and Devex; the Dantzig rule sweeps every arc on every pivot and never drops).  Same pivots. */
    int32_t pivot_run;       /* candidate-list handles on the blocked list: a list period is one sweep + this many pairs of (k_pivot_run:
                                pivots back to back in ONE workgroup, each followed by its update in place, until an update is too large
                                for one workgroup or the list is used up; k_update_bpl: that update on the grid).  0 = off (measured
                                slower than the pair shape at 1M/16M: 30.9-35.7 K against 41.1 K pivots/s; the
                                environment variable MCF_PIVOT_RUN=k turns it on for an A/B), k > 0 = k pairs (the handle goes back
                                to one k_pivot per slot once its run launches end after fewer than three pivots on average).
                                Same pivots in the same order. */
} mcf_options;

typedef struct mcf_stats {
    int64_t pivots;           /* FlowResult.iterations (degenerate pivots included) */
    int64_t degenerate;       /* theta == 0 */
    int64_t bound_flips;      /* leaving arc == entering arc */
    int64_t arcs_priced;      /* sum over pricing passes of the arcs the pass covers */
    int64_t nodes_moved;      /* preorder positions rewritten by the apply pass */
    int64_t subtree_nodes;    /* sum of re-hung subtree sizes */
    int64_t cycle_arcs;       /* sum of cycle lengths */
    int64_t batches;          /* host round trips */
    int64_t unbounded_arc;    /* entering arc when status is MCF_ST_UNBOUNDED, else -1 */
    int64_t unbounded_rc;     /* its reduced cost in the push direction (< 0) */
    double solve_seconds;     /* wall time inside mcf_solve */
    double price_ms;          /* with options.profile: summed kernel durations */
    double pivot_ms;
    double apply_ms;
    int64_t price_launches;
    int64_t pivot_launches;
    int64_t apply_launches;
    int64_t price_bytes;      /* compulsory bytes of one pricing launch of this handle's sweep kernel: 4 B/arc (k_price_v, key codes),
                                 9 B/arc (k_price_rc; 13 Devex), or SURVEY 8d's 13 B/arc + 8 B/node (k_price gather; 17 Devex) */
    int64_t artificial_flow;  /* flow still on artificial arcs (> 0 at optimality = infeasible, simplex.py:1573-1624);
                                 -1 when mcf_get_result was asked for neither status, objective nor flow */
    int64_t pricing_mode;     /* 0 = gather sweep (k_price), 1 = resident reduced costs (k_price_rc + k_rcupd),
                                 2 = fused LDS-resident pivot loop (k_solve_small),
                                 3 = persistent single-workgroup loop over global memory (k_solve_mid) */
    int64_t cycle_scans;      /* pivots whose cycle was completed by the position-space scan */
    int64_t scan_rounds;      /* chunk iterations of those scans */
    int64_t arcs_swept;       /* arcs whose reduced cost the grid sweeps actually read (<= arcs_priced with incremental pricing) */
    double loop_ms;           /* persistent pivot loops (pricing_mode 2, and 3 outside a graph): summed kernel durations, by HIP
                                 events on the engine's stream around every launch */
    int64_t loop_launches;    /* ... and the number of launches (one launch runs many pivots) */
    int64_t sweep_variant;    /* which grid sweep the handle launches: bit 0 = 4-byte key codes (k_price_v), bit 1 = non-temporal
                                 loads (k_price_v<.., true>), bit 2 = incremental (clean workgroups keep their candidate) */
    int64_t tree_blocks;      /* log2 of the block size of the blocked preorder list, 0 = dense preorder array */
    int64_t tree_rebuilds;    /* blocked list: pivots whose update rewrote the whole list densely (the block pool had run out) */
    int64_t rc_dropped_at;    /* pivot count at which the handle gave up its resident reduced costs (mcf_options.rc_drop), 0 = it has not */
    int64_t run_pairs;        /* (k_pivot_run, k_update_bpl) pairs per list period right now (mcf_options.pivot_run), 0 = one k_pivot per slot */
    int64_t run_left_at;      /* pivot count at which the handle went back to one k_pivot per slot, 0 = it has not */
} mcf_stats;

/* Called from mcf_solve every cb_interval pivots (simplex.py:1143-1154).
 * Return non-zero to stop the solve (status becomes MCF_ST_ITERATION_LIMIT). */
typedef int (*mcf_progress_cb)(void* user, int64_t pivots, int64_t max_pivots, double elapsed_seconds);

/* Fill *opt with defaults. */
void mcf_default_options(mcf_options* opt);

/* Build the device-resident problem: arc SoA, potentials, preorder spanning tree with the
 * all-artificial start basis.  n = real nodes (ids 0..n-1), m = arcs, lower bounds already
 * shifted out.  cap[i] < 0 or >= 2^60 means uncapacitated.  sum(supply) must be 0.
 * |cost| must fit int32 and m + n must stay below 2^30. */
int mcf_create(int32_t n, int64_t m, const int32_t* tail, const int32_t* head, const int64_t* cost,
               const int64_t* cap, const int64_t* supply, const mcf_options* opt, mcf_handle** out);

/* Pivot until optimal / unbounded / max_pivots more pivots were made (max_pivots < 0:
 * the reference's default budget max(100, 20 * (m + n)), simplex.py:1470). */
int mcf_solve(mcf_handle* h, int64_t max_pivots, mcf_progress_cb cb, void* user, int64_t cb_interval);

/* Solve `count` INDEPENDENT instances side by side: one persistent workgroup (one CU) per handle, each running its whole
 * solve as mcf_solve would (same pivot sequence, same budget rule; no progress callback), all in one launch per engine path.
 * The reference solves instances one after the other on one core (benchmarks/runners/run_benchmark.py; its published
 * per-instance figures are all for <= 4 096 nodes): a single such instance can only ever occupy one CU of 256, a batch
 * fills the chip.  Every handle must run as one persistent workgroup -- the fused LDS path (mcf_stats.pricing_mode == 2:
 * about <= 300 nodes / 2 500 arcs) or the persistent loop over global state (pricing_mode == 3; mcf_options.mid_loop = 1
 * asks for it at any size; a candidate-list loop then does its full sweeps itself) -- on the same device; max_pivots: one budget per handle (< 0: the
 * reference's default) or NULL for the default everywhere; kernel_ms (optional) <- duration of the launches.  Results per
 * handle through mcf_get_result as usual. */
int mcf_solve_batch(mcf_handle* const* handles, int32_t count, const int64_t* max_pivots, double* kernel_ms);

/* Copy the solution out.  Any pointer may be NULL.  objective_hi_lo[0..1] = high and low
 * 64 bits of the exact 128-bit sum(flow*cost); flow[m]; potential[n] (root excluded);
 * in_tree[m]. */
int mcf_get_result(mcf_handle* h, int32_t* status, int64_t* objective_hi_lo, int64_t* flow,
                   int64_t* potential, int8_t* in_tree, mcf_stats* stats);

/* One pricing pass over arcs [start, end) (caller's arc indices) with the current potentials,
 * without pivoting: the kernel-level parity hook.  Ties go to the lowest arc index.  *arc = -1 when no arc is eligible; *dir = +1 forward /
 * -1 backward; *key = violation |rc| (Dantzig) or the f64 merit's bit pattern (Devex). */
int mcf_price_once(mcf_handle* h, int32_t rule, int64_t start, int64_t end, int64_t* arc, int32_t* dir,
                   int64_t* key);

/* Back to the all-artificial start basis (flows, potentials, tree, counters). */
int mcf_reset(mcf_handle* h);

/* Warm start (NetworkSimplex._apply_warm_start_basis, simplex.py:740-903, and _recompute_tree_flows,
 * :905-1010): restart from the caller's basis instead of the all-artificial one.  in_tree[m] marks the basic
 * arcs (they must form a forest; every component gets one artificial arc to the root, as in the reference);
 * at_upper[m] (may be NULL) marks non-basic arcs sitting at their capacity rather than at zero (the reference
 * keeps no such information in a Basis and starts them at zero).  Tree flows are recomputed from conservation
 * with the handle's supplies / capacities.  Returns MCF_OK, or MCF_E_STATE when the basis cannot be used (cycle,
 * empty, flows outside the bounds): the handle is then at the cold start and can be solved as usual -- the
 * reference's fall-back (simplex.py:1527-1531).  Counters are reset either way. */
int mcf_set_basis(mcf_handle* h, const int8_t* in_tree, const int8_t* at_upper);

/* ---- arc-sharded multi-GPU pivoting: one handle per rank, every rank holds the full
 * replicated state and prices only its shard (options.shard_rank / shard_count).  Per pivot:
 *   mcf_enqueue_price   local best candidate -> cand_out (device, 2 x int64: key, packed arc id)
 *   <RCCL all-gather of the 16-byte candidates, by the caller, on the same stream>
 *   mcf_enqueue_pivot   every rank applies the same winning pivot to its replica
 * `stream` is a hipStream_t (0 = default stream).  Nothing here synchronises. */
int mcf_enqueue_price(mcf_handle* h, void* stream, int64_t* cand_out_dev);
int mcf_enqueue_pivot(mcf_handle* h, void* stream, const int64_t* cands_dev, int32_t ncand);
/* Candidate-list rule over several ranks -- the amortisation lever of SURVEY.md section 8e: ONE collective per
 * (minor_cap + 1) pivots instead of one per pivot.
 *   mcf_shard_info          list_len = candidates one sweep of this handle leaves (one per pricing workgroup);
 *                           minor_cap = pivots that may re-price a list before the next sweep (simplex_pricing.py:398-400)
 *   mcf_enqueue_price_list  sweep the shard; cands_out (device, list_len x {key, packed arc id}) <- its candidates
 *   <all-gather of the lists, by the caller>
 *   mcf_enqueue_pivots      `count` pivots on the gathered list: the first takes the sweep's keys, the others re-price
 *                           the listed arcs against the current potentials (minor iterations, simplex_pricing.py:419-456)
 * With shard_count > 1 a handle patches the resident reduced costs of its own shard only (1 / shard_count of the
 * update work); listed arcs of other shards are re-priced from the replicated potentials. */
int mcf_shard_info(mcf_handle* h, int32_t* list_len, int32_t* minor_cap);
int mcf_enqueue_price_list(mcf_handle* h, void* stream, int64_t* cands_out_dev);
int mcf_enqueue_pivots(mcf_handle* h, void* stream, const int64_t* cands_dev, int32_t ncand, int32_t count);
/* Read the control block (synchronises `stream`): status (MCF_ST_* or -1 = still running). */
int mcf_poll(mcf_handle* h, void* stream, int32_t* status_or_running, int64_t* pivots);
int mcf_set_max_pivots(mcf_handle* h, int64_t max_total_pivots);

/* ---- measurement helpers (bench.py) */
/* Launch the pricing kernel `reps` times back to back on the engine stream between two HIP
 * events; *ms_per_launch = average duration.  Read-only with respect to the solver state. */
int mcf_time_pricing(mcf_handle* h, int32_t rule, int32_t reps, double* ms_per_launch);
/* Device-to-device copy of `bytes` bytes, `reps` times, between two HIP events: the measured
 * HBM copy ceiling quoted beside the datasheet peak. */
int mcf_time_copy(int32_t device, int64_t bytes, int32_t reps, double* ms_per_copy);

/* ---- introspection for the parity tests: raw tree state, host copies.
 * parent[n+1], pred_arc[n+1] (-1 for the root), size[n+1], pos[n+1], order[n+1], state[m],
 * potential_with_root[n+1], depth[n+1], psize[n+1] (subtree size of the node at each preorder
 * position).  Any pointer may be NULL. */
int mcf_get_tree(mcf_handle* h, int32_t* parent, int32_t* pred_arc, int32_t* size, int32_t* pos,
                 int32_t* order, int8_t* state, int64_t* potential_with_root, int32_t* depth, int32_t* psize);

/* Reduced cost of every arc (caller's order) as the pricing kernel sees it: the resident copy
 * when the handle keeps one (*resident = 1), else cost + pi[tail] - pi[head] computed on the host.
 * Tests use it to check the invariant resident rc == cost + pi[tail] - pi[head]. */
int mcf_get_reduced_costs(mcf_handle* h, int64_t* rc_out, int32_t* resident);

/* The compressed Dantzig key of every arc (caller's order) as the sweep reads it; *present = 0 (and zeros) when the
 * handle keeps none.  Tests check the invariant key == code(-state * reduced cost) and decode it with the documented
 * scheme (csrc/mcf_core.h: mcf_vkey). */
int mcf_get_pricing_keys(mcf_handle* h, int32_t* keys_out, int32_t* present);

/* Devex reference weights of every arc (caller's order; 1.0 for a handle that never priced with the Devex rule).
 * With mcf_get_tree / mcf_get_result this is the full input of one block selection, so that a test can replay
 * mcf_price_once(MCF_RULE_DEVEX_BLOCK, ...) on the oracle's restated _select_entering_arc_vectorized. */
int mcf_get_weights(mcf_handle* h, float* weight_out);

/* ---- native DIMACS "p min" reader (host only; replaces benchmarks/parsers/dimacs.py:105-286 for
 * instances too large for one Python object per arc).  Two calls: mcf_dimacs_scan returns the
 * counts, mcf_dimacs_load fills caller-allocated arrays (0-based node ids; cap -1 = uncapacitated,
 * also for the reference's "-1" / "inf" / >= 1e15 conventions, dimacs.py:216-221; 4-field arc lines
 * mean lower = 0, dimacs.py:202-207).  Integer data only: a non-integral token is an error.
 * Return 0 or MCF_E_BAD_ARG with a message in err (may be NULL). */
int mcf_dimacs_scan(const char* path, int64_t* n_nodes, int64_t* n_arcs, char* err, int32_t err_len);
int mcf_dimacs_load(const char* path, int64_t n_nodes, int64_t n_arcs, int32_t* tail, int32_t* head,
                    int64_t* lower, int64_t* cap, int64_t* cost, int64_t* supply, char* err, int32_t err_len);

const char* mcf_last_error(mcf_handle* h); /* NULL handle: last create-time error of this thread */
void mcf_destroy(mcf_handle* h);
int mcf_abi_version(void);
/* Number of usable HIP devices (0 when none); never initialises a context. */
int mcf_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* MCF_H */
