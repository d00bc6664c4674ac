// mcf_core.h -- data layout and per-pivot logic of the MI355X network-simplex engine.
//
// Everything here is integer and deterministic.  The functions are
// __host__ __device__ so that (a) the HIP kernels in mcf_engine.hip and (b) the
// host-only emulation build used by the CPU test-suite (oracle/emul_engine.cpp,
// test infrastructure) execute the very same pivot logic.
//
// Reference path being replaced (file:line under /root/reference/src/network_solver/):
//   reduced cost / eligibility ....... simplex.py:498-512, simplex_pricing.py:110-137
//   cycle + ratio test ............... basis.py:178-241, simplex.py:1198-1229
//   flow update ...................... simplex.py:1255-1283
//   basis swap + tree/potential ...... simplex.py:1285-1425, basis.py:82-122
//
// Design differences from the reference (SURVEY.md section 8a "Native:" notes):
//   * int64 flows/potentials, single-phase big-M start instead of two float phases;
//   * a strongly feasible spanning tree (leaving-arc tie rule below) replaces the
//     1e-10 * 1.00001^idx cost perturbation as the anti-cycling device;
//   * the spanning tree is stored as a PREORDER ARRAY: order[pos] = node, and the
//     subtree of v is the contiguous slice order[pos[v] .. pos[v]+size[v]).  A basis
//     swap re-hangs one subtree; that becomes a block permutation of `order`
//     (apply_map below) which, like the potential update of the moved subtree, is a
//     data-parallel pass over a contiguous range -- no per-pivot BFS
//     (basis.py:89-122) and no sequential thread-list walk.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MCF_HD __host__ __device__ __forceinline__
#else
#define MCF_HD inline
#endif

#define MCF_INF ((int64_t)1 << 60)  // "uncapacitated" sentinel; also ratio-test infinity

// bit pattern of a double (positive doubles order like their bit patterns: merits travel as 64-bit integer keys)
#if defined(__HIP_DEVICE_COMPILE__)
#define mcf_double_bits(x) ((int64_t)__double_as_longlong(x))
#else
static inline int64_t mcf_double_bits(double x) { int64_t k; __builtin_memcpy(&k, &x, 8); return k; }
#endif

// Team primitives of the cooperative passes (mcf_pivot_scan): a workgroup on the device, a single
// "lane" in the host emulation build (where a barrier is nothing and an atomic is a plain update).
#if defined(__HIP_DEVICE_COMPILE__)
#define MCF_TEAM_BARRIER() __syncthreads()
#define MCF_ATOMIC_MIN64(p, x) atomicMin(reinterpret_cast<long long*>(p), (long long)(x))
#define MCF_ATOMIC_MIN32(p, x) atomicMin(reinterpret_cast<int*>(p), (int)(x))
#define MCF_ATOMIC_MAX32(p, x) atomicMax(reinterpret_cast<int*>(p), (int)(x))
#define MCF_ATOMIC_ADD32(p, x) atomicAdd(reinterpret_cast<int*>(p), (int)(x))
#define MCF_ATOMIC_MAX64(p, x) atomicMax(reinterpret_cast<long long*>(p), (long long)(x))
#define MCF_ATOMIC_OR32(p, x) atomicOr(reinterpret_cast<int*>(p), (int)(x))
#else
#define MCF_TEAM_BARRIER() ((void)0)
#define MCF_ATOMIC_MIN64(p, x) do { if ((int64_t)(x) < *(p)) *(p) = (int64_t)(x); } while (0)
#define MCF_ATOMIC_MIN32(p, x) do { if ((int32_t)(x) < *(p)) *(p) = (int32_t)(x); } while (0)
#define MCF_ATOMIC_MAX32(p, x) do { if ((int32_t)(x) > *(p)) *(p) = (int32_t)(x); } while (0)
static inline int32_t mcf_host_fetch_add32(int32_t* p, int32_t x) { const int32_t o = *p; *p = o + x; return o; }
#define MCF_ATOMIC_ADD32(p, x) mcf_host_fetch_add32((p), (x))
#define MCF_ATOMIC_MAX64(p, x) do { if ((int64_t)(x) > *(p)) *(p) = (int64_t)(x); } while (0)
#define MCF_ATOMIC_OR32(p, x) do { *(p) |= (int32_t)(x); } while (0)
#endif

enum McfStatus : int32_t {
    MCF_RUNNING = 0,
    MCF_OPTIMAL = 1,        // no eligible arc left
    MCF_PIVOT_LIMIT = 2,    // max_pivots reached
    MCF_UNBOUNDED = 3,      // ratio test found no blocking arc
    MCF_INTERNAL_ERROR = 4
};

// pricing rules (same numbering as MCF_RULE_* in include/mcf.h)
#define MCF_RULE_DANTZIG 0
#ifndef MCF_RULE_DEVEX_BLOCK
#define MCF_RULE_DEVEX_BLOCK 1
#endif
// Candidate list (simplex_pricing.py:375-542): a full Dantzig sweep leaves one candidate per
// pricing workgroup; the next `minor_cap` pivots only re-price that list ("minor iterations",
// simplex_pricing.py:419-456) before the next full sweep.  Optimality is only ever declared by a
// full sweep.
#ifndef MCF_RULE_CANDIDATE_LIST
#define MCF_RULE_CANDIDATE_LIST 2
#endif

#if defined(__HIPCC__)
// Wave-wide max of a signed 64-bit value with DPP moves (row-local butterflies, then the two row broadcasts of
// gfx9) instead of ds_bpermute shuffles: ~6 x (2 dpp movs + a 64-bit compare/select) of a few cycles each, versus
// twelve dependent trips through the LDS crossbar.  All 64 lanes of the wave must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int64_t mcf_dpp_max_step(int64_t x) {
    const int lo = (int)(uint32_t)x, hi = (int)(uint32_t)((uint64_t)x >> 32);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const int64_t o = (int64_t)(((uint64_t)(uint32_t)ohi << 32) | (uint32_t)olo);
    return o > x ? o : x;
}
__device__ __forceinline__ int64_t mcf_wave_max64(int64_t x) {
    x = mcf_dpp_max_step<0xB1, 0xf>(x);   // quad_perm [1,0,3,2]
    x = mcf_dpp_max_step<0x4E, 0xf>(x);   // quad_perm [2,3,0,1]
    x = mcf_dpp_max_step<0x141, 0xf>(x);  // row_half_mirror
    x = mcf_dpp_max_step<0x140, 0xf>(x);  // row_mirror: every lane of a row of 16 now holds the row's max
    x = mcf_dpp_max_step<0x142, 0xa>(x);  // row_bcast15 into rows 1 and 3
    x = mcf_dpp_max_step<0x143, 0xc>(x);  // row_bcast31 into rows 2 and 3: lane 63 holds the max of the wave
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), 63);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
#endif

// Diagnostic build (-DMCF_STAMPS): cycle stamps of the one-workgroup pivot kernel, accumulated per phase.
#if defined(MCF_STAMPS) && defined(__HIPCC__)
__shared__ unsigned long long mcf_stamp_acc[24];
__shared__ unsigned long long mcf_stamp_last;
#define MCF_PSTAMP(slot)                                                     \
    do {                                                                      \
        if (threadIdx.x == 0) {                                               \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
            mcf_stamp_acc[slot] += now_ - mcf_stamp_last;                     \
            mcf_stamp_last = now_;                                            \
        }                                                                     \
    } while (0)
#else
#define MCF_PSTAMP(slot) do {} while (0)
#endif

// One 16-byte record per node: a cycle walk needs exactly one load per step.
struct alignas(16) McfNode {
    int32_t parent;  // parent node (root: -1)
    int32_t pred;    // (tree arc to parent << 1) | up, up = 1 when tail[arc] == this node
    int32_t size;    // nodes in the subtree rooted here (>= 1)
    int32_t depth;   // arcs between this node and the root
};
// Preorder positions are NOT part of the record: the apply pass rewrites ~0.4 n of them per pivot,
// and keeping them out means the records the cycle walk chases are only written along the cycle
// and inside the re-hung subtree.  They live in McfView::posbuf (double buffered like order[]).

// Walk-side arc data, 16 bytes: residuals of a cycle arc come from one load.
struct alignas(16) McfArcW {
    int64_t cap;   // MCF_INF when uncapacitated
    int64_t flow;
};

// Entering-arc candidate: larger key wins, ties -> smaller `arc` word.
// `arc` packs (caller's original arc index << 32) | engine arc index, so ties fall to the
// lowest ORIGINAL index: the pivot sequence does not depend on the engine's internal arc
// layout (mcf_host.h permutes arcs into XCD head buckets).
struct alignas(16) McfCand {
    int64_t key;   // Dantzig: violation = -state*rc > 0;  Devex: bit pattern of the f64 merit;  <= 0: none
    int64_t arc;   // packed ids, -1 when none
};

// Candidate-list rule: what a minor iteration needs to know about a listed arc, kept current by the update pass of every
// pivot (rc changes by +-sigma iff exactly one end point is in the re-hung subtree; the state changes only for the entering
// and the leaving arc), so that re-pricing the list is ONE coalesced read of these records instead of two dependent
// random loads per listed arc -- and the winner's end points are known without looking the arc up.
struct alignas(16) McfCandX {
    int64_t rc;      // exact reduced cost under the current potentials
    int64_t arc;     // packed ids as in McfCand, -1: none
    int32_t tail, head;
    int32_t state;   // +1 / -1 / 0 (basic)
    int32_t pad;
};

#define MCF_NUM_BUCKETS 8  // one head-node bucket per XCD (8 XCDs, private 4 MiB L2 each)

// ---- Devex block search (simplex_pricing.py:310-357) with the reference's block-size tuner
// (simplex_adaptive.py:98-151) and its periodic weight reset (simplex.py:1370-1400, every 64 basis swaps).
// A block is a run of `block_granules` consecutive GRANULES; granule g is the g-th 1/64 of every head bucket
// (so a block spans all 8 XCDs and a block of 64 granules is the whole arc list).  The tuner moves the block
// size in granule units: x1.5 when more than 30 % of the last >= 50 pivots were degenerate, x0.75 below 10 %.
// Measured on the reference itself (oracle with the tuner / the reset switched off): both are essential --
// netgen_8_12a takes 10 113 pivots with both, 34 757 without the reset and does not finish without the tuner.
#define MCF_GRANULES 64
#define MCF_WLIST_CAP 1024       // arcs whose weight differs from 1 between two resets (overflow forces a reset)
#define MCF_DEVEX_RESET_SWAPS 64 // ft_update_limit (data.py / simplex.py:1370-1373)
#define MCF_TUNER_MAX_ARCS 16384  // largest block the tuner grows to (= what one persistent workgroup prices per pivot)
#define MCF_DIR_FLAG (1 << 30)   // Devex tie rule: bit 30 of the caller's arc index word marks a FORWARD candidate
struct McfDevex {
    int32_t gran[MCF_NUM_BUCKETS][MCF_GRANULES + 1];  // engine arc index where granule g of bucket x starts
    int32_t gtot[MCF_GRANULES + 1];                   // sum over the buckets of gran[x][g] (arcs before granule g; accounting)
    int32_t wlist[MCF_WLIST_CAP];                     // engine arcs whose weight was set since the last reset
};

// Segment of the block permutation that re-roots the moved subtree:
// new positions [dst, dst+len) take old positions [src, src+len).
struct McfSeg {
    int32_t dst, src, len;
    int32_t ddepth;  // what the depth of every node of the segment changes by
};

// Per-solve control block in device memory; the host only reads it between batches.
struct McfCtx {
    // ---- read by the host between batches
    int32_t status;
    int32_t unbounded_arc;
    int64_t pivots;            // completed pivots (degenerate ones included, as the reference counts them)
    int64_t degenerate;        // pivots with theta == 0
    int64_t bound_flips;       // pivots whose leaving arc is the entering arc (tree unchanged)
    int64_t max_pivots;
    int64_t arcs_priced;       // sum over pricing passes of the arcs the pass covers (clean blocks of an incremental sweep included)
    int64_t nodes_moved;       // sum of preorder positions rewritten (diagnostic)
    int64_t subtree_nodes;     // sum of |T2| (diagnostic)
    int64_t cycle_arcs;        // sum of cycle lengths (diagnostic)
    // ---- block-search state (Devex rule): block k = granules [k * block_granules, (k + 1) * block_granules) of every bucket
    int64_t block_size;        // nominal arcs per block (reporting only)
    int64_t block_index;       // block the next pricing pass scans
    int32_t empty_blocks;      // consecutive blocks without a candidate
    int32_t num_blocks;        // ceil(MCF_GRANULES / block_granules)
    int32_t block_granules;    // 1 .. MCF_GRANULES
    int32_t auto_tune;         // 1: the tuner below adapts block_granules (block_size "auto", simplex_adaptive.py:70-96)
    int32_t tn_total, tn_degenerate;  // pivots / degenerate pivots since the last adaptation (record_pivot)
    int64_t tn_last_adapt;     // pivot count at the last adaptation
    int32_t swaps_since_reset; // basis swaps since the Devex weights were last reset
    int32_t wlist_n;           // entries of McfDevex::wlist in use
    int32_t wreset;            // hand-over decide -> finish: reset the listed weights to 1
    int32_t devex_cyclic;      // 1: the block advances after every pivot; 0: stay on a block until it is empty (the reference)
    int32_t max_granules;      // the tuner never grows a block beyond this (bounds the arcs one pricing pass reads)
    int32_t limit_checked;     // status == MCF_PIVOT_LIMIT was confirmed on the device: an eligible arc is left
                               // (simplex.py:1678-1699 re-prices once at the budget to tell optimal from iteration_limit)
    // ---- candidate-list state
    int32_t minor_left;        // > 0: the next pass re-prices the candidate list instead of sweeping
    int32_t minor_cap;         // minor pivots allowed per full sweep
    int64_t minor_pivots;      // pivots taken from a re-priced list (diagnostic)
    int64_t major_sweeps;      // full sweeps (diagnostic)
    // ---- preorder double buffer
    int32_t cur;               // which order[] copy is current
    int32_t pending_flip;      // the last apply wrote order[cur^1]; flip before the next pivot
    int32_t prev_lo, prev_hi;  // range order[cur^1] is stale on
    // ---- descriptor of the tree update the apply pass has to perform
    int32_t apply;             // 0 = nothing, 1 = re-hang
    int32_t lo, hi;            // affected preorder range
    int32_t t2_old, t2_new, t2_size;  // old start a, new start b, |T2|
    int32_t nseg;              // entries in seg[] (sorted by dst, cover [t2_new, t2_new+t2_size))
    int32_t nchg;              // entries in chg[]
    int32_t pad2;
    int64_t sigma;             // potential shift of the moved subtree
    // ---- hand-over from mcf_pivot_walk (one lane) to mcf_pivot_finish (all lanes)
    int32_t stage;             // 0 = nothing to finish, 1 = bound flip, 2 = basis swap
    int32_t pv_e, pv_s;        // entering arc (engine index), its state
    int32_t pv_n1, pv_n2;      // recorded path lengths (first / second side)
    int32_t pv_result, pv_k;   // blocking side (1 / 2), stem index of q on that side
    int32_t pv_vin, pv_leave;  // new parent of the re-hung subtree, leaving arc
    int32_t pv_leave_state;    // state the leaving arc takes (+1 ends at 0 flow, -1 at capacity)
    int32_t pv_tail_in_t2;
    int32_t pv_vin_depth;      // depth of the new parent v_in
    int32_t pv_first, pv_second;  // end points of the entering arc in push order
    int64_t pv_rc;             // its exact reduced cost
    int64_t pv_cap;            // its capacity
    int64_t pv_flow;           // its flow before the pivot
    int64_t pv_delta;
    // ---- cycle search: round trips the pointer-chasing climb may take before the position-space scan
    // takes over (only when the view carries psz[]); diagnostics
    int32_t climb_budget;
    int32_t climb_depth;       // end points no deeper than this are climbed outright (at most that many round trips: cheaper
                               // than the scan's fixed passes while the tree is shallow, e.g. the first pivots of a cold start)
    int64_t scans;             // pivots whose cycle was completed by the scan
    int64_t scan_rounds;       // chunk iterations of those scans
    // ---- blocked preorder list (McfView::bmeta != nullptr; see "blocked preorder list" below)
    int32_t arena;             // which tok / psz / ext arena is current
    int32_t alloc_next;        // physical blocks of the current arena handed out so far (bump allocation; the scan covers [0, alloc_next))
    int32_t alloc_prev;        // blocks that existed before this pivot (the ones its update has to look at)
    int32_t alloc_lo;          // first block this pivot's update may fill: two copy blocks, then the blocks of the re-hung subtree
    int32_t rebuild;           // this pivot's update writes the whole list densely into the other arena (the pool ran out)
    int32_t t_ins;             // insertion point of the re-hung subtree in OLD logical coordinates
    int32_t t2_blk;            // the block that holds the first element of T2 (known to the pivot: the update starts on it at once)
    int32_t ins_blk;           // the block the insertion point may cut (the new parent's block), -1: to be found by the update
    int32_t pv_t2n;            // preview of T2 for the reduced-cost patch: 1 = T2 is the single node pv_t2node (its adjacency range
    int32_t pv_t2node;         //   rides along, fetched while the cycle was searched), 0 = the update finds T2's nodes itself
    int64_t pv_adj[4];         // adjacency ranges [beg, end) of the entering arc's end points: first, second
    int32_t dense_blocks;      // blocks a dense list needs: ceil(n_nodes / block)
    int32_t pv_dd0;            // depth change of the first piece of the re-rooted subtree (all of it when pv_k == 0: nseg == 1)
    int64_t rebuilds;          // (diagnostic)
};

// ---- blocked preorder list: per-block record and the encodings of loc[] / bext[] (see McfView::bmeta)
struct McfBlkMeta {
    int32_t base;   // logical position of the block's slot 0 (MCF_BLK_FREE: the block holds nothing)
    int32_t rrel;   // max over the live slots o of o + size[o]  (may be too high, never too low)
};
#define MCF_BLK_FREE 0x3fffffff          // "base" of a block that holds nothing: no position is ever >= it
#define MCF_LOC_ARENA (1 << 30)          // loc[node] = slot | arena bit
#define MCF_LOC_SLOT (MCF_LOC_ARENA - 1)
#define MCF_EXT_FLAG (1 << 24)           // bext: a subtree that starts in the block shrank, rrel is due for re-indexing
#define MCF_BLK_COPIES 2                 // copy blocks reserved per pivot (a block is cut at most at the hole T2 leaves and at the insertion point)
#if defined(MCF_NO_BPL)   // A/B build: what do the blocked list's branches cost the dense paths?
#define MCF_HAS_BPL(v) false
#else
#define MCF_HAS_BPL(v) ((v).bmeta[0] != nullptr)
#endif

// Raw views the core functions operate on (device pointers in the kernels,
// host pointers in the emulation build).
struct McfView {
    int32_t n_nodes;        // including the artificial root (= n_nodes - 1)
    int64_t m;              // real arcs; artificial arc of node v is m + v
    const int32_t* tail;    // [m_pad]  arcs in ENGINE order: bucketed by head range, tail-sorted inside
    const int32_t* head;    // [m_pad]
    const int32_t* cost;    // [m_pad]
    const int32_t* orig;    // [m_pad]  engine arc index -> caller's arc index
    int64_t bucket_off[MCF_NUM_BUCKETS + 1];  // engine arcs of bucket x: [bucket_off[x], bucket_off[x+1])
    int8_t* state;          // [m_pad]  +1 at lower bound, -1 at upper bound, 0 basic / padding
    float* weight;          // [m_pad]  Devex reference weights (nullptr for Dantzig)
    McfDevex* dx;           // Devex granule table + list of touched weights (nullptr for the other rules)
    McfArcW* arcw;          // [m + n_nodes - 1]
    int64_t* pi;            // [n_nodes]
    McfNode* node;          // [n_nodes]
    int32_t* order[2];      // [n_nodes] each
    int32_t* path1;         // [n_nodes] scratch: nodes on the `first` side of the cycle
    int32_t* path2;         // [n_nodes] scratch: nodes on the `second` side
    McfNode* rec1;          // [n_nodes] scratch: their node records as read during the walk
    McfNode* rec2;          // [n_nodes]
    int32_t* ppos1;         // [n_nodes] scratch: their preorder positions (old view), so that the finish pass
    int32_t* ppos2;         // [n_nodes]   needs no dependent position look-ups
    McfSeg* seg;            // [2*n_nodes + 2] scratch
    McfCtx* ctx;
    // ---- resident reduced costs (large instances; nullptr = price by gathering potentials)
    int32_t key_mode;       // MCF_KEY_*: how the Dantzig / candidate-list key is formed from an eligible arc's violation (the
                            // reference's specialised entering rules as variants of the one sweep, specialized_pivots.py:69-424)
    int32_t rc_partial;     // 1: a sharded handle keeps rcache exact for ITS OWN shard only (the patch walks a per-rank
                            // adjacency: 1/G of the work); single arcs outside the sweep are then priced from the potentials
    int64_t* rcache;        // [m_pad] rc of every arc under the current potentials, engine order
    // Compressed Dantzig keys: vkey[e] = mcf_vkey(-state[e] * rcache[e]) -- 0 for an ineligible arc, otherwise a
    // 31-bit code that orders like the violation (equal codes <=> equal violations), or MCF_VKEY_SAT when the
    // violation does not fit: then the sweep looks the exact value up.  A Dantzig / candidate-list sweep reads these
    // 4 bytes per arc instead of 8 B reduced cost + 1 B state.  Kept exact by the same passes that keep rcache exact.
    int32_t* vkey;          // [m_pad] or nullptr
    const int8_t* prio;     // [m_pad] MCF_KEY_PRIORITY: bit 0 = the arc is preferred as a forward candidate, bit 1 = as a backward one
    int64_t vk_bigm;        // big-M of the instance (level spacing of the code)
    int32_t vk_half;        // half width of a level: violations within +-vk_half of a multiple of big-M are coded exactly
    int32_t blk_shift;      // blocked preorder list: log2(slots per physical block), 6 .. 10 (0 with the dense array)
    const int64_t* adj_off; // [n_nodes] CSR over real nodes: entries of node u are adj[adj_off[u] .. adj_off[u+1])
    const int64_t* adj;     // [2m] (other end point << 32) | (engine arc << 1) | (1 when u is the arc's tail)
    // preorder position of every node, double buffered with the same flip as order[]: while the apply
    // pass writes posbuf[cur ^ 1], posbuf[cur] stays the OLD, stable view (the walk's descriptor and the
    // reduced-cost update's membership test read it)
    int32_t* posbuf[2];     // [n_nodes] each
    // subtree sizes in POSITION space (psz[cur][pos[v]] == size[v]), double buffered like order[], or
    // nullptr.  With it, "the node at position i is an ancestor of u" is i <= pos[u] < i + psz[i]: the
    // cycle can be found by a coalesced team-wide scan over preorder positions instead of a pointer
    // chase whose length is the tree depth (mcf_pivot_scan).
    int32_t* psz[2];        // [n_nodes rounded up to a coarse block + 4] each
    // Coarse index over psz: reach[b] >= max over the positions i of block b of i + psz[i] (the end of the farthest-
    // reaching subtree that starts in the block).  Preorder intervals are nested, so block b holds an ancestor of the
    // node at position p exactly when b * 64 <= p < reach[b]: one pass over n / 64 words finds every block that can
    // matter, whatever the depth of the tree, and only those are scanned position by position.  An entry may be stale
    // on the HIGH side (a shrunken subtree not yet re-indexed): that only costs a wasted block visit.
    int32_t* reach;         // [ceil(n_nodes / 64)] or nullptr
    int32_t* chg;           // [n_nodes] scratch: positions whose subtree shrank in this pivot (their blocks are re-indexed by the apply pass)
    // ---- blocked preorder list (large trees; nullptr = the dense preorder array above).  The logical preorder is the
    // same, but it lives in physical blocks of 1 << blk_shift slots with a logical base each, so that re-hanging a
    // subtree costs O(|T2| + block + blocks / lane) instead of a shift of everything between its old and new place:
    //   order[a], psz[a]   become the two ARENAS of slots (node id / subtree size per slot, size 0 = empty slot);
    //   posbuf[0]          becomes loc[node] = slot | arena << 30;  posbuf[1] is unused;
    //   bmeta[k][b]        {base, rrel} of block b: the slot at offset o holds logical position base + o, and
    //                      base + rrel >= the end of every subtree that starts in the block (the coarse index);
    //                      double buffered with ctx.cur like the dense arrays (the update reads [cur], writes [cur ^ 1]);
    //   bext[a][b]         live slot range of block b in arena a: beg | end << 12 (| MCF_EXT_FLAG: re-index rrel).
    struct McfBlkMeta* bmeta[2];
    int32_t* bext[2];
    int32_t blk_cap;        // physical blocks per arena
    int32_t ncandx;         // entries of candx[] (= pricing workgroups)
    McfCandX* candx;        // candidate-list rule, single GPU: the live list's records (nullptr: minor iterations look the arcs up)
    // ---- incremental pricing (nullptr = every sweep prices every block).  A pricing workgroup's best candidate
    // only changes when an arc of its block changes reduced cost or state; the passes that change arcs raise the
    // block's flag, and a full Dantzig sweep (also the candidate-list rule's) skips the blocks whose flag is down.
    // (one pointer only: the view travels in scalar registers, and 136 more bytes of it cost every kernel ~1.3 us)
    struct McfDirty* dirty;
    const struct McfDirty* dirty_hdr;   // a copy of *dirty's header (nlb, lo, hi) that is cheaper to read (LDS in the pivot kernel), or nullptr
};

#define MCF_MAX_PRICE_BLOCKS 2048
struct McfDirty {
    int32_t nlb;                      // pricing workgroups per bucket
    int32_t lo[MCF_NUM_BUCKETS];      // first engine arc of this rank's share of bucket x (full-sweep slicing)
    int32_t hi[MCF_NUM_BUCKETS];      // one past its last
    int32_t pad[3];
    int32_t flag[MCF_MAX_PRICE_BLOCKS];  // != 0: the workgroup has to sweep its block again
};

// ---- blocked preorder list: small accessors
MCF_HD int32_t mcf_ext_beg(int32_t x) { return x & 0xfff; }
MCF_HD int32_t mcf_ext_end(int32_t x) { return (x >> 12) & 0xfff; }
MCF_HD int32_t mcf_ext_make(int32_t beg, int32_t end) { return beg | (end << 12); }

// Preorder position of `node` under the current view, and the slot that holds it (dense array: the position itself).
MCF_HD int32_t mcf_node_pos(const McfView& v, const McfCtx* c, int32_t node, int32_t* slot) {
    if (MCF_HAS_BPL(v)) {
        const int32_t s = v.posbuf[0][node] & MCF_LOC_SLOT;
        const McfBlkMeta* bm = c->cur ? v.bmeta[1] : v.bmeta[0];
        *slot = s;
        return bm[s >> v.blk_shift].base + (s & ((1 << v.blk_shift) - 1));
    }
    const int32_t p = (c->cur ? v.posbuf[1] : v.posbuf[0])[node];
    *slot = p;
    return p;
}
// ... the slot alone (one load; the blocked list's second, dependent one -- the block's base -- is left to the caller)
// (if / else, not a nested select of view members: that once made the compiler index the view in private memory)
MCF_HD int32_t mcf_node_slot(const McfView& v, const McfCtx* c, int32_t node) {
    if (MCF_HAS_BPL(v)) return v.posbuf[0][node] & MCF_LOC_SLOT;
    return (c->cur ? v.posbuf[1] : v.posbuf[0])[node];
}
// the slot arrays of the current view: node id per slot, subtree size per slot
MCF_HD const int32_t* mcf_slot_nodes(const McfView& v, const McfCtx* c) {
    const int32_t sel = MCF_HAS_BPL(v) ? c->arena : c->cur;
    return sel ? v.order[1] : v.order[0];
}
MCF_HD int32_t* mcf_slot_sizes(const McfView& v, const McfCtx* c) {
    const int32_t sel = MCF_HAS_BPL(v) ? c->arena : c->cur;
    return sel ? v.psz[1] : v.psz[0];
}

// ---- compressed Dantzig keys (McfView::vkey).  Violations cluster around 0, big-M and 2 big-M (an end point that
// still hangs on its artificial arc carries a potential of +-big-M), so:
//   big-M < 2^29:   every violation is < 2^31 -> the code is the violation itself;
//   otherwise:      level j = round(viol / big-M) in {0, 1, 2} and offset d = viol - j * big-M with |d| < half:
//                   code = j * 2^29 + (d + 2^28)   (levels cannot overlap because big-M >= 2^29 > 2 * half);
//   anything else:  MCF_VKEY_SAT = "not coded" (NOT "larger than every code": a violation between two levels is
//                   saturated too): the sweep fetches the exact reduced cost of such an arc and compares exact values.
#define MCF_VKEY_SAT 0x7fffffff
MCF_HD int32_t mcf_vkey(int64_t viol, int64_t bigm, int32_t half) {
    if (viol <= 0) return 0;
    if (bigm < ((int64_t)1 << 29) && half >= (1 << 28)) return viol < MCF_VKEY_SAT ? (int32_t)viol : MCF_VKEY_SAT;
    int32_t j;
    int64_t d;
    if (2 * viol < bigm) { j = 0; d = viol; if (d >= half) return MCF_VKEY_SAT; }   // level 0 holds [1, half)
    else if (2 * viol < 3 * bigm) { j = 1; d = viol - bigm; }
    else if (2 * viol < 5 * bigm) { j = 2; d = viol - 2 * bigm; }
    else return MCF_VKEY_SAT;
    if (d >= half || d <= -half) return MCF_VKEY_SAT;
    return (int32_t)(((int64_t)j << 29) + d + ((int64_t)1 << 28));
}

// the violation a (non-saturated, non-zero) code stands for
MCF_HD int64_t mcf_vkey_decode(int32_t code, int64_t bigm, int32_t half) {
    if (bigm < ((int64_t)1 << 29) && half >= (1 << 28)) return code;
    const int64_t j = code >> 29, d = (int64_t)(code & ((1 << 29) - 1)) - ((int64_t)1 << 28);
    return j * bigm + d;
}

// Dantzig key of an eligible arc (engine index i, violation viol > 0, state +-1).  The reference's specialised entering rules
// (specialized_pivots.py) are all "scan every arc, keep the best by some merit, else fall back to the general rule"; here
// each is a way of forming the key, so one sweep serves them all:
//   MCF_KEY_PLAIN          the violation (Dantzig; the row scan for transportation problems, :69-117, is exactly this)
//   MCF_KEY_FORWARD_FIRST  every forward candidate outranks every backward one, the violation orders each group (min-cost
//                          rule for assignment problems: forward arcs only, the rest left to the pricing strategy, :191-223)
//   MCF_KEY_PRIORITY       candidates flagged in prio[] (bit 0 as forward, bit 1 as backward candidate) outrank the others
//                          (shortest path, :338-424: forward arcs whose tail the source reaches, and every backward arc;
//                          bipartite matching, :233-281: arcs leaving the unit-supply side)
//   MCF_KEY_CAPACITY       merit = capacity x violation as a double (max flow, :284-335: residual x |reduced cost|; an
//                          eligible non-basic arc's residual is its capacity); uncapacitated arcs: +inf, as in the reference
// Bit 61 marks the preferred group: violations stay below 2^46.  Positive doubles order like their bit patterns.
#define MCF_FWD_BIT ((int64_t)1 << 61)
#define MCF_KEY_PLAIN 0
#define MCF_KEY_FORWARD_FIRST 1
#define MCF_KEY_PRIORITY 2
#define MCF_KEY_CAPACITY 3
MCF_HD int64_t mcf_dantzig_key(const McfView& v, int64_t i, int64_t viol, int32_t state) {
    switch (v.key_mode) {
        case MCF_KEY_FORWARD_FIRST: return state > 0 ? (viol | MCF_FWD_BIT) : viol;
        case MCF_KEY_PRIORITY: return (v.prio[i] & (state > 0 ? 1 : 2)) ? (viol | MCF_FWD_BIT) : viol;
        case MCF_KEY_CAPACITY: {
            const int64_t cap = v.arcw[i].cap;
            const double merit = cap >= MCF_INF ? __builtin_huge_val() : (double)cap * (double)viol;
            return mcf_double_bits(merit);
        }
        default: return viol;
    }
}

MCF_HD bool mcf_cand_better(int64_t key, int64_t arc, int64_t bkey, int64_t barc) {
    return key > bkey || (key == bkey && key > 0 && arc < barc);
}

MCF_HD int64_t mcf_pack_arc(int32_t orig, int64_t engine_idx) { return ((int64_t)orig << 32) | engine_idx; }

// The pricing workgroup that sweeps engine arc e (k_price_rc / k_price: workgroup lb * 8 + x takes the groups of
// four arcs  g_lo(x) + lb * 256 + lane + j * nlb * 256  of bucket x), or -1 when e is another rank's arc.
MCF_HD int32_t mcf_price_block_of(const McfView& v, const McfDirty* d, int64_t e) {
    int32_t x = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 1; k < MCF_NUM_BUCKETS; ++k) x += e >= v.bucket_off[k] ? 1 : 0;
    const int32_t plo = d->lo[x], phi = d->hi[x];
    if (e < plo || e >= phi) return -1;
    const int32_t q = (int32_t)(e >> 2) - (plo >> 2);
    return ((q >> 8) % d->nlb) * MCF_NUM_BUCKETS + x;
}
MCF_HD void mcf_mark_dirty(const McfView& v, int64_t e) {
    McfDirty* d = v.dirty;
    if (!d || e >= v.m) return;
    const int32_t b = mcf_price_block_of(v, v.dirty_hdr ? v.dirty_hdr : d, e);
    if (b >= 0) d->flag[b] = 1;
}

// The arcs one pricing pass of one rank looks at inside bucket x: the Devex block k of nb
// (a Devex "block" is slice k of every bucket, so a block search spans all 8 XCDs), and of
// that block the rank's share r of G (every GPU keeps all its XCDs busy).  Block first, shard
// second: the union over ranks is the same arc set for any number of GPUs, so the pivot
// sequence does not depend on the GPU count.
MCF_HD void mcf_bucket_slice(const int64_t* bucket_off, int x, int64_t r, int64_t G, int64_t k, int64_t nb,
                             int64_t* lo, int64_t* hi) {
    const int64_t s = bucket_off[x], len = bucket_off[x + 1] - s;
    if (G == 1 && nb == 1) { *lo = s; *hi = s + len; return; }  // the whole bucket: no (emulated 64-bit) divisions
    const int64_t a = s + len * k / nb, b = s + len * (k + 1) / nb;
    const int64_t len2 = b - a;
    *lo = a + len2 * r / G;
    *hi = a + len2 * (r + 1) / G;
}

// Devex: the arcs of block `k` (of `bg` granules each) inside bucket x, and of those the rank's share r of G.
MCF_HD void mcf_devex_slice(const McfDevex* dx, int x, int64_t r, int64_t G, int32_t k, int32_t bg, int64_t* lo, int64_t* hi) {
    int32_t g0 = k * bg;
    if (g0 >= MCF_GRANULES) g0 = 0;  // (simplex_pricing.py:329-331: wrap to the first block)
    const int32_t g1 = g0 + bg < MCF_GRANULES ? g0 + bg : MCF_GRANULES;
    const int64_t a = dx->gran[x][g0], b = dx->gran[x][g1];
    if (G == 1) { *lo = a; *hi = b; return; }
    const int64_t len = b - a;
    *lo = a + len * r / G;
    *hi = a + len * (r + 1) / G;
}
// host-side fill of the granule table
MCF_HD void mcf_devex_fill_granules(McfDevex* dx, const int64_t* bucket_off) {
    for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
        const int64_t s0 = bucket_off[x], len = bucket_off[x + 1] - s0;
        for (int g = 0; g <= MCF_GRANULES; ++g) dx->gran[x][g] = (int32_t)(s0 + len * g / MCF_GRANULES);
    }
    for (int g = 0; g <= MCF_GRANULES; ++g) {
        int64_t t = 0;
        for (int x = 0; x < MCF_NUM_BUCKETS; ++x) t += dx->gran[x][g] - bucket_off[x];
        dx->gtot[g] = (int32_t)t;
    }
}
// arcs of block k (all buckets, all shards)
MCF_HD int64_t mcf_devex_block_arcs(const McfDevex* dx, int32_t k, int32_t bg) {
    int32_t g0 = k * bg;
    if (g0 >= MCF_GRANULES) g0 = 0;
    const int32_t g1 = g0 + bg < MCF_GRANULES ? g0 + bg : MCF_GRANULES;
    return (int64_t)dx->gtot[g1] - dx->gtot[g0];
}
// block size in granules for a nominal block of `bs` arcs out of m
MCF_HD int32_t mcf_devex_granules_for(int64_t bs, int64_t m) {
    if (m <= 0) return MCF_GRANULES;
    int64_t g = (bs * MCF_GRANULES + m / 2) / m;
    if (g < 1) g = 1;
    if (g > MCF_GRANULES) g = MCF_GRANULES;
    return (int32_t)g;
}
// Block-search state of a fresh solve: the reference's initial block size m/4, m/8, m/16 (simplex_adaptive.py:89-96)
// in granule units with the tuner on, or the caller's fixed block size with the tuner off (data.py: block_size int).
MCF_HD void mcf_init_block_state(McfCtx* c, int32_t rule, int64_t m, int64_t block_size) {
    c->auto_tune = 0;
    c->block_granules = MCF_GRANULES;
    c->num_blocks = 1;
    c->block_size = m > 0 ? m : 1;
    c->block_index = 0;
    c->empty_blocks = 0;
    c->tn_total = 0; c->tn_degenerate = 0; c->tn_last_adapt = 0;
    c->swaps_since_reset = 0; c->wlist_n = 0; c->wreset = 0;
    c->devex_cyclic = 1;
    c->max_granules = MCF_GRANULES;
    if (rule != MCF_RULE_DEVEX_BLOCK) return;
    int64_t bs = block_size;
    if (bs <= 0) {
        c->auto_tune = 1;
        bs = m < 1000 ? m / 4 : (m < 10000 ? m / 8 : m / 16);
    }
    if (bs < 1) bs = 1;
    c->block_granules = mcf_devex_granules_for(bs, m);
    c->num_blocks = (MCF_GRANULES + c->block_granules - 1) / c->block_granules;
    c->block_size = m * c->block_granules / MCF_GRANULES > 0 ? m * c->block_granules / MCF_GRANULES : 1;
    // the tuner may grow a block up to MCF_TUNER_MAX_ARCS arcs (never below the initial size): beyond that a
    // pricing pass stops being a latency-bound hop and starts to cost bandwidth, while the pivot count no longer
    // improves (measured: netgen_8_14a 41 826 pivots with whole-list blocks, 42 571 with m/16 blocks)
    int32_t cap = m > 0 ? (int32_t)((int64_t)MCF_TUNER_MAX_ARCS * MCF_GRANULES / m) : MCF_GRANULES;
    if (cap > MCF_GRANULES) cap = MCF_GRANULES;
    c->max_granules = cap > c->block_granules ? cap : c->block_granules;
}

// Devex candidates carry their direction in the id word: among equal merits the reference's vectorised selection
// prefers a BACKWARD arc (forward wins only when strictly greater, simplex.py:596), then the lowest index
// (np.argmax).  state > 0 = forward.
MCF_HD int32_t mcf_devex_tie_id(int32_t orig, int32_t state) { return state > 0 ? (orig | MCF_DIR_FLAG) : orig; }

// Reduced cost of arc i under the current potentials (simplex.py:508-512):
// rc = cost + pi[tail] - pi[head].  An arc is eligible when state * rc < 0
// (simplex_pricing.py:124-131 with residuals expressed through `state`).
MCF_HD int64_t mcf_violation(const McfView& v, int64_t i) {
    const int64_t rc = (int64_t)v.cost[i] + v.pi[v.tail[i]] - v.pi[v.head[i]];
    return -(int64_t)v.state[i] * rc;
}

// Candidate-list minor iteration: current violation of a listed arc (packed id), 0 when it is no
// longer eligible.  Reads the resident reduced cost when there is one.
MCF_HD int64_t mcf_minor_key(const McfView& v, int64_t packed_arc) {
    if (packed_arc < 0) return 0;
    const int64_t e = packed_arc & 0xffffffff;
    const int64_t s = v.state[e];
    if (s == 0) return 0;
    const int64_t rc = (v.rcache && !v.rc_partial) ? v.rcache[e] : (int64_t)v.cost[e] + v.pi[v.tail[e]] - v.pi[v.head[e]];
    const int64_t viol = -s * rc;
    return viol > 0 ? mcf_dantzig_key(v, e, viol, (int32_t)s) : 0;
}

// ---------------------------------------------------------------------------
// One pivot = mcf_pivot_walk (ONE lane) + mcf_pivot_finish (ALL lanes of the workgroup,
// after a barrier).
//
// walk:   the only inherently sequential piece -- O(cycle length) dependent loads: join
//         search by subtree size, ratio test with the strongly-feasible tie rule.  It records
//         the path nodes and their records, takes every scalar decision, and leaves a
//         descriptor in the control block (stage / pv_* and the apply descriptor).
// finish: everything that touches arrays -- flow augmentation, state swap, subtree-size
//         bookkeeping, stem re-parenting, the segments of the block permutation.  Each path
//         element has a closed form from the recorded records, so lanes just stride over them.
// ---------------------------------------------------------------------------
// Where a cycle is recorded: node ids, their records as read, their preorder positions, per side.
struct McfPaths {
    int32_t *path1, *path2;
    McfNode *rec1, *rec2;
    int32_t *ppos1, *ppos2;  // (logical) preorder positions
    int64_t *flow1, *flow2;  // the arcs' flows as read by the hit pass, or null (then the finish pass re-reads them)
    int32_t *slot1, *slot2;  // the slots that hold them (dense array: the very same arrays as ppos1 / ppos2)
};
MCF_HD McfPaths mcf_view_paths(const McfView& v) {
    // blocked list: the slot scratch sits behind the position scratch (ppos arrays of 2 * n_nodes entries)
    const int32_t off = MCF_HAS_BPL(v) ? v.n_nodes : 0;
    return McfPaths{v.path1, v.path2, v.rec1, v.rec2, v.ppos1, v.ppos2, nullptr, nullptr, v.ppos1 + off, v.ppos2 + off};
}

// One-sided ancestor noted by the scan rounds: (preorder position << 1) | side.  (Fetching the node record right
// in the round was measured and lost: the wave that finds a hit stalls on two dependent loads per hit while the
// other waves wait for it at the round's barrier -- rounds 5.3 K -> 11.1 K ticks on netgen_8_14a.)
typedef int32_t McfHit;   // (carrying the node id along -- read together with the sizes -- was tried in round 3: it spares the hit pass a
                          //  dependent load but doubles the bytes of the fine pass, which is bound by one CU's memory pipeline: net slower)

// State of the cycle search, shared by the climb (one lane) and the scan (the whole team).
struct McfCycle {
    int64_t d1, d2;        // smallest residual so far on the first / second side (MCF_INF: none)
    int32_t k1, k2;        // index of that blocking node in path1 / path2 (-1: none)
    int32_t n1, n2;        // path elements recorded so far
    int32_t u, w;          // where the two climbs stand; u == w: that node is the join
    int32_t pu, pw;        // their preorder positions
    int32_t su, sw;        // ... and the slots that hold them
    McfNode ru, rw;        // their records
    int32_t p0u, p0w;      // positions and records of the entering arc's end points (first / second)
    int32_t s0u, s0w;      // ... and their slots
    McfNode r0u, r0w;
    int32_t small;         // the scan recorded the cycle in the small (LDS) path buffers, not in v.path*/rec*/ppos*
};

// Accumulators of the team-wide scan (LDS on the device).
#define MCF_REACH_SHIFT 6                  // coarse blocks of 64 preorder positions
#define MCF_REACH_BLOCK (1 << MCF_REACH_SHIFT)
#define MCF_SCAN_BLK_CAP 2048             // flagged coarse blocks one scan can hold (more: the plain rounds take over)
struct McfScanAcc {
    int64_t jpos[2];       // deepest common ancestor met so far: (preorder position << 32) | slot, -1: none yet
                           // (one entry per round parity: a round needs a single barrier)
    int32_t nhits;         // one-sided ancestors noted so far
    int32_t nblk;          // coarse pass: blocks that may hold an ancestor of either end point
    int32_t blk[MCF_SCAN_BLK_CAP];
    int32_t blkbase[MCF_SCAN_BLK_CAP];   // ... and the logical position of their slot 0
    int32_t jnode;         // the join and its record (fetched while the hit pass runs)
    McfNode join;
    int64_t wr1[16], wr2[16];  // per-wave results of the ratio reduction
    int32_t wi1[16], wi2[16];
};

// Step 1 (one lane): bookkeeping + the entering arc.  Returns false when there is nothing to pivot on
// in this slot (limit reached, no candidate, ...); the control block then already says why.
// `wx`: the winner's candidate record when the caller has it (candidate cache): state, end points and reduced cost then
// need no look-up, and the arc's capacity / flow load does not depend on anything else the cycle search fetches.
// `cy` != null: the records and slots of the arc's end points are fetched here too -- together with the arc's capacity /
// flow, before anything is stored -- and *cy is left as mcf_cycle_init would leave it (the caller then skips that call).
// (a compile-time switch, not a null test: with a run-time one the two node records took a detour through private memory)
template <bool WITH_CY>
MCF_HD bool mcf_pivot_begin_t(const McfView& v, int64_t best_key, int64_t best_arc, int32_t rule, const McfCandX* wx, McfCycle* cy) {
    McfCtx* c = v.ctx;
    c->apply = 0;
    c->stage = 0;
    c->nchg = 0;
    if (c->wreset) { c->wreset = 0; c->wlist_n = 0; }  // the previous finish pass reset the listed Devex weights
    if (c->pending_flip) {  // the previous apply pass wrote order[cur ^ 1]
        c->cur ^= 1;
        c->prev_lo = c->lo;
        c->prev_hi = c->hi;
        c->pending_flip = 0;
        if (c->rebuild) { c->arena ^= 1; c->rebuild = 0; }  // ... and, blocked list, the whole list into the other arena
    }
    if (c->pivots >= c->max_pivots) { c->status = MCF_PIVOT_LIMIT; return false; }

    const bool minor = rule == MCF_RULE_CANDIDATE_LIST && c->minor_left > 0;
    if (rule == MCF_RULE_CANDIDATE_LIST && !minor) c->major_sweeps += 1;
    if (best_arc < 0 || best_key <= 0) {
        if (minor) {
            // the list is exhausted, which says nothing about optimality: sweep again next pass
            c->minor_left = 0;
        } else if (rule == MCF_RULE_DEVEX_BLOCK) {
            // this block held no eligible arc: move on (simplex_pricing.py:325-355)
            c->empty_blocks += 1;
            c->block_index += 1;
            if (c->block_index >= c->num_blocks) c->block_index = 0;
            if (c->empty_blocks >= c->num_blocks) c->status = MCF_OPTIMAL;
        } else {
            c->status = MCF_OPTIMAL;
        }
        return false;
    }
    c->empty_blocks = 0;
    if (rule == MCF_RULE_CANDIDATE_LIST) {
        if (minor) { c->minor_left -= 1; c->minor_pivots += 1; }
        else c->minor_left = c->minor_cap;
    }
    if (rule == MCF_RULE_DEVEX_BLOCK && c->devex_cyclic) {
        // cyclic variant: the next pass looks at the NEXT block whether or not this one had a candidate
        // (the reference stays on a block until it is exhausted, simplex_pricing.py:325-355)
        c->block_index += 1;
        if (c->block_index >= c->num_blocks) c->block_index = 0;
    }

    if (rule == MCF_RULE_DEVEX_BLOCK && c->block_index >= c->num_blocks) c->block_index = 0;  // (a tuner step may have shrunk num_blocks)
    const int32_t e = (int32_t)(best_arc & 0xffffffff);  // engine index (low word of the packed id)
    int32_t s, first, second;
    int64_t rc;
    // ---- everything that needs only e, requested together: one round trip
    const McfArcW ae = v.arcw[e];  // capacity for the ratio test, flow for the store-only update
    if (wx) {
        s = wx->state;
        first = s > 0 ? wx->tail : wx->head;
        second = s > 0 ? wx->head : wx->tail;
        rc = wx->rc;
    } else {
        // both end points are fetched whatever the state says (a select between two LOADED values, not a load from a selected
        // address, which would have to wait for the state); so is the resident reduced cost
        const bool resident = v.rcache && !v.rc_partial;
        const int32_t te = v.tail[e], he = v.head[e];
        s = v.state[e];  // +1: flow rises from 0; -1: flow falls from cap
        int64_t rcv = 0;
        if (resident) rcv = v.rcache[e];
        first = s > 0 ? te : he;
        second = s > 0 ? he : te;
        // exact reduced cost of the entering arc (the Devex key is a merit, not a violation)
        rc = resident ? rcv : (int64_t)v.cost[e] + v.pi[te] - v.pi[he];
    }

    // ---- then what hangs on the end points, all independent of each other: one more round trip
    int64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const bool adjp = v.rcache && v.adj_off;   // ([r3] both layouts: the dense update's patch starts from the range too)
    if (adjp) {
        // the re-hung subtree contains exactly one end point of the entering arc; when it is that node alone (most pivots of
        // a large sparse instance) the update's reduced-cost patch can start from here instead of looking the node up
        a0 = v.adj_off[first]; a1 = v.adj_off[first + 1];
        a2 = v.adj_off[second]; a3 = v.adj_off[second + 1];
    }
    if (WITH_CY) {
        // (records in consts that every store below reads: copied through a mutable temporary they took a detour
        //  through private memory)
        const McfNode ru = v.node[first], rw = v.node[second];
        const int32_t su = mcf_node_slot(v, c, first), sw = mcf_node_slot(v, c, second);
        cy->u = first; cy->w = second;
        cy->ru = ru; cy->rw = rw;
        cy->r0u = ru; cy->r0w = rw;
        cy->su = su; cy->sw = sw; cy->s0u = su; cy->s0w = sw;
        cy->pu = su; cy->pw = sw; cy->p0u = su; cy->p0w = sw;   // dense array: the slot is the position; blocked list: resolved after the climb
        cy->small = 0;
        cy->d1 = MCF_INF; cy->d2 = MCF_INF;
        cy->k1 = -1; cy->k2 = -1;
        cy->n1 = 0; cy->n2 = 0;
    }
    // ---- then the stores
    c->pv_e = e; c->pv_s = s; c->pv_first = first; c->pv_second = second; c->pv_rc = rc;
    c->pv_cap = ae.cap;
    c->pv_flow = ae.flow;
    c->pv_t2n = 0;
    if (adjp) { c->pv_adj[0] = a0; c->pv_adj[1] = a1; c->pv_adj[2] = a2; c->pv_adj[3] = a3; }
    return true;
}
MCF_HD bool mcf_pivot_begin(const McfView& v, int64_t best_key, int64_t best_arc, int32_t rule) {
    return mcf_pivot_begin_t<false>(v, best_key, best_arc, rule, nullptr, nullptr);
}

// Step 2a (one lane): the cycle by pointer chasing, at most `budget` round trips.
// --- join search + ratio test (basis.py:207-241, simplex.py:1201-1229).
// Flow is pushed second -> join -> first -> (entering arc) -> second.
// Tie rule for a strongly feasible tree: of all blocking arcs take the LAST one
// met on that route starting from the join, i.e. first-side arcs lose ties to
// the entering arc, which loses ties to second-side arcs.
MCF_HD void mcf_cycle_init(const McfView& v, McfCycle* cy) {
    const McfCtx* c = v.ctx;
    cy->u = c->pv_first;
    cy->w = c->pv_second;
    cy->ru = v.node[cy->u];
    cy->rw = v.node[cy->w];
    cy->su = mcf_node_slot(v, c, cy->u);
    cy->sw = mcf_node_slot(v, c, cy->w);
    cy->pu = cy->su; cy->pw = cy->sw;   // dense array: the slot is the position; blocked list: resolved after the climb (mcf_pivot_climb)
    cy->r0u = cy->ru; cy->r0w = cy->rw;
    cy->p0u = cy->pu; cy->p0w = cy->pw;
    cy->s0u = cy->su; cy->s0w = cy->sw;
    cy->small = 0;
    cy->d1 = MCF_INF; cy->d2 = MCF_INF;
    cy->k1 = -1; cy->k2 = -1;
    cy->n1 = 0; cy->n2 = 0;
}

// Round trips the pointer-chasing climb may take before the scan takes over: unlimited without position-space sizes
// and for shallow end points (the climb then needs at most climb_depth trips), else the handle's budget.
MCF_HD int32_t mcf_climb_budget(const McfView& v, const McfCycle& cy) {
    if (!v.psz[0]) return INT32_MAX;
    const int32_t deep = cy.ru.depth > cy.rw.depth ? cy.ru.depth : cy.rw.depth;
    return deep <= v.ctx->climb_depth ? INT32_MAX : v.ctx->climb_budget;
}

// Returns false on an internal error (depths out of sync).  On return cy->u == cy->w means joined.
// `pb`: where the path is recorded (the global scratch, or -- a climb that is known to be short -- the small LDS buffers).
MCF_HD bool mcf_pivot_climb(const McfView& v, const McfPaths& pb, McfCycle* cy, int32_t budget) {
    McfCtx* c = v.ctx;
    // Depth-balanced climb: per round trip every side that is at least as deep as the other moves
    // up one arc (both when they are level), so the walk costs max(d1, d2) dependent loads instead of
    // d1 + d2; both parents' records and both arcs are requested before anything is looked at.
    // Only the SLOT of a node is fetched on the way (one load that nothing waits for); the blocked list turns slots into
    // positions -- a second, dependent load per node -- after the walk, all of them in flight together.
    const bool bpl = MCF_HAS_BPL(v);
    int32_t u = cy->u, w = cy->w, su = cy->su, sw = cy->sw;
    McfNode ru = cy->ru, rw = cy->rw;
    int64_t d1 = cy->d1, d2 = cy->d2;
    int32_t k1 = cy->k1, k2 = cy->k2, n1 = cy->n1, n2 = cy->n2;
    const int32_t n1_in = n1, n2_in = n2;
    int32_t trips = 0;
    while (u != w && trips < budget) {
        const bool step_u = ru.depth >= rw.depth, step_w = rw.depth >= ru.depth;
        McfNode nu = ru, nw = rw;
        int32_t nsu = su, nsw = sw;
        McfArcW au = McfArcW{0, 0}, aw = McfArcW{0, 0};
        if (step_u) { nu = v.node[ru.parent]; nsu = mcf_node_slot(v, c, ru.parent); au = v.arcw[ru.pred >> 1]; }
        if (step_w) { nw = v.node[rw.parent]; nsw = mcf_node_slot(v, c, rw.parent); aw = v.arcw[rw.pred >> 1]; }
        if (step_u) {
            // first side is walked against the flow: an up arc loses flow, a down arc gains
            const int64_t r = (ru.pred & 1) ? au.flow : (au.cap >= MCF_INF ? MCF_INF : au.cap - au.flow);
            if (r < d1) { d1 = r; k1 = n1; }
            pb.path1[n1] = u;
            pb.rec1[n1] = ru;
            pb.slot1[n1] = su;
            if (!bpl) pb.ppos1[n1] = su;   // dense array: the slot is the position (blocked list: resolved after the walk)
            if (pb.flow1) pb.flow1[n1] = au.flow;
            ++n1;
            u = ru.parent;
            ru = nu;
            su = nsu;
        }
        if (step_w) {
            const int64_t r = (rw.pred & 1) ? (aw.cap >= MCF_INF ? MCF_INF : aw.cap - aw.flow) : aw.flow;
            if (r <= d2) { d2 = r; k2 = n2; }
            pb.path2[n2] = w;
            pb.rec2[n2] = rw;
            pb.slot2[n2] = sw;
            if (!bpl) pb.ppos2[n2] = sw;
            if (pb.flow2) pb.flow2[n2] = aw.flow;
            ++n2;
            w = rw.parent;
            rw = nw;
            sw = nsw;
        }
        if (++trips > v.n_nodes) { c->status = MCF_INTERNAL_ERROR; return false; }  // depths out of sync: never spin
    }
    int32_t pu = su, pw = sw;
    if (bpl) {
        const McfBlkMeta* bm = c->cur ? v.bmeta[1] : v.bmeta[0];
        const int32_t bs = v.blk_shift, bmask = (1 << bs) - 1;
        for (int32_t i = n1_in; i < n1; ++i) { const int32_t sl = pb.slot1[i]; pb.ppos1[i] = bm[sl >> bs].base + (sl & bmask); }
        for (int32_t i = n2_in; i < n2; ++i) { const int32_t sl = pb.slot2[i]; pb.ppos2[i] = bm[sl >> bs].base + (sl & bmask); }
        pu = bm[su >> bs].base + (su & bmask);
        pw = bm[sw >> bs].base + (sw & bmask);
        // (the end points' own positions: the climb is always the first thing that runs on a fresh McfCycle)
        cy->p0u = bm[cy->s0u >> bs].base + (cy->s0u & bmask);
        cy->p0w = bm[cy->s0w >> bs].base + (cy->s0w & bmask);
    }
    cy->u = u; cy->w = w; cy->ru = ru; cy->rw = rw; cy->pu = pu; cy->pw = pw; cy->su = su; cy->sw = sw;
    cy->d1 = d1; cy->d2 = d2; cy->k1 = k1; cy->k2 = k2; cy->n1 = n1; cy->n2 = n2;
    return true;
}

// Step 2b (the whole team): the rest of the cycle by a scan over preorder positions.
//
// The node at position i is an ancestor of x (x included) iff i <= pos[x] < i + psz[i].  The nodes
// still missing from the two paths are exactly the ancestors of cy->u that are not ancestors of
// cy->w (first side) and vice versa; common ancestors lie at lower positions than all of them and
// the join is the common ancestor with the highest position.
//   rounds:  the team sweeps the positions downwards from max(pos[u], pos[w]) in chunks of
//            nlanes * 16 (one 16-byte load of four sizes per lane, four of them in flight) and stops after
//            the first chunk that holds a common ancestor (one barrier per round; position 0, the root, ends
//            the sweep at the latest).  One-sided ancestors are only noted (position + side) in a hit list.
//   hits:    one dense pass over the list: node id, record, arc -> three dependent loads for the whole
//            cycle, whatever its length.  A found ancestor a of u goes to path index
//            n1 + depth[u] - depth[a] (the depth field makes a compaction unnecessary), its record next
//            to it -- exactly what the climb would have recorded.
//   ratio:   every lane keeps the best residual of the elements it met (side 1: lowest index among
//            ties, side 2: highest -- the climb's `<` / `<=`); hit t is handled by lane t, so only the
//            first ceil(nhits / 64) waves hold anything: they reduce by shuffles, lane 0 combines their results.
#ifndef MCF_SCAN_GROUPS
#define MCF_SCAN_GROUPS 4  // 16-byte loads in flight per lane and round
#endif
#ifndef MCF_COARSE_GROUPS
#define MCF_COARSE_GROUPS 8  // ... in the blocked list's coarse pass
#endif
struct McfScanBest { int64_t b1r, b2r; int32_t b1i, b2i; };

MCF_HD void mcf_scan_best_merge(McfScanBest* a, int64_t r1, int32_t i1, int64_t r2, int32_t i2) {
    if (i1 >= 0 && (a->b1i < 0 || r1 < a->b1r || (r1 == a->b1r && i1 < a->b1i))) { a->b1r = r1; a->b1i = i1; }
    if (i2 >= 0 && (a->b2i < 0 || r2 < a->b2r || (r2 == a->b2r && i2 > a->b2i))) { a->b2r = r2; a->b2i = i2; }
}

// lane 0 of the thread-0 section that precedes the scan: reset the accumulators (saves the scan a barrier)
MCF_HD void mcf_scan_init(McfScanAcc* acc) {
    acc->jpos[0] = -1; acc->jpos[1] = -1;   // (64-bit)
    acc->nhits = 0;
    acc->nblk = 0;
}

// The dense pass over the hit list: node id, record, tree arc of every one-sided ancestor (three dependent loads
// for the whole cycle) -> its slot in the path buffers `pb`, and this lane's best residual per side.
MCF_HD void mcf_scan_hit_pass(const McfView& v, const McfPaths& pb, const int32_t* ord, const McfHit* hits, int32_t hits_cap,
                              const McfHit* spill, int32_t nhits, int32_t base1, int32_t base2, int32_t du, int32_t dw,
                              int32_t lane, int32_t nlanes, McfScanBest* out) {
    int64_t b1r = 0, b2r = 0;
    int32_t b1i = -1, b2i = -1;
    const bool bpl = MCF_HAS_BPL(v);
    const McfBlkMeta* bm = v.ctx->cur ? v.bmeta[1] : v.bmeta[0];
    const int32_t bs = v.blk_shift, bmask = (1 << bs) - 1;
    for (int32_t t = lane; t < nhits; t += nlanes) {
        int32_t slot_side;
        if (t < hits_cap) slot_side = hits[t]; else slot_side = spill[t - hits_cap];
        const int32_t slot = slot_side >> 1;
        const int32_t node = ord[slot];
        // blocked list: the logical position from the block's base (independent of the node load: same round trip)
        const int32_t pos = bpl ? bm[slot >> bs].base + (slot & bmask) : slot;
        const McfNode rec = v.node[node];
        const McfArcW a = v.arcw[rec.pred >> 1];
        if (!(slot_side & 1)) {
            const int32_t idx = base1 + du - rec.depth;
            pb.path1[idx] = node;
            pb.rec1[idx] = rec;
            pb.slot1[idx] = slot;   // (before the position: with the dense array both are the same word)
            pb.ppos1[idx] = pos;
            if (pb.flow1) pb.flow1[idx] = a.flow;
            const int64_t r = (rec.pred & 1) ? a.flow : (a.cap >= MCF_INF ? MCF_INF : a.cap - a.flow);
            if (b1i < 0 || r < b1r || (r == b1r && idx < b1i)) { b1r = r; b1i = idx; }
        } else {
            const int32_t idx = base2 + dw - rec.depth;
            pb.path2[idx] = node;
            pb.rec2[idx] = rec;
            pb.slot2[idx] = slot;
            pb.ppos2[idx] = pos;
            if (pb.flow2) pb.flow2[idx] = a.flow;
            const int64_t r = (rec.pred & 1) ? (a.cap >= MCF_INF ? MCF_INF : a.cap - a.flow) : a.flow;
            if (b2i < 0 || r < b2r || (r == b2r && idx > b2i)) { b2r = r; b2i = idx; }
        }
    }
    out->b1r = b1r; out->b2r = b2r; out->b1i = b1i; out->b2i = b2i;
}

// `hits` holds the first `hits_cap` entries of the hit list (LDS on the device), the scratch behind v.seg the rest.
// `sp`: small buffers of `small_cap` entries each (LDS on the device; cap 0 = none); a cycle found by the
// scan alone that fits is recorded there (cy->small = 1), which spares the decide / finish passes a
// global round trip per look-up.  The caller has run mcf_scan_init(acc) before the barrier in front of this call.
// Four consecutive positions i0 .. i0 + 3 with their subtree sizes: note the ancestors of the node at position pu
// and / or pw among them (common ancestor -> jpos, one-sided -> hit list).
MCF_HD void mcf_scan_group(int32_t i0, int32_t s0, const int32_t* sz, int32_t pu, int32_t pw, int32_t pmin, int64_t* jpos_slot,
                           McfScanAcc* acc, McfHit* hits, int32_t hits_cap, McfHit* spill, bool dense) {
    // Cheap reject of the whole group first: a subtree can hold pu or pw only if it reaches past the lower
    // of the two, and almost every position is a small subtree far to the left of both.
    int32_t zmax = sz[0] > sz[1] ? sz[0] : sz[1];
    const int32_t z23 = sz[2] > sz[3] ? sz[2] : sz[3];
    zmax = zmax > z23 ? zmax : z23;
    if (i0 + 3 + zmax <= pmin) return;
    for (int e = 0; e < 4; ++e) {
        const int32_t i = i0 + e;   // logical position; the slot that holds it is s0 + e
        // i <= p < i + size  <=>  unsigned(p - i) < unsigned(size)   (sizes are positive; 0 marks "no position")
        const bool au = (uint32_t)(pu - i) < (uint32_t)sz[e];
        const bool aw = (uint32_t)(pw - i) < (uint32_t)sz[e];
        if (au && aw) {
            // dense array: slot == position, a 32-bit max on the low word does (the high word stays -1 / 0: see the reader);
            // blocked list: position and slot travel together.  Every common ancestor up to the root passes here -- hundreds
            // in a deep tree, each an atomic on ONE word, which the LDS serialises -- but only one that beats the value already
            // there can change it: look first (a plain read; in any order of arrival ~ln(n) of n candidates get through).
            if (dense) {
                if (i > *reinterpret_cast<volatile int32_t*>(jpos_slot)) MCF_ATOMIC_MAX32(reinterpret_cast<int32_t*>(jpos_slot), i);
            } else {
                const int64_t cand = ((int64_t)i << 32) | (uint32_t)(s0 + e);
                if (cand > *reinterpret_cast<volatile int64_t*>(jpos_slot)) MCF_ATOMIC_MAX64(jpos_slot, cand);
            }
        }
        const bool hit = (au || aw) && !(au && aw);
#if defined(__HIP_DEVICE_COMPILE__)
        // one append per wave and step instead of one per hit (the hit counter is one LDS word too): the lanes of this wave
        // that found a one-sided ancestor at this step take consecutive entries
        const unsigned long long hm = __ballot(hit);
        if (hit) {
            const int32_t lane_id = (int32_t)__lane_id();
            const int32_t leader = __ffsll((unsigned long long)hm) - 1;
            int32_t base = 0;
            if (lane_id == leader) base = atomicAdd(&acc->nhits, (int)__popcll(hm));
            base = __shfl(base, leader);
            const int32_t slot = base + (int32_t)__popcll(hm & ((1ull << lane_id) - 1ull));
            const McfHit hrec = ((s0 + e) << 1) | (aw ? 1 : 0);
            if (slot < hits_cap) hits[slot] = hrec; else spill[slot - hits_cap] = hrec;
        }
#else
        if (hit) {
            const int32_t slot = MCF_ATOMIC_ADD32(&acc->nhits, 1);
            const McfHit hrec = ((s0 + e) << 1) | (aw ? 1 : 0);
            if (slot < hits_cap) hits[slot] = hrec; else spill[slot - hits_cap] = hrec;
        }
#endif
    }
}

MCF_HD void mcf_pivot_scan(const McfView& v, const McfPaths& sp, int32_t small_cap, McfCycle* cy, McfScanAcc* acc,
                           McfHit* hits, int32_t hits_cap, int32_t lane, int32_t nlanes) {
    McfCtx* c = v.ctx;
    const int32_t* ord = mcf_slot_nodes(v, c);
    const int32_t* psz = mcf_slot_sizes(v, c);
    McfHit* spill = reinterpret_cast<McfHit*>(v.seg);  // scratch the finish pass only fills later; <= n_nodes entries
    const int32_t pu = cy->pu, pw = cy->pw, du = cy->ru.depth, dw = cy->rw.depth;
    const int32_t pmin = pu < pw ? pu : pw, pmax = pu > pw ? pu : pw;
    const int32_t base1 = cy->n1, base2 = cy->n2;
    MCF_PSTAMP(4);
    int32_t rounds = 0;
    int64_t jpk = -1;   // (position << 32) | slot of the join
    bool done = false;
    const bool bpl = MCF_HAS_BPL(v);
    // ---- coarse pass (trees too large for a round or two of the plain sweep): which blocks can hold an ancestor of
    // either end point?  Dense array: one pass over reach[0 .. pmax / 64]; blocked list: one pass over the {base, rrel}
    // records of the blocks handed out so far.  Then only those blocks are looked at: two dependent round trips whatever
    // the depth of the tree (the plain sweep needs (pmax - pos[join]) / 16 384).
    if (bpl || (v.reach && pmax >= 2 * nlanes * 4 * MCF_SCAN_GROUPS)) {
        int32_t* const blk_spill = v.chg;   // blocked list: flagged blocks beyond the LDS list, (block, base) pairs
        if (bpl) {
            const McfBlkMeta* bm = c->cur ? v.bmeta[1] : v.bmeta[0];
            const int32_t nb = c->alloc_next;
            // two records per 16-byte load (bmeta[] is padded to an even number of records); MCF_COARSE_GROUPS loads in flight
            // per lane: 16 384 blocks per trip of a 1 024-lane team -- the 1 M-node list (13-14 K blocks of 128) in ONE round trip
            for (int32_t base = 0; base < nb; base += nlanes * 2 * MCF_COARSE_GROUPS) {
                int32_t rr[MCF_COARSE_GROUPS][4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int k = 0; k < MCF_COARSE_GROUPS; ++k) {
                    const int32_t b0 = base + (k * nlanes + lane) * 2;
                    rr[k][0] = rr[k][2] = MCF_BLK_FREE; rr[k][1] = rr[k][3] = 0;
                    if (b0 < nb) {
#if defined(__HIP_DEVICE_COMPILE__)
                        const int4 q = *reinterpret_cast<const int4*>(bm + b0);
                        rr[k][0] = q.x; rr[k][1] = q.y; rr[k][2] = q.z; rr[k][3] = q.w;
#else
                        rr[k][0] = bm[b0].base; rr[k][1] = bm[b0].rrel; rr[k][2] = bm[b0 + 1].base; rr[k][3] = bm[b0 + 1].rrel;
#endif
                    }
                }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int k = 0; k < MCF_COARSE_GROUPS; ++k) {
                    const int32_t b0 = base + (k * nlanes + lane) * 2;
                    for (int e = 0; e < 2; ++e) {
                        const int32_t b = b0 + e, s0 = rr[k][2 * e], r = s0 + rr[k][2 * e + 1];
                        if (b < nb && ((s0 <= pu && r > pu) || (s0 <= pw && r > pw))) {
                            const int32_t slot = MCF_ATOMIC_ADD32(&acc->nblk, 1);
                            if (slot < MCF_SCAN_BLK_CAP) { acc->blk[slot] = b; acc->blkbase[slot] = s0; }
                            else { blk_spill[2 * (slot - MCF_SCAN_BLK_CAP)] = b; blk_spill[2 * (slot - MCF_SCAN_BLK_CAP) + 1] = s0; }
                        }
                    }
                }
            }
        } else {
            const int32_t nb = (pmax >> MCF_REACH_SHIFT) + 1;
            // four entries per 16-byte load, MCF_SCAN_GROUPS loads in flight per lane: the pass is one memory round trip per
            // nlanes * 16 entries (a lane-at-a-time loop of dependent 4-byte loads took 30 us at 4 M nodes)
            for (int32_t base = 0; base < nb; base += nlanes * 4 * MCF_SCAN_GROUPS) {
                int32_t rr[MCF_SCAN_GROUPS][4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int k = 0; k < MCF_SCAN_GROUPS; ++k) {
                    const int32_t b0 = base + (k * nlanes + lane) * 4;
                    rr[k][0] = rr[k][1] = rr[k][2] = rr[k][3] = 0;
                    if (b0 < nb) {   // (reach[] is padded to a multiple of four entries)
#if defined(__HIP_DEVICE_COMPILE__)
                        const int4 q = *reinterpret_cast<const int4*>(v.reach + b0);
                        rr[k][0] = q.x; rr[k][1] = q.y; rr[k][2] = q.z; rr[k][3] = q.w;
#else
                        for (int e = 0; e < 4; ++e) rr[k][e] = v.reach[b0 + e];
#endif
                    }
                }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int k = 0; k < MCF_SCAN_GROUPS; ++k) {
                    const int32_t b0 = base + (k * nlanes + lane) * 4;
                    for (int e = 0; e < 4; ++e) {
                        const int32_t b = b0 + e, r = rr[k][e], s0 = b << MCF_REACH_SHIFT;
                        if (b < nb && ((s0 <= pu && r > pu) || (s0 <= pw && r > pw))) {
                            const int32_t slot = MCF_ATOMIC_ADD32(&acc->nblk, 1);
                            if (slot < MCF_SCAN_BLK_CAP) { acc->blk[slot] = b; acc->blkbase[slot] = s0; }
                        }
                    }
                }
            }
        }
        MCF_TEAM_BARRIER();
        MCF_PSTAMP(21);   // (diagnostic build: the coarse pass on its own)
        const int32_t nf = acc->nblk;
        if (bpl || nf <= MCF_SCAN_BLK_CAP) {
            // a group of lanes takes one flagged block: four slots (one 16-byte load) per lane
            const int32_t bshift = bpl ? v.blk_shift : MCF_REACH_SHIFT;
            const int32_t per = (1 << bshift) / 4;                      // lanes per block
            const int32_t ngroups = nlanes >= per ? nlanes / per : 1;
            const int32_t g = nlanes >= per ? lane / per : 0, sub0 = nlanes >= per ? lane % per : 0;
            const int32_t nsub = nlanes >= per ? 1 : per;              // a team of one lane walks the block itself
            // MCF_SCAN_GROUPS blocks per lane group and trip: all their loads are in flight before the first is looked at
            // (a long cycle flags hundreds of blocks; one block per trip made the pass a chain of dependent round trips)
            for (int32_t t0 = g; t0 < nf; t0 += ngroups * MCF_SCAN_GROUPS) {
                for (int32_t q = 0; q < nsub; ++q) {
                    int32_t szk[MCF_SCAN_GROUPS][4], s0k[MCF_SCAN_GROUPS], i0k[MCF_SCAN_GROUPS];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                    for (int k = 0; k < MCF_SCAN_GROUPS; ++k) {
                        const int32_t t = t0 + k * ngroups;
                        s0k[k] = -1; i0k[k] = 0;
                        szk[k][0] = szk[k][1] = szk[k][2] = szk[k][3] = 0;
                        if (t >= nf) continue;
                        int32_t b, lbase;
                        if (t < MCF_SCAN_BLK_CAP) { b = acc->blk[t]; lbase = acc->blkbase[t]; }
                        else { b = blk_spill[2 * (t - MCF_SCAN_BLK_CAP)]; lbase = blk_spill[2 * (t - MCF_SCAN_BLK_CAP) + 1]; }
                        s0k[k] = (b << bshift) + (sub0 + q) * 4;   // slots
                        i0k[k] = lbase + (sub0 + q) * 4;           // their logical positions
#if defined(__HIP_DEVICE_COMPILE__)
                        const int4 w4 = *reinterpret_cast<const int4*>(psz + s0k[k]);
                        szk[k][0] = w4.x; szk[k][1] = w4.y; szk[k][2] = w4.z; szk[k][3] = w4.w;
#else
                        for (int e = 0; e < 4; ++e) szk[k][e] = psz[s0k[k] + e];
#endif
                    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                    for (int k = 0; k < MCF_SCAN_GROUPS; ++k)
                        if (s0k[k] >= 0) mcf_scan_group(i0k[k], s0k[k], szk[k], pu, pw, pmin, &acc->jpos[0], acc, hits, hits_cap, spill, !bpl);
                }
            }
            MCF_TEAM_BARRIER();
            jpk = bpl ? acc->jpos[0] : (int64_t)(int32_t)(acc->jpos[0] & 0xffffffff);   // (dense array: the low word is the position, -1 = none)
            rounds = 2;
            done = true;
        }
        // (dense array, more flagged blocks than the list holds: the plain rounds below redo the search; the hit list is
        //  still empty because the fine pass did not run)
    }
    // ---- plain rounds (dense array): groups of four positions, aligned; psz[] is padded so that the group holding max(pu, pw) can be read whole
    int32_t top = ((pmax + 1) + 3) & ~3;  // exclusive
    int32_t par = 0;
    while (!done) {
        const int32_t lo = top - nlanes * 4 * MCF_SCAN_GROUPS;
        int32_t sz[MCF_SCAN_GROUPS][4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < MCF_SCAN_GROUPS; ++k) {
            const int32_t slice = lo + k * nlanes * 4;  // uniform
            const int32_t i = slice + lane * 4;
            sz[k][0] = sz[k][1] = sz[k][2] = sz[k][3] = 0;
            if (slice + nlanes * 4 <= 0) continue;      // the whole slice lies below position 0 (small trees: most do)
            if (i >= 0) {
#if defined(__HIP_DEVICE_COMPILE__)
                const int4 q = *reinterpret_cast<const int4*>(psz + i);
                sz[k][0] = q.x; sz[k][1] = q.y; sz[k][2] = q.z; sz[k][3] = q.w;
#else
                for (int e = 0; e < 4; ++e) sz[k][e] = psz[i + e];
#endif
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < MCF_SCAN_GROUPS; ++k) {
            const int32_t slice = lo + k * nlanes * 4;
            if (slice + nlanes * 4 <= 0) continue;
            mcf_scan_group(slice + lane * 4, slice + lane * 4, sz[k], pu, pw, pmin, &acc->jpos[par], acc, hits, hits_cap, spill, true);
        }
        MCF_TEAM_BARRIER();
        // the next round's atomics go to the other slot: nobody can overtake a lane still reading this one
        jpk = (int64_t)(int32_t)(acc->jpos[par] & 0xffffffff);   // (dense array only gets here)
        par ^= 1;
        ++rounds;
        if (jpk >= 0 || lo <= 0) break;
        top = lo;
    }
    const int32_t jslot = jpk < 0 ? -1 : (int32_t)(jpk & 0xffffffff);
    if (jslot < 0) {  // position 0 is the root, a common ancestor: cannot happen
        if (lane == 0) c->status = MCF_INTERNAL_ERROR;
        return;
    }
    MCF_PSTAMP(5);
    const int32_t nhits = acc->nhits;  // final: every append precedes the last barrier
#if defined(MCF_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
    if (lane == 0) {   // diagnostic build: scan volume (blocks handed out, blocks flagged, cycle nodes found, scans)
        mcf_stamp_acc[17] += (unsigned long long)(bpl ? c->alloc_next : (pmax >> MCF_REACH_SHIFT) + 1);
        mcf_stamp_acc[18] += (unsigned long long)acc->nblk;
        mcf_stamp_acc[19] += (unsigned long long)nhits;
        mcf_stamp_acc[20] += 1;
        if (acc->nblk > 128) mcf_stamp_acc[22] += 1;   // fine passes that need more than one trip ...
        if (acc->nblk > 512) mcf_stamp_acc[11] += 1;   // ... more than four
    }
#endif
    if (lane == nlanes - 1) {  // the join's record rides along with the hit pass (this lane is the last to get a hit)
        const int32_t jn = ord[jslot];
        acc->jnode = jn;
        acc->join = v.node[jn];
    }
    // every path index is < base + nhits: with nothing climbed and few hits the small buffers hold the cycle.
    // Two calls rather than selected pointers: each inlined copy then has one known address space.
    const bool small = base1 == 0 && base2 == 0 && nhits <= small_cap;
    McfScanBest best;
    if (small) mcf_scan_hit_pass(v, sp, ord, hits, hits_cap, spill, nhits, base1, base2, du, dw, lane, nlanes, &best);
    else mcf_scan_hit_pass(v, mcf_view_paths(v), ord, hits, hits_cap, spill, nhits, base1, base2, du, dw, lane, nlanes, &best);
    MCF_PSTAMP(6);
#if defined(__HIP_DEVICE_COMPILE__)
    // hit t was handled by lane t % nlanes: waves beyond ceil(nhits / 64) hold nothing.  Per side two DPP max
    // reductions: the smallest residual, then the wanted path index among the lanes that hold it.
    const int32_t wave = lane >> 6, nwaves_hit = nhits >= nlanes ? nlanes >> 6 : (nhits + 63) >> 6;
    if (wave < nwaves_hit) {
        const int64_t kNone = INT64_MIN;
        const int64_t m1 = mcf_wave_max64(best.b1i >= 0 ? -best.b1r : kNone);
        const int64_t x1 = mcf_wave_max64(best.b1i >= 0 && -best.b1r == m1 ? -(int64_t)best.b1i : kNone);  // lowest index
        const int64_t m2 = mcf_wave_max64(best.b2i >= 0 ? -best.b2r : kNone);
        const int64_t x2 = mcf_wave_max64(best.b2i >= 0 && -best.b2r == m2 ? (int64_t)best.b2i : kNone);   // highest index
        if ((lane & 63) == 0) {
            acc->wr1[wave] = m1 == kNone ? 0 : -m1; acc->wi1[wave] = m1 == kNone ? -1 : (int32_t)(-x1);
            acc->wr2[wave] = m2 == kNone ? 0 : -m2; acc->wi2[wave] = m2 == kNone ? -1 : (int32_t)x2;
        }
    }
    MCF_TEAM_BARRIER();
    if (lane == 0) {
        best.b1i = -1; best.b2i = -1;
        for (int32_t q = 0; q < nwaves_hit; ++q) mcf_scan_best_merge(&best, acc->wr1[q], acc->wi1[q], acc->wr2[q], acc->wi2[q]);
    }
#endif
    if (lane == 0) {
        // merge with what the climb found: its elements have the lower path indices
        if (best.b1i >= 0 && best.b1r < cy->d1) { cy->d1 = best.b1r; cy->k1 = best.b1i; }
        if (best.b2i >= 0 && best.b2r <= cy->d2) { cy->d2 = best.b2r; cy->k2 = best.b2i; }
        const int32_t jn = acc->jnode;
        const McfNode rj = acc->join;
        cy->n1 = base1 + du - rj.depth;
        cy->n2 = base2 + dw - rj.depth;
        cy->u = jn; cy->w = jn;
        cy->ru = rj; cy->rw = rj;
        cy->small = small ? 1 : 0;
        c->scans += 1;
        c->scan_rounds += rounds;
    }
    MCF_PSTAMP(7);
}

// Step 3 (one lane): ratio-test decision + everything the finish / apply passes need to know.
MCF_HD void mcf_pivot_decide(const McfView& v, const McfPaths& pp, const McfCycle& cy) {
    McfCtx* c = v.ctx;
    const int32_t e = c->pv_e, s = c->pv_s, first = c->pv_first, second = c->pv_second;
    const int64_t rc = c->pv_rc;
    const int64_t d1 = cy.d1, d2 = cy.d2;
    const int32_t k1 = cy.k1, k2 = cy.k2, n1 = cy.n1, n2 = cy.n2;
    const McfNode ru = cy.ru;  // joined: ru == rw == the join's record
    const int64_t de = c->pv_cap;  // residual of the entering arc in its push direction (a non-basic arc sits at a bound)
    int32_t result;
    int64_t delta;
    if (d2 <= de && d2 <= d1) { result = 2; delta = d2; }
    else if (de <= d1) { result = 0; delta = de; }
    else { result = 1; delta = d1; }

    if (delta >= MCF_INF) {  // simplex.py:1231-1246
        c->status = MCF_UNBOUNDED;
        c->unbounded_arc = e;
        return;
    }
    if (delta == 0) c->degenerate += 1;
    c->pivots += 1;
    c->cycle_arcs += n1 + n2 + 1;
    c->pv_e = e; c->pv_s = s; c->pv_n1 = n1; c->pv_n2 = n2; c->pv_delta = delta; c->pv_result = result;
    if (v.weight) {
        // Devex bookkeeping, all of it scalar (lane 0).
        // (1) periodic reset (simplex.py:1370-1400: after ft_update_limit = 64 rank-one updates the basis is
        //     refactorised and the pricing strategy reset: weights 1, block 0).  Only basis swaps count.
        bool reset = false;
        if (result != 0) {
            if (c->swaps_since_reset >= MCF_DEVEX_RESET_SWAPS) { reset = true; c->swaps_since_reset = 0; }
            else c->swaps_since_reset += 1;
        }
        // (2) deferred weight update: only the selected arc's weight is refreshed (simplex_pricing.py:271-292);
        //     ||B^-1 a||^2 of a tree basis = number of tree arcs between the end points.  The touched arcs are
        //     listed so that a reset costs a pass over the list instead of over all arcs.
        if (!reset && v.dx) {
            if (c->wlist_n >= MCF_WLIST_CAP) reset = true;  // list full: reset early (heuristic state only)
            else {
                const int32_t len = n1 + n2;
                v.weight[e] = (float)(len > 0 ? len : 1);
                v.dx->wlist[c->wlist_n] = e;
                c->wlist_n += 1;
            }
        }
        if (reset) { c->wreset = 1; c->block_index = 0; }
        // (3) block-size tuner (simplex_adaptive.py:98-151): every >= 50 pivots, on the share of degenerate pivots
        //     (record_pivot counts theta == 0 and entering == leaving, simplex.py:1316-1318)
        c->tn_total += 1;
        if (delta == 0 || result == 0) c->tn_degenerate += 1;
        if (c->auto_tune && c->pivots - c->tn_last_adapt >= 50 && c->tn_total >= 10) {
            int32_t bg = c->block_granules;
            if (10 * c->tn_degenerate > 3 * c->tn_total) {         // > 30 % degenerate: larger blocks
                const int32_t nb = bg * 3 / 2 > bg + 1 ? bg * 3 / 2 : bg + 1;
                bg = nb < c->max_granules ? nb : (bg > c->max_granules ? bg : c->max_granules);
            } else if (10 * c->tn_degenerate < c->tn_total) {      // < 10 %: smaller blocks
                const int32_t nb = bg * 3 / 4;
                bg = nb > 1 ? nb : 1;
            }
            if (bg != c->block_granules) {
                c->block_granules = bg;
                c->num_blocks = (MCF_GRANULES + bg - 1) / bg;
                c->block_size = v.m * bg / MCF_GRANULES;
                if (c->block_index >= c->num_blocks) c->block_index = 0;
            }
            c->tn_degenerate = 0; c->tn_total = 0; c->tn_last_adapt = c->pivots;
        }
    }
    if (result == 0) {  // entering arc is also the leaving arc (simplex.py:1320-1334)
        c->bound_flips += 1;
        c->stage = 1;
        return;
    }

    // --- basis swap (simplex.py:1335-1425): scalar decisions only
    const int32_t k = result == 1 ? k1 : k2;                 // stem[0] = u_in ... stem[k] = q
    const McfNode* srec = result == 1 ? pp.rec1 : pp.rec2;
    const int32_t v_in = result == 1 ? second : first;
    const McfNode rq = srec[k];
    const bool tail_in_t2 = (result == 1) == (s > 0);
    c->pv_k = k;
    c->pv_vin = v_in;
    c->pv_leave = rq.pred >> 1;
    // state the leaving arc takes: the first side is walked against the flow (an up arc ends empty),
    // the second side with it (an up arc ends full)
    c->pv_leave_state = result == 1 ? ((rq.pred & 1) ? 1 : -1) : ((rq.pred & 1) ? -1 : 1);
    c->pv_tail_in_t2 = tail_in_t2 ? 1 : 0;
    c->sigma = tail_in_t2 ? -rc : rc;  // potential shift that zeroes the entering arc's reduced cost

    const int32_t S = rq.size, a0 = (result == 1 ? pp.ppos1 : pp.ppos2)[k];
    // v_in is the entering arc's end point on the other side: its record and position were read at the start
    const McfNode rvin = result == 1 ? cy.r0w : cy.r0u;
    const int32_t pvin = result == 1 ? cy.p0w : cy.p0u;
    c->pv_vin_depth = rvin.depth;
    // depth change of the first piece of the re-rooted subtree (u_in and what hangs below it): with k == 0 that piece is all
    // of T2 and the segment table is {t2_new, t2_old, S, this} -- the apply pass then needs no look-up in it (mcf_apply_source)
    c->pv_dd0 = rvin.depth + 1 - (result == 1 ? cy.r0u : cy.r0w).depth;
    // insertion point in OLD coordinates: directly behind v_in, or at the end of
    // v_in's block -- whichever moves fewer array elements
    const int32_t tA = pvin + 1, tB = pvin + rvin.size;
    const int32_t costA = tA <= a0 ? a0 - tA : tA - (a0 + S);
    const int32_t costB = tB <= a0 ? a0 - tB : tB - (a0 + S);
    const int32_t t = costA <= costB ? tA : tB;
    const int32_t b = t <= a0 ? t : t - S;
    c->t2_old = a0;
    c->t2_new = b;
    c->t2_size = S;
    c->lo = t < a0 ? t : a0;
    c->hi = t > a0 + S ? t : a0 + S;
    c->nseg = 2 * k + 1;
    c->nchg = (result == 1 ? n1 : n2) - k - 1;  // the old ancestors of q below the join: their subtrees shrink
    c->apply = 1;
    c->pending_flip = 1;
    c->subtree_nodes += S;
    c->stage = 2;
    if (S == 1) {   // T2 = {u_in}: first (stem on the first side) or second (the adjacency range only rides along with resident reduced costs)
        c->pv_t2n = 1;
        c->pv_t2node = result == 1 ? first : second;
        if (result != 1) { c->pv_adj[0] = c->pv_adj[2]; c->pv_adj[1] = c->pv_adj[3]; }
    }
    if (MCF_HAS_BPL(v)) {
        // blocked list: T2 goes into fresh blocks (dense-packed), in front of them two blocks for what has to be cut off
        // existing blocks; nothing else moves.  The pool is a bump allocator: when it runs out, this pivot's update writes
        // the WHOLE list densely into the other arena instead (every position is then rewritten once).
        c->t_ins = t;
        c->nchg = 0;   // (shrunken subtrees are flagged in bext[], not listed)
        c->t2_blk = (result == 1 ? pp.slot1 : pp.slot2)[k] >> v.blk_shift;
        // inserted directly behind v_in: the cut, if any, is in v_in's block; at the end of v_in's subtree: wherever that is
        c->ins_blk = t == tA ? ((result == 1 ? cy.s0w : cy.s0u) >> v.blk_shift) : -1;
        const int32_t need = MCF_BLK_COPIES + ((S + (1 << v.blk_shift) - 1) >> v.blk_shift);
        c->alloc_prev = c->alloc_next;
        if (c->alloc_next + need > v.blk_cap) {
            c->rebuild = 1;
            c->rebuilds += 1;
            c->alloc_lo = 0;
            c->alloc_next = c->dense_blocks;
            c->nodes_moved += v.n_nodes;
        } else {
            c->rebuild = 0;
            c->alloc_lo = c->alloc_next;
            c->alloc_next += need;
            c->nodes_moved += S;   // (+ the cut-off runs, counted by the update pass)
        }
    } else {
        c->nodes_moved += c->hi - c->lo;
    }
}

// ---------------------------------------------------------------------------
// Blocked preorder list: the tree update in O(|T2| + block) element moves.
//
// The LOGICAL preorder (what positions, subtree intervals, segments and the insertion point t are expressed in) is
// exactly the dense array's.  Physically the list lives in blocks of B = 1 << blk_shift slots; block b holds the
// logical positions base[b] + [beg, end) in its slots [beg, end).  A basis swap takes T2 = [a0, a0 + S) out and puts it
// back (re-rooted: the segment table) in front of old position t; every other position p moves by
//     shift(p) = +S for t <= p < a0,   -S for a0 + S <= p < t,   0 otherwise.
// Per block that is pure interval arithmetic on [L0, L1) = base + [beg, end):
//   * no element of T2 inside and t not strictly inside: the whole block lies in one zone -> base += shift, nothing moves;
//   * otherwise the survivors form at most three runs (cut at the hole T2 leaves and at t).  The first run stays where
//     it is (new base, new [beg, end)); a run right of the cut at t goes to copy block alloc_lo, a run right of the
//     hole to copy block alloc_lo + 1 (each can only happen in one block of the whole list);
//   * T2's elements go, dense-packed, to blocks alloc_lo + 2 ...: new logical position j -> slot (j - t2_new).
// Blocks are never refilled (bump allocation); when the pool runs out the update writes the whole list densely into the
// other arena (ctx.rebuild).  Freed blocks keep base = MCF_BLK_FREE, empty slots size 0: the cycle scan skips both.
// ---------------------------------------------------------------------------
MCF_HD int32_t mcf_bpl_shift(const McfCtx& c, int32_t p) {
    const int32_t a0 = c.t2_old, S = c.t2_size, t = c.t_ins;
    if (t <= a0) return (p >= t && p < a0) ? S : 0;
    return (p >= a0 + S && p < t) ? -S : 0;
}

struct McfBlkPlan {
    int32_t touched;            // 0: the block keeps its slots, only the base moves (nbase)
    int32_t nbase;              // new base of the block (MCF_BLK_FREE: nothing stays)
    int32_t r0lo, r0hi;         // logical interval (old coordinates) that stays in the block
    int32_t ilo[2], ihi[2];     // intervals cut off: [0] -> copy block alloc_lo (right of t), [1] -> alloc_lo + 1 (right of the hole)
                                // (only ever indexed by constants)
};

MCF_HD McfBlkPlan mcf_bpl_plan(const McfCtx& c, int32_t base, int32_t L0, int32_t L1) {
    McfBlkPlan P;
    const int32_t a0 = c.t2_old, S = c.t2_size, t = c.t_ins;
    P.ilo[0] = P.ihi[0] = P.ilo[1] = P.ihi[1] = 0;
    const bool hit_t2 = L0 < a0 + S && L1 > a0;
    const bool cut_t = L0 < t && t < L1;
    if (L1 <= L0) { P.touched = 0; P.nbase = MCF_BLK_FREE; P.r0lo = P.r0hi = 0; return P; }
    if (!hit_t2 && !cut_t) {
        P.touched = 0; P.nbase = base + mcf_bpl_shift(c, L0); P.r0lo = L0; P.r0hi = L1;
        return P;
    }
    P.touched = 1;
    // survivors: A = [L0, min(L1, a0)) and Bv = [max(L0, a0 + S), L1); t cuts one of them strictly inside.
    // Three candidate runs in logical order (scalars, not arrays: the plan lives in registers); kind k: what lies to the
    // left of run k -- 0 the cut at t, 1 the hole
    const int32_t Ahi = L1 < a0 ? L1 : a0, Blo = L0 > a0 + S ? L0 : a0 + S;
    int32_t lo0, hi0, lo1, hi1, lo2, hi2, kind1, kind2;
    if (t <= a0) {
        lo0 = L0; hi0 = t < Ahi ? t : Ahi;
        lo1 = L0 > t ? L0 : t; hi1 = Ahi; kind1 = 0;
        lo2 = Blo; hi2 = L1; kind2 = 1;
    } else {
        lo0 = L0; hi0 = Ahi;
        lo1 = Blo; hi1 = t < L1 ? t : L1; kind1 = 1;
        lo2 = Blo > t ? Blo : t; hi2 = L1; kind2 = 0;
    }
    bool have0 = false;
    P.r0lo = P.r0hi = 0;
    int32_t prev_hi = -1;
    if (hi0 > lo0) { have0 = true; P.r0lo = lo0; P.r0hi = hi0; prev_hi = hi0; }
    if (hi1 > lo1) {
        if (!have0) { have0 = true; P.r0lo = lo1; P.r0hi = hi1; }
        else if (lo1 == prev_hi && kind1 == 0) { P.ilo[0] = lo1; P.ihi[0] = hi1; }   // directly behind the previous run: cut off by t
        else { P.ilo[1] = lo1; P.ihi[1] = hi1; }                                      // ... by the hole
        prev_hi = hi1;
    }
    if (hi2 > lo2) {
        if (!have0) { have0 = true; P.r0lo = lo2; P.r0hi = hi2; }
        else if (lo2 == prev_hi && kind2 == 0) { P.ilo[0] = lo2; P.ihi[0] = hi2; }
        else { P.ilo[1] = lo2; P.ihi[1] = hi2; }
    }
    P.nbase = have0 ? base + mcf_bpl_shift(c, P.r0lo) : MCF_BLK_FREE;
    return P;
}

// Old logical position i inside T2 -> its new logical position (and the depth change of its piece).  The segment table is
// sorted by destination; by SOURCE the order is: the left pieces from the outermost stem node inwards (seg[2k-1], seg[2k-3],
// ..., seg[1]), the innermost block seg[0], then the right pieces outwards (seg[2], seg[4], ..., seg[2k]).  Empty pieces
// share their source with the next one: the LAST entry whose source is <= i is the one that holds i.
MCF_HD int32_t mcf_bpl_seg_at(int32_t r, int32_t k) { return r < k ? 2 * (k - r) - 1 : (r == k ? 0 : 2 * (r - k)); }
MCF_HD int32_t mcf_bpl_new_pos(const McfCtx& c, const McfSeg* seg, int32_t i, int32_t* ddepth) {
    const int32_t k = (c.nseg - 1) >> 1;
    int32_t lo = 0, hi = c.nseg - 1;
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (seg[mcf_bpl_seg_at(mid, k)].src <= i) lo = mid; else hi = mid - 1;
    }
    const McfSeg sg = seg[mcf_bpl_seg_at(lo, k)];
    *ddepth = sg.ddepth;
    return sg.dst + (i - sg.src);
}

// End of the finish pass (all lanes): the records of the blocks this pivot's update will fill, in the meta copy the update
// writes (bmeta[cur ^ 1]) and in the extents of the arena it writes to.  The update's pushes then only raise rrel.
MCF_HD void mcf_bpl_prepare(const McfView& v, const McfCtx& c, int32_t lane, int32_t nlanes) {
    if (c.stage != 2) return;
    McfBlkMeta* bn = c.cur ? v.bmeta[0] : v.bmeta[1];
    const int32_t bs = v.blk_shift, B = 1 << bs;
    if (c.rebuild) {
        int32_t* xn = c.arena ? v.bext[0] : v.bext[1];   // the OTHER arena
        const int32_t nd = c.dense_blocks;
        for (int32_t b = lane; b < nd; b += nlanes) {
            const int32_t left = v.n_nodes - (b << bs);
            bn[b] = McfBlkMeta{b << bs, 0};
            xn[b] = mcf_ext_make(0, left < B ? left : B);
        }
        return;
    }
    int32_t* xa = c.arena ? v.bext[1] : v.bext[0];
    const int32_t first = c.alloc_lo, S = c.t2_size;
    const int32_t nt2 = (S + B - 1) >> bs;
    for (int32_t q = lane; q < MCF_BLK_COPIES + nt2; q += nlanes) {
        const int32_t b = first + q;
        if (q < MCF_BLK_COPIES) { bn[b] = McfBlkMeta{MCF_BLK_FREE, 0}; xa[b] = 0; }
        else {
            const int32_t o = (q - MCF_BLK_COPIES) << bs;
            bn[b] = McfBlkMeta{c.t2_new + o, 0};
            xa[b] = mcf_ext_make(0, S - o < B ? S - o : B);
        }
    }
}

// One slot of a block the update has to take apart (host: scalar loop; device: one lane per slot).  `p` = old logical
// position, `nd` / `z` = node and size in the slot.  Returns what stays in the block as o + z (0: nothing), for the
// block's new rrel.  `cnt` counts elements copied to the cut-off blocks (diagnostic).
struct McfBplOut { int32_t keep_reach; int32_t copy_reach[2]; int32_t moved;
                   int32_t push_blk, push_reach; };   // an element pushed to a block of T2 / of the dense rewrite: the block and o + z there (-1: none)
MCF_HD void mcf_bpl_slot(const McfView& v, const McfCtx& c, const McfBlkPlan& P, int32_t slot, int32_t p, int32_t nd, int32_t z,
                         McfBplOut* out) {
    const int32_t bs = v.blk_shift, bmask = (1 << bs) - 1;
    const int32_t a0 = c.t2_old, S = c.t2_size;
    int32_t* const tok_old = c.arena ? v.order[1] : v.order[0];
    int32_t* const psz_old = c.arena ? v.psz[1] : v.psz[0];
    int32_t* const tok_new = c.rebuild ? (c.arena ? v.order[0] : v.order[1]) : tok_old;
    int32_t* const psz_new = c.rebuild ? (c.arena ? v.psz[0] : v.psz[1]) : psz_old;
    const int32_t arena_new = (c.rebuild ? (c.arena ^ 1) : c.arena) ? MCF_LOC_ARENA : 0;
    McfBlkMeta* const bn = c.cur ? v.bmeta[0] : v.bmeta[1];
    out->keep_reach = 0; out->copy_reach[0] = 0; out->copy_reach[1] = 0; out->moved = 0; out->push_blk = -1; out->push_reach = 0;
    const bool in_t2 = p >= a0 && p < a0 + S;
    int32_t dslot = -1;
    if (in_t2) {
        int32_t dd;
        const int32_t j = mcf_bpl_new_pos(c, v.seg, p, &dd);
        v.pi[nd] += c.sigma;
        if (dd) v.node[nd].depth += dd;
        dslot = c.rebuild ? j : ((c.alloc_lo + MCF_BLK_COPIES) << bs) + (j - c.t2_new);
        out->push_blk = dslot >> bs; out->push_reach = (dslot & bmask) + z;   // (the caller raises that block's rrel: wave-wide on the device)
    } else if (c.rebuild) {
        dslot = p + mcf_bpl_shift(c, p);
        out->push_blk = dslot >> bs; out->push_reach = (dslot & bmask) + z;
    } else if (p >= P.r0lo && p < P.r0hi) {
        out->keep_reach = (slot & bmask) + z;
        return;
    } else {
        const int kd = (p >= P.ilo[0] && p < P.ihi[0]) ? 0 : 1;
        const int32_t o = p - (kd ? P.ilo[1] : P.ilo[0]);
        dslot = ((c.alloc_lo + kd) << bs) + o;
        out->copy_reach[kd] = o + z;
        out->moved = 1;
    }
    tok_new[dslot] = nd;
    psz_new[dslot] = z;
    psz_old[slot] = 0;                       // the old slot is empty from now on (size 0: the scan skips it)
    v.posbuf[0][nd] = dslot | arena_new;
    (void)tok_old; (void)bn;
}

// Is `node` (a neighbour of a T2 node) inside T2?  Asked by the reduced-cost patch while other lanes may be moving it:
// loc[] is one word, and a slot in a block this pivot handed out says what the node is (T2's blocks hold T2, the copy
// blocks do not); an old slot gives the old position, tested against T2's old interval.
MCF_HD bool mcf_bpl_in_t2(const McfView& v, const McfCtx& c, int32_t node) {
    const int32_t lw = v.posbuf[0][node];
    const int32_t sl = lw & MCF_LOC_SLOT, bs = v.blk_shift;
    if (c.rebuild) {
        if (((lw & MCF_LOC_ARENA) != 0) != (c.arena != 0)) return sl >= c.t2_new && sl < c.t2_new + c.t2_size;   // moved: dense, slot = new position
    } else if ((sl >> bs) >= c.alloc_lo) {
        return (sl >> bs) >= c.alloc_lo + MCF_BLK_COPIES;
    }
    const McfBlkMeta* bm = c.cur ? v.bmeta[1] : v.bmeta[0];
    const int32_t p = bm[sl >> bs].base + (sl & ((1 << bs) - 1));
    return p >= c.t2_old && p < c.t2_old + c.t2_size;
}

// The whole update of one pivot, scalar (CPU emulation; the kernels run the same per-block / per-slot functions with one
// lane per block, then one lane per slot of the blocks that have to be taken apart).
MCF_HD void mcf_bpl_update_seq(const McfView& v, const McfCtx& c) {
    const int32_t bs = v.blk_shift, B = 1 << bs;
    const McfBlkMeta* bm = c.cur ? v.bmeta[1] : v.bmeta[0];
    McfBlkMeta* bn = c.cur ? v.bmeta[0] : v.bmeta[1];
    int32_t* xa = c.arena ? v.bext[1] : v.bext[0];
    const int32_t* tok = c.arena ? v.order[1] : v.order[0];
    const int32_t* psz = c.arena ? v.psz[1] : v.psz[0];
    const int32_t nold = c.alloc_prev;   // blocks that existed before this pivot
    for (int32_t b = 0; b < nold; ++b) {
        const McfBlkMeta m = bm[b];
        const int32_t x = xa[b], beg = mcf_ext_beg(x), end = mcf_ext_end(x);
        if (m.base == MCF_BLK_FREE || end <= beg) { if (!c.rebuild) bn[b] = McfBlkMeta{MCF_BLK_FREE, 0}; continue; }
        const McfBlkPlan P = mcf_bpl_plan(c, m.base, m.base + beg, m.base + end);
        if (!c.rebuild && !P.touched && !(x & MCF_EXT_FLAG)) { bn[b] = McfBlkMeta{P.nbase, m.rrel}; continue; }
        int32_t keep = 0, cr[2] = {0, 0}, moved = 0;
        for (int32_t o = beg; o < end; ++o) {
            const int32_t slot = (b << bs) + o;
            McfBplOut r;
            if (!c.rebuild && !P.touched) { r.keep_reach = o + psz[slot]; r.copy_reach[0] = r.copy_reach[1] = 0; r.moved = 0; }   // re-index only
            else {
                mcf_bpl_slot(v, c, P, slot, m.base + o, tok[slot], psz[slot], &r);
                if (r.push_blk >= 0 && r.push_reach > bn[r.push_blk].rrel) bn[r.push_blk].rrel = r.push_reach;
            }
            if (r.keep_reach > keep) keep = r.keep_reach;
            if (r.copy_reach[0] > cr[0]) cr[0] = r.copy_reach[0];
            if (r.copy_reach[1] > cr[1]) cr[1] = r.copy_reach[1];
            moved += r.moved;
        }
        if (c.rebuild) continue;   // (the old arena is left empty; its records die with the flip)
        bn[b] = McfBlkMeta{P.nbase, keep};
        xa[b] = P.nbase == MCF_BLK_FREE ? 0 : mcf_ext_make(P.r0lo - m.base, P.r0hi - m.base);
        for (int kd = 0; kd < 2; ++kd) {
            if (P.ihi[kd] <= P.ilo[kd]) continue;
            const int32_t cb = c.alloc_lo + kd;
            bn[cb] = McfBlkMeta{P.ilo[kd] + mcf_bpl_shift(c, P.ilo[kd]), cr[kd]};
            xa[cb] = mcf_ext_make(0, P.ihi[kd] - P.ilo[kd]);
        }
        v.ctx->nodes_moved += moved;
    }
    (void)B;
}

// The whole cycle search + decision for single-threaded callers (CPU emulation): climb within the
// budget, then -- when the view carries psz[] -- the scan as a team of one.
MCF_HD void mcf_pivot_walk(const McfView& v, int64_t best_key, int64_t best_arc, int32_t rule) {
    MCF_PSTAMP(12);
    if (!mcf_pivot_begin(v, best_key, best_arc, rule)) return;
    MCF_PSTAMP(13);
    McfCycle cy;
    mcf_cycle_init(v, &cy);
    MCF_PSTAMP(14);
    if (!mcf_pivot_climb(v, mcf_view_paths(v), &cy, mcf_climb_budget(v, cy))) return;
    MCF_PSTAMP(15);
    if (cy.u != cy.w) {
        McfScanAcc acc;
        mcf_scan_init(&acc);
        mcf_pivot_scan(v, mcf_view_paths(v), 0, &cy, &acc, nullptr, 0, 0, 1);  // hit list: the scratch behind v.seg
        if (v.ctx->status != MCF_RUNNING) return;
    }
    mcf_pivot_decide(v, mcf_view_paths(v), cy);
    MCF_PSTAMP(16);
}

MCF_HD void mcf_pivot_finish(const McfView& v, const McfPaths& pp, int32_t lane, int32_t nlanes) {
    const McfCtx* c = v.ctx;
    const int32_t stage = c->stage;
    if (stage == 0) return;
    const int32_t e = c->pv_e, s = c->pv_s, n1 = c->pv_n1, n2 = c->pv_n2;
    const int64_t delta = c->pv_delta;
    if (c->wreset && v.dx) {  // Devex weight reset: the touched arcs go back to 1 (the caller clears wreset / wlist_n afterwards)
        const int32_t nw = c->wlist_n;
        for (int32_t i = lane; i < nw; i += nlanes) v.weight[v.dx->wlist[i]] = 1.0f;
    }

    // --- flow update along the cycle (simplex.py:1255-1283): distinct arcs, one per lane
    if (delta > 0) {
        for (int32_t i = lane; i < n1 + n2; i += nlanes) {
            const bool side1 = i < n1;
            const int32_t p = side1 ? pp.rec1[i].pred : pp.rec2[i - n1].pred;
            const bool up = (p & 1) != 0;
            // first side is walked against the flow, second side with it
            const int64_t dlt = (side1 == up) ? -delta : delta;
            if (pp.flow1) v.arcw[p >> 1].flow = (side1 ? pp.flow1[i] : pp.flow2[i - n1]) + dlt;  // read by the hit pass already
            else v.arcw[p >> 1].flow += dlt;
        }
        if (lane == 0) v.arcw[e].flow = c->pv_flow + (int64_t)s * delta;  // (read in mcf_pivot_begin: no load here)
    }
    if (stage == 1) {
        if (lane == 0) {
            v.state[e] = (int8_t)(-s);
            if (v.vkey) v.vkey[e] = 0;  // its reduced cost keeps its sign, the state flipped: no longer eligible
            mcf_mark_dirty(v, e);
        }
        return;
    }

    const int32_t result = c->pv_result, k = c->pv_k, v_in = c->pv_vin;
    const int32_t* stem = result == 1 ? pp.path1 : pp.path2;
    const McfNode* srec = result == 1 ? pp.rec1 : pp.rec2;
    const int32_t nstem_side = result == 1 ? n1 : n2;
    const int32_t* other = result == 1 ? pp.path2 : pp.path1;
    const McfNode* orec = result == 1 ? pp.rec2 : pp.rec1;
    const int32_t nother = result == 1 ? n2 : n1;
    const int32_t S = c->t2_size, b = c->t2_new;
    if (lane == 0) {
        v.state[e] = 0;
        mcf_mark_dirty(v, e);
        if (c->pv_leave < v.m) { v.state[c->pv_leave] = (int8_t)c->pv_leave_state; mcf_mark_dirty(v, c->pv_leave); }
    }
    // subtree sizes outside T2: the old ancestors of q lose S, v_in and its ancestors gain S
    // (join and above keep their size)
    // The position-space copies are written at the OLD position into BOTH buffers: a node outside the
    // affected range keeps its position (both buffers must agree there), one inside is moved -- together
    // with this value -- by the apply pass, which reads psz[cur] and overwrites psz[cur ^ 1].
    const int32_t* spos = result == 1 ? pp.ppos1 : pp.ppos2;
    const int32_t* opos = result == 1 ? pp.ppos2 : pp.ppos1;
    // blocked list: sizes live at the SLOT that holds the node (one copy per arena, moved by the update pass), the coarse
    // index is the block's rrel, and a shrunken subtree raises its block's re-index flag
    const bool bpl = MCF_HAS_BPL(v);
    const int32_t* sslot = result == 1 ? pp.slot1 : pp.slot2;
    const int32_t* oslot = result == 1 ? pp.slot2 : pp.slot1;
    int32_t* const zarena = bpl ? mcf_slot_sizes(v, c) : nullptr;
    McfBlkMeta* const bmc = bpl ? (c->cur ? v.bmeta[1] : v.bmeta[0]) : nullptr;
    int32_t* const bxa = bpl ? (c->arena ? v.bext[1] : v.bext[0]) : nullptr;
    const int32_t bs = v.blk_shift, bmask = (1 << bs) - 1;
    for (int32_t i = k + 1 + lane; i < nstem_side; i += nlanes) {
        v.node[stem[i]].size = srec[i].size - S;
        if (bpl) {
            const int32_t sl = sslot[i];
            zarena[sl] = srec[i].size - S;
            MCF_ATOMIC_OR32(&bxa[sl >> bs], MCF_EXT_FLAG);
        } else if (v.psz[0]) {
            const int32_t p = spos[i];
            v.psz[0][p] = srec[i].size - S; v.psz[1][p] = srec[i].size - S;
            if (v.reach) v.chg[i - k - 1] = p;  // its block's reach entry may now be too high: the apply pass re-indexes it
        }
    }
    for (int32_t i = lane; i < nother; i += nlanes) {
        v.node[other[i]].size = orec[i].size + S;
        if (bpl) {
            const int32_t sl = oslot[i];
            zarena[sl] = orec[i].size + S;
            MCF_ATOMIC_MAX32(&bmc[sl >> bs].rrel, (sl & bmask) + orec[i].size + S);  // must never be too low
        } else if (v.psz[0]) {
            const int32_t p = opos[i];
            v.psz[0][p] = orec[i].size + S; v.psz[1][p] = orec[i].size + S;
            if (v.reach) MCF_ATOMIC_MAX32(&v.reach[p >> MCF_REACH_SHIFT], p + orec[i].size + S);  // must never be too low
        }
    }

    // re-root T2 at u_in: reverse the stem and emit the block permutation.
    // Old layout: block(s_i) = [p_i, p_i + z_i), nested, s_k = q.  New layout: block(s_0), then
    // for i = 1..k: s_i + what precedes block(s_{i-1}) inside block(s_i), then what follows it;
    // piece i starts at b + z_{i-1} because pieces 0..i-1 are exactly old block(s_{i-1}).
    const int32_t base_depth = c->pv_vin_depth + 1;
    for (int32_t i = lane; i <= k; i += nlanes) {
        const McfNode r = srec[i];
        const int32_t p = spos[i];
        const int32_t dd = base_depth + i - r.depth;  // piece i moves from depth(s_i) to depth(v_in) + 1 + i
        McfNode nr;
        nr.depth = r.depth;  // the apply pass adds dd to every node of the piece, s_i included
        if (i == 0) {
            nr.parent = v_in;
            nr.pred = (e << 1) | c->pv_tail_in_t2;
            nr.size = S;
            v.seg[0] = McfSeg{b, p, r.size, dd};
        } else {
            const McfNode rp = srec[i - 1];
            const int32_t pp = spos[i - 1];
            nr.parent = stem[i - 1];
            nr.pred = rp.pred ^ 1;  // the arc s_{i-1} used to hang on now carries s_i: direction bit flips
            nr.size = S - rp.size;  // all of T2 except what stays below s_{i-1}
            const int32_t left = pp - p;                            // s_i itself + blocks before block(s_{i-1})
            const int32_t right = (p + r.size) - (pp + rp.size);   // blocks after it (may be empty)
            const int32_t dst = b + rp.size;
            v.seg[2 * i - 1] = McfSeg{dst, p, left, dd};
            v.seg[2 * i] = McfSeg{dst + left, pp + rp.size, right, dd};
        }
        v.node[stem[i]] = nr;
        if (bpl) zarena[sslot[i]] = nr.size;   // moved to its new slot by the update pass
        else if (v.psz[0]) { v.psz[0][p] = nr.size; v.psz[1][p] = nr.size; }  // moved to its new position by the apply pass
    }
    if (bpl) mcf_bpl_prepare(v, *c, lane, nlanes);
}

// Convenience for single-threaded callers (CPU emulation): the whole pivot.
MCF_HD void mcf_pivot_seq(const McfView& v, int64_t best_key, int64_t best_arc, int32_t rule) {
    mcf_pivot_walk(v, best_key, best_arc, rule);
    mcf_pivot_finish(v, mcf_view_paths(v), 0, 1);
}

// ---------------------------------------------------------------------------
// Data-parallel apply pass: new preorder position j  ->  old position.
// Valid for j in [ctx.lo, ctx.hi).
// ---------------------------------------------------------------------------
MCF_HD int32_t mcf_apply_source(const McfCtx& c, const McfSeg* seg, int32_t j, bool* in_t2, int32_t* ddepth) {
    const int32_t b = c.t2_new, S = c.t2_size, a0 = c.t2_old;
    if (j >= b && j < b + S) {
        *in_t2 = true;
        if (c.nseg == 1) {   // (the leaving arc is u_in's own tree arc: one piece, described by the control block alone)
            *ddepth = c.pv_dd0;
            return a0 + (j - b);
        }
        int32_t lo = 0, hi = c.nseg - 1;  // last segment with dst <= j
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (seg[mid].dst <= j) lo = mid; else hi = mid - 1;
        }
        *ddepth = seg[lo].ddepth;
        return seg[lo].src + (j - seg[lo].dst);
    }
    *in_t2 = false;
    *ddepth = 0;
    // untouched nodes slide over the gap T2 leaves behind
    return b <= a0 ? j - S : j + S;
}

// One element of the apply pass (the HIP kernel runs this for a grid-strided j).
// Returns the subtree size now stored at position j (0 when the view keeps none): the apply pass re-indexes reach[] from it.
MCF_HD int32_t mcf_apply_one(const McfView& v, const McfCtx& c, int32_t j) {
    // selects, not a runtime-indexed member array: a view held in registers must not be
    // forced into private memory
    const int32_t* src = c.cur ? v.order[1] : v.order[0];
    int32_t* dst = c.cur ? v.order[0] : v.order[1];
    int32_t* pnext = c.cur ? v.posbuf[0] : v.posbuf[1];
    const int32_t* zsrc = c.cur ? v.psz[1] : v.psz[0];
    int32_t* zdst = c.cur ? v.psz[0] : v.psz[1];
    if (j >= c.lo && j < c.hi) {
        bool in_t2;
        int32_t dd;
        const int32_t i = mcf_apply_source(c, v.seg, j, &in_t2, &dd);
        const int32_t nd = src[i];
        dst[j] = nd;
        pnext[nd] = j;
        int32_t z = 0;
        if (zsrc) { z = zsrc[i]; zdst[j] = z; }  // subtree sizes travel with their nodes (the finish pass ran before)
        if (in_t2) {
            v.pi[nd] += c.sigma;
            if (dd) v.node[nd].depth += dd;
        }
        return z;
    } else {
        const int32_t nd = src[j];
        dst[j] = nd;  // catch up on what the previous apply changed in the other copy
        pnext[nd] = j;
        int32_t z = 0;
        if (zsrc) { z = zsrc[j]; zdst[j] = z; }
        return z;
    }
}

// Scalar re-indexing of the coarse blocks one pivot touched (host emulation; the kernels do the same wave-wide inside
// their apply pass): every block that meets [lo, hi) or the catch-up range, and the blocks of the shrunken subtrees.
MCF_HD void mcf_reach_reindex_block(const McfView& v, const int32_t* zs, int32_t b) {
    int32_t m = 0;
    for (int32_t j = b << MCF_REACH_SHIFT; j < ((b + 1) << MCF_REACH_SHIFT) && j < v.n_nodes; ++j) {
        const int32_t e = j + zs[j];
        if (e > m) m = e;
    }
    v.reach[b] = m;
}
