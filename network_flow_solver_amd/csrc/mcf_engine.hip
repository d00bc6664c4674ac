// mcf_engine.hip -- MI355X (gfx950) network-simplex pivot engine: HIP kernels + C ABI.
//
// One pivot, by instance size (mcf_create picks; all paths share mcf_core.h and give the same pivot sequence):
//
//   k_solve_small   whole instance in LDS: one persistent workgroup prices, pivots (two lanes climb the cycle in
//                   lock step) and updates until the solve ends.
//   k_solve_mid     <= 1 536 nodes: one persistent workgroup over global (L2-resident) state -- prices a Devex
//                   block / re-prices the candidate list / sweeps a small arc list, pivots, permutes, patches.
//                   (both also as k_solve_small_batch / k_solve_mid_batch: MANY independent instances in one launch, one
//                   persistent workgroup = one CU per instance -- mcf_solve_batch)
//   otherwise three kernels per pivot on one stream, 64 pivots per captured hipGraph:
//     k_price_v     Dantzig sweep over 4-BYTE KEY CODES (one per arc, ordering like the violation -state * rc): HBM-bound at
//                   scale, 13.7 us for 16 M arcs; full-sweep Dantzig handles from 4 M arcs on.
//     k_price_rc    the same sweep over RESIDENT reduced costs (8 B rc + 1 B state per arc, +4 B Devex weight):
//                   per-lane best, DPP wave max, one candidate per workgroup.  Devex: the block comes from a granule
//                   table and moves / resizes under the reference's tuner.  Incremental from 4 M arcs: workgroups
//                   whose arcs did not change keep their candidate.
//                   (k_price: the same sweep by gathering pi[tail], pi[head]; mode 0 / parity hook.)
//                   Replaces simplex.py:498-617, simplex_pricing.py:97-137, 310-357, 375-542, simplex_adaptive.py:98-151.
//     k_pivot       one workgroup of 1024: final arg-max over the workgroup candidates (or the candidates
//                   all-gathered from the other ranks), cycle by a workgroup-wide scan over preorder positions
//                   (mcf_pivot_scan; through the coarse index `reach` on large trees; shallow end points are climbed),
//                   ratio test, then mcf_pivot_finish on all lanes (flow update, stem re-parenting, segment table).
//                   Replaces basis.py:178-241, simplex.py:1198-1425.
//     k_update      grid-wide block permutation of the preorder array + potential shift (pi += sigma) + position /
//                   size rewrite + re-indexing of the coarse index, and -- in other workgroups of the same launch --
//                   the patch of the resident reduced costs / key codes of the arcs incident to the re-hung subtree.
//                   Replaces the per-pivot BFS rebuild (basis.py:82-122) and _update_tree_sets (simplex.py:1103-1107).
//
// The host enqueues `batch_pivots` pivots (optionally as one captured hipGraph), then
// reads the small control block back once.  Kernels of a finished solve early-exit.
//
// gfx950 only; no CPU path: without a device every compute entry point returns
// MCF_E_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mcf.h"
#include "mcf_core.h"
#include "mcf_host.h"

namespace {

constexpr int kPriceThreads = 256;
constexpr int kPivotThreads = 1024;  // one workgroup: final arg-max, cycle search (climb by one lane / scan by all), finish
constexpr int kReduceThreads = 256;
constexpr int kMidMaxNodes = 1536;           // persistent single-workgroup loop (k_solve_mid): auto up to this many nodes (measured crossover
                                             // with the kernel-per-phase graph: +12..21 % at 512 / 1 024 nodes, even at 2 048, -25 % at 4 096;
                                             // profiles/r02_ab_engine_modes.txt) ...
constexpr int kMidMaxArcsPerPivot = MCF_TUNER_MAX_ARCS + 1024;  // ... and this many arcs priced inside the loop per pivot
constexpr int64_t kIncrementalMinArcs = (int64_t)1 << 22;  // incremental sweeps by default from this many arcs
constexpr int kScanMaxNodes = 1 << 27;  // (with the coarse index the scan's cost no longer grows with the tree: no practical limit)
constexpr int kApplyThreads = 256;
constexpr int kBplMinNodes = 200000;  // blocked preorder list from this many nodes on (auto)
constexpr int kMaxPriceBlocks = 2048;  // 8 workgroups per CU on 256 CUs
constexpr int kMaxApplyBlocks = 512;   // (1 024 workgroups: +5 % in the first 40 K pivots at 1 M nodes, nothing over the whole solve -- 124.5 s either way; MCF_APPLY_BLOCKS, scripts/ab_apply_blocks.py)
#ifndef MCF_PRICE_UNROLL
#define MCF_PRICE_UNROLL 2
#endif
constexpr int kUnroll = MCF_PRICE_UNROLL;  // 4-arc groups in flight per lane in k_price
constexpr int kCtxWords = (int)(sizeof(McfCtx) / 4);   // the control block is staged in LDS word by word

thread_local std::string g_create_error;

// ------------------------------------------------------------------ device helpers
// Wave-wide arg-max of (key, id): max-reduce the 64-bit key with xor shuffles, then resolve the
// id among the lanes that hold the maximum with a ballot + scalar readlanes (almost always one
// lane), instead of shuffling both words through every step.  Result is wave-uniform.
__device__ __forceinline__ int64_t readlane64(int64_t x, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)x >> 32), lane);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void wave_argmax(int64_t& key, int64_t& arc) {
    const int64_t mx = mcf_wave_max64(key);  // DPP reduction (mcf_core.h)
    uint64_t mask = __ballot(key == mx && key > 0);
    int64_t best = -1;
    while (mask) {  // uniform loop: one iteration unless several lanes tie on the key
        const int lane = __ffsll((unsigned long long)mask) - 1;
        const int64_t a = readlane64(arc, lane);
        if (best < 0 || a < best) best = a;
        mask &= mask - 1;
    }
    key = best < 0 ? 0 : mx;
    arc = best;
}

// Block-wide arg-max; result valid in thread 0 (and in all of wave 0).
template <int THREADS>
__device__ __forceinline__ void block_argmax(int64_t& key, int64_t& arc) {
    __shared__ int64_t s_key[THREADS / 64];
    __shared__ int64_t s_arc[THREADS / 64];
    wave_argmax(key, arc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_key[wave] = key; s_arc[wave] = arc; }
    __syncthreads();
    if (wave == 0) {
        const bool has = lane < THREADS / 64;
        key = has ? s_key[has ? lane : 0] : 0;
        arc = has ? s_arc[has ? lane : 0] : -1;
        wave_argmax(key, arc);
    }
}

// Candidate cache (McfView::candx): the record of this pricing workgroup's candidate -- end points, state and exact
// reduced cost as of this sweep; the update pass of every following pivot keeps it current.  Thread 0, after the arg-max.
__device__ __forceinline__ void write_candx(const McfView& v, int64_t arc) {
    if (!v.candx) return;
    McfCandX x;
    x.arc = arc; x.rc = 0; x.tail = 0; x.head = 0; x.state = 0; x.pad = 0;
    if (arc >= 0) {
        const int64_t e = arc & 0xffffffff;
        x.tail = v.tail[e]; x.head = v.head[e]; x.state = v.state[e];
        x.rc = (v.rcache && !v.rc_partial) ? v.rcache[e] : (int64_t)v.cost[e] + v.pi[x.tail] - v.pi[x.head];
    }
    v.candx[blockIdx.x] = x;
}

// ------------------------------------------------------------------ k_price
// XCD-aware sweep.  Workgroups are dealt to XCDs round-robin (blockIdx % 8 share an XCD --
// a performance observation, not a correctness assumption), so workgroup b sweeps head-bucket
// b % 8: its random pi[head] gathers stay inside one eighth of the potential array, which the
// XCD's private L2 holds; pi[tail] runs almost sequentially (arcs are tail-sorted inside a
// bucket) and coalesces.  Each lane owns groups of 4 consecutive arcs: one 16-byte load per
// SoA stream (tail, head, cost[, weight]) + 4 state bytes = 13 (17) B per arc.
// Inside the bucket the pass covers the rank's shard and, for Devex, the current block
// (mcf_bucket_slice), both resolved on the device so a captured graph can be replayed.
// FILTER restricts the pass to arcs whose ORIGINAL index lies in [f_lo, f_hi) (parity hook).
template <int RULE, bool FILTER>
__global__ __launch_bounds__(kPriceThreads) void k_price(McfView v, int64_t shard, int64_t shards, int use_block,
                                                          int64_t f_lo, int64_t f_hi, McfCand* __restrict__ cand) {
    int64_t key = 0, arc = -1;
    const McfCtx* c = v.ctx;
    // candidate-list rule: while minor iterations are pending the list of the last sweep must
    // survive, so the whole launch is a no-op (use_block == 2 marks that rule)
    if (use_block == 2 && (c->minor_left > 0 || c->status != MCF_RUNNING)) return;  // (also: a finished batch's
    // trailing launches must not wipe the list a resumed solve will want)
    if (c->status == MCF_RUNNING) {
        const int x = blockIdx.x & (MCF_NUM_BUCKETS - 1);
        const int64_t lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
        int64_t lo, hi;
        if (use_block == 1 && v.dx) mcf_devex_slice(v.dx, x, shard, shards, (int32_t)c->block_index, c->block_granules, &lo, &hi);
        else mcf_bucket_slice(v.bucket_off, x, shard, shards, 0, 1, &lo, &hi);
        const int64_t g_lo = lo >> 2, g_hi = (hi + 3) >> 2;  // groups of 4 arcs
        const int4* __restrict__ tail4 = reinterpret_cast<const int4*>(v.tail);
        const int4* __restrict__ head4 = reinterpret_cast<const int4*>(v.head);
        const int4* __restrict__ cost4 = reinterpret_cast<const int4*>(v.cost);
        const int32_t* __restrict__ state4 = reinterpret_cast<const int32_t*>(v.state);
        const float4* __restrict__ w4 = reinterpret_cast<const float4*>(v.weight);
        const int32_t* __restrict__ orig = v.orig;
        const int64_t* __restrict__ pi = v.pi;
        const int64_t stride = nlb * kPriceThreads;
        // kUnroll groups of 4 arcs in flight per lane: all SoA loads are issued first, then all
        // 8 * kUnroll potential gathers, so a wave exposes one memory round trip per phase
        // instead of one per group (the sweep is latency-bound otherwise).
        for (int64_t g0 = g_lo + lb * kPriceThreads + threadIdx.x; g0 < g_hi; g0 += stride * kUnroll) {
            int32_t st[kUnroll];
            int4 t[kUnroll], h[kUnroll], cc[kUnroll];
            float4 w[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t g = g0 + u * stride;
                const bool in = g < g_hi;
                st[u] = in ? state4[g] : 0;
                const int64_t gs = in ? g : g_lo;  // clamp: keep the loads unconditional and in range
                t[u] = tail4[gs];
                h[u] = head4[gs];
                cc[u] = cost4[gs];
                if (RULE == MCF_RULE_DEVEX_BLOCK) w[u] = w4[gs];
            }
            int64_t pt[kUnroll][4], ph[kUnroll][4];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int32_t ts[4] = {t[u].x, t[u].y, t[u].z, t[u].w}, hs[4] = {h[u].x, h[u].y, h[u].z, h[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool live = ((st[u] >> (8 * k)) & 0xff) != 0;
                    pt[u][k] = live ? pi[ts[k]] : 0;
                    ph[u][k] = live ? pi[hs[k]] : 0;
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int32_t cs[4] = {cc[u].x, cc[u].y, cc[u].z, cc[u].w};
                const float ws[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
                const int64_t g = g0 + u * stride;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t s = (int32_t)(int8_t)(st[u] >> (8 * k));
                    const int64_t i = (g << 2) + k;
                    if (s == 0 || i < lo || i >= hi) continue;
                    const int64_t rc = (int64_t)cs[k] + pt[u][k] - ph[u][k];
                    const int64_t viol = -(int64_t)s * rc;
                    if (viol <= 0) continue;
                    int64_t kk = mcf_dantzig_key(v, i, viol, s);
                    if (RULE == MCF_RULE_DEVEX_BLOCK) {
                        const double merit = ((double)viol * (double)viol) / (double)ws[k];
                        kk = __double_as_longlong(merit);
                    }
                    if (kk < key) continue;              // cannot win: skip the id lookup
                    const int32_t o = orig[i];
                    if (FILTER && (o < f_lo || o >= f_hi)) continue;
                    const int64_t id = mcf_pack_arc(RULE == MCF_RULE_DEVEX_BLOCK ? mcf_devex_tie_id(o, s) : o, i);
                    if (mcf_cand_better(kk, id, key, arc)) { key = kk; arc = id; }
                }
            }
        }
    }
    block_argmax<kPriceThreads>(key, arc);
    if (threadIdx.x == 0) { cand[blockIdx.x] = McfCand{key, arc}; write_candx(v, arc); }
}

// ------------------------------------------------------------------ k_price_rc: sweep over RESIDENT reduced costs
// Large instances keep rc[e] = cost + pi[tail] - pi[head] resident (8 B/arc, exact at all
// times, see k_rcupd).  The sweep is then a pure coalesced stream -- 8 B rc + 1 B state per arc
// (+4 B Devex weight), no gathers, no dependence on the potentials -- instead of 13 B/arc plus
// two random 8-byte gathers that each cost a 128-B L2->L1 line.  Same arc sets (bucket slices),
// same keys, same tie rule as k_price, so it selects the identical entering arc.
template <int RULE, bool FILTER, bool INC>
__global__ __launch_bounds__(kPriceThreads) void k_price_rc(McfView v, int64_t shard, int64_t shards, int use_block,
                                                             int64_t f_lo, int64_t f_hi, McfCand* __restrict__ cand,
                                                             int64_t* __restrict__ swept, const int32_t* __restrict__ blk_tab) {
    // On small and mid-size instances this kernel is a latency chain (control block -> arc data -> ids -> arg-max),
    // so: (1) the first batch of arc data is requested BEFORE the control block is looked at (full sweeps: the
    // slice does not depend on it); (2) the caller's arc id, needed only to break ties and to name the winner, is
    // looked up once per lane at the end instead of once per improvement.
    const McfCtx* c = v.ctx;
    const int x = blockIdx.x & (MCF_NUM_BUCKETS - 1);
    const int64_t lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
    using rc2_t = long2;                                  // two int64 reduced costs per 16-byte load
    const rc2_t* __restrict__ rc2 = reinterpret_cast<const rc2_t*>(v.rcache);
    const int32_t* __restrict__ state4 = reinterpret_cast<const int32_t*>(v.state);
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(v.weight);
    const int32_t* __restrict__ orig = v.orig;
    const int64_t stride = nlb * kPriceThreads;
    constexpr int U = 4;
    int32_t st[U];
    rc2_t ra[U], rb[U];
    float4 w[U];
    int64_t lo = 0, hi = 0, g_lo = 0, g_hi = 0;
    auto load_batch = [&](int64_t g0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t g = g0 + u * stride;
            const bool in = g < g_hi;
            const int64_t gs = in ? g : g_lo;  // clamp: keep the loads unconditional and in range
            st[u] = in ? state4[gs] : 0;
            ra[u] = rc2[2 * gs];
            rb[u] = rc2[2 * gs + 1];
            if (RULE == MCF_RULE_DEVEX_BLOCK) w[u] = w4[gs];
        }
    };
    const bool early = use_block != 1;  // full sweep: the slice is known without the control block
    int64_t g0 = 0;
    if (early) {
        // this rank's share of bucket x: a host-made table when the arc list is sharded (divisions otherwise)
        if (blk_tab) { lo = blk_tab[x * 2]; hi = blk_tab[x * 2 + 1]; }
        else mcf_bucket_slice(v.bucket_off, x, shard, shards, 0, 1, &lo, &hi);
        g_lo = lo >> 2; g_hi = (hi + 3) >> 2;  // groups of 4 arcs
        g0 = g_lo + lb * kPriceThreads + threadIdx.x;
        if (g0 < g_hi) load_batch(g0);
    }
    // candidate-list rule: while minor iterations are pending the list of the last sweep must
    // survive, so the whole launch is a no-op (use_block == 2 marks that rule)
    if (use_block == 2 && (c->minor_left > 0 || c->status != MCF_RUNNING)) return;  // (also: a finished batch's
    // trailing launches must not wipe the list a resumed solve will want)
    // incremental full sweep: no arc of this block changed since the block was last swept -> cand[blockIdx.x] still
    // holds.  (A finished batch's trailing launches must leave the candidates alone too: a resumed solve relies on them.)
    if (INC && v.dirty && use_block != 1 && (c->status != MCF_RUNNING || !v.dirty->flag[blockIdx.x])) return;
    int64_t key = 0, arc = -1;
    int64_t best_i = -1;  // engine index of this lane's best arc (its caller's id is looked up at the end)
    int32_t best_s = 0;   // its state (Devex: the direction takes part in the tie rule)
    if (c->status == MCF_RUNNING) {
        if (!early) {
            // Devex block search: this rank's share of block k of bucket x, from the host-made granule table (two
            // dependent 4-byte loads; the block size changes under the tuner, so blocks themselves cannot be tabulated)
            if (v.dx) mcf_devex_slice(v.dx, x, shard, shards, (int32_t)c->block_index, c->block_granules, &lo, &hi);
            else mcf_bucket_slice(v.bucket_off, x, shard, shards, 0, 1, &lo, &hi);
            g_lo = lo >> 2; g_hi = (hi + 3) >> 2;
            g0 = g_lo + lb * kPriceThreads + threadIdx.x;
            if (g0 < g_hi) load_batch(g0);
        }
        if (INC && swept && threadIdx.x == 0) {
            // accounting off the critical path: 32-bit arithmetic, and a no-return atomic on the workgroup's private slot
            // (nothing waits for it; one shared word would serialise 2 048 atomics, ~10 us per sweep).
            // This workgroup's share of the slice: groups g_lo + lb * 256 + [0, 256) + j * nlb * 256.
            const uint32_t ng = (uint32_t)(g_hi - g_lo), per = (uint32_t)nlb * 256u, full = ng / per, rem = ng - full * per;
            const uint32_t off = (uint32_t)lb * 256u;
            const uint32_t mine = full * 256u + (rem > off ? (rem - off < 256u ? rem - off : 256u) : 0u);
            atomicAdd(reinterpret_cast<unsigned long long*>(swept) + blockIdx.x, (unsigned long long)mine * 4ull);
        }
        while (g0 < g_hi) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t rcs[4] = {ra[u].x, ra[u].y, rb[u].x, rb[u].y};
                const float ws[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
                const int64_t g = g0 + u * stride;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t s = (int32_t)(int8_t)(st[u] >> (8 * k));
                    const int64_t i = (g << 2) + k;
                    if (s == 0 || i < lo || i >= hi) continue;
                    const int64_t viol = -(int64_t)s * rcs[k];
                    if (viol <= 0) continue;
                    int64_t kk = mcf_dantzig_key(v, i, viol, s);
                    if (RULE == MCF_RULE_DEVEX_BLOCK) {
                        const double merit = ((double)viol * (double)viol) / (double)ws[k];
                        kk = __double_as_longlong(merit);
                    }
                    if (kk < key) continue;
                    if (FILTER) {  // parity hook: the caller's index range decides eligibility
                        const int32_t o = orig[i];
                        if (o < f_lo || o >= f_hi) continue;
                    }
                    if (kk > key) { key = kk; best_i = i; best_s = s; }
                    else if (best_i < 0) { best_i = i; best_s = s; }
                    else {  // tie on the key (rare): Devex prefers a backward arc, then the lowest caller's index
                        const int32_t a = RULE == MCF_RULE_DEVEX_BLOCK ? mcf_devex_tie_id(orig[i], s) : orig[i];
                        const int32_t b = RULE == MCF_RULE_DEVEX_BLOCK ? mcf_devex_tie_id(orig[best_i], best_s) : orig[best_i];
                        if (a < b) { best_i = i; best_s = s; }
                    }
                }
            }
            g0 += stride * U;
            if (g0 < g_hi) load_batch(g0);
        }
        if (best_i >= 0) arc = mcf_pack_arc(RULE == MCF_RULE_DEVEX_BLOCK ? mcf_devex_tie_id(orig[best_i], best_s) : orig[best_i], best_i);
    }
    block_argmax<kPriceThreads>(key, arc);
    if (threadIdx.x == 0) {
        cand[blockIdx.x] = McfCand{key, arc};
        write_candx(v, arc);
        // every lane has passed the gate above (the arg-max has a barrier): the flag may go down now
        if (INC && v.dirty && use_block != 1 && c->status == MCF_RUNNING) v.dirty->flag[blockIdx.x] = 0;
    }
}

// ------------------------------------------------------------------ k_price_v: Dantzig sweep over COMPRESSED keys
// 4 bytes per arc (McfView::vkey, see mcf_core.h) instead of 8 B reduced cost + 1 B state: the sweep is HBM-bound at
// scale, so bytes are time.  Same arc sets, same arg-max, same tie rule as k_price_rc<DANTZIG>: a code orders like the
// violation it stands for and equal codes mean equal violations; an arc whose violation does not fit the code
// (MCF_VKEY_SAT) is compared by its exact value, fetched on the spot.  The workgroup's candidate carries the exact
// violation (one look-up per lane at the end, together with the caller's arc id), so k_pivot / k_reduce see what
// they always saw.
// NT: the code stream is loaded with the non-temporal hint -- for sweeps larger than the Infinity Cache, where no line
// is ever re-used before it is evicted (below that size the next sweep re-reads the lines from the cache: no hint).
template <bool INC, bool NT = false>
__global__ __launch_bounds__(kPriceThreads) void k_price_v(McfView v, int64_t shard, int64_t shards, int use_block,
                                                            McfCand* __restrict__ cand, int64_t* __restrict__ swept,
                                                            const int32_t* __restrict__ blk_tab) {
    const McfCtx* c = v.ctx;
    const int x = blockIdx.x & (MCF_NUM_BUCKETS - 1);
    const int64_t lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
    const int4* __restrict__ vk4 = reinterpret_cast<const int4*>(v.vkey);
    const int32_t* __restrict__ orig = v.orig;
    const int64_t stride = nlb * kPriceThreads;
    constexpr int U = 8;   // 16-byte loads in flight per lane (4 arcs each)
    int4 q[U];
    int64_t lo, hi;
    if (blk_tab) { lo = blk_tab[x * 2]; hi = blk_tab[x * 2 + 1]; }
    else mcf_bucket_slice(v.bucket_off, x, shard, shards, 0, 1, &lo, &hi);
    const int64_t g_lo = lo >> 2, g_hi = (hi + 3) >> 2;  // groups of 4 arcs
    auto load_batch = [&](int64_t g0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t g = g0 + u * stride;
            if (g >= g_hi) q[u] = make_int4(0, 0, 0, 0);
            else if (NT) {
                typedef int nt_int4 __attribute__((ext_vector_type(4)));
                const nt_int4 w = __builtin_nontemporal_load(reinterpret_cast<const nt_int4*>(vk4) + g);
                q[u] = make_int4(w.x, w.y, w.z, w.w);
            } else q[u] = vk4[g];
        }
    };
    int64_t g0 = g_lo + lb * kPriceThreads + threadIdx.x;
    if (g0 < g_hi) load_batch(g0);   // before the control block is looked at: a full sweep's slice does not depend on it
    if (use_block == 2 && (c->minor_left > 0 || c->status != MCF_RUNNING)) return;   // candidate list still live
    if (INC && v.dirty && (c->status != MCF_RUNNING || !v.dirty->flag[blockIdx.x])) return;  // clean block: candidate stands
    int32_t best = 0;      // best code of this lane (MCF_VKEY_SAT: its violation is not coded, best_x holds it)
    int64_t best_i = -1;   // its engine arc
    int64_t best_x = 0;    // exact violation while best == MCF_VKEY_SAT
    const int64_t bigm = v.vk_bigm;
    const int32_t half = v.vk_half;
    auto exact = [&](int64_t i) { return -(int64_t)v.state[i] * v.rcache[i]; };
    if (c->status == MCF_RUNNING) {
        if (INC && swept && threadIdx.x == 0) {
            const uint32_t ng = (uint32_t)(g_hi - g_lo), per = (uint32_t)nlb * 256u, full = ng / per, rem = ng - full * per;
            const uint32_t off = (uint32_t)lb * 256u;
            const uint32_t mine = full * 256u + (rem > off ? (rem - off < 256u ? rem - off : 256u) : 0u);
            atomicAdd(reinterpret_cast<unsigned long long*>(swept) + blockIdx.x, (unsigned long long)mine * 4ull);
        }
        while (g0 < g_hi) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t ks[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                const int64_t g = g0 + u * stride;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t kk = ks[k];
                    if (kk == 0 || (kk < best && best != MCF_VKEY_SAT)) continue;   // ineligible / cannot win
                    const int64_t i = (g << 2) + k;
                    if (i < lo || i >= hi) continue;
                    if (kk != MCF_VKEY_SAT && best != MCF_VKEY_SAT) {   // the common case: codes order like violations
                        if (kk > best) { best = kk; best_i = i; }
                        else if (orig[i] < orig[best_i]) best_i = i;    // equal codes = equal violations: lowest caller's index
                    } else {                                             // an arc that is not coded is involved: exact values
                        const int64_t xv = kk == MCF_VKEY_SAT ? exact(i) : mcf_vkey_decode(kk, bigm, half);
                        const int64_t cur = best_i < 0 ? 0 : (best == MCF_VKEY_SAT ? best_x : mcf_vkey_decode(best, bigm, half));
                        if (xv > cur || (xv == cur && best_i >= 0 && orig[i] < orig[best_i])) { best = kk; best_i = i; best_x = xv; }
                    }
                }
            }
            g0 += stride * U;
            if (g0 < g_hi) load_batch(g0);
        }
    }
    int64_t key = 0, arc = -1;
    if (best_i >= 0) {
        key = best == MCF_VKEY_SAT ? best_x : mcf_vkey_decode(best, bigm, half);   // the exact violation, as k_price_rc reports it
        arc = mcf_pack_arc(orig[best_i], best_i);
    }
    block_argmax<kPriceThreads>(key, arc);
    if (threadIdx.x == 0) {
        cand[blockIdx.x] = McfCand{key, arc};
        write_candx(v, arc);
        if (INC && v.dirty && c->status == MCF_RUNNING) v.dirty->flag[blockIdx.x] = 0;
    }
}

// ------------------------------------------------------------------ keeping the resident reduced costs exact
// The swap shifted the potentials of the re-hung subtree T2 by sigma, so an arc changes iff exactly
// one end point is in T2: +sigma when its tail is inside, -sigma when its head is.  16 lanes per T2
// node walk its CSR adjacency; every such arc is visited exactly once (from its inside end), so no
// atomics.  The pass lives in k_update below.
constexpr int kRcupdThreads = 256;
constexpr int kMaxRcupdBlocks = 1024;
constexpr int kCandxBlocks = 8;   // workgroups of the dense update launches that keep the candidate cache current (2 048 entries)

// ------------------------------------------------------------------ k_update = tree/potential apply + reduced-cost update in ONE launch
// Resident-rc engines only.  The two halves touch disjoint data once T2 membership is tested
// against the OLD positions (posbuf[cur], which this launch never writes) and T2 is enumerated
// through the OLD order (order[cur][a0 .. a0+S)): the apply half writes order[cur^1],
// posbuf[cur^1], pi and the depths of T2; the update half writes rcache.  One launch boundary less per pivot.
// `c` is the control block (kernel argument memory / LDS); the pass is spread over `stride` lanes of which this
// one is `tid`, and over `ngroups` 16-lane groups of which this lane belongs to `group` (sub-lane `sub`).
__device__ __forceinline__ int32_t wave_max32(int32_t x) { return (int32_t)mcf_wave_max64((int64_t)x); }  // DPP steps, no LDS crossbar

// `tid` of `stride` lanes, both multiples of 64 apart: whole waves.  With a coarse index (v.reach) every wave takes
// aligned blocks of 64 positions, so that it holds all 64 new subtree sizes of a block and re-indexes reach[] in the
// same pass (one wave max + one store per block).
__device__ __forceinline__ void apply_pass(const McfView& v, const McfCtx& c, int64_t tid, int64_t stride) {
    const int32_t lo = c.lo, hi = c.hi, plo = c.prev_lo, phi = c.prev_hi;
    if (!v.reach) {
        for (int64_t j = lo + tid; j < hi; j += stride) mcf_apply_one(v, c, (int32_t)j);
        for (int64_t j = plo + tid; j < phi; j += stride)
            if (j < lo || j >= hi) mcf_apply_one(v, c, (int32_t)j);
        return;
    }
    const int32_t* zsrc = c.cur ? v.psz[1] : v.psz[0];  // the stable old view (outside [lo, hi) old == new)
    const int32_t lane = (int32_t)(tid & 63);
    const int64_t gw = tid >> 6, nw = stride >> 6;
    const int32_t B0 = lo >> MCF_REACH_SHIFT, B1 = hi > lo ? (hi - 1) >> MCF_REACH_SHIFT : B0 - 1;
    const int32_t P0 = plo >> MCF_REACH_SHIFT, P1 = phi > plo ? (phi - 1) >> MCF_REACH_SHIFT : P0 - 1;
    // Up to KB blocks of a wave are permuted first and re-indexed afterwards: the sizes a block's re-indexing needs are
    // loaded by the permutation itself, and using them right away would make every block wait for its own loads
    // (measured at 1 M nodes: k_update 8.6 -> 11.1 us); deferred, the loads of all KB blocks are in flight together.
    constexpr int KB = 4;
    auto blocks = [&](int64_t first, int64_t last, int64_t skip_lo, int64_t skip_hi) {
        for (int64_t b0 = first + gw; b0 <= last; b0 += nw * KB) {
            int32_t z[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int64_t b = b0 + k * nw;
                z[k] = 0;
                if (b > last || (b >= skip_lo && b <= skip_hi)) continue;   // (wave-uniform)
                const int32_t j = ((int32_t)b << MCF_REACH_SHIFT) + lane;
                if (j < v.n_nodes) {
                    const bool in_cur = j >= lo && j < hi, in_prev = j >= plo && j < phi;
                    z[k] = (in_cur || in_prev) ? mcf_apply_one(v, c, j) : zsrc[j];
                }
            }
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const int64_t b = b0 + k * nw;
                if (b > last || (b >= skip_lo && b <= skip_hi)) continue;
                const int32_t j = ((int32_t)b << MCF_REACH_SHIFT) + lane;
                const int32_t m = wave_max32(j < v.n_nodes ? j + z[k] : 0);
                if (lane == 0) v.reach[b] = m;
            }
        }
    };
    blocks(B0, B1, 1, 0);        // the moved range
    blocks(P0, P1, B0, B1);      // the catch-up range, minus what the first call covered
    // shrunken subtrees outside both ranges: their blocks from the stable view (the finish pass wrote the new sizes
    // into both copies)
    for (int64_t t = gw; t < c.nchg; t += nw) {
        const int32_t b = v.chg[t] >> MCF_REACH_SHIFT;
        if ((b < B0 || b > B1) && (b < P0 || b > P1)) {
            const int32_t j = (b << MCF_REACH_SHIFT) + lane;
            const int32_t m = wave_max32(j < v.n_nodes ? j + zsrc[j] : 0);
            if (lane == 0) v.reach[b] = m;
        }
    }
}

__device__ __forceinline__ void rcupd_pass(const McfView& v, const McfCtx& c, int64_t group, int64_t ngroups, int32_t sub) {
    if (!v.rcache) return;
    const int32_t a0 = c.t2_old, S = c.t2_size;
    const int64_t sigma = c.sigma;
    const int32_t* __restrict__ ord = c.cur ? v.order[1] : v.order[0];     // old order
    const int32_t* __restrict__ pold = c.cur ? v.posbuf[1] : v.posbuf[0];  // old positions
    const int64_t* __restrict__ adj_off = v.adj_off;
    const int64_t* __restrict__ adj = v.adj;
    int64_t* __restrict__ rcache = v.rcache;
    if (c.pv_t2n == 1) {
        // [r3] T2 is the one node the pivot kernel named, and its adjacency range came along with the control block: control
        // block -> adjacency -> reduced cost, instead of -> order -> adjacency offsets -> adjacency -> other end's position ->
        // reduced cost (the longest chain of the launch on the first pivots of a cold start, when almost every T2 is one node)
        if (group != 0) return;
        const int32_t node = c.pv_t2node;
        for (int64_t p = c.pv_adj[0] + sub; p < c.pv_adj[1]; p += 16) {
            const int64_t ent = adj[p];
            if ((int32_t)(ent >> 32) == node) continue;  // a loop: both ends inside T2
            const int32_t e = (int32_t)((uint32_t)ent >> 1);
            const int64_t r = rcache[e] + ((ent & 1) ? sigma : -sigma);
            rcache[e] = r;
            if (v.vkey) v.vkey[e] = mcf_vkey(-(int64_t)v.state[e] * r, v.vk_bigm, v.vk_half);
            mcf_mark_dirty(v, e);
        }
        return;
    }
    for (int64_t t = group; t < S; t += ngroups) {
        const int32_t u = ord[a0 + t];
        const int64_t beg = adj_off[u], end = adj_off[u + 1];
        for (int64_t p = beg + sub; p < end; p += 16) {
            const int64_t ent = adj[p];
            const int32_t pw = pold[(int32_t)(ent >> 32)];
            if (pw >= a0 && pw < a0 + S) continue;  // both ends inside T2: unchanged
            const int32_t e = (int32_t)((uint32_t)ent >> 1);
            const int64_t r = rcache[e] + ((ent & 1) ? sigma : -sigma);
            rcache[e] = r;
            // the compressed key follows (the finish pass has already given the entering / leaving arc its new state)
            if (v.vkey) v.vkey[e] = mcf_vkey(-(int64_t)v.state[e] * r, v.vk_bigm, v.vk_half);
            mcf_mark_dirty(v, e);
        }
    }
}

// Candidate cache: every listed arc's reduced cost follows the potential shift of the re-hung subtree (+sigma when its tail
// is inside T2, -sigma when its head is), and the leaving arc's entry takes its new state.  Membership against the OLD view,
// like the reduced-cost patch; a single-node T2 whose node the pivot kernel named needs no look-up at all.
__device__ __forceinline__ void candx_pass(const McfView& v, const McfCtx& c, int32_t tid, int32_t stride) {
    if (!v.candx) return;
    const bool bpl = MCF_HAS_BPL(v);
    const int32_t a0 = c.t2_old, S = c.t2_size;
    const int32_t* __restrict__ pold = c.cur ? v.posbuf[1] : v.posbuf[0];  // dense array: old positions
    const bool one = c.pv_t2n == 1;   // (either layout)
    for (int32_t i = tid; i < v.ncandx; i += stride) {
        McfCandX x = v.candx[i];
        if (x.arc < 0) continue;
        bool tin, hin;
        if (one) { tin = x.tail == c.pv_t2node; hin = x.head == c.pv_t2node; }
        else if (bpl) { tin = mcf_bpl_in_t2(v, c, x.tail); hin = mcf_bpl_in_t2(v, c, x.head); }
        else { const int32_t pt = pold[x.tail], ph = pold[x.head]; tin = pt >= a0 && pt < a0 + S; hin = ph >= a0 && ph < a0 + S; }
        const bool leave = (int32_t)(x.arc & 0xffffffff) == c.pv_leave;
        if (tin == hin && !leave) continue;
        if (tin != hin) x.rc += tin ? c.sigma : -c.sigma;
        if (leave) x.state = c.pv_leave_state;
        v.candx[i] = x;
    }
}

// The two halves are independent, and each is a chain of dependent loads: different workgroups run them side
// by side (the first `apply_blocks` the permutation, the others the reduced-cost patch) instead of one after the other.
template <bool MARK>  // MARK: the handle prices incrementally, the patched arcs' blocks are flagged
__global__ __launch_bounds__(kRcupdThreads) void k_update(McfView g, int apply_blocks) {
    McfView v = g;
    if (!MARK) v.dirty = nullptr;  // folds the marking away
    const McfCtx c = *v.ctx;  // uniform: scalar loads
    if (!c.apply) return;
    const int b = blockIdx.x;
    const int xb = v.candx ? kCandxBlocks : 0;   // the last workgroups keep the candidate cache current
    if (b < apply_blocks) {
        apply_pass(v, c, (int64_t)b * kRcupdThreads + threadIdx.x, (int64_t)apply_blocks * kRcupdThreads);
    } else if (b >= (int)gridDim.x - xb) {
        candx_pass(v, c, (b - ((int)gridDim.x - xb)) * kRcupdThreads + (int)threadIdx.x, xb * kRcupdThreads);
    } else {
        const int64_t rb = b - apply_blocks, nrb = (int64_t)gridDim.x - apply_blocks - xb;
        rcupd_pass(v, c, rb * (kRcupdThreads / 16) + (threadIdx.x >> 4), nrb * (kRcupdThreads / 16), threadIdx.x & 15);
    }
}


// ------------------------------------------------------------------ k_update_bpl: the update over the BLOCKED preorder list
// (mcf_core.h "blocked preorder list").  Workgroups 0 .. G-1, three phases:
//   A  one lane per physical block handed out before this pivot (block b belongs to workgroup b % G, so that the
//      consecutive blocks of a large subtree spread over the grid): read {base, rrel} and the live range, write the record's
//      new copy -- for almost every block just base + shift -- or list the block as one to take apart;
//   B  one wave per listed block, one lane per slot: T2's elements go to their new blocks (segment table, potential and
//      depth shift on the way), cut-off runs to the copy blocks, the block's records are rewritten; the T2 nodes met are
//      listed in LDS;
//   C  16 lanes per listed T2 node walk its adjacency and patch the resident reduced costs / key codes of the arcs whose
//      other end is outside T2 (mcf_bpl_in_t2: valid whether or not that node has been moved yet).
// Workgroup G (the DIRECT one) skips A: the pivot kernel already knows two blocks that have to be taken apart -- the one T2
// starts in and the new parent's -- so this workgroup starts on them the moment the control block has arrived, with the
// segment table in LDS.  Almost every pivot re-hangs a handful of nodes that sit in that one block: then the whole launch
// is   control block -> slots -> adjacency offsets -> adjacency -> reduced costs   (five dependent round trips), and the
// membership test of the patch is a look-up in the LDS list instead of two more loads per arc.
// With ctx.rebuild every block is taken apart (no direct workgroup) and every element lands at slot = new position in the
// other arena.
constexpr int kBplThreads = 1024;
constexpr int kBplTouchedCap = 2048;   // blocks one workgroup can list (mcf_create sizes the grid so that a workgroup owns no more)
constexpr int kBplT2Cap = 8192;        // T2 nodes one workgroup can list (more: patched on the spot by the lane that moved them)
constexpr int kBplSegLds = 1024;       // segment-table entries a workgroup stages in LDS (one per thread) before it moves T2 elements
constexpr int kBplListMember = 32;     // a T2 this small that sits in one block: membership by scanning the LDS list

// `whole` != null: the listed nodes are ALL of T2 (n of them): the other end is inside iff it is in the list
template <bool MARK>
__device__ __forceinline__ void bpl_patch_node(const McfView& v, const McfCtx& c, int32_t u, int32_t sub, int32_t nsub,
                                               const int32_t* whole = nullptr, int32_t n = 0) {
    const int64_t sigma = c.sigma;
    const int64_t beg = v.adj_off[u], end = v.adj_off[u + 1];
    int64_t* __restrict__ rcache = v.rcache;
    for (int64_t p = beg + sub; p < end; p += nsub) {
        const int64_t ent = v.adj[p];
        const int32_t other = (int32_t)(ent >> 32);
        bool inside;
        if (whole) { inside = false; for (int32_t q = 0; q < n; ++q) inside |= whole[q] == other; }
        else inside = mcf_bpl_in_t2(v, c, other);
        if (inside) continue;  // both ends inside T2: unchanged
        const int32_t e = (int32_t)((uint32_t)ent >> 1);
        const int64_t r = rcache[e] + ((ent & 1) ? sigma : -sigma);
        rcache[e] = r;
        if (v.vkey) v.vkey[e] = mcf_vkey(-(int64_t)v.state[e] * r, v.vk_bigm, v.vk_half);
        if (MARK) mcf_mark_dirty(v, e);
    }
}

// One wave takes block b apart (phase B).  `segv`: the view with the segment table wherever it is cheapest to search.
template <bool MARK, bool RC>
__device__ __forceinline__ void bpl_block(const McfView& v, const McfView& segv, const McfCtx& c, McfCtx* gctx, int32_t b, int32_t lane,
                                          int32_t* s_t2, int32_t* s_nn, bool list = true, int32_t t2_cap = kBplT2Cap) {
    const int32_t bs = v.blk_shift;
    const McfBlkMeta* __restrict__ bm = c.cur ? v.bmeta[1] : v.bmeta[0];
    McfBlkMeta* __restrict__ bn = c.cur ? v.bmeta[0] : v.bmeta[1];
    int32_t* __restrict__ xa = c.arena ? v.bext[1] : v.bext[0];
    const int32_t* __restrict__ tok = c.arena ? v.order[1] : v.order[0];
    const int32_t* __restrict__ psz = c.arena ? v.psz[1] : v.psz[0];
    // the slots are requested together with the block's record (they do not depend on it): one round trip
    constexpr int kChunks = 4;   // 64-slot chunks held in registers (blocks up to 256 slots; larger ones loop on)
    int32_t zr[kChunks], nr[kChunks];
#pragma unroll
    for (int q = 0; q < kChunks; ++q) {
        const int32_t o = q * 64 + lane;
        const bool in = o < (1 << bs);
        zr[q] = in ? psz[(b << bs) + o] : 0;
        nr[q] = in ? tok[(b << bs) + o] : 0;
    }
    const McfBlkMeta m = bm[b];
    const int32_t x = xa[b], beg = mcf_ext_beg(x), end = mcf_ext_end(x);
    if (m.base == MCF_BLK_FREE || end <= beg) { if (!c.rebuild && lane == 0) bn[b] = McfBlkMeta{MCF_BLK_FREE, 0}; return; }
    const McfBlkPlan P = mcf_bpl_plan(c, m.base, m.base + beg, m.base + end);
    if (!c.rebuild && !P.touched && !(x & MCF_EXT_FLAG)) { if (lane == 0) bn[b] = McfBlkMeta{P.nbase, m.rrel}; return; }
    const bool reindex_only = !c.rebuild && !P.touched;
    int32_t keep = 0, cr0 = 0, cr1 = 0;
    for (int32_t o0 = beg & ~63; o0 < end; o0 += 64) {
        const int32_t o = o0 + lane, q = o0 >> 6;
        int32_t pblk = -1, pval = 0;   // this lane's pushed element: destination block and o + z there
        if (o >= beg && o < end) {
            const int32_t slot = (b << bs) + o;
            int32_t z, nd;
            if (q < kChunks) { z = q == 0 ? zr[0] : (q == 1 ? zr[1] : (q == 2 ? zr[2] : zr[3])); nd = q == 0 ? nr[0] : (q == 1 ? nr[1] : (q == 2 ? nr[2] : nr[3])); }
            else { z = psz[slot]; nd = tok[slot]; }
            if (reindex_only) { keep = keep > o + z ? keep : o + z; }
            else {
                const int32_t p = m.base + o;
                McfBplOut r;
                mcf_bpl_slot(segv, c, P, slot, p, nd, z, &r);
                pblk = r.push_blk; pval = r.push_reach;
                keep = keep > r.keep_reach ? keep : r.keep_reach;
                cr0 = cr0 > r.copy_reach[0] ? cr0 : r.copy_reach[0];
                cr1 = cr1 > r.copy_reach[1] ? cr1 : r.copy_reach[1];
                if (RC && list && p >= c.t2_old && p < c.t2_old + c.t2_size) {
                    const int32_t li = atomicAdd(s_nn, 1);
                    if (li < t2_cap) s_t2[li] = nd;
                    else bpl_patch_node<MARK>(v, c, nd, 0, 1);   // list full: this lane patches the node's arcs itself
                }
            }
        }
        // the pushed elements raise the rrel of the blocks they land in: one atomic per destination block and chunk (consecutive
        // old positions mostly land in one or two blocks), not one per element -- thousands of them on a few dozen words otherwise
        for (uint64_t todo = __ballot(pblk >= 0); todo;) {
            const int32_t b0 = __builtin_amdgcn_readlane(pblk, (int)__ffsll((unsigned long long)todo) - 1);
            const bool mine = pblk == b0;
            const int32_t mx = wave_max32(mine ? pval : 0);
            if (lane == 0) atomicMax(&bn[b0].rrel, mx);
            todo &= ~__ballot(mine);
        }
    }
    if (c.rebuild) return;   // (the old arena is left empty; its records die with the flip)
    keep = wave_max32(keep); cr0 = wave_max32(cr0); cr1 = wave_max32(cr1);
    if (lane == 0) {
        bn[b] = McfBlkMeta{P.nbase, keep};
        xa[b] = P.nbase == MCF_BLK_FREE ? 0 : mcf_ext_make(P.r0lo - m.base, P.r0hi - m.base);
        int32_t nm = 0;
#pragma unroll
        for (int kd = 0; kd < 2; ++kd) {
            const int32_t ilo = kd ? P.ilo[1] : P.ilo[0], ihi = kd ? P.ihi[1] : P.ihi[0];
            if (ihi <= ilo) continue;
            const int32_t cb = c.alloc_lo + kd;
            bn[cb] = McfBlkMeta{ilo + mcf_bpl_shift(c, ilo), kd ? cr1 : cr0};
            xa[cb] = mcf_ext_make(0, ihi - ilo);
            nm += ihi - ilo;
        }
        if (nm) atomicAdd(reinterpret_cast<unsigned long long*>(&gctx->nodes_moved), (unsigned long long)nm);
    }
}

template <bool MARK, bool RC>
__global__ __launch_bounds__(kBplThreads) void k_update_bpl(McfView g, int G) {
    __shared__ McfCtx cs;
    __shared__ int32_t s_nt, s_nn;
    __shared__ int32_t s_tb[kBplTouchedCap];
    __shared__ int32_t s_t2[RC ? kBplT2Cap : 1];
    __shared__ McfSeg s_seg[kBplSegLds];
    const bool direct = (int32_t)blockIdx.x == G;
    if (threadIdx.x < kCtxWords) reinterpret_cast<int32_t*>(&cs)[threadIdx.x] = reinterpret_cast<const int32_t*>(g.ctx)[threadIdx.x];
    // the direct workgroup will search the segment table for sure: it fetches it together with the control block (the scratch
    // holds 2 n + 2 entries: reading the first 64 is always in bounds; longer tables are staged below, once nseg is known)
    if (direct && threadIdx.x >= 128 && threadIdx.x < 128 + 64) s_seg[threadIdx.x - 128] = g.seg[threadIdx.x - 128];
    if (threadIdx.x == 0) { s_nt = 0; s_nn = 0; }
    __syncthreads();
    const McfCtx& c = cs;
    if (!c.apply) return;
    McfView v = g;
    if (!MARK) v.dirty = nullptr;
    if ((int32_t)blockIdx.x > G) { candx_pass(v, c, (int32_t)threadIdx.x, kBplThreads); return; }   // the candidate cache's workgroup
    const int32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t known0 = c.rebuild ? -1 : c.t2_blk, known1 = c.rebuild ? -1 : c.ins_blk;   // taken apart by the direct workgroup
    if (direct) {
        if (c.rebuild) return;
        McfView segv = v;
        if (c.nseg > 64 && c.nseg <= kBplSegLds) {   // (uniform) a longer table: the rest of it, one entry per thread
            if ((int32_t)threadIdx.x >= 64 && (int32_t)threadIdx.x < c.nseg) s_seg[threadIdx.x] = g.seg[threadIdx.x];
            __syncthreads();
        }
        if (c.nseg <= kBplSegLds) segv.seg = s_seg;
        if (RC && c.pv_t2n == 1) {
            // T2 is one node and the pivot kernel sent its adjacency range along: the patch needs no look-up at all
            // (control block -> adjacency -> reduced costs); the node has no arc to itself, so every incident arc changes
            if (wave == 0) bpl_block<MARK, false>(v, segv, c, g.ctx, known0, lane, s_t2, &s_nn);
            else if (wave == 1 && known1 >= 0 && known1 != known0) bpl_block<MARK, false>(v, segv, c, g.ctx, known1, lane, s_t2, &s_nn);
            else if (wave >= 2) {
                const int64_t sigma = c.sigma;
                for (int64_t p = c.pv_adj[0] + (threadIdx.x - 128); p < c.pv_adj[1]; p += kBplThreads - 128) {
                    const int64_t ent = v.adj[p];
                    const int32_t e = (int32_t)((uint32_t)ent >> 1);
                    const int64_t r = v.rcache[e] + ((ent & 1) ? sigma : -sigma);
                    v.rcache[e] = r;
                    if (v.vkey) v.vkey[e] = mcf_vkey(-(int64_t)v.state[e] * r, v.vk_bigm, v.vk_half);
                    if (MARK) mcf_mark_dirty(v, e);
                }
            }
            return;
        }
        if (wave == 0) bpl_block<MARK, RC>(v, segv, c, g.ctx, known0, lane, s_t2, &s_nn);
        else if (wave == 1 && known1 >= 0 && known1 != known0) bpl_block<MARK, RC>(v, segv, c, g.ctx, known1, lane, s_t2, &s_nn);
        if (!RC) return;
        __syncthreads();
        const int32_t nn = s_nn < kBplT2Cap ? s_nn : kBplT2Cap;
        const bool whole = nn == c.t2_size && nn <= kBplListMember;   // all of T2 sat in the known block
        for (int32_t t = threadIdx.x >> 4; t < nn; t += kBplThreads / 16)
            bpl_patch_node<MARK>(v, c, s_t2[t], threadIdx.x & 15, 16, whole ? s_t2 : nullptr, nn);
        return;
    }
    const McfBlkMeta* __restrict__ bm = c.cur ? v.bmeta[1] : v.bmeta[0];
    McfBlkMeta* __restrict__ bn = c.cur ? v.bmeta[0] : v.bmeta[1];
    const int32_t* __restrict__ xa = c.arena ? v.bext[1] : v.bext[0];
    const int32_t nold = c.alloc_prev;
    // ---- A
    for (int32_t q = threadIdx.x;; q += kBplThreads) {
        const int32_t b = q * G + (int32_t)blockIdx.x;
        if (b >= nold) break;
        if (b == known0 || b == known1) continue;
        const McfBlkMeta m = bm[b];
        const int32_t x = xa[b], beg = mcf_ext_beg(x), end = mcf_ext_end(x);
        if (m.base == MCF_BLK_FREE || end <= beg) { if (!c.rebuild) bn[b] = McfBlkMeta{MCF_BLK_FREE, 0}; continue; }
        bool apart = c.rebuild != 0 || (x & MCF_EXT_FLAG) != 0;
        if (!apart) {
            const McfBlkPlan P = mcf_bpl_plan(c, m.base, m.base + beg, m.base + end);
            if (P.touched) apart = true; else bn[b] = McfBlkMeta{P.nbase, m.rrel};
        }
        if (apart) {
            const int32_t slot = atomicAdd(&s_nt, 1);
            if (slot < kBplTouchedCap) s_tb[slot] = b;   // (cannot overflow: grid sized by mcf_create)
        }
    }
    __syncthreads();
    // ---- B
    const int32_t nt = s_nt < kBplTouchedCap ? s_nt : kBplTouchedCap;
    if (nt == 0) return;   // (uniform: the usual case for every workgroup but a few)
    // the elements of T2 look their new position up in the segment table (a binary search: log2(nseg) dependent loads each):
    // from LDS when the table fits (a stem of <= 511 nodes), staged by the whole workgroup in one round trip
    McfView segv = v;
    if (c.nseg <= kBplSegLds) {
        if ((int32_t)threadIdx.x < c.nseg) s_seg[threadIdx.x] = g.seg[threadIdx.x];
        __syncthreads();
        segv.seg = s_seg;
    }
    for (int32_t t = wave; t < nt; t += kBplThreads / 64) bpl_block<MARK, RC>(v, segv, c, g.ctx, s_tb[t], lane, s_t2, &s_nn);
    if (!RC) return;
    __syncthreads();
    // ---- C
    const int32_t nn = s_nn < kBplT2Cap ? s_nn : kBplT2Cap;
    for (int32_t t = threadIdx.x >> 4; t < nn; t += kBplThreads / 16) bpl_patch_node<MARK>(v, c, s_t2[t], threadIdx.x & 15, 16);
}

// ------------------------------------------------------------------ k_reduce (multi-GPU: local best -> 16 bytes)
__global__ __launch_bounds__(kReduceThreads) void k_reduce(const McfCand* __restrict__ cand, int ncand,
                                                            McfCand* __restrict__ out) {
    int64_t key = 0, arc = -1;
    for (int i = threadIdx.x; i < ncand; i += kReduceThreads) {
        const McfCand cd = cand[i];
        if (mcf_cand_better(cd.key, cd.arc, key, arc)) { key = cd.key; arc = cd.arc; }
    }
    block_argmax<kReduceThreads>(key, arc);
    if (threadIdx.x == 0) *out = McfCand{key, arc};
}

// ------------------------------------------------------------------ k_pivot
// One workgroup; latency-bound, so the point is to keep the chain of dependent global round trips short:
// the control block is staged in LDS for the whole kernel (begin / decide / finish read and write dozens
// of its fields), the candidate loads are in flight while it arrives, the cycle comes from the
// position-space scan (~4 dependent round trips whatever its length) with the hit list in LDS.
#ifdef MCF_STAMPS
__device__ unsigned long long g_pivot_stamps[24];
#endif
constexpr int kHitsLds = 4096;  // hit-list entries kept in LDS (a longer cycle spills to global scratch)
constexpr int kSmallPath = 512;  // cycles up to this many nodes are recorded in LDS instead of the global path scratch

// LDS state of one pivoting workgroup (k_pivot, k_solve_mid)
struct PivotShared {
    McfCtx ctx;
    McfCycle cy;
    McfScanAcc acc;
    int go;  // 0 = nothing to pivot on, 1 = the climb reached the join, 2 = the scan has to finish the cycle
    // candidate cache: the winning entry (its index in the list, -1: none / not cached) and its record
    int64_t win_arc;
    int32_t win_idx;
    McfCandX winx;
    int32_t dirty_hdr[20];   // incremental sweeps: nlb, lo[8], hi[8] of McfDirty (the marking's look-ups then stay in LDS)
    McfHit hits[kHitsLds];
    int32_t path[2][kSmallPath], ppos[2][kSmallPath], pslot[2][kSmallPath];
    McfNode rec[2][kSmallPath];
    int64_t flow[2][kSmallPath];
};
static_assert(sizeof(McfCtx) % 4 == 0 && kCtxWords <= kPivotThreads, "control block staging");

// From the chosen entering arc to the updated flows / tree records / apply descriptor.  `v.ctx` must point at
// S.ctx; (key, arc) valid in thread 0; `priced` = arcs this pass evaluated (accounting).  All threads call it.
// `cached`: S.win_idx / S.winx hold the winner's candidate record (candidate cache)
template <int NT = kPivotThreads>   // NT: threads of the calling workgroup
__device__ __forceinline__ void pivot_core(const McfView& v, PivotShared& S, int64_t key, int64_t arc, int32_t rule,
                                            int64_t priced, bool cached = false) {
    const McfPaths gp = mcf_view_paths(v);                                                    // cycle scratch in global memory
    // ... and in LDS
    const McfPaths sp = McfPaths{S.path[0], S.path[1], S.rec[0], S.rec[1], S.ppos[0], S.ppos[1], S.flow[0], S.flow[1], S.pslot[0], S.pslot[1]};
    if (threadIdx.x == 0) {
        McfCtx* c = v.ctx;
        if (c->pivots < c->max_pivots) c->arcs_priced += priced;  // whole-job accounting: the arcs of this pass over ALL shards
        int go = 0;
        const bool have_wx = cached && arc >= 0 && S.win_idx >= 0;
        if (mcf_pivot_begin_t<true>(v, key, arc, rule, have_wx ? &S.winx : nullptr, &S.cy)) {   // (also what mcf_cycle_init would do)
            MCF_PSTAMP(2);
            // sequential part: at most climb_budget dependent round trips.  Shallow end points (the depth gate) are climbed
            // outright: that path is short, so it is recorded in LDS, with the arcs' flows -- decide and finish then work
            // from LDS and the flow update is store-only, as after a scan.
            const int32_t deep = S.cy.ru.depth > S.cy.rw.depth ? S.cy.ru.depth : S.cy.rw.depth;
            const bool gate = v.psz[0] && deep <= c->climb_depth && deep < kSmallPath;
            bool ok;
            if (gate) { ok = mcf_pivot_climb(v, sp, &S.cy, INT32_MAX); S.cy.small = 1; }
            else ok = mcf_pivot_climb(v, gp, &S.cy, mcf_climb_budget(v, S.cy));
            if (ok) go = S.cy.u == S.cy.w ? 1 : 2;
            if (go == 2) mcf_scan_init(&S.acc);
        }
        S.go = go;
    }
    __syncthreads();
    MCF_PSTAMP(3);
    const int go = S.go;
    if (go == 2) mcf_pivot_scan(v, sp, kSmallPath, &S.cy, &S.acc, S.hits, kHitsLds, threadIdx.x, NT);  // barriers inside
    // S.cy.small is written by lane 0 at the very end of the scan: lane 0 may use it at once, the others after the barrier.
    // Separate calls for the two scratch locations: each inlined copy works on one known address space.
    if (go && threadIdx.x == 0 && S.ctx.status == MCF_RUNNING) {
        // (two calls, one per scratch location, were merged by the compiler into one body that fetched its pointers from a
        //  table in private memory -- two dependent scratch loads on the serial path; selected here they stay in registers)
        const bool sm = S.cy.small != 0;
        const McfPaths pp = McfPaths{sm ? sp.path1 : gp.path1, sm ? sp.path2 : gp.path2, sm ? sp.rec1 : gp.rec1, sm ? sp.rec2 : gp.rec2,
                                     sm ? sp.ppos1 : gp.ppos1, sm ? sp.ppos2 : gp.ppos2, sm ? sp.flow1 : gp.flow1, sm ? sp.flow2 : gp.flow2,
                                     sm ? sp.slot1 : gp.slot1, sm ? sp.slot2 : gp.slot2};
        mcf_pivot_decide(v, pp, S.cy);
    }
    MCF_PSTAMP(8);
    __syncthreads();
    if (go && S.cy.small) mcf_pivot_finish(v, sp, threadIdx.x, NT);  // array updates, one path element per lane
    else mcf_pivot_finish(v, gp, threadIdx.x, NT);
    // candidate cache: the entering arc's entry follows its new state (basic after a swap, the other bound after a flip);
    // the leaving arc's entry, if it is listed, and every entry's reduced cost are the update pass's business
    if (cached && threadIdx.x == 0 && S.win_idx >= 0 && S.ctx.stage != 0)
        v.candx[S.win_idx].state = S.ctx.stage == 1 ? -S.ctx.pv_s : 0;
    MCF_PSTAMP(9);
}

// MARK: the handle prices incrementally, the entering / leaving arc's blocks are flagged.  BPL: the tree lives in the blocked
// preorder list (a dense-array handle's instantiation folds every branch of it away: they cost the dense path ~4 %).
// `g_list`: the handle's own candidate list -- only for it does the candidate cache hold the records
template <bool MARK, bool BPL>
__global__ __launch_bounds__(kPivotThreads) void k_pivot(McfView g, const McfCand* __restrict__ cand, int ncand,
                                                          int32_t rule, int have_sweep, const McfCand* g_list) {
    __shared__ PivotShared S;
#ifdef MCF_STAMPS
    if (threadIdx.x == 0) { for (int i = 0; i < 24; ++i) mcf_stamp_acc[i] = 0; mcf_stamp_last = __builtin_amdgcn_s_memtime(); }
#endif
    if (threadIdx.x < kCtxWords) reinterpret_cast<int32_t*>(&S.ctx)[threadIdx.x] = reinterpret_cast<const int32_t*>(g.ctx)[threadIdx.x];
    // the first two candidates of every lane are in flight together with the control block (the grid leaves at most 2 048);
    // candidate cache (g.candx): the records instead -- they hold everything a minor iteration needs
    const bool cached = g.candx != nullptr && rule == MCF_RULE_CANDIDATE_LIST && ncand <= 2 * kPivotThreads && cand == g_list;
    McfCand first = McfCand{0, -1}, second = McfCand{0, -1};
    McfCandX x0, x1;
    x0.arc = -1; x1.arc = -1; x0.rc = 0; x1.rc = 0; x0.state = 0; x1.state = 0; x0.tail = x0.head = x1.tail = x1.head = 0;
    if ((int)threadIdx.x < ncand) { if (cached) x0 = g.candx[threadIdx.x]; else first = cand[threadIdx.x]; }
    if ((int)threadIdx.x + kPivotThreads < ncand) { if (cached) x1 = g.candx[threadIdx.x + kPivotThreads]; else second = cand[threadIdx.x + kPivotThreads]; }
    if (threadIdx.x == 0) S.win_idx = -1;
    if (MARK && g.dirty && threadIdx.x >= 512 && threadIdx.x < 512 + 17) S.dirty_hdr[threadIdx.x - 512] = reinterpret_cast<const int32_t*>(g.dirty)[threadIdx.x - 512];
    __syncthreads();
    McfView v = g;
    v.ctx = &S.ctx;
    if (!MARK) v.dirty = nullptr;  // folds the marking away
    else v.dirty_hdr = reinterpret_cast<const McfDirty*>(S.dirty_hdr);
    if (!BPL) { v.bmeta[0] = nullptr; v.bmeta[1] = nullptr; }  // folds the blocked list away
    MCF_PSTAMP(0);
    // candidate-list rule: slots without a pricing launch in front (have_sweep == 0) can only run
    // minor iterations; once the list is exhausted they idle until the next slot that sweeps
    if (S.ctx.status != MCF_RUNNING || (!have_sweep && S.ctx.minor_left <= 0)) {
        if (threadIdx.x == 0) g.ctx->apply = 0;
        return;
    }
    int64_t key = 0, arc = -1;
    // candidate-list rule, minor iteration: no sweep ran; the listed arcs are re-priced here
    // against the current state (resident reduced cost or potentials)
    const bool minor = rule == MCF_RULE_CANDIDATE_LIST && S.ctx.minor_left > 0;
    // accounting: arcs this pass covers over ALL shards.  Devex: the block's size comes from the host-made granule
    // totals (two loads, in flight during the arg-max)
    int64_t priced = minor ? ncand : v.m;
    if (threadIdx.x == 0 && rule == MCF_RULE_DEVEX_BLOCK && v.dx && S.ctx.num_blocks > 1)
        priced = mcf_devex_block_arcs(v.dx, (int32_t)S.ctx.block_index, S.ctx.block_granules);
    if (cached) {
        // keys straight from the records (a fresh sweep's record gives the sweep's key; an entry a clean incremental block
        // kept, or one that pivots have touched since, gives what a sweep would find now)
        const int64_t v0 = -(int64_t)x0.state * x0.rc, v1 = -(int64_t)x1.state * x1.rc;
        const int64_t k0 = (x0.arc >= 0 && v0 > 0) ? mcf_dantzig_key(v, x0.arc & 0xffffffff, v0, x0.state) : 0;
        const int64_t k1 = (x1.arc >= 0 && v1 > 0) ? mcf_dantzig_key(v, x1.arc & 0xffffffff, v1, x1.state) : 0;
        if (mcf_cand_better(k0, x0.arc, key, arc)) { key = k0; arc = x0.arc; }
        if (mcf_cand_better(k1, x1.arc, key, arc)) { key = k1; arc = x1.arc; }
    } else if (minor && v.rcache && !v.rc_partial) {
        // re-pricing from the resident reduced costs: state and reduced cost of both listed arcs of a lane are requested
        // at once, unconditionally (a dead entry reads arc 0) -- ONE round trip for the whole list instead of a chain of
        // candidate -> state -> reduced cost per entry (measured at 2 048 entries: 5.6 us of a 17 us launch)
        const int64_t e0 = first.arc < 0 ? 0 : (first.arc & 0xffffffff), e1 = second.arc < 0 ? 0 : (second.arc & 0xffffffff);
        const int32_t s0 = v.state[e0], s1 = v.state[e1];
        const int64_t r0 = v.rcache[e0], r1 = v.rcache[e1];
        const int64_t v0 = -(int64_t)s0 * r0, v1 = -(int64_t)s1 * r1;
        const int64_t k0 = (first.arc >= 0 && v0 > 0) ? mcf_dantzig_key(v, e0, v0, s0) : 0;
        const int64_t k1 = (second.arc >= 0 && v1 > 0) ? mcf_dantzig_key(v, e1, v1, s1) : 0;
        if (mcf_cand_better(k0, first.arc, key, arc)) { key = k0; arc = first.arc; }
        if (mcf_cand_better(k1, second.arc, key, arc)) { key = k1; arc = second.arc; }
        for (int i = threadIdx.x + 2 * kPivotThreads; i < ncand; i += kPivotThreads) {   // (longer gathered lists)
            const McfCand cd = cand[i];
            const int64_t kk = mcf_minor_key(v, cd.arc);
            if (mcf_cand_better(kk, cd.arc, key, arc)) { key = kk; arc = cd.arc; }
        }
    } else
    for (int i = threadIdx.x; i < ncand; i += kPivotThreads) {
        const McfCand cd = i == (int)threadIdx.x ? first : (i == (int)threadIdx.x + kPivotThreads ? second : cand[i]);
        const int64_t kk = minor ? mcf_minor_key(v, cd.arc) : cd.key;
        if (mcf_cand_better(kk, cd.arc, key, arc)) { key = kk; arc = cd.arc; }
    }
    if (ncand <= 64) {  // one wave holds every candidate: no cross-wave stage
        if (threadIdx.x < 64) wave_argmax(key, arc);
    } else {
        block_argmax<kPivotThreads>(key, arc);
    }
    if (cached) {   // who won?  its lane hands the record over (two barriers: cheaper than looking the arc up again)
        if (threadIdx.x == 0) S.win_arc = key > 0 ? arc : -1;
        __syncthreads();
        const int64_t wa = S.win_arc;
        if (wa >= 0 && x0.arc == wa) { S.winx = x0; S.win_idx = (int32_t)threadIdx.x; }
        else if (wa >= 0 && x1.arc == wa) { S.winx = x1; S.win_idx = (int32_t)threadIdx.x + kPivotThreads; }
        __syncthreads();
    }
    MCF_PSTAMP(1);
    pivot_core(v, S, key, arc, rule, priced, cached);
    // publish the control block for the apply / pricing launches that follow (finish only reads it)
    if (threadIdx.x < kCtxWords) reinterpret_cast<int32_t*>(g.ctx)[threadIdx.x] = reinterpret_cast<const int32_t*>(&S.ctx)[threadIdx.x];
#ifdef MCF_STAMPS
    __syncthreads();
    MCF_PSTAMP(10);
    if (threadIdx.x == 0) { for (int i = 0; i < 24; ++i) g_pivot_stamps[i] += mcf_stamp_acc[i]; g_pivot_stamps[23] += 1; }
#endif
}

// ------------------------------------------------------------------ k_pivot_run: pivots of a candidate list back to back
// Blocked list + candidate cache + candidate-list rule.  Almost every pivot of a large sparse instance re-hangs a handful of
// nodes: its whole update is a few block records, a few dozen reduced costs and the list's records -- work for ONE workgroup.
// This kernel therefore keeps going: pivot, update in place (bpl_update_inline), next minor iteration, ... until the list is
// used up, a pivot's update is too large for one workgroup (the descriptor is then left for the k_update_bpl launch that
// follows, exactly as k_pivot leaves it) or `max_steps` is reached.  What it saves per pivot: two launch boundaries, the
// control block's trip through memory and back, the staging of the list.  The pivots are the same ones in the same order:
// every step is k_pivot's step and k_update_bpl's passes, run by one workgroup instead of a grid.
constexpr int kRunTouchedCap = 2048, kRunT2Cap = 2048, kRunSegCap = 256;
constexpr int kRunMaxSubtree = 64;     // larger re-hung subtrees go to the grid

struct RunShared {
    int32_t nt, nn;
    int32_t tb[kRunTouchedCap];
    int32_t t2[kRunT2Cap];
    McfSeg seg[kRunSegCap];
};

// k_update_bpl's passes by the calling workgroup alone (NT threads; not for ctx.rebuild).  `c` is the control block in LDS.
template <bool MARK, bool RC, int NT>
__device__ __forceinline__ void bpl_update_inline(const McfView& v, McfCtx& c, RunShared& R) {
    const int32_t tid = (int32_t)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) { R.nt = 0; R.nn = 0; }
    if (tid < c.nseg && tid < kRunSegCap) R.seg[tid] = v.seg[tid];
    __syncthreads();
    McfView segv = v;
    if (c.nseg <= kRunSegCap) segv.seg = R.seg;
    const int32_t known0 = c.t2_blk, known1 = c.ins_blk;
    const bool preview = RC && c.pv_t2n == 1;
    const McfBlkMeta* bm = c.cur ? v.bmeta[1] : v.bmeta[0];
    McfBlkMeta* bn = c.cur ? v.bmeta[0] : v.bmeta[1];
    const int32_t* xa = c.arena ? v.bext[1] : v.bext[0];
    if (wave < 2) {
        const int32_t kb = wave == 0 ? known0 : ((known1 >= 0 && known1 != known0) ? known1 : -1);
        if (kb >= 0) bpl_block<MARK, RC>(v, segv, c, &c, kb, lane, R.t2, &R.nn, !preview, kRunT2Cap);
    } else if (preview && wave >= NT / 128) {
        // the one node of T2: its adjacency range came with the pivot (second half of the workgroup)
        const int64_t sigma = c.sigma;
        const int32_t t0 = tid - NT / 2, nt = NT / 2;
        for (int64_t p = c.pv_adj[0] + t0; p < c.pv_adj[1]; p += nt) {
            const int64_t ent = v.adj[p];
            const int32_t e = (int32_t)((uint32_t)ent >> 1);
            const int64_t r = v.rcache[e] + ((ent & 1) ? sigma : -sigma);
            v.rcache[e] = r;
            if (v.vkey) v.vkey[e] = mcf_vkey(-(int64_t)v.state[e] * r, v.vk_bigm, v.vk_half);
            if (MARK) mcf_mark_dirty(v, e);
        }
    } else {
        // phase A: the other blocks, one lane per block (the lanes of waves 2 .. that are not patching)
        const int32_t a_lanes = (preview ? NT / 2 : NT) - 128, a0 = tid - 128;
        const int32_t nold = c.alloc_prev;
        // eight blocks per lane and trip, all their loads in flight together (one block per trip was a chain of dependent
        // round trips: 20 K blocks over 400-900 lanes = 25-50 of them per pivot)
        constexpr int KA = 8;
        for (int32_t b0 = a0; b0 < nold; b0 += a_lanes * KA) {
            McfBlkMeta mm[KA];
            int32_t xx[KA];
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                const int32_t b = b0 + k * a_lanes;
                const int32_t bc = b < nold ? b : a0;   // clamp: keep the loads unconditional
                mm[k] = bm[bc];
                xx[k] = xa[bc];
            }
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                const int32_t b = b0 + k * a_lanes;
                if (b >= nold || b == known0 || b == known1) continue;
                const McfBlkMeta m = mm[k];
                const int32_t x = xx[k], beg = mcf_ext_beg(x), end = mcf_ext_end(x);
                if (m.base == MCF_BLK_FREE || end <= beg) { bn[b] = McfBlkMeta{MCF_BLK_FREE, 0}; continue; }
                bool apart = (x & MCF_EXT_FLAG) != 0;
                if (!apart) {
                    const McfBlkPlan P = mcf_bpl_plan(c, m.base, m.base + beg, m.base + end);
                    if (P.touched) apart = true; else bn[b] = McfBlkMeta{P.nbase, m.rrel};
                }
                if (apart) {
                    const int32_t slot = atomicAdd(&R.nt, 1);
                    if (slot < kRunTouchedCap) R.tb[slot] = b;
                }
            }
        }
    }
    __syncthreads();
    const int32_t nt = R.nt < kRunTouchedCap ? R.nt : kRunTouchedCap;   // (at most the re-index flags of the cycle + T2's other blocks)
    for (int32_t t = wave; t < nt; t += NT / 64) bpl_block<MARK, RC>(v, segv, c, &c, R.tb[t], lane, R.t2, &R.nn, !preview, kRunT2Cap);
    __syncthreads();
    if (RC && !preview) {
        const int32_t nn = R.nn < kRunT2Cap ? R.nn : kRunT2Cap;
        const bool whole = nn == c.t2_size && nn <= kBplListMember;
        for (int32_t t = tid >> 4; t < nn; t += NT / 16) bpl_patch_node<MARK>(v, c, R.t2[t], tid & 15, 16, whole ? R.t2 : nullptr, nn);
    }
    candx_pass(v, c, tid, NT);
}

template <bool MARK, bool RC>
__global__ __launch_bounds__(kPivotThreads) void k_pivot_run(McfView g, int ncand, int have_sweep, int max_steps) {
    __shared__ PivotShared S;
    __shared__ RunShared R;
#ifdef MCF_STAMPS
    if (threadIdx.x == 0) { for (int i = 0; i < 24; ++i) mcf_stamp_acc[i] = 0; mcf_stamp_last = __builtin_amdgcn_s_memtime(); }
#endif
    if (threadIdx.x < kCtxWords) reinterpret_cast<int32_t*>(&S.ctx)[threadIdx.x] = reinterpret_cast<const int32_t*>(g.ctx)[threadIdx.x];
    if (MARK && g.dirty && threadIdx.x >= 512 && threadIdx.x < 512 + 17) S.dirty_hdr[threadIdx.x - 512] = reinterpret_cast<const int32_t*>(g.dirty)[threadIdx.x - 512];
    __syncthreads();
    McfView v = g;
    v.ctx = &S.ctx;
    if (!MARK) v.dirty = nullptr;
    else v.dirty_hdr = reinterpret_cast<const McfDirty*>(S.dirty_hdr);
    if (threadIdx.x == 0) S.ctx.apply = 0;   // (a descriptor left by an earlier launch has been carried out by the update launch in between)
    const int32_t rule = MCF_RULE_CANDIDATE_LIST;
    int steps = 0;
    for (;; ++steps) {
        // uniform control values are read BEFORE a barrier (lane 0 rewrites them later in the iteration)
        const int32_t status_now = S.ctx.status, minor_left = S.ctx.minor_left;
        const bool sweep_now = steps == 0 && have_sweep;
        __syncthreads();
        if (steps >= max_steps || status_now != MCF_RUNNING || (!sweep_now && minor_left <= 0)) break;
        MCF_PSTAMP(0);
        // the list's records (kept current by the update passes; after this workgroup's own in-place updates they come from its
        // L2): state and exact reduced cost give the key a sweep would find now
        McfCandX x0, x1;
        x0.arc = -1; x1.arc = -1; x0.rc = 0; x1.rc = 0; x0.state = 0; x1.state = 0; x0.tail = x0.head = x1.tail = x1.head = 0;
        if ((int)threadIdx.x < ncand) x0 = g.candx[threadIdx.x];
        if ((int)threadIdx.x + kPivotThreads < ncand) x1 = g.candx[threadIdx.x + kPivotThreads];
        if (threadIdx.x == 0) S.win_idx = -1;
        int64_t key = 0, arc = -1;
        const int64_t v0 = -(int64_t)x0.state * x0.rc, v1 = -(int64_t)x1.state * x1.rc;
        const int64_t k0 = (x0.arc >= 0 && v0 > 0) ? mcf_dantzig_key(v, x0.arc & 0xffffffff, v0, x0.state) : 0;
        const int64_t k1 = (x1.arc >= 0 && v1 > 0) ? mcf_dantzig_key(v, x1.arc & 0xffffffff, v1, x1.state) : 0;
        if (mcf_cand_better(k0, x0.arc, key, arc)) { key = k0; arc = x0.arc; }
        if (mcf_cand_better(k1, x1.arc, key, arc)) { key = k1; arc = x1.arc; }
        const int64_t priced = minor_left > 0 ? ncand : v.m;
        if (ncand <= 64) { if (threadIdx.x < 64) wave_argmax(key, arc); }
        else block_argmax<kPivotThreads>(key, arc);
        if (threadIdx.x == 0) S.win_arc = key > 0 ? arc : -1;
        __syncthreads();
        const int64_t wa = S.win_arc;
        if (wa >= 0 && x0.arc == wa) { S.winx = x0; S.win_idx = (int32_t)threadIdx.x; }
        else if (wa >= 0 && x1.arc == wa) { S.winx = x1; S.win_idx = (int32_t)threadIdx.x + kPivotThreads; }
        __syncthreads();
        MCF_PSTAMP(1);
        pivot_core(v, S, key, arc, rule, priced, true);
        __syncthreads();
        if (!S.ctx.apply) continue;   // a bound flip (or nothing to pivot on): no tree change
        // the update: here when it is small, else by the grid launch that follows
        if (S.ctx.rebuild || S.ctx.t2_size > kRunMaxSubtree) break;
        bpl_update_inline<MARK, RC, kPivotThreads>(v, S.ctx, R);
        __syncthreads();
        if (threadIdx.x == 0) S.ctx.apply = 0;   // (pending_flip stays: the next begin switches to the copies just written)
        // what this workgroup has just stored must not be served from a stale line of the scalar cache (uniform loads of the
        // block records); the vector L1 is write-through and shared by the whole workgroup
        __builtin_amdgcn_s_dcache_inv();
    }
    __syncthreads();
    if (threadIdx.x < kCtxWords) reinterpret_cast<int32_t*>(g.ctx)[threadIdx.x] = reinterpret_cast<const int32_t*>(&S.ctx)[threadIdx.x];
#ifdef MCF_STAMPS
    __syncthreads();
    MCF_PSTAMP(10);
    if (threadIdx.x == 0) { for (int i = 0; i < 24; ++i) g_pivot_stamps[i] += mcf_stamp_acc[i]; g_pivot_stamps[23] += steps; }
#endif
}

// ------------------------------------------------------------------ k_solve_mid: persistent single-workgroup pivot loop, state in global memory
// Between two kernels of the three-kernel path every first touch of the tree arrays is a miss to the memory-side
// cache (the previous kernel wrote them from other CUs / XCDs): ~0.7 us per dependent round trip, ~12 of them in
// k_pivot alone.  For mid-size instances one workgroup can do the WHOLE pivot -- pricing of a Devex block (or, for
// the candidate-list rule, re-pricing the list), cycle scan, finish, block permutation + potential shift, resident
// reduced-cost update -- and then nothing crosses a kernel boundary: the state stays in this CU's L1 / this XCD's
// L2 for as many pivots as the rule allows.  Same core functions, same arc sets and tie rules as the other paths,
// hence the identical pivot sequence.
//   Devex block / small Dantzig:  one launch runs until optimal / pivot limit.
//   candidate list:               the launch ends when a full sweep is due (k_price_rc over the whole grid builds
//                                 the next list); `fresh` says such a sweep ran right before this launch.

// `arm`: take the pivot cap from the kernel argument (what k_ctl does for the other paths: one launch less per call;
// only for launches that are not replayed from a captured graph, whose arguments are frozen).
__device__ __forceinline__ void arm_ctx(McfCtx* c, int64_t cap) {
    c->max_pivots = cap;
    c->limit_checked = 0;
    if (c->status == MCF_PIVOT_LIMIT && c->pivots < cap) c->status = MCF_RUNNING;
}

// The candidate-list rule's full sweep done by ONE workgroup (batched persistent loops: no grid to go back to): wave w
// plays the pricing workgroups w, w + nwaves, ... of the grid sweep -- the same arc sets (workgroup lb * 8 + x: the groups of
// four arcs g_lo(x) + lb * 256 + [0, 256) + j * nlb * 256 of bucket x, as in k_price_rc / mcf_price_block_of), the same keys,
// the same tie rule -- and writes the same list, one candidate per (virtual) workgroup.
template <int NT>
__device__ __forceinline__ void virtual_list_sweep(const McfView& v, McfCand* list, int nlist) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nlb = nlist / MCF_NUM_BUCKETS;
    const int8_t* __restrict__ state = v.state;
    const int64_t* __restrict__ rcache = v.rcache;
    const int32_t* __restrict__ orig = v.orig;
    for (int b = wave; b < nlist; b += NT / 64) {
        const int x = b & (MCF_NUM_BUCKETS - 1), lb = b >> 3;
        const int64_t lo = v.bucket_off[x], hi = v.bucket_off[x + 1];
        const int64_t g_lo = lo >> 2, g_hi = (hi + 3) >> 2;
        int64_t key = 0, best_i = -1;
        for (int64_t gbase = g_lo + (int64_t)lb * 256; gbase < g_hi; gbase += (int64_t)nlb * 256) {
            for (int t = lane; t < 256; t += 64) {
                const int64_t g = gbase + t;
                if (g >= g_hi) break;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t i = (g << 2) + k;
                    if (i < lo || i >= hi) continue;
                    const int32_t s = state[i];
                    if (s == 0) continue;
                    const int64_t viol = -(int64_t)s * rcache[i];
                    if (viol <= 0) continue;
                    const int64_t kk = mcf_dantzig_key(v, i, viol, s);
                    if (kk < key) continue;
                    if (kk > key || best_i < 0 || orig[i] < orig[best_i]) { key = kk; best_i = i; }
                }
            }
        }
        int64_t arc = best_i >= 0 ? mcf_pack_arc(orig[best_i], best_i) : -1;
        wave_argmax(key, arc);
        if (lane == 0) list[b] = McfCand{key, arc};
    }
}

// NT: threads of the workgroup (the batched launch also runs narrower workgroups, two per CU).  SELF: a candidate-list loop does
// the list's full sweeps itself (cand is then written as well as read).  A template parameter, not a run-time flag: with the
// sweep inlined into every instance the loop's register demand went up (spills 9 -> 21 VGPRs, scratch 72 -> 488 B) and
// every rule lost 30-45 % -- only the kernels that need it carry it.
template <int NT = kPivotThreads, bool SELF = false>
__device__ __forceinline__ void solve_mid_body(const McfView& g, int32_t rule, McfCand* cand,
                                               int ncand, int fresh, int max_iters, int arm, int64_t cap, McfCtx* host_ctx = nullptr) {
    __shared__ PivotShared S;
    __shared__ int32_t s_gran[MCF_NUM_BUCKETS][MCF_GRANULES + 1];  // Devex: the granule table (blocks move and resize under the tuner)
    __shared__ int32_t s_lo[MCF_NUM_BUCKETS], s_hi[MCF_NUM_BUCKETS];  // other rules: the whole buckets
#ifdef MCF_STAMPS
    if (threadIdx.x == 0) { for (int i = 0; i < 24; ++i) mcf_stamp_acc[i] = 0; mcf_stamp_last = __builtin_amdgcn_s_memtime(); }
    __syncthreads();
#endif
    if (threadIdx.x < kCtxWords) reinterpret_cast<int32_t*>(&S.ctx)[threadIdx.x] = reinterpret_cast<const int32_t*>(g.ctx)[threadIdx.x];
    const bool devex = rule == MCF_RULE_DEVEX_BLOCK && g.dx;
    if (devex) {
        for (int q = threadIdx.x; q < MCF_NUM_BUCKETS * (MCF_GRANULES + 1); q += NT)
            (&s_gran[0][0])[q] = (&g.dx->gran[0][0])[q];
    } else if (threadIdx.x < MCF_NUM_BUCKETS) {
        s_lo[threadIdx.x] = (int32_t)g.bucket_off[threadIdx.x];
        s_hi[threadIdx.x] = (int32_t)g.bucket_off[threadIdx.x + 1];
    }
    __syncthreads();
    if (arm && threadIdx.x == 0) arm_ctx(&S.ctx, cap);
    __syncthreads();
    McfView v = g;
    v.ctx = &S.ctx;
    v.dirty = nullptr;  // mcf_create never combines incremental sweeps with this loop: the marking folds away
    v.dirty_hdr = nullptr; v.bmeta[0] = nullptr; v.bmeta[1] = nullptr; v.candx = nullptr;   // (nor the blocked list, nor the candidate cache)
    const bool listing = rule == MCF_RULE_CANDIDATE_LIST;
    // the sweep in front of this launch was a no-op if the list was still live (k_price_rc, use_block == 2)
    bool have_fresh = listing && fresh && S.ctx.minor_left <= 0 && S.ctx.status == MCF_RUNNING;
    for (int it = 0; it < max_iters; ++it) {
        // uniform control values are read BEFORE a barrier: lane 0 rewrites them later in this very iteration
        const int32_t status_now = S.ctx.status;
        const int32_t minor_left = S.ctx.minor_left;
        int32_t bg0 = 0, bg1 = MCF_GRANULES;  // Devex: granule range of the current block
        if (devex) {
            const int32_t bg = S.ctx.block_granules;
            bg0 = (int32_t)S.ctx.block_index * bg;
            if (bg0 >= MCF_GRANULES) bg0 = 0;
            bg1 = bg0 + bg < MCF_GRANULES ? bg0 + bg : MCF_GRANULES;
        }
        __syncthreads();
        if (status_now != MCF_RUNNING) break;
        MCF_PSTAMP(0);
        int64_t key = 0, arc = -1, priced = 0;
        if (listing) {
            const bool minor = minor_left > 0;
            if (!minor && !have_fresh) {
                if constexpr (!SELF) break;  // a full sweep is due: back to the grid
                else {
                    virtual_list_sweep<NT>(v, cand, ncand);
                    __syncthreads();         // the list (global memory, this workgroup's own writes) before anyone reads it
                }
            }
            have_fresh = false;
            for (int i = threadIdx.x; i < ncand; i += NT) {
                const McfCand cd = cand[i];
                const int64_t kk = minor ? mcf_minor_key(v, cd.arc) : cd.key;
                if (mcf_cand_better(kk, cd.arc, key, arc)) { key = kk; arc = cd.arc; }
            }
            priced = minor ? ncand : v.m;
            if (ncand <= 64) { if (threadIdx.x < 64) wave_argmax(key, arc); }
            else block_argmax<NT>(key, arc);
        } else {
            // the arc set of k_price_rc for shard 0 of 1: 128 lanes per head bucket, all eight buckets at once,
            // streaming the resident reduced costs
            constexpr int kPer = NT / MCF_NUM_BUCKETS;
            const int x = threadIdx.x / kPer, l = threadIdx.x % kPer;
            const int64_t lo = devex ? s_gran[x][bg0] : s_lo[x], hi = devex ? s_gran[x][bg1] : s_hi[x];
            const int64_t* __restrict__ rcache = v.rcache;
            const int8_t* __restrict__ state = v.state;
            int64_t best_i = -1;  // the caller's id is looked up once at the end (and on ties)
            int32_t best_s = 0;
            constexpr int UM = 4;  // arcs in flight per lane: all loads first (unconditional, clamped), then the arithmetic
            if (hi - lo <= kPer) {  // at most one arc per lane (small Devex blocks): nothing to overlap
                const int64_t i = lo + l;
                if (i < hi && state[i]) {
                    const int64_t viol = -(int64_t)state[i] * rcache[i];
                    if (viol > 0) {
                        key = mcf_dantzig_key(v, i, viol, (int32_t)state[i]);
                        if (rule == MCF_RULE_DEVEX_BLOCK) key = __double_as_longlong(((double)viol * (double)viol) / (double)v.weight[i]);
                        best_i = i;
                        best_s = state[i];
                    }
                }
            } else
            for (int64_t i0 = lo + l; i0 < hi; i0 += (int64_t)kPer * UM) {
                int32_t sts[UM];
                int64_t rcs[UM];
                float wts[UM];
#pragma unroll
                for (int u = 0; u < UM; ++u) {
                    const int64_t i = i0 + (int64_t)u * kPer;
                    const int64_t ic = i < hi ? i : lo;
                    sts[u] = i < hi ? (int32_t)state[ic] : 0;
                    rcs[u] = rcache[ic];
                    wts[u] = rule == MCF_RULE_DEVEX_BLOCK ? v.weight[ic] : 1.0f;
                }
#pragma unroll
                for (int u = 0; u < UM; ++u) {
                    const int64_t i = i0 + (int64_t)u * kPer;
                    if (!sts[u]) continue;
                    const int64_t viol = -(int64_t)sts[u] * rcs[u];
                    if (viol <= 0) continue;
                    int64_t kk = mcf_dantzig_key(v, i, viol, sts[u]);
                    if (rule == MCF_RULE_DEVEX_BLOCK) {
                        const double merit = ((double)viol * (double)viol) / (double)wts[u];
                        kk = __double_as_longlong(merit);
                    }
                    if (kk < key) continue;
                    if (kk > key) { key = kk; best_i = i; best_s = sts[u]; }
                    else if (best_i < 0) { best_i = i; best_s = sts[u]; }
                    else {
                        const int32_t a = devex ? mcf_devex_tie_id(v.orig[i], sts[u]) : v.orig[i];
                        const int32_t b = devex ? mcf_devex_tie_id(v.orig[best_i], best_s) : v.orig[best_i];
                        if (a < b) { best_i = i; best_s = sts[u]; }
                    }
                }
            }
            if (best_i >= 0) arc = mcf_pack_arc(devex ? mcf_devex_tie_id(v.orig[best_i], best_s) : v.orig[best_i], best_i);
            if (threadIdx.x == 0)
                for (int x2 = 0; x2 < MCF_NUM_BUCKETS; ++x2) priced += devex ? s_gran[x2][bg1] - s_gran[x2][bg0] : s_hi[x2] - s_lo[x2];
            block_argmax<NT>(key, arc);
        }
        MCF_PSTAMP(1);
        pivot_core<NT>(v, S, key, arc, rule, priced);
        __syncthreads();  // the finish pass's writes (records, sizes, segment table) before the apply pass reads them
        if (S.ctx.apply) {  // (splitting the lanes between the two halves was measured and lost at 4 096 nodes)
            apply_pass(v, S.ctx, threadIdx.x, NT);
            rcupd_pass(v, S.ctx, threadIdx.x >> 4, NT / 16, threadIdx.x & 15);
        }
        __syncthreads();
        MCF_PSTAMP(10);
#ifdef MCF_STAMPS
        if (threadIdx.x == 0) mcf_stamp_acc[23] += 1;
#endif
        if (threadIdx.x == 0) S.ctx.apply = 0;
    }
    __syncthreads();
    // at the budget: is any arc still eligible?  (simplex.py:1678-1699; saves the host a pricing pass + three syncs)
    if (S.ctx.status == MCF_PIVOT_LIMIT && !S.ctx.limit_checked) {
        __shared__ int s_any;
        if (threadIdx.x == 0) s_any = 0;
        __syncthreads();
        int any = 0;
        for (int64_t i = threadIdx.x; i < g.m && !any; i += NT) {
            const int64_t st = g.state[i];
            if (st != 0 && -st * g.rcache[i] > 0) any = 1;
        }
        if (any) s_any = 1;
        __syncthreads();
        if (threadIdx.x == 0) { if (s_any) S.ctx.limit_checked = 1; else S.ctx.status = MCF_OPTIMAL; }
        __syncthreads();
    }
    if (threadIdx.x < kCtxWords) reinterpret_cast<int32_t*>(g.ctx)[threadIdx.x] = reinterpret_cast<const int32_t*>(&S.ctx)[threadIdx.x];
    if (host_ctx && threadIdx.x < kCtxWords) reinterpret_cast<int32_t*>(host_ctx)[threadIdx.x] = reinterpret_cast<const int32_t*>(&S.ctx)[threadIdx.x];
#ifdef MCF_STAMPS
    if (threadIdx.x == 0) for (int i = 0; i < 24; ++i) g_pivot_stamps[i] += mcf_stamp_acc[i];
#endif
}

__global__ __launch_bounds__(kPivotThreads) void k_solve_mid(McfView g, int32_t rule, McfCand* cand, int ncand, int fresh,
                                                              int max_iters, int arm, int64_t cap) {
    solve_mid_body(g, rule, cand, ncand, fresh, max_iters, arm, cap);
}

// Many independent instances side by side, state in global memory: workgroup b runs the whole solve of jobs[b]
// (Dantzig / Devex handles of the persistent loop; mcf_solve_batch).
struct MidJob {
    McfView g;
    int32_t rule;
    int32_t nlist;     // candidate list: its length (= the grid's pricing workgroups) ...
    McfCand* list;     // ... and where it lives
    int64_t cap;
    McfCtx* host_ctx;  // pinned: the final control block goes straight to the host
};

// 4 waves per SIMD (<= 128 VGPRs): one workgroup of 1 024 or TWO of 512 per CU -- the loop is a chain of dependent memory
// round trips, so a second instance on the CU fills the first one's waiting time (scripts/batch_mid.py, MCF_BATCH_THREADS).
template <int NT, bool SELF>   // SELF: the candidate-list jobs of a batch (their own launch)
__global__ __launch_bounds__(NT, 4) void k_solve_mid_batch(const MidJob* __restrict__ jobs) {
    const MidJob& J = jobs[blockIdx.x];   // uniform per workgroup: scalar loads
    solve_mid_body<NT, SELF>(J.g, J.rule, J.list, J.nlist, 0, 1 << 22, 1, J.cap, J.host_ctx);
}

// The reduced-cost half alone (overlapped graphs: the next pivot's pricing waits for this half only, see build_graph)
template <bool MARK>
__global__ __launch_bounds__(kRcupdThreads) void k_rcupd(McfView g) {
    McfView v = g;
    if (!MARK) v.dirty = nullptr;
    const McfCtx c = *v.ctx;
    if (!c.apply) return;
    rcupd_pass(v, c, (int64_t)blockIdx.x * (kRcupdThreads / 16) + (threadIdx.x >> 4), (int64_t)gridDim.x * (kRcupdThreads / 16), threadIdx.x & 15);
}

// ------------------------------------------------------------------ k_apply
__global__ __launch_bounds__(kApplyThreads) void k_apply(McfView v) {
    const McfCtx c = *v.ctx;  // uniform: scalar loads
    if (!c.apply) return;
    const int xb = v.candx ? kCandxBlocks : 0;   // (kApplyThreads == kRcupdThreads)
    const int nb = (int)gridDim.x - xb;
    if ((int)blockIdx.x >= nb) { candx_pass(v, c, ((int)blockIdx.x - nb) * kApplyThreads + (int)threadIdx.x, xb * kApplyThreads); return; }
    apply_pass(v, c, (int64_t)blockIdx.x * kApplyThreads + threadIdx.x, (int64_t)nb * kApplyThreads);
}

// ------------------------------------------------------------------ two-lane cycle climb (LDS loop)
// With the tree in LDS the climb is bound by instruction issue of ONE lane (~150 instructions per round trip for
// the two sides), not by latency.  Lanes 0 and 1 of a wave therefore climb one side each in lock step -- the same
// instruction stream serves both sides -- and swap (node, depth) with one DPP move per word and round trip.  The
// steps, the recorded paths and the ratio-test winners are exactly those of mcf_pivot_climb (depth-balanced: a side
// moves when it is at least as deep as the other); lane 0 gets the finished McfCycle.  Called by lanes 0 and 1 only.
__device__ __forceinline__ int swap01(int x) { return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false); }  // quad_perm [1,0,3,2]

__device__ __forceinline__ bool pivot_climb_2lanes(const McfView& v, McfCycle* out) {
    const McfCtx* c = v.ctx;
    const int side = threadIdx.x & 1;  // 0: first side, 1: second side
    const int32_t* pcur = c->cur ? v.posbuf[1] : v.posbuf[0];
    int32_t node = side ? c->pv_second : c->pv_first;
    McfNode rec = v.node[node];
    int32_t pos = pcur[node];
    const McfNode rec0 = rec;
    const int32_t pos0 = pos;
    int32_t* const path = side ? v.path2 : v.path1;
    int32_t* const ppos = side ? v.ppos2 : v.ppos1;
    McfNode* const recs = side ? v.rec2 : v.rec1;
    int64_t d = MCF_INF;
    int32_t k = -1, n = 0, trips = 0;
    bool ok = true;
    for (;;) {
        const int32_t onode = swap01(node), odepth = swap01(rec.depth);
        if (node == onode) break;
        if (rec.depth >= odepth) {
            const McfNode nrec = v.node[rec.parent];
            const int32_t npos = pcur[rec.parent];
            const McfArcW a = v.arcw[rec.pred >> 1];
            // first side is walked against the flow (an up arc loses flow), second side with it
            const bool loses = ((rec.pred & 1) != 0) != (side != 0);
            const int64_t r = loses ? a.flow : (a.cap >= MCF_INF ? MCF_INF : a.cap - a.flow);
            if (side ? r <= d : r < d) { d = r; k = n; }
            path[n] = node;
            recs[n] = rec;
            ppos[n] = pos;
            ++n;
            node = rec.parent;
            rec = nrec;
            pos = npos;
        }
        if (++trips > v.n_nodes) { ok = false; break; }  // depths out of sync: never spin (both lanes count alike)
    }
    // hand the second side to lane 0
    const int32_t od_lo = swap01((int)(uint32_t)d), od_hi = swap01((int)(uint32_t)((uint64_t)d >> 32));
    const int32_t ok2 = swap01(k), on2 = swap01(n), opos0 = swap01(pos0);
    McfNode orec0;
    orec0.parent = swap01(rec0.parent); orec0.pred = swap01(rec0.pred); orec0.size = swap01(rec0.size); orec0.depth = swap01(rec0.depth);
    if (side == 0) {
        out->d1 = d; out->k1 = k; out->n1 = n;
        out->d2 = (int64_t)(((uint64_t)(uint32_t)od_hi << 32) | (uint32_t)od_lo); out->k2 = ok2; out->n2 = on2;
        out->u = node; out->w = node; out->pu = pos; out->pw = pos; out->ru = rec; out->rw = rec;
        out->p0u = pos0; out->p0w = opos0; out->r0u = rec0; out->r0w = orec0;
        out->small = 0;
    }
    return ok;
}

// ------------------------------------------------------------------ k_solve_small: fused LDS-resident pivot loop
// Instances whose whole state fits in one CU's 160 KiB of LDS (netgen_8_08a: ~96 KiB) are
// latency-bound, not bandwidth-bound: 27 KB per sweep is nothing, three kernel boundaries and
// a dozen dependent global loads per pivot are everything.  This kernel copies the instance
// into LDS once, runs pivots back to back in ONE persistent workgroup (price: all 1024 lanes
// over LDS; pivot: the same mcf_pivot_walk / mcf_pivot_finish; apply: all lanes through the same
// mcf_apply_one) and copies the state back.  Same arc sets, same tie rule, same core functions
// as the three-kernel path, so the pivot sequence is identical.
struct SmallLayout {
    uint32_t tail, head, cost, orig, state, weight, arcw, pi, node, order0, order1, pos0, pos1, path1, path2, ppos1, ppos2, rec1, rec2, seg, ctx, total;
};

constexpr int kSmallThreads = 1024;
constexpr int kSmallMaxLds = 158 * 1024;   // dynamic LDS of the fused small-instance loop (160 KB per CU, a little static on top)

__device__ __forceinline__ void copy_words(void* dst, const void* src, uint32_t bytes) {
    const uint32_t n = bytes >> 2;
    const int32_t* s = static_cast<const int32_t*>(src);
    int32_t* d = static_cast<int32_t*>(dst);
    for (uint32_t i = threadIdx.x; i < n; i += kSmallThreads) d[i] = s[i];
}

#ifdef MCF_STAMPS
#define STAMP(slot)                                                                       \
    do {                                                                                  \
        if (threadIdx.x == 0) {                                                           \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                 \
            stamps_[slot] += now_ - last_;                                                \
            last_ = now_;                                                                 \
        }                                                                                 \
    } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

// ------------------------------------------------------------------ node-parallel cycle search (LDS loop)
// A tree that fits one workgroup's lanes needs no walk at all: lane x asks whether NODE x is an ancestor of the entering arc's
// first / second end point -- pos[x] <= pos[u] < pos[x] + size[x], both from LDS -- and, if it is an ancestor of exactly one,
// files itself at path index depth[u] - depth[x] with its record and position (what the climb would have recorded) and
// takes part in that side's ratio test; the deepest common ancestor is the join.  Three dependent LDS reads and two wave
// reductions for the whole cycle, whatever its length, instead of one dependent round per tree level on one or two lanes
// (the 2-lane climb was 53 % of a pivot on netgen_8_08a).  Same path arrays, same ratio-test winners (first side: lowest
// index among equal residuals, second side: highest), hence the same pivots.  All threads call it; one barrier inside;
// lane 0 gets the finished McfCycle.  Requires n_nodes <= THREADS.
struct SmallCycleAcc {
    int32_t c1[16], c2[16];   // one-sided ancestors per wave, first / second side
};

template <int THREADS>
__device__ __forceinline__ void small_cycle_parallel(const McfView& v, SmallCycleAcc& A, McfCycle* out) {
    const McfCtx* c = v.ctx;
    const int32_t* pcur = c->cur ? v.posbuf[1] : v.posbuf[0];
    const int32_t first = c->pv_first, second = c->pv_second;      // (uniform: LDS broadcast reads)
    // (only the depths are carried through the pass; lane 0 reads the whole records again at the end -- kept alive across the
    //  pass they were spilled to scratch, the only private memory this kernel used)
    const int32_t du0 = v.node[first].depth, dw0 = v.node[second].depth;
    const int32_t pu = pcur[first], pw = pcur[second];
    const int32_t x = (int32_t)threadIdx.x, wave = x >> 6;
    const int32_t N = v.n_nodes, nwaves = (N + 63) >> 6;
    // scratch in the segment table's LDS (the finish pass only fills it afterwards): residual per path index and side, and
    // the common ancestors by depth (root .. join occupy depths 0 .. depth[join], one node each)
    int64_t* const res1 = reinterpret_cast<int64_t*>(v.seg);
    int64_t* const res2 = res1 + N;
    int32_t* const by_depth = reinterpret_cast<int32_t*>(res2 + N);
    bool h1 = false, h2 = false;
    if (x < N) {
        const McfNode rec = v.node[x];
        const int32_t px = pcur[x];
        const bool au = (uint32_t)(pu - px) < (uint32_t)rec.size, aw = (uint32_t)(pw - px) < (uint32_t)rec.size;
        if (au && aw) by_depth[rec.depth] = x;
        else if (au) {
            const McfArcW a = v.arcw[rec.pred >> 1];
            const int32_t idx = du0 - rec.depth;
            v.path1[idx] = x; v.rec1[idx] = rec; v.ppos1[idx] = px;
            // first side is walked against the flow: an up arc loses flow, a down arc gains
            res1[idx] = (rec.pred & 1) ? a.flow : (a.cap >= MCF_INF ? MCF_INF : a.cap - a.flow);
            h1 = true;
        } else if (aw) {
            const McfArcW a = v.arcw[rec.pred >> 1];
            const int32_t idx = dw0 - rec.depth;
            v.path2[idx] = x; v.rec2[idx] = rec; v.ppos2[idx] = px;
            res2[idx] = (rec.pred & 1) ? (a.cap >= MCF_INF ? MCF_INF : a.cap - a.flow) : a.flow;
            h2 = true;
        }
    }
    if (wave < nwaves) {
        const int32_t n1w = (int32_t)__popcll(__ballot(h1)), n2w = (int32_t)__popcll(__ballot(h2));
        if ((x & 63) == 0) { A.c1[wave] = n1w; A.c2[wave] = n2w; }
    }
    __syncthreads();
    if (x == 0) {
        int32_t n1 = 0, n2 = 0;
        for (int32_t q = 0; q < nwaves; ++q) { n1 += A.c1[q]; n2 += A.c2[q]; }
        const int32_t jn = by_depth[du0 - n1];   // the join: the common ancestor right above the first side's path
        const McfNode rj = v.node[jn];
        // the ratio tests, exactly the climb's: first side -- strictly smaller wins (lowest index among equals), second side --
        // smaller or equal wins (highest index among equals)
        int64_t d1 = MCF_INF, d2 = MCF_INF;
        int32_t k1 = -1, k2 = -1;
        for (int32_t i = 0; i < n1; ++i) { const int64_t r = res1[i]; if (r < d1) { d1 = r; k1 = i; } }
        for (int32_t i = 0; i < n2; ++i) { const int64_t r = res2[i]; if (r <= d2) { d2 = r; k2 = i; } }
        out->d1 = d1; out->k1 = k1; out->d2 = d2; out->k2 = k2;
        out->n1 = n1; out->n2 = n2;
        out->u = jn; out->w = jn; out->ru = rj; out->rw = rj;
        out->pu = pcur[jn]; out->pw = out->pu; out->su = out->pu; out->sw = out->pu;
        out->p0u = pu; out->p0w = pw; out->s0u = pu; out->s0w = pw; out->r0u = v.node[first]; out->r0w = v.node[second];
        out->small = 0;
    }
}

// The whole solve of one LDS-resident instance by one workgroup (k_solve_small: one instance per launch;
// k_solve_small_batch: one instance per workgroup of the launch)
__device__ __forceinline__ void solve_small_body(const McfView& g, const SmallLayout& L, int32_t rule,
                                                 McfCand* __restrict__ list, int64_t cap, char* smem, McfCtx* host_ctx) {
#ifdef MCF_STAMPS
    unsigned long long stamps_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { for (int i = 0; i < 24; ++i) mcf_stamp_acc[i] = 0; mcf_stamp_last = last_; }
#endif
    const uint32_t m_pad4 = (uint32_t)((g.m + 1023) / 1024 * 1024) * 4u;
    const uint32_t m_padb = m_pad4 >> 2;
    const uint32_t N = (uint32_t)g.n_nodes;
    const uint32_t arcw_b = (uint32_t)(g.m + N - 1) * 16u;
    McfView v = g;
    v.tail = reinterpret_cast<const int32_t*>(smem + L.tail);
    v.head = reinterpret_cast<const int32_t*>(smem + L.head);
    v.cost = reinterpret_cast<const int32_t*>(smem + L.cost);
    v.orig = reinterpret_cast<const int32_t*>(smem + L.orig);
    v.state = reinterpret_cast<int8_t*>(smem + L.state);
    v.weight = g.weight ? reinterpret_cast<float*>(smem + L.weight) : nullptr;
    v.arcw = reinterpret_cast<McfArcW*>(smem + L.arcw);
    v.pi = reinterpret_cast<int64_t*>(smem + L.pi);
    v.node = reinterpret_cast<McfNode*>(smem + L.node);
    v.order[0] = reinterpret_cast<int32_t*>(smem + L.order0);
    v.order[1] = reinterpret_cast<int32_t*>(smem + L.order1);
    v.path1 = reinterpret_cast<int32_t*>(smem + L.path1);
    v.path2 = reinterpret_cast<int32_t*>(smem + L.path2);
    v.ppos1 = reinterpret_cast<int32_t*>(smem + L.ppos1);
    v.ppos2 = reinterpret_cast<int32_t*>(smem + L.ppos2);
    v.rec1 = reinterpret_cast<McfNode*>(smem + L.rec1);
    v.rec2 = reinterpret_cast<McfNode*>(smem + L.rec2);
    v.seg = reinterpret_cast<McfSeg*>(smem + L.seg);
    v.ctx = reinterpret_cast<McfCtx*>(smem + L.ctx);
    v.posbuf[0] = reinterpret_cast<int32_t*>(smem + L.pos0);
    v.posbuf[1] = reinterpret_cast<int32_t*>(smem + L.pos1);
    v.psz[0] = nullptr;  // always the climb here (see the pivot step); mcf_create keeps no sizes for such a handle
    v.psz[1] = nullptr;
    v.reach = nullptr;
    v.bmeta[0] = nullptr; v.bmeta[1] = nullptr; v.candx = nullptr; v.dirty = nullptr; v.dirty_hdr = nullptr;   // (folds the blocked list etc. away)

    // (All fourteen arrays' loads in flight together -- one fused loop, 4- or 16-byte loads -- was measured SLOWER than these
    //  plain loops: 131 / 125 us against 117 us per 20-pivot launch.)
    copy_words(smem + L.tail, g.tail, m_pad4);
    copy_words(smem + L.head, g.head, m_pad4);
    copy_words(smem + L.cost, g.cost, m_pad4);
    copy_words(smem + L.orig, g.orig, m_pad4);
    copy_words(smem + L.state, g.state, m_padb);
    if (g.weight) copy_words(smem + L.weight, g.weight, m_pad4);
    copy_words(smem + L.arcw, g.arcw, arcw_b);
    copy_words(smem + L.pi, g.pi, N * 8u);
    copy_words(smem + L.node, g.node, N * 16u);
    copy_words(smem + L.order0, g.order[0], N * 4u);
    copy_words(smem + L.order1, g.order[1], N * 4u);
    copy_words(smem + L.pos0, g.posbuf[0], N * 4u);
    copy_words(smem + L.pos1, g.posbuf[1], N * 4u);
    copy_words(smem + L.ctx, g.ctx, (uint32_t)sizeof(McfCtx));
    __syncthreads();
    STAMP(0);

    McfCtx* c = v.ctx;
    if (threadIdx.x == 0) arm_ctx(c, cap);  // (what k_ctl does for the other paths; visible after the barrier below)
    // Devex: the host-made granule table in LDS (blocks move and resize under the tuner); other rules: whole buckets
    __shared__ int32_t s_gran[MCF_NUM_BUCKETS][MCF_GRANULES + 1];
    const bool devex = rule == MCF_RULE_DEVEX_BLOCK && g.dx;
    if (devex) {
        for (int q = threadIdx.x; q < MCF_NUM_BUCKETS * (MCF_GRANULES + 1); q += kSmallThreads)
            (&s_gran[0][0])[q] = (&g.dx->gran[0][0])[q];
    } else if (threadIdx.x < MCF_NUM_BUCKETS) {
        s_gran[threadIdx.x][0] = (int32_t)g.bucket_off[threadIdx.x];
        s_gran[threadIdx.x][MCF_GRANULES] = (int32_t)g.bucket_off[threadIdx.x + 1];
    }
    __syncthreads();
    // candidate-list rule: the list is the best arc of each head bucket (= of each of the 8 pricing
    // workgroups the three-kernel path would use at this size); minor iterations re-price just
    // those 8 arcs.  The list survives between launches in `list` (global).
    const bool listing = rule == MCF_RULE_CANDIDATE_LIST;
    __shared__ int64_t s_lk[MCF_NUM_BUCKETS], s_la[MCF_NUM_BUCKETS];
    if (listing && threadIdx.x < MCF_NUM_BUCKETS) { s_lk[threadIdx.x] = list[threadIdx.x].key; s_la[threadIdx.x] = list[threadIdx.x].arc; }
    __syncthreads();
    for (;;) {
        // uniform control values are read BEFORE a barrier: lane 0 rewrites them later in this very
        // iteration (mcf_pivot_walk), and a lagging wave must not see the new values
        const int32_t status_now = c->status;
        const bool minor = listing && c->minor_left > 0;
        int32_t bg0 = 0, bg1 = MCF_GRANULES;  // Devex: granule range of the current block
        if (devex) {
            const int32_t bg = c->block_granules;
            bg0 = (int32_t)c->block_index * bg;
            if (bg0 >= MCF_GRANULES) bg0 = 0;
            bg1 = bg0 + bg < MCF_GRANULES ? bg0 + bg : MCF_GRANULES;
        }
        __syncthreads();
        if (status_now != MCF_RUNNING) break;
        // ---- price: the arc set of k_price for shard 0 of 1
        int64_t key = 0, arc = -1;
        if (!minor) {
            // 128 lanes per head bucket, all eight buckets at once
            constexpr int kPer = kSmallThreads / MCF_NUM_BUCKETS;
            const int x = threadIdx.x / kPer, l = threadIdx.x % kPer;
            const int64_t lo = s_gran[x][bg0], hi = s_gran[x][bg1];
            for (int64_t i = lo + l; i < hi; i += kPer) {
                if (!v.state[i]) continue;
                const int64_t viol = mcf_violation(v, i);
                if (viol <= 0) continue;
                int64_t kk = mcf_dantzig_key(v, i, viol, (int32_t)v.state[i]);
                if (rule == MCF_RULE_DEVEX_BLOCK) {
                    const double merit = ((double)viol * (double)viol) / (double)v.weight[i];
                    kk = __double_as_longlong(merit);
                }
                const int64_t id = mcf_pack_arc(devex ? mcf_devex_tie_id(v.orig[i], v.state[i]) : v.orig[i], i);
                if (mcf_cand_better(kk, id, key, arc)) { key = kk; arc = id; }
            }
        }
        STAMP(1);
        if (listing) {
            __shared__ int64_t s_wk[kSmallThreads / 64], s_wa[kSmallThreads / 64];
            if (!minor) {
                wave_argmax(key, arc);
                if ((threadIdx.x & 63) == 0) { s_wk[threadIdx.x >> 6] = key; s_wa[threadIdx.x >> 6] = arc; }
                __syncthreads();
                if (threadIdx.x < MCF_NUM_BUCKETS) {  // two waves per bucket
                    const int w = 2 * threadIdx.x;
                    const bool second = mcf_cand_better(s_wk[w + 1], s_wa[w + 1], s_wk[w], s_wa[w]);
                    s_lk[threadIdx.x] = second ? s_wk[w + 1] : s_wk[w];
                    s_la[threadIdx.x] = second ? s_wa[w + 1] : s_wa[w];
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                key = 0; arc = -1;
                for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
                    const int64_t kk = minor ? mcf_minor_key(v, s_la[x]) : s_lk[x];
                    if (mcf_cand_better(kk, s_la[x], key, arc)) { key = kk; arc = s_la[x]; }
                }
            }
        } else {
            block_argmax<kSmallThreads>(key, arc);
        }
        STAMP(2);
        // ---- pivot: one lane walks, everything it touches is in LDS
        if (threadIdx.x == 0) {
            if (c->pivots < c->max_pivots) {
                int64_t priced = 0;
                for (int x = 0; x < MCF_NUM_BUCKETS; ++x) priced += s_gran[x][bg1] - s_gran[x][bg0];
                c->arcs_priced += minor ? MCF_NUM_BUCKETS : priced;
            }
        }
        // Cycle search: every lane is a node (small_cycle_parallel) when the tree fits the workgroup; else lanes 0 and 1 climb
        // one side each (pivot_climb_2lanes).  Lane 0 does the scalar rest.
        __shared__ int s_go, s_deep;
        __shared__ McfCycle s_cy;
        __shared__ SmallCycleAcc s_acc;
        if (threadIdx.x == 0) {
            MCF_PSTAMP(12);
            s_go = mcf_pivot_begin(v, key, arc, rule) ? 1 : 0;
            if (s_go) { const int32_t du = v.node[c->pv_first].depth, dw = v.node[c->pv_second].depth; s_deep = du > dw ? du : dw; }
            MCF_PSTAMP(13);
        }
        __syncthreads();
        if (s_go) {
            // (end points that hang close to the root -- the first pivots of a cold start -- are climbed: a level or two of
            //  LDS round trips beat the parallel search's fixed cost: 144 K vs 126 K pivots/s over the first 25 pivots)
#if defined(MCF_SMALL_CLIMB)   // A/B build: always the two-lane climb
            if (false) {
#else
            if (v.n_nodes <= kSmallThreads && s_deep > 3) {
#endif
                small_cycle_parallel<kSmallThreads>(v, s_acc, &s_cy);
                if (threadIdx.x == 0) { MCF_PSTAMP(15); mcf_pivot_decide(v, mcf_view_paths(v), s_cy); MCF_PSTAMP(16); }
            } else if (threadIdx.x < 2) {
                const bool ok = pivot_climb_2lanes(v, &s_cy);
                if (threadIdx.x == 0) {
                    if (ok) mcf_pivot_decide(v, mcf_view_paths(v), s_cy);
                    else c->status = MCF_INTERNAL_ERROR;
                }
            }
        }
        STAMP(3);
        __syncthreads();
        mcf_pivot_finish(v, mcf_view_paths(v), threadIdx.x, kSmallThreads);
        __syncthreads();
        STAMP(4);
        // ---- apply: block permutation of the preorder array + potential shift
        if (c->status == MCF_RUNNING || c->apply) {
            if (c->apply) {  // the descriptor stays in LDS (broadcast reads); a private copy would spill
                const int32_t lo = c->lo, hi = c->hi, plo = c->prev_lo, phi = c->prev_hi;
                for (int32_t j = lo + threadIdx.x; j < hi; j += kSmallThreads) mcf_apply_one(v, *c, j);
                for (int32_t j = plo + threadIdx.x; j < phi; j += kSmallThreads)
                    if (j < lo || j >= hi) mcf_apply_one(v, *c, j);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) c->apply = 0;
        __syncthreads();
        STAMP(5);
    }

    // at the budget: is any arc still eligible?  (simplex.py:1678-1699; saves the host a pricing pass + three syncs)
    if (c->status == MCF_PIVOT_LIMIT && !c->limit_checked) {
        __shared__ int s_any;
        if (threadIdx.x == 0) s_any = 0;
        __syncthreads();
        int any = 0;
        for (int64_t i = threadIdx.x; i < g.m && !any; i += kSmallThreads)
            if (v.state[i] && mcf_violation(v, i) > 0) any = 1;
        if (any) s_any = 1;
        __syncthreads();
        if (threadIdx.x == 0) { if (s_any) c->limit_checked = 1; else c->status = MCF_OPTIMAL; }
        __syncthreads();
    }
    copy_words(g.state, smem + L.state, m_padb);
    if (g.weight) copy_words(g.weight, smem + L.weight, m_pad4);
    copy_words(g.arcw, smem + L.arcw, arcw_b);
    copy_words(g.pi, smem + L.pi, N * 8u);
    copy_words(g.node, smem + L.node, N * 16u);
    copy_words(g.order[0], smem + L.order0, N * 4u);
    copy_words(g.order[1], smem + L.order1, N * 4u);
    copy_words(g.posbuf[0], smem + L.pos0, N * 4u);
    copy_words(g.posbuf[1], smem + L.pos1, N * 4u);
    copy_words(g.ctx, smem + L.ctx, (uint32_t)sizeof(McfCtx));
    // ... and straight into the host's pinned copy: the host then only waits for the stream, no read-back copy
    if (host_ctx) copy_words(host_ctx, smem + L.ctx, (uint32_t)sizeof(McfCtx));
    if (listing && threadIdx.x < MCF_NUM_BUCKETS) list[threadIdx.x] = McfCand{s_lk[threadIdx.x], s_la[threadIdx.x]};
#ifdef MCF_STAMPS
    STAMP(6);
    if (threadIdx.x == 0) {  // diagnostic build only: cycle sums into the (global) path scratch, read by mcf_debug_stamps
        unsigned long long* out = reinterpret_cast<unsigned long long*>(g.rec1);
        for (int i = 0; i < 8; ++i) out[i] = stamps_[i];
        for (int i = 0; i < 24; ++i) g_pivot_stamps[i] += mcf_stamp_acc[i];
    }
#endif
}

__global__ __launch_bounds__(kSmallThreads) void k_solve_small(McfView g, SmallLayout L, int32_t rule,
                                                                McfCand* __restrict__ list, int64_t cap, McfCtx* host_ctx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    solve_small_body(g, L, rule, list, cap, smem, host_ctx);
}

// Many independent small instances side by side: workgroup b solves jobs[b] from start to finish in its own CU's LDS.
// Nothing is shared between the workgroups, so a batch of >= 256 instances keeps every CU of the chip busy with
// latency-bound work that a single instance can only ever give one CU of (mcf_solve_batch).
struct SmallJob {
    McfView g;
    SmallLayout L;
    int32_t rule;
    int32_t pad;
    McfCand* list;
    int64_t cap;
    McfCtx* host_ctx;   // pinned: the final control block goes straight to the host
};

__global__ __launch_bounds__(kSmallThreads) void k_solve_small_batch(const SmallJob* __restrict__ jobs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SmallJob& J = jobs[blockIdx.x];   // uniform per workgroup: scalar loads
    solve_small_body(J.g, J.L, J.rule, J.list, J.cap, smem, J.host_ctx);
}

// ------------------------------------------------------------------ k_ctl: (re)arm the control block
__global__ void k_ctl(McfCtx* c, int64_t max_pivots, int resume) {
    c->max_pivots = max_pivots;
    c->limit_checked = 0;
    if (resume && c->status == MCF_PIVOT_LIMIT && c->pivots < max_pivots) c->status = MCF_RUNNING;
}

__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace

// ====================================================================== handle
struct mcf_handle {
    McfHostImage im;
    mcf_options opt{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool stream_owned = true;   // false: borrowed from the per-device pool (small / mid-size instances)
    // device arrays
    int32_t *d_tail = nullptr, *d_head = nullptr, *d_cost = nullptr, *d_orig = nullptr;
    int8_t* d_state = nullptr;
    int8_t* d_prio = nullptr;    // MCF_KEY_PRIORITY: per-arc preference bits, engine order
    float* d_weight = nullptr;
    McfArcW* d_arcw = nullptr;
    int64_t* d_pi = nullptr;
    McfNode* d_node = nullptr;
    int32_t *d_order0 = nullptr, *d_order1 = nullptr, *d_path1 = nullptr, *d_path2 = nullptr, *d_ppos1 = nullptr, *d_ppos2 = nullptr;
    McfNode *d_rec1 = nullptr, *d_rec2 = nullptr;
    int64_t *d_rcache = nullptr, *d_adj_off = nullptr, *d_adj = nullptr;
    int32_t* d_vkey = nullptr;   // compressed Dantzig keys (4 B per arc), see McfView::vkey
    bool nt_sweep = false;       // the key-code sweep is larger than the Infinity Cache: non-temporal loads
    int32_t *d_pos0 = nullptr, *d_pos1 = nullptr, *d_psz0 = nullptr, *d_psz1 = nullptr;
    int32_t *d_reach = nullptr, *d_chg = nullptr;  // coarse index over the position-space sizes + its scratch
    // blocked preorder list (large trees): d_order0/1 and d_psz0/1 are then the two slot arenas, d_pos0 is loc[]
    bool bpl = false;
    int bpl_shift = 0, bpl_cap = 0, bpl_dense = 0, bpl_grid = 1;
    McfBlkMeta* d_bmeta[2] = {nullptr, nullptr};
    int32_t* d_bext[2] = {nullptr, nullptr};
    McfDirty* d_dirty = nullptr;
    int64_t* d_swept = nullptr;  // arcs swept per pricing workgroup (summed by mcf_get_result)
    McfDevex* d_dx = nullptr;        // Devex: granule table + touched-weight list
    int32_t* d_full_tab = nullptr;   // sharded full sweeps: (lo, hi) of this rank's share of bucket x, [8][2]
    bool rcached = false;     // large instance: resident reduced costs + k_rcupd
    int rcupd_blocks = 1;
    McfSeg* d_seg = nullptr;
    McfCtx* d_ctx = nullptr;
    McfCand* d_cand = nullptr;
    McfCandX* d_candx = nullptr;    // candidate cache (candidate-list rule, single GPU, grid path)
    McfCand* d_cand_aux = nullptr;  // scratch for mcf_price_once / mcf_time_pricing: the live list in d_cand must survive them
    McfCand* d_one = nullptr;
    McfCtx* h_ctx = nullptr;   // pinned
    McfCand* h_one = nullptr;  // pinned
    McfView view{};
    int price_blocks = 1;
    int apply_blocks = 1;
    int32_t climb_budget = INT32_MAX;  // round trips the cycle climb may take before the scan takes over
    bool small = false;       // whole instance fits in LDS: fused single-workgroup pivot loop
    bool mid = false;         // mid-size instance: persistent single-workgroup pivot loop over global memory (k_solve_mid)
    SmallLayout small_layout{};
    int64_t shard = 0, shards = 1;
    int64_t shard_arcs = 0;  // arcs of this rank's shard (all buckets)
    int64_t priced_per_pass = 0;
    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_batch = 0;
    // overlapped graphs: pricing of pivot t+1 runs beside the tree permutation of pivot t (needs only the reduced-cost half)
    bool overlap = false;
    bool overlap_cfg = false;   // what mcf_create decided (dropping the reduced costs turns it off until the next reset)
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> fork_ev;   // two per slot: pivot done (main -> side), priced (side -> main)
    // profiling events
    std::vector<hipEvent_t> events;
    hipEvent_t loop_ev[2] = {nullptr, nullptr};  // around every launch of a persistent pivot loop
    // bookkeeping
    mcf_stats stats{};
    int64_t total_cap = 0;
    std::string err;
    bool solved_once = false;
    // resident reduced costs -> pricing from the potentials, once the re-hung subtrees have grown so large that patching
    // the reduced costs of their incident arcs costs more per pivot than the dearer sweeps (mcf_solve decides between batches)
    // candidate-list handles on the blocked list: pivots back to back in one workgroup while the updates are small (k_pivot_run)
    int run_pairs = 0;         // > 0: a list period is one sweep + this many (k_pivot_run, k_update_bpl) pairs; 0: one k_pivot per slot
    int run_low = 0;           // consecutive batches that made poor use of their run launches
    int run_pairs_cfg = 0;     // what mcf_create decided (a reset goes back to it)
    int64_t run_seen = 0;      // pivot count at the last look
    bool rc_able = false;      // the handle was created with resident reduced costs
    bool rc_dropped = false;   // ... and has stopped keeping them (until the next reset / warm start)
    int64_t rc_drop_subtree = 0;   // average |T2| over a batch from which on they are dropped (0: never)
    int64_t sw_pivots = 0, sw_subtree = 0;   // counters at the last look
    bool ctx_current = false;  // *h_ctx equals the device control block (no kernel was enqueued since it was read)
    bool external_driver = false;  // the caller enqueues the pivots itself (mcf_enqueue_*), possibly by replaying a graph it
                                   // captured: the library cannot know when the control block changes, so every read goes to the device
};

namespace {

#define HIP_TRY(h, expr)                                                                         \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                        \
            return MCF_E_HIP;                                                                    \
        }                                                                                        \
    } while (0)

int usable_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

template <typename T>
hipError_t dalloc(T** p, size_t count) { return hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T)); }

// Measured on this stack (scripts/bin/hipcost): hipStreamCreate 3.7 ms, hipStreamDestroy 2.5 ms, hipHostFree 0.22 ms --
// against 4 ms for a whole netgen_8_08a solve and 16 us per instance in a batch (hipMalloc / hipFree: microseconds).
// Handles of small and mid-size instances therefore borrow a stream from a per-device pool (round robin, never
// destroyed) and their pinned control-block copy from a pool of slots; large handles keep a stream of their own.
constexpr int kPoolStreams = 4;
constexpr int kPooledMaxNodes = 1 << 14;
constexpr size_t kPinnedSlot = ((sizeof(McfCtx) + 63) & ~(size_t)63) + 64;   // control block + one candidate
std::mutex g_pool_mu;
std::vector<hipStream_t> g_pool_streams[64];    // per device
unsigned g_pool_next[64];
std::vector<char*> g_pinned_free;               // free slots

hipError_t pool_stream(int device, hipStream_t* out) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    std::vector<hipStream_t>& v = g_pool_streams[device & 63];
    if ((int)v.size() < kPoolStreams) {
        hipStream_t s = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        v.push_back(s);
        *out = s;
        return hipSuccess;
    }
    *out = v[g_pool_next[device & 63]++ % kPoolStreams];
    return hipSuccess;
}

hipError_t pinned_take(char** out) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (g_pinned_free.empty()) {
        constexpr int kSlots = 256;
        char* block = nullptr;
        const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&block), kSlots * kPinnedSlot, hipHostMallocPortable);
        if (e != hipSuccess) return e;
        for (int i = kSlots - 1; i >= 0; --i) g_pinned_free.push_back(block + (size_t)i * kPinnedSlot);
    }
    *out = g_pinned_free.back();
    g_pinned_free.pop_back();
    return hipSuccess;
}
void pinned_give(char* slot) {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    g_pinned_free.push_back(slot);
}

// Host-to-device copies of an image go through a pinned staging buffer (one per host thread) when they fit: a copy from
// pageable memory costs ~34 us here whatever its size, from pinned memory a few (mcf_create of a small instance: ~15 copies).
constexpr size_t kStageBytes = (size_t)2 << 20;
struct Stage {
    char* base = nullptr;
    size_t used = 0;
    bool tried = false;
};
Stage g_stage;              // one per process (a per-thread one leaked 2 MiB of pinned memory for every host thread that ever created a handle)
std::mutex g_stage_mu;      // held by upload_image from its first staged copy to the synchronisation that ends it

hipError_t h2d(mcf_handle* h, void* dst, const void* src, size_t bytes) {
    Stage& st = g_stage;
    if (!st.tried) {
        st.tried = true;
        if (hipHostMalloc(reinterpret_cast<void**>(&st.base), kStageBytes, hipHostMallocPortable) != hipSuccess) { st.base = nullptr; (void)hipGetLastError(); }
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (st.base && st.used + need <= kStageBytes) {
        char* p = st.base + st.used;
        st.used += need;
        std::memcpy(p, src, bytes);
        return hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, h->stream);
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream);
}

int upload_image(mcf_handle* h) {
    const McfHostImage& im = h->im;
    std::lock_guard<std::mutex> stage_lock(g_stage_mu);
    g_stage.used = 0;   // (everything staged below is synchronised before this function returns)
    if (h->rc_dropped) {   // a fresh start keeps the reduced costs resident again
        h->rc_dropped = false;
        h->rcached = true;
        h->overlap = h->overlap_cfg;
        h->view.rcache = h->d_rcache; h->view.vkey = h->d_vkey; h->view.dirty = h->d_dirty;
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
        if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
        h->graph_batch = 0;
    }
    h->sw_pivots = 0; h->sw_subtree = 0;
    if (h->run_pairs != h->run_pairs_cfg) {   // (a fresh start gets the run shape back)
        h->run_pairs = h->run_pairs_cfg;
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
        if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
        h->graph_batch = 0;
    }
    h->run_low = 0; h->run_seen = 0;
    HIP_TRY(h, h2d(h, h->d_tail, im.tail.data(), im.m_pad * 4));
    HIP_TRY(h, h2d(h, h->d_head, im.head.data(), im.m_pad * 4));
    HIP_TRY(h, h2d(h, h->d_cost, im.cost.data(), im.m_pad * 4));
    HIP_TRY(h, h2d(h, h->d_orig, im.orig.data(), im.m_pad * 4));
    HIP_TRY(h, h2d(h, h->d_state, im.state.data(), im.m_pad));
    HIP_TRY(h, h2d(h, h->d_weight, im.weight.data(), im.m_pad * 4));
    HIP_TRY(h, h2d(h, h->d_arcw, im.arcw.data(), im.arcw.size() * sizeof(McfArcW)));
    HIP_TRY(h, h2d(h, h->d_pi, im.pi.data(), im.pi.size() * 8));
    HIP_TRY(h, h2d(h, h->d_node, im.node.data(), im.node.size() * sizeof(McfNode)));
    std::vector<int32_t> reach;  // (outlives the asynchronous copy: synchronised at the end of this function)
    McfBplImage bp;              // (likewise)
    if (h->bpl) {
        // blocked preorder list: the dense image cut into blocks (slot = position), both meta copies alike, arena 1 empty
        mcf_bpl_build(im, h->bpl_shift, h->bpl_cap - h->bpl_dense > 0 ? h->bpl_cap - h->bpl_dense : -1, bp);
        if (bp.cap != h->bpl_cap || bp.shift != h->bpl_shift) { h->err = "internal: blocked-list geometry"; return MCF_E_INTERNAL; }
        HIP_TRY(h, hipMemcpyAsync(h->d_order0, bp.tok[0].data(), bp.tok[0].size() * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_psz0, bp.psz[0].data(), bp.psz[0].size() * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_psz1, 0, bp.psz[1].size() * 4, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_pos0, bp.loc.data(), bp.loc.size() * 4, hipMemcpyHostToDevice, h->stream));
        for (int a = 0; a < 2; ++a) {
            HIP_TRY(h, hipMemcpyAsync(h->d_bmeta[a], bp.meta[a].data(), bp.meta[a].size() * sizeof(McfBlkMeta), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->d_bext[a], bp.ext[a].data(), bp.ext[a].size() * 4, hipMemcpyHostToDevice, h->stream));
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));   // (pageable sources)
    } else {
    HIP_TRY(h, h2d(h, h->d_order0, im.order.data(), im.order.size() * 4));
    HIP_TRY(h, h2d(h, h->d_order1, im.order.data(), im.order.size() * 4));
    HIP_TRY(h, h2d(h, h->d_pos0, im.pos.data(), im.pos.size() * 4));
    HIP_TRY(h, h2d(h, h->d_pos1, im.pos.data(), im.pos.size() * 4));
    if (h->d_psz0) {
        HIP_TRY(h, h2d(h, h->d_psz0, im.psize.data(), im.psize.size() * 4));
        HIP_TRY(h, h2d(h, h->d_psz1, im.psize.data(), im.psize.size() * 4));
        if (h->d_reach) {
            const int64_t nb = ((int64_t)im.n_nodes + MCF_REACH_BLOCK - 1) / MCF_REACH_BLOCK;
            reach.assign((size_t)nb, 0);
            for (int64_t j = 0; j < im.n_nodes; ++j) {
                const int32_t end = (int32_t)(j + im.psize[j]);
                if (end > reach[j >> MCF_REACH_SHIFT]) reach[j >> MCF_REACH_SHIFT] = end;
            }
            HIP_TRY(h, h2d(h, h->d_reach, reach.data(), reach.size() * 4));
        }
    }
    }
    if (h->rcached)
        HIP_TRY(h, h2d(h, h->d_rcache, im.rcache.data(), im.m_pad * 8));
    std::vector<int32_t> vk;  // (synchronised before this function returns)
    if (h->d_vkey) {
        vk.assign(im.m_pad, 0);
        for (int64_t e = 0; e < im.m; ++e) vk[e] = mcf_vkey(-(int64_t)im.state[e] * im.rcache[e], h->view.vk_bigm, h->view.vk_half);
        HIP_TRY(h, h2d(h, h->d_vkey, vk.data(), vk.size() * 4));
    }
    if (h->d_dirty) HIP_TRY(h, hipMemsetAsync(h->d_dirty->flag, 1, sizeof(h->d_dirty->flag), h->stream));  // every block is due (any non-zero word)
    if (h->d_swept) HIP_TRY(h, hipMemsetAsync(h->d_swept, 0, kMaxPriceBlocks * sizeof(int64_t), h->stream));
    McfCtx c;
    std::memset(&c, 0, sizeof c);
    c.unbounded_arc = -1;
    const int64_t m = im.m;
    mcf_init_block_state(&c, h->opt.rule, m, h->opt.block_size);
    if (h->opt.rule == MCF_RULE_DEVEX_BLOCK) {
        if (h->opt.devex_tuner > 0) c.auto_tune = 1; else if (h->opt.devex_tuner < 0) c.auto_tune = 0;
        if (h->opt.devex_stay > 0) c.devex_cyclic = 0;
    }
    c.minor_cap = mcf_minor_cap(h->price_blocks);
    c.climb_budget = h->climb_budget;
    // shallow end points are climbed outright: one round trip per level beats the scan's fixed passes up to ~3 levels at
    // mid size (one plain round) and ~8 levels where the coarse index is used (two passes over up to 16 K words first)
    c.climb_depth = h->opt.climb_depth > 0 ? h->opt.climb_depth : (h->opt.climb_depth < 0 ? 0 : (im.n_nodes > 32768 ? 8 : 3));
    if (h->bpl) { c.arena = 0; c.alloc_next = h->bpl_dense; c.dense_blocks = h->bpl_dense; }
    *h->h_ctx = c;
    if (h->shards > 1 && !h->d_full_tab) {
        int32_t tab[MCF_NUM_BUCKETS * 2];
        for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
            int64_t lo, hi;
            mcf_bucket_slice(im.bucket_off, x, h->shard, h->shards, 0, 1, &lo, &hi);
            tab[x * 2] = (int32_t)lo; tab[x * 2 + 1] = (int32_t)hi;
        }
        if (dalloc(&h->d_full_tab, MCF_NUM_BUCKETS * 2) != hipSuccess) { h->err = "hipMalloc slice table"; return MCF_E_ALLOC; }
        HIP_TRY(h, hipMemcpy(h->d_full_tab, tab, sizeof tab, hipMemcpyHostToDevice));
    }
    if (h->opt.rule == MCF_RULE_DEVEX_BLOCK && !h->d_dx) {  // granule table, once
        McfDevex* dx = new (std::nothrow) McfDevex();
        if (!dx) { h->err = "host allocation"; return MCF_E_ALLOC; }
        std::memset(dx, 0, sizeof *dx);
        mcf_devex_fill_granules(dx, im.bucket_off);
        hipError_t de = dalloc(&h->d_dx, 1);
        if (de == hipSuccess) de = hipMemcpy(h->d_dx, dx, sizeof *dx, hipMemcpyHostToDevice);
        delete dx;
        if (de != hipSuccess) { h->err = "hipMalloc / copy granule table"; return MCF_E_ALLOC; }
        h->view.dx = h->d_dx;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_ctx, h->h_ctx, sizeof(McfCtx), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // per-pass accounting: a Devex pass prices one block of the shard, Dantzig the whole shard
    h->shard_arcs = 0;
    for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
        int64_t lo, hi;
        mcf_bucket_slice(im.bucket_off, x, h->shard, h->shards, 0, 1, &lo, &hi);
        h->shard_arcs += hi - lo;
    }
    h->priced_per_pass = h->opt.rule == MCF_RULE_DEVEX_BLOCK ? h->shard_arcs * c.block_granules / MCF_GRANULES : h->shard_arcs;
    std::memset(&h->stats, 0, sizeof h->stats);
    h->stats.unbounded_arc = -1;
    {   // compulsory bytes of ONE pricing launch of this handle's sweep kernel
        const bool devex = h->opt.rule == MCF_RULE_DEVEX_BLOCK;
        if (h->d_vkey && !devex) h->stats.price_bytes = 4 * h->priced_per_pass;                       // k_price_v: key codes
        else if (h->rcached) h->stats.price_bytes = (devex ? 13 : 9) * h->priced_per_pass;            // k_price_rc: rc + state (+ weight)
        else h->stats.price_bytes = (devex ? 17 : 13) * h->priced_per_pass + 8 * (int64_t)im.n_nodes; // k_price: SURVEY 8d's gather figure
    }
    h->total_cap = 0;
    h->solved_once = false;
    h->ctx_current = true;  // *h_ctx was just copied to the device
    h->external_driver = false;
    return MCF_OK;
}

void launch_price(mcf_handle* h, hipStream_t s, const McfView& v_in, int32_t rule, int use_block, McfCand* out = nullptr) {
    int64_t* swept = out ? nullptr : h->d_swept;  // measurement / parity launches (own candidate buffer) are not accounted
    McfView vx = v_in;
    if (out) vx.candx = nullptr;   // ... and leave the live list's records alone
    const McfView& v = vx;
    if (!out) out = h->d_cand;
    const dim3 grid(h->price_blocks), block(kPriceThreads);
    const int64_t z = 0;
    // use_block: 0 = whole shard, 1 = current Devex block, 2 = whole shard unless minor iterations are pending
    if (rule == MCF_RULE_CANDIDATE_LIST && use_block) use_block = 2;
    if (h->rcached && v.vkey && rule != MCF_RULE_DEVEX_BLOCK) {
        const bool nt = h->nt_sweep;
        if (v.dirty) hipLaunchKernelGGL((k_price_v<true>), grid, block, 0, s, v, h->shard, h->shards, use_block, out, swept, (const int32_t*)h->d_full_tab);
        else if (nt) hipLaunchKernelGGL((k_price_v<false, true>), grid, block, 0, s, v, h->shard, h->shards, use_block, out, swept, (const int32_t*)h->d_full_tab);
        else hipLaunchKernelGGL((k_price_v<false>), grid, block, 0, s, v, h->shard, h->shards, use_block, out, swept, (const int32_t*)h->d_full_tab);
    } else if (h->rcached) {
        if (rule == MCF_RULE_DEVEX_BLOCK)
            hipLaunchKernelGGL((k_price_rc<MCF_RULE_DEVEX_BLOCK, false, false>), grid, block, 0, s, v, h->shard, h->shards, use_block, z, z, out, swept, (const int32_t*)nullptr);
        else
            if (v.dirty) hipLaunchKernelGGL((k_price_rc<MCF_RULE_DANTZIG, false, true>), grid, block, 0, s, v, h->shard, h->shards, use_block, z, z, out, swept, (const int32_t*)h->d_full_tab);
            else hipLaunchKernelGGL((k_price_rc<MCF_RULE_DANTZIG, false, false>), grid, block, 0, s, v, h->shard, h->shards, use_block, z, z, out, swept, (const int32_t*)h->d_full_tab);
    } else {
        if (rule == MCF_RULE_DEVEX_BLOCK)
            hipLaunchKernelGGL((k_price<MCF_RULE_DEVEX_BLOCK, false>), grid, block, 0, s, v, h->shard, h->shards, use_block, z, z, out);
        else
            hipLaunchKernelGGL((k_price<MCF_RULE_DANTZIG, false>), grid, block, 0, s, v, h->shard, h->shards, use_block, z, z, out);
    }
}

void launch_k_pivot(mcf_handle* h, hipStream_t s, const McfCand* cand, int ncand, int32_t rule, int have_sweep) {
    const McfCand* own = h->d_cand;
    const dim3 one(1), block(kPivotThreads);
    if (h->view.dirty) {
        if (h->bpl) hipLaunchKernelGGL((k_pivot<true, true>), one, block, 0, s, h->view, cand, ncand, rule, have_sweep, own);
        else hipLaunchKernelGGL((k_pivot<true, false>), one, block, 0, s, h->view, cand, ncand, rule, have_sweep, own);
    } else {
        if (h->bpl) hipLaunchKernelGGL((k_pivot<false, true>), one, block, 0, s, h->view, cand, ncand, rule, have_sweep, own);
        else hipLaunchKernelGGL((k_pivot<false, false>), one, block, 0, s, h->view, cand, ncand, rule, have_sweep, own);
    }
}

void launch_apply(mcf_handle* h, hipStream_t s) {
    if (h->bpl) {  // blocked preorder list: block records, T2 move and reduced-cost patch in one launch
        const dim3 grid(h->bpl_grid + 1 + (h->view.candx ? 1 : 0)), block(kBplThreads);   // (+1: the direct workgroup, +1: the candidate cache)
        if (!h->rcached) hipLaunchKernelGGL((k_update_bpl<false, false>), grid, block, 0, s, h->view, h->bpl_grid);
        else if (h->view.dirty) hipLaunchKernelGGL((k_update_bpl<true, true>), grid, block, 0, s, h->view, h->bpl_grid);
        else hipLaunchKernelGGL((k_update_bpl<false, true>), grid, block, 0, s, h->view, h->bpl_grid);
        return;
    }
    const int xb = h->view.candx ? kCandxBlocks : 0;
    if (h->rcached) {  // tree/potential update and reduced-cost update in one launch
        if (h->view.dirty) hipLaunchKernelGGL(k_update<true>, dim3(h->apply_blocks + h->rcupd_blocks + xb), dim3(kRcupdThreads), 0, s, h->view, h->apply_blocks);
        else hipLaunchKernelGGL(k_update<false>, dim3(h->apply_blocks + h->rcupd_blocks + xb), dim3(kRcupdThreads), 0, s, h->view, h->apply_blocks);
    } else {
        hipLaunchKernelGGL(k_apply, dim3(h->apply_blocks + xb), dim3(kApplyThreads), 0, s, h->view);
    }
}

// One list period in "run" shape: the sweep (a no-op while the list is live), then `run_pairs` pairs of (pivots back to back
// until an update is too large for one workgroup or the list is used up, that update on the grid).
void launch_run_period(mcf_handle* h, hipStream_t s) {
    const int32_t rule = h->opt.rule;
    const int steps = mcf_minor_cap(h->price_blocks) + 1;
    launch_price(h, s, h->view, rule, 1);
    for (int i = 0; i < h->run_pairs; ++i) {
        const dim3 one(1), block(kPivotThreads);
        const int sweep = i == 0 ? 1 : 0;
        if (h->rcached) {
            if (h->view.dirty) hipLaunchKernelGGL((k_pivot_run<true, true>), one, block, 0, s, h->view, h->price_blocks, sweep, steps);
            else hipLaunchKernelGGL((k_pivot_run<false, true>), one, block, 0, s, h->view, h->price_blocks, sweep, steps);
        } else {
            hipLaunchKernelGGL((k_pivot_run<false, false>), one, block, 0, s, h->view, h->price_blocks, sweep, steps);
        }
        launch_apply(h, s);
    }
}

// One pivot slot.  Candidate-list rule: only every (minor_cap + 1)-th slot carries a pricing launch;
// the slots in between go straight to k_pivot, which re-prices the list (a pricing launch there
// would be a no-op anyway -- this just saves its launch boundary).
// arm_cap >= 0 (persistent loop outside a graph): the kernel takes the pivot cap from its argument, no k_ctl launch
void launch_pivot_triplet(mcf_handle* h, hipStream_t s, int slot = 0, int64_t arm_cap = -1) {
    const int32_t rule = h->opt.rule;
    if (h->mid) {
        // candidate list: one grid sweep (a no-op while the list is live), then the persistent loop runs the major
        // pivot and every minor pivot the list yields.  Other rules: the loop prices by itself until the solve ends.
        const bool listing = rule == MCF_RULE_CANDIDATE_LIST;
        if (listing) launch_price(h, s, h->view, rule, 1);
        hipLaunchKernelGGL(k_solve_mid, dim3(1), dim3(kPivotThreads), 0, s, h->view, rule, h->d_cand, h->price_blocks,
                           listing ? 1 : 0, listing ? mcf_minor_cap(h->price_blocks) + 2 : (1 << 22), arm_cap >= 0 ? 1 : 0,
                           arm_cap >= 0 ? arm_cap : (int64_t)0);
        return;
    }
    int have_sweep = 1;
    if (rule == MCF_RULE_CANDIDATE_LIST) have_sweep = slot % (mcf_minor_cap(h->price_blocks) + 1) == 0;
    if (have_sweep) launch_price(h, s, h->view, rule, rule != MCF_RULE_DANTZIG);
    launch_k_pivot(h, s, h->d_cand, h->price_blocks, rule, have_sweep);
    launch_apply(h, s);
}

int build_graph(mcf_handle* h, int batch) {
    if (h->graph_exec && h->graph_batch == batch) return MCF_OK;
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    if (h->overlap) {
        while (h->fork_ev.size() < (size_t)batch * 2) {
            hipEvent_t e;
            HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            h->fork_ev.push_back(e);
        }
    }
    HIP_TRY(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    if (h->run_pairs > 0) {
        const int period = mcf_minor_cap(h->price_blocks) + 1;
        for (int i = 0; i < batch / period; ++i) launch_run_period(h, h->stream);
    } else if (!h->overlap) {
        for (int i = 0; i < batch; ++i) launch_pivot_triplet(h, h->stream, i);
    } else {
        // main:  [price 0] pivot t -> permutation t ------------------------> pivot t+1 ...
        // side:            \-> reduced-cost patch t -> price t+1 ---------/
        // k_price reads what k_pivot and the patch wrote (control block, states, weights, reduced costs / key codes, dirty
        // flags); the permutation writes order / pos / psz / pi / reach only, which pricing from resident values never reads.
        const int32_t rule = h->opt.rule;
        launch_price(h, h->stream, h->view, rule, rule != MCF_RULE_DANTZIG);
        for (int i = 0; i < batch; ++i) {
            hipEvent_t pivoted = h->fork_ev[(size_t)i * 2], priced = h->fork_ev[(size_t)i * 2 + 1];
            launch_k_pivot(h, h->stream, h->d_cand, h->price_blocks, rule, 1);
            HIP_TRY(h, hipEventRecord(pivoted, h->stream));
            HIP_TRY(h, hipStreamWaitEvent(h->side, pivoted, 0));
            if (h->view.dirty) hipLaunchKernelGGL(k_rcupd<true>, dim3(h->rcupd_blocks), dim3(kRcupdThreads), 0, h->side, h->view);
            else hipLaunchKernelGGL(k_rcupd<false>, dim3(h->rcupd_blocks), dim3(kRcupdThreads), 0, h->side, h->view);
            if (i + 1 < batch) launch_price(h, h->side, h->view, rule, rule != MCF_RULE_DANTZIG);
            HIP_TRY(h, hipEventRecord(priced, h->side));
            hipLaunchKernelGGL(k_apply, dim3(h->apply_blocks), dim3(kApplyThreads), 0, h->stream, h->view);
            HIP_TRY(h, hipStreamWaitEvent(h->stream, priced, 0));
        }
    }
    HIP_TRY(h, hipStreamEndCapture(h->stream, &h->graph));
    HIP_TRY(h, hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
    h->graph_batch = batch;
    return MCF_OK;
}

// One batch of pivots with every kernel bracketed by events (profile mode).
int run_batch_profiled(mcf_handle* h, int batch) {
    const size_t need = (size_t)batch * 4;
    while (h->events.size() < need) {
        hipEvent_t e;
        HIP_TRY(h, hipEventCreate(&e));
        h->events.push_back(e);
    }
    const int32_t rule = h->opt.rule;
    for (int i = 0; i < batch; ++i) {
        hipEvent_t* ev = &h->events[(size_t)i * 4];
        HIP_TRY(h, hipEventRecord(ev[0], h->stream));
        launch_price(h, h->stream, h->view, rule, rule != MCF_RULE_DANTZIG);
        HIP_TRY(h, hipEventRecord(ev[1], h->stream));
        launch_k_pivot(h, h->stream, h->d_cand, h->price_blocks, rule, 1);
        HIP_TRY(h, hipEventRecord(ev[2], h->stream));
        launch_apply(h, h->stream);
        HIP_TRY(h, hipEventRecord(ev[3], h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < batch; ++i) {
        hipEvent_t* ev = &h->events[(size_t)i * 4];
        float a = 0, b = 0, c = 0;
        (void)hipEventElapsedTime(&a, ev[0], ev[1]);
        (void)hipEventElapsedTime(&b, ev[1], ev[2]);
        (void)hipEventElapsedTime(&c, ev[2], ev[3]);
        h->stats.price_ms += a; h->stats.pivot_ms += b; h->stats.apply_ms += c;
    }
    h->stats.price_launches += batch; h->stats.pivot_launches += batch; h->stats.apply_launches += batch;
    return MCF_OK;
}

int read_ctx(mcf_handle* h, hipStream_t s) {
    HIP_TRY(h, hipMemcpyAsync(h->h_ctx, h->d_ctx, sizeof(McfCtx), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    h->ctx_current = true;
    return MCF_OK;
}
// the host copy of the control block, refreshed only when something ran since it was last read
int sync_ctx(mcf_handle* h, hipStream_t s) { return (h->ctx_current && !h->external_driver) ? MCF_OK : read_ctx(h, s); }

void free_all(mcf_handle* h) {
    if (h->stream) (void)hipStreamSynchronize(h->stream);   // (a borrowed stream may still hold this handle's work)
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->fork_ev) (void)hipEventDestroy(e);
    if (h->side) (void)hipStreamDestroy(h->side);
    for (hipEvent_t e : h->loop_ev) if (e) (void)hipEventDestroy(e);
    (void)hipFree(h->d_tail); (void)hipFree(h->d_head); (void)hipFree(h->d_cost); (void)hipFree(h->d_orig); (void)hipFree(h->d_state); (void)hipFree(h->d_prio); (void)hipFree(h->d_weight);
    (void)hipFree(h->d_arcw); (void)hipFree(h->d_pi); (void)hipFree(h->d_node); (void)hipFree(h->d_order0); (void)hipFree(h->d_order1);
    (void)hipFree(h->d_path1); (void)hipFree(h->d_path2); (void)hipFree(h->d_ppos1); (void)hipFree(h->d_ppos2); (void)hipFree(h->d_rec1); (void)hipFree(h->d_rec2); (void)hipFree(h->d_seg); (void)hipFree(h->d_ctx); (void)hipFree(h->d_cand); (void)hipFree(h->d_cand_aux); (void)hipFree(h->d_one);
    (void)hipFree(h->d_pos0); (void)hipFree(h->d_pos1); (void)hipFree(h->d_psz0); (void)hipFree(h->d_psz1); (void)hipFree(h->d_reach); (void)hipFree(h->d_chg); (void)hipFree(h->d_dirty); (void)hipFree(h->d_swept); (void)hipFree(h->d_dx); (void)hipFree(h->d_full_tab);
    (void)hipFree(h->d_rcache); (void)hipFree(h->d_adj_off); (void)hipFree(h->d_adj); (void)hipFree(h->d_vkey); (void)hipFree(h->d_candx);
    for (int a = 0; a < 2; ++a) { (void)hipFree(h->d_bmeta[a]); (void)hipFree(h->d_bext[a]); }
    if (h->h_ctx) pinned_give(reinterpret_cast<char*>(h->h_ctx));   // (h_one lives in the same slot)
    if (h->stream && h->stream_owned) (void)hipStreamDestroy(h->stream);
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

int mcf_abi_version(void) { return MCF_ABI_VERSION; }

int mcf_device_count(void) { return usable_devices(); }

void mcf_default_options(mcf_options* opt) {
    if (!opt) return;
    std::memset(opt, 0, sizeof *opt);
    opt->abi_version = MCF_ABI_VERSION;
    opt->device = -1;
    opt->rule = MCF_RULE_DANTZIG_FULL;
    opt->batch_pivots = 64;
    opt->use_graph = 1;
}

const char* mcf_last_error(mcf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mcf_create(int32_t n, int64_t m, const int32_t* tail, const int32_t* head, const int64_t* cost, const int64_t* cap,
               const int64_t* supply, const mcf_options* opt_in, mcf_handle** out) {
    if (!out) return MCF_E_BAD_ARG;
    *out = nullptr;
    mcf_options opt;
    if (opt_in) opt = *opt_in; else mcf_default_options(&opt);
    if (opt.abi_version != MCF_ABI_VERSION) { g_create_error = "mcf_options.abi_version mismatch"; return MCF_E_BAD_ARG; }
    if (opt.key_mode < 0 || opt.key_mode > MCF_KEY_CAPACITY || (opt.key_mode == MCF_KEY_PRIORITY && !opt.arc_priority) ||
        (opt.key_mode > 0 && opt.rule == MCF_RULE_DEVEX_BLOCK)) {
        g_create_error = "mcf_options.key_mode: 0..3, MCF_KEY_PRIORITY needs arc_priority, and the key variants belong to the Dantzig / candidate-list rules";
        return MCF_E_BAD_ARG;
    }
    if (opt.rule != MCF_RULE_DANTZIG_FULL && opt.rule != MCF_RULE_DEVEX_BLOCK && opt.rule != MCF_RULE_CANDIDATE_LIST) {
        g_create_error = "unknown pricing rule";
        return MCF_E_BAD_ARG;
    }
    const int ndev = usable_devices();
    if (ndev <= 0) { g_create_error = "no HIP device available (the engine has no CPU path)"; return MCF_E_NO_DEVICE; }
    mcf_handle* h = new (std::nothrow) mcf_handle();
    if (!h) return MCF_E_ALLOC;
    int err = 0;
    const std::string msg = mcf_build_image(n, m, tail, head, cost, cap, supply, h->im, &err);
    if (err) { g_create_error = msg; delete h; return err; }
    h->opt = opt;
    if (h->opt.batch_pivots <= 0) h->opt.batch_pivots = 64;
    if (opt.device >= 0) {
        if (opt.device >= ndev) { g_create_error = "device ordinal out of range"; delete h; return MCF_E_BAD_ARG; }
        if (hipSetDevice(opt.device) != hipSuccess) { g_create_error = "hipSetDevice failed"; delete h; return MCF_E_HIP; }
    }
    if (hipGetDevice(&h->device) != hipSuccess) { g_create_error = "hipGetDevice failed"; delete h; return MCF_E_HIP; }
    // shard
    h->shards = opt.shard_count > 0 ? opt.shard_count : 1;
    h->shard = opt.shard_rank;
    if (h->shard < 0 || h->shard >= h->shards) {
        g_create_error = "bad shard_rank / shard_count";
        delete h;
        return MCF_E_BAD_ARG;
    }
    const McfHostImage& im = h->im;
    {
        // Tree layout: the blocked preorder list (mcf_core.h) from kBplMinNodes nodes on -- below that the dense array's block
        // permutation moves few enough positions, and the persistent loops keep the dense array anyway.
        // tree_blocks: 0 = auto, -1 = dense array, k >= 2 = blocks of 2^k slots;  tree_pool: spare blocks (0 = auto, -1 = none).
        const bool forced = opt.tree_blocks > 0, never = opt.tree_blocks < 0;
        h->bpl = !never && opt.mid_loop <= 0 && opt.overlap_update <= 0 && (forced || im.n_nodes >= kBplMinNodes);
        if (h->bpl) {
            int32_t sh = 0, cap = 0, dense = 0;
            mcf_bpl_geometry(im.n_nodes, forced ? opt.tree_blocks : 0, opt.tree_pool, &sh, &cap, &dense);
            h->bpl_shift = sh; h->bpl_cap = cap; h->bpl_dense = dense;
            // (a large grid costs dispatch time on every pivot: 1 M / 16 M, 256 -> 64 workgroups: k_update_bpl 7.5 -> 6.0 us, +4 % pivots/s)
            int g = (cap + 7) / 8;
            if (g > 64) g = 64;
            if (const char* ge = std::getenv("MCF_BPL_GRID")) { const int vv = std::atoi(ge); if (vv >= 1 && vv <= 4096) g = vv; }   // A/B switch
            if (g < 1) g = 1;
            while ((cap + g - 1) / g > kBplTouchedCap) g *= 2;
            h->bpl_grid = g;
        }
    }
    {
        // 8 * k workgroups, one group of k per XCD head bucket (formula shared with the CPU emulation)
        h->price_blocks = mcf_price_blocks(m, h->shards, opt.price_blocks);
        const int64_t ab = ((int64_t)im.n_nodes + kApplyThreads - 1) / kApplyThreads;
        int max_ab = kMaxApplyBlocks;
        if (const char* ab_env = std::getenv("MCF_APPLY_BLOCKS")) { const int vv = std::atoi(ab_env); if (vv >= 64 && vv <= 4096) max_ab = vv; }   // A/B switch
        h->apply_blocks = (int)(ab < max_ab ? ab : max_ab);
    }

    auto fail = [&](const char* what, hipError_t e) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(e);
        free_all(h);
        delete h;
        return e == hipErrorOutOfMemory ? MCF_E_ALLOC : MCF_E_HIP;
    };
    hipError_t e;
    h->stream_owned = n > kPooledMaxNodes;
    if ((e = h->stream_owned ? hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) : pool_stream(h->device, &h->stream)) != hipSuccess) return fail("hipStreamCreate", e);
    const size_t N = im.n_nodes;
    if ((e = dalloc(&h->d_tail, im.m_pad)) != hipSuccess) return fail("hipMalloc tail", e);
    if ((e = dalloc(&h->d_head, im.m_pad)) != hipSuccess) return fail("hipMalloc head", e);
    if ((e = dalloc(&h->d_cost, im.m_pad)) != hipSuccess) return fail("hipMalloc cost", e);
    if ((e = dalloc(&h->d_orig, im.m_pad)) != hipSuccess) return fail("hipMalloc orig", e);
    if ((e = dalloc(&h->d_state, im.m_pad)) != hipSuccess) return fail("hipMalloc state", e);
    if ((e = dalloc(&h->d_weight, im.m_pad)) != hipSuccess) return fail("hipMalloc weight", e);
    if ((e = dalloc(&h->d_arcw, im.arcw.size())) != hipSuccess) return fail("hipMalloc arcw", e);
    if ((e = dalloc(&h->d_pi, N)) != hipSuccess) return fail("hipMalloc pi", e);
    if ((e = dalloc(&h->d_node, N)) != hipSuccess) return fail("hipMalloc node", e);
    const size_t bpl_slots = h->bpl ? ((size_t)h->bpl_cap << h->bpl_shift) + 4 : 0;   // slots per arena (+4: groups of four)
    // (+4: the cycle scan reads the nodes in aligned groups of four, like the sizes)
    if ((e = dalloc(&h->d_order0, h->bpl ? bpl_slots : N + 4)) != hipSuccess) return fail("hipMalloc order", e);
    if ((e = dalloc(&h->d_order1, h->bpl ? bpl_slots : N + 4)) != hipSuccess) return fail("hipMalloc order", e);
    if ((e = dalloc(&h->d_pos0, N)) != hipSuccess) return fail("hipMalloc pos", e);
    if ((e = dalloc(&h->d_pos1, h->bpl ? 1 : N)) != hipSuccess) return fail("hipMalloc pos", e);
    if ((e = dalloc(&h->d_path1, N)) != hipSuccess) return fail("hipMalloc path", e);
    if ((e = dalloc(&h->d_path2, N)) != hipSuccess) return fail("hipMalloc path", e);
    if ((e = dalloc(&h->d_ppos1, h->bpl ? 2 * N : N)) != hipSuccess) return fail("hipMalloc path", e);   // (blocked list: positions, then slots)
    if ((e = dalloc(&h->d_ppos2, h->bpl ? 2 * N : N)) != hipSuccess) return fail("hipMalloc path", e);
    if ((e = dalloc(&h->d_rec1, N)) != hipSuccess) return fail("hipMalloc rec", e);
    if ((e = dalloc(&h->d_rec2, N)) != hipSuccess) return fail("hipMalloc rec", e);
    if ((e = dalloc(&h->d_seg, 2 * N + 2)) != hipSuccess) return fail("hipMalloc seg", e);
    if ((e = dalloc(&h->d_ctx, 1)) != hipSuccess) return fail("hipMalloc ctx", e);
    if ((e = dalloc(&h->d_cand, kMaxPriceBlocks)) != hipSuccess) return fail("hipMalloc cand", e);
    if ((e = dalloc(&h->d_cand_aux, kMaxPriceBlocks)) != hipSuccess) return fail("hipMalloc cand", e);
    if ((e = hipMemset(h->d_cand, 0xff, kMaxPriceBlocks * sizeof(McfCand))) != hipSuccess) return fail("hipMemset cand", e);
    if ((e = dalloc(&h->d_one, 1)) != hipSuccess) return fail("hipMalloc one", e);
    if ((e = dalloc(&h->d_swept, kMaxPriceBlocks)) != hipSuccess) return fail("hipMalloc swept", e);
    {   // pinned: the control block's host copy and the one-candidate read-back buffer (one pooled slot)
        char* slot = nullptr;
        if ((e = pinned_take(&slot)) != hipSuccess) return fail("hipHostMalloc", e);
        h->h_ctx = reinterpret_cast<McfCtx*>(slot);
        h->h_one = reinterpret_cast<McfCand*>(slot + ((sizeof(McfCtx) + 63) & ~(size_t)63));
    }

    McfView& v = h->view;
    v.n_nodes = im.n_nodes;
    v.m = im.m;
    v.tail = h->d_tail; v.head = h->d_head; v.cost = h->d_cost; v.orig = h->d_orig; v.state = h->d_state;
    for (int x = 0; x <= MCF_NUM_BUCKETS; ++x) v.bucket_off[x] = im.bucket_off[x];
    v.weight = opt.rule == MCF_RULE_DEVEX_BLOCK ? h->d_weight : nullptr;
    v.dx = nullptr;  // (allocated and filled by upload_image for the Devex rule)
    // key variant of the Dantzig / candidate-list sweep (the reference's specialised entering rules, mcf_core.h: mcf_dantzig_key)
    v.key_mode = opt.key_mode > 0 ? opt.key_mode : (opt.forward_first ? MCF_KEY_FORWARD_FIRST : MCF_KEY_PLAIN);
    v.prio = nullptr;
    if (v.key_mode == MCF_KEY_PRIORITY) {
        std::vector<int8_t> pr((size_t)im.m_pad, 0);   // the caller's order -> engine order
        for (int64_t i = 0; i < im.m; ++i) pr[(size_t)i] = (int8_t)(opt.arc_priority[im.orig[(size_t)i]] & 3);
        if ((e = dalloc(&h->d_prio, im.m_pad)) != hipSuccess) return fail("hipMalloc priority", e);
        if ((e = hipMemcpy(h->d_prio, pr.data(), pr.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy priority", e);
        v.prio = h->d_prio;
    }
    v.arcw = h->d_arcw; v.pi = h->d_pi; v.node = h->d_node;
    v.order[0] = h->d_order0; v.order[1] = h->d_order1;
    v.posbuf[0] = h->d_pos0; v.posbuf[1] = h->d_pos1;
    v.psz[0] = nullptr; v.psz[1] = nullptr;
    v.reach = nullptr; v.chg = nullptr;
    v.path1 = h->d_path1; v.path2 = h->d_path2; v.ppos1 = h->d_ppos1; v.ppos2 = h->d_ppos2; v.rec1 = h->d_rec1; v.rec2 = h->d_rec2; v.seg = h->d_seg; v.ctx = h->d_ctx;

    {
        // LDS plan of the fused small-instance path (every offset a multiple of 16)
        SmallLayout& L = h->small_layout;
        uint32_t off = 0;
        auto take = [&](uint64_t bytes) { const uint32_t o = off; off += (uint32_t)((bytes + 15) / 16 * 16); return o; };
        const uint64_t mp = (uint64_t)im.m_pad, Nn = (uint64_t)im.n_nodes;
        L.tail = take(mp * 4); L.head = take(mp * 4); L.cost = take(mp * 4); L.orig = take(mp * 4);
        L.state = take(mp); L.weight = take(opt.rule == MCF_RULE_DEVEX_BLOCK ? mp * 4 : 0);
        L.arcw = take((uint64_t)im.arcw.size() * 16); L.pi = take(Nn * 8); L.node = take(Nn * 16);
        L.order0 = take(Nn * 4); L.order1 = take(Nn * 4); L.pos0 = take(Nn * 4); L.pos1 = take(Nn * 4);
        L.path1 = take(Nn * 4); L.path2 = take(Nn * 4); L.ppos1 = take(Nn * 4); L.ppos2 = take(Nn * 4);
        L.rec1 = take(Nn * 16); L.rec2 = take(Nn * 16);
        L.seg = take((2 * Nn + 2) * sizeof(McfSeg)); L.ctx = take(sizeof(McfCtx));
        L.total = off;
        const uint64_t need = mp * 21 + (uint64_t)im.arcw.size() * 16 + Nn * 112 + 4096;
        h->small = !h->bpl && !opt.no_fused && !opt.profile && h->shards == 1 && need < 150 * 1024 && L.total <= kSmallMaxLds;
        if (h->small) {
            // (the limit is a property of the kernel, not of the handle: it only ever grows, so that handles of different
            //  sizes can be alive together -- and share one batched launch)
            static std::mutex lds_mu;
            static int lds_limits[64] = {0};   // per device: the attribute belongs to the device's copy of the function
            int& lds_limit = lds_limits[h->device & 63];
            std::lock_guard<std::mutex> lock(lds_mu);
            hipError_t fe = hipSuccess;
            if ((int)L.total > lds_limit) {
                fe = hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve_small), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
                if (fe == hipSuccess) fe = hipFuncSetAttribute(reinterpret_cast<const void*>(k_solve_small_batch),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total);
                if (fe == hipSuccess) lds_limit = (int)L.total;
                else (void)hipGetLastError();   // (the refusal must not surface later as some launch's error)
            }
            if (fe != hipSuccess) h->small = false;  // fall back to the three-kernel GPU path
        }
    }
    // position-space subtree sizes for the cycle scan: every handle but the LDS-resident ones
    const bool scan_ok = opt.cycle_scan >= 0 && im.n_nodes <= kScanMaxNodes && !h->small;  // -1: never scan
    v.bmeta[0] = v.bmeta[1] = nullptr; v.bext[0] = v.bext[1] = nullptr; v.blk_shift = 0; v.blk_cap = 0; v.ncandx = 0; v.candx = nullptr;
    if (h->bpl) {
        // the two slot arenas (sizes; node ids are d_order0/1), the block records (two copies) and extents (one per arena),
        // and the scratch the scan spills flagged blocks to
        if ((e = dalloc(&h->d_psz0, bpl_slots)) != hipSuccess) return fail("hipMalloc psz", e);
        if ((e = dalloc(&h->d_psz1, bpl_slots)) != hipSuccess) return fail("hipMalloc psz", e);
        if ((e = hipMemset(h->d_psz0, 0, bpl_slots * 4)) != hipSuccess || (e = hipMemset(h->d_psz1, 0, bpl_slots * 4)) != hipSuccess) return fail("hipMemset psz", e);
        for (int a = 0; a < 2; ++a) {
            if ((e = dalloc(&h->d_bmeta[a], (size_t)h->bpl_cap + 2)) != hipSuccess) return fail("hipMalloc block records", e);
            if ((e = dalloc(&h->d_bext[a], (size_t)h->bpl_cap)) != hipSuccess) return fail("hipMalloc block extents", e);
            v.bmeta[a] = h->d_bmeta[a]; v.bext[a] = h->d_bext[a];
        }
        const size_t nchg = std::max(N, 2 * (size_t)h->bpl_cap + 2);
        if ((e = dalloc(&h->d_chg, nchg)) != hipSuccess) return fail("hipMalloc chg", e);
        v.psz[0] = h->d_psz0; v.psz[1] = h->d_psz1;
        v.chg = h->d_chg;
        v.blk_shift = h->bpl_shift; v.blk_cap = h->bpl_cap;
    } else if (scan_ok) {
        const size_t NP = (N + MCF_REACH_BLOCK - 1) / MCF_REACH_BLOCK * MCF_REACH_BLOCK + 4;  // whole coarse blocks (+4: groups of four)
        if ((e = dalloc(&h->d_psz0, NP)) != hipSuccess) return fail("hipMalloc psz", e);
        if ((e = dalloc(&h->d_psz1, NP)) != hipSuccess) return fail("hipMalloc psz", e);
        if ((e = hipMemset(h->d_psz0, 0, NP * 4)) != hipSuccess || (e = hipMemset(h->d_psz1, 0, NP * 4)) != hipSuccess) return fail("hipMemset psz", e);
        v.psz[0] = h->d_psz0; v.psz[1] = h->d_psz1;
        if ((e = dalloc(&h->d_reach, NP / MCF_REACH_BLOCK + 8)) != hipSuccess) return fail("hipMalloc reach", e);   // (+8: read four at a time)
        if ((e = hipMemset(h->d_reach, 0, (NP / MCF_REACH_BLOCK + 8) * 4)) != hipSuccess) return fail("hipMemset reach", e);
        if ((e = dalloc(&h->d_chg, N)) != hipSuccess) return fail("hipMalloc chg", e);
        v.reach = h->d_reach; v.chg = h->d_chg;
    }
    // cycle search: how many round trips the one-lane climb takes before the workgroup-wide scan over
    // preorder positions finishes the cycle.  Auto = none: measured on MI355X (profiles/r01_f_*), scanning
    // at once beats every hybrid from 256 to 65 536 nodes -- a dependent global round trip per tree level
    // costs more than the whole scan's ~4 -- by 1.1x (netgen_8_08a) to 4.5x (goto_8_16a, cycles of ~450 arcs).
    if (!scan_ok && !(h->bpl && opt.cycle_scan >= 0)) h->climb_budget = INT32_MAX;
    else if (opt.cycle_scan > 0) h->climb_budget = opt.cycle_scan - 1;
    else h->climb_budget = 0;
    // resident reduced costs for everything that does not take the fused LDS path
    h->rcached = !h->small && !opt.no_rcache && im.m > 0;
    if (h->rcached) {
        // a rank patches only the reduced costs it sweeps -- for the rules whose sweeps cover a FIXED share of every bucket.
        // Devex cuts the block first and the shard second (so that the union over ranks does not depend on the rank
        // count), and blocks move and resize: there every rank keeps all reduced costs exact, as a single GPU does.
        const bool partial = h->shards > 1 && opt.rule != MCF_RULE_DEVEX_BLOCK;
        mcf_build_rcache(h->im, partial ? h->shard : 0, partial ? h->shards : 1);
        v.rc_partial = partial ? 1 : 0;
        if ((e = dalloc(&h->d_rcache, im.m_pad)) != hipSuccess) return fail("hipMalloc rcache", e);
        if ((e = dalloc(&h->d_adj_off, im.adj_off.size())) != hipSuccess) return fail("hipMalloc adj_off", e);
        if ((e = dalloc(&h->d_adj, im.adj.size())) != hipSuccess) return fail("hipMalloc adj", e);
        if ((e = hipMemcpy(h->d_adj_off, im.adj_off.data(), im.adj_off.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail("copy adj_off", e);
        if ((e = hipMemcpy(h->d_adj, im.adj.data(), im.adj.size() * 8, hipMemcpyHostToDevice)) != hipSuccess) return fail("copy adj", e);
        h->im.adj.clear(); h->im.adj.shrink_to_fit();  // the device copy is the only one needed from here on
        v.rcache = h->d_rcache; v.adj_off = h->d_adj_off; v.adj = h->d_adj;
        const int64_t rb = ((int64_t)im.n_nodes + 15) / 16;  // one 16-lane group per node of the largest possible T2
        int max_rb = kMaxRcupdBlocks;
        if (const char* re = std::getenv("MCF_RCUPD_BLOCKS")) { const int vv = std::atoi(re); if (vv >= 1 && vv <= 4096) max_rb = vv; }   // A/B switch
        h->rcupd_blocks = (int)(rb < max_rb ? (rb > 0 ? rb : 1) : max_rb);
    } else {
        v.rcache = nullptr; v.adj_off = nullptr; v.adj = nullptr; v.rc_partial = 0;
    }
    // persistent single-workgroup loop: the tree work of one pivot must be small enough for one CU, and so must
    // the arcs it prices per pivot (a Devex block / the whole arc list for Dantzig; the candidate list's full
    // sweeps stay on the grid)
    {
        int64_t per_pivot_arcs = 0;
        if (opt.rule == MCF_RULE_DEVEX_BLOCK) {
            McfCtx probe;  // the largest block the tuner may grow to (mcf_core.h: never beyond MCF_TUNER_MAX_ARCS or the initial size)
            std::memset(&probe, 0, sizeof probe);
            mcf_init_block_state(&probe, opt.rule, im.m, opt.block_size);
            per_pivot_arcs = im.m * probe.max_granules / MCF_GRANULES;
        } else if (opt.rule == MCF_RULE_DANTZIG_FULL) {
            per_pivot_arcs = im.m;
        }
        // (the candidate list never gains from the loop -- its full sweeps run on the grid either way -- and stays on the graph)
        const bool fits = im.n_nodes <= kMidMaxNodes && per_pivot_arcs <= kMidMaxArcsPerPivot && opt.rule != MCF_RULE_CANDIDATE_LIST;
        h->mid = !h->bpl && h->rcached && scan_ok && h->shards == 1 && !opt.profile && opt.mid_loop >= 0 && (fits || opt.mid_loop > 0);
    }
    // overlapped graphs (build_graph), on request only: the idea -- on large trees the permutation outlasts the reduced-cost
    // patch, so the next pricing could hide beside it -- loses to the cost of the two cross-queue edges per pivot
    // (netgen_8_14a 55 K -> 33 K pivots/s, 1 M / 16 M 26 K -> 22 K; profiles/r02_ab_overlapped_graph.txt)
    {
        const bool able = !h->bpl && h->rcached && !h->small && !h->mid && h->shards == 1 && !opt.profile && opt.rule != MCF_RULE_CANDIDATE_LIST;
        h->overlap = able && opt.overlap_update > 0;
        h->overlap_cfg = h->overlap;
        if (h->overlap && (e = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
    }
    // compressed Dantzig keys for the grid sweeps of the Dantzig / candidate-list rules (4 B per arc instead of 9)
    v.vkey = nullptr;
    v.vk_bigm = im.big_m;
    v.vk_half = 1 << (opt.vkey_half_log2 > 0 && opt.vkey_half_log2 <= 28 ? opt.vkey_half_log2 : 28);
    // They pay where a sweep reads the whole shard and is bandwidth-bound: full (non-incremental) sweeps of the Dantzig rule
    // from kIncrementalMinArcs arcs on (1 M / 16 M: sweep 24 -> 14 us).  Keeping the codes exact costs the reduced-cost patch a state
    // byte and a 4-byte store per patched arc (k_update +1..3 us per pivot), which a candidate-list handle (one sweep per ~33
    // pivots) or an incremental sweep (1 % of the arcs re-read per pivot) never earns back.  compressed_keys: 1 = on, -1 = off.
    const bool keys_auto = opt.rule == MCF_RULE_DANTZIG_FULL && opt.full_sweeps > 0 && im.m >= kIncrementalMinArcs;
    if (h->rcached && !h->mid && opt.rule != MCF_RULE_DEVEX_BLOCK && v.key_mode == MCF_KEY_PLAIN &&
        (opt.compressed_keys > 0 || (opt.compressed_keys == 0 && keys_auto))) {
        if ((e = dalloc(&h->d_vkey, im.m_pad)) != hipSuccess) return fail("hipMalloc vkey", e);
        v.vkey = h->d_vkey;
        const char* nt_env = std::getenv("MCF_NT_SWEEP");   // A/B switch: 0 / 1 force, unset = by size
        h->nt_sweep = nt_env ? nt_env[0] == '1' : (im.m_pad / (h->shards > 0 ? h->shards : 1)) * 4 > (int64_t)200 * 1024 * 1024;
    }
    // incremental pricing for the rules whose sweeps cover the whole shard (Dantzig, candidate list); not for the
    // persistent loop, whose instances are far too small for it (its kernel compiles the marking away)
    v.dirty = nullptr;
    // (auto: only where a sweep is a large part of a pivot -- below ~4 M arcs the flag look-up in front of every
    // sweep and the marking in k_update cost more than the skipped blocks save: netgen_8_14a 52.7 K -> 46.7 K pivots/s;
    // at 16 M arcs: 23.9 K -> 37 K)
    if (h->rcached && !h->mid && opt.rule != MCF_RULE_DEVEX_BLOCK && (opt.full_sweeps < 0 || (opt.full_sweeps == 0 && im.m >= kIncrementalMinArcs))) {
        static_assert(MCF_MAX_PRICE_BLOCKS == kMaxPriceBlocks, "flag array size");
        if ((e = dalloc(&h->d_dirty, 1)) != hipSuccess) return fail("hipMalloc dirty", e);
        McfDirty head;  // (the flags are raised by upload_image)
        head.nlb = h->price_blocks / MCF_NUM_BUCKETS;
        for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
            int64_t lo, hi;
            mcf_bucket_slice(im.bucket_off, x, h->shard, h->shards, 0, 1, &lo, &hi);
            head.lo[x] = (int32_t)lo; head.hi[x] = (int32_t)hi;
        }
        if ((e = hipMemcpy(h->d_dirty, &head, offsetof(McfDirty, flag), hipMemcpyHostToDevice)) != hipSuccess) return fail("copy dirty", e);
        v.dirty = h->d_dirty;
    }
    // candidate cache: the grid path of the candidate-list rule on one GPU, keys that need nothing but the reduced cost
    v.candx = nullptr; v.ncandx = 0;
    if (opt.rule == MCF_RULE_CANDIDATE_LIST && h->shards == 1 && !h->small && !h->mid && !opt.profile && v.key_mode <= MCF_KEY_FORWARD_FIRST &&
        h->price_blocks <= 2 * kPivotThreads && !std::getenv("MCF_NO_CANDX")) {
        if ((e = dalloc(&h->d_candx, kMaxPriceBlocks)) != hipSuccess) return fail("hipMalloc candidate records", e);
        if ((e = hipMemset(h->d_candx, 0xff, kMaxPriceBlocks * sizeof(McfCandX))) != hipSuccess) return fail("hipMemset candidate records", e);
        v.candx = h->d_candx; v.ncandx = h->price_blocks;
    }
    // pivots back to back (k_pivot_run): candidate-list handles on the blocked list that keep the candidate cache
    h->run_pairs_cfg = 0;
    // Opt-in: measured at 1M/16M (first 300 K pivots, same box) 30.9-35.7 K pivots/s in the run shape against 41.1 K in the
    // pivot+update pair shape -- one workgroup's phase A over 8-20 K blocks costs more than the launches it saves.
    const char* run_env = std::getenv("MCF_PIVOT_RUN");
    const int want_run = opt.pivot_run > 0 ? opt.pivot_run : (run_env ? std::atoi(run_env) : 0);
    if (h->bpl && v.candx && want_run > 0 && opt.use_graph && !std::getenv("MCF_NO_RUN")) h->run_pairs_cfg = want_run;
    h->run_pairs = h->run_pairs_cfg;
    h->rc_able = h->rcached;
    {
        // auto: candidate lists (one sweep per ~33 pivots) on large instances (from 100 000 nodes: measured at 1 M / 16 M) from an
        // average |T2| of 384 nodes, Devex there from 1 500; never for the Dantzig rule, whose every pivot sweeps all arcs, for
        // shards (their driver owns the launches) or the persistent loops.  rc_drop: -1 = never, k > 0 = that threshold.
        int64_t thr = 0;
        if (h->rcached && !h->mid && !h->small && h->shards == 1 && opt.key_mode == 0 && !opt.forward_first) {
            if (opt.rc_drop > 0) thr = opt.rc_drop;
            else if (opt.rc_drop == 0 && im.n_nodes >= 100000)
                // (Devex block search sweeps one block per pivot: the dearer sweep is paid every time, so it waits for larger
                //  subtrees -- 1 M / 16 M: 139.8 s without, 134.0 / 134.8 s with a threshold of 1 500 / 384)
                thr = opt.rule == MCF_RULE_CANDIDATE_LIST ? 384 : (opt.rule == MCF_RULE_DEVEX_BLOCK ? 1500 : 0);
            if (opt.rule == MCF_RULE_DANTZIG_FULL) thr = 0;
        }
        if (const char* env = std::getenv("MCF_RC_DROP")) { const long vv = std::atol(env); thr = vv > 0 ? vv : 0; }   // A/B switch
        h->rc_drop_subtree = h->rcached && h->shards == 1 ? thr : 0;
    }
    const int rc = upload_image(h);   // (ends with a synchronisation of h->stream)
    if (rc != MCF_OK) { g_create_error = h->err; free_all(h); delete h; return rc; }
    // A handle that captures graphs needs a stream nobody else captures on: only the persistent-loop paths that launch
    // plain kernels keep the borrowed stream.
    if (!h->stream_owned && !(h->small || (h->mid && opt.rule != MCF_RULE_CANDIDATE_LIST))) {
        hipStream_t own = nullptr;
        if ((e = hipStreamCreateWithFlags(&own, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
        h->stream = own;
        h->stream_owned = true;
    }
    *out = h;
    return MCF_OK;
}

int mcf_reset(mcf_handle* h) {
    if (!h) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    mcf_init_cold_basis(h->im);  // a warm start may have replaced the start basis in the host image
    mcf_refresh_rcache(h->im);
    return upload_image(h);
}

int mcf_set_basis(mcf_handle* h, const int8_t* in_tree, const int8_t* at_upper) {
    if (!h || !in_tree) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    const std::string msg = mcf_apply_basis(h->im, in_tree, at_upper);
    if (!msg.empty()) mcf_init_cold_basis(h->im);  // rejected: the image may be half written
    mcf_refresh_rcache(h->im);
    const int rc = upload_image(h);
    if (rc) return rc;
    if (!msg.empty()) { h->err = "warm-start basis rejected: " + msg; return MCF_E_STATE; }
    return MCF_OK;
}

int mcf_set_max_pivots(mcf_handle* h, int64_t max_total_pivots) {
    if (!h) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_ctl, dim3(1), dim3(1), 0, h->stream, h->d_ctx, max_total_pivots, 1);
    HIP_TRY(h, hipGetLastError());
    // the caller enqueues the pivots on a stream of its own: make the new cap visible first
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->total_cap = max_total_pivots;
    h->ctx_current = false;
    h->external_driver = true;
    return MCF_OK;
}

int mcf_solve(mcf_handle* h, int64_t max_pivots, mcf_progress_cb cb, void* user, int64_t cb_interval) {
    if (!h) return MCF_E_BAD_ARG;
    if (h->shards != 1) {
        // a sharded handle prices 1/shard_count of the arcs: on its own it would declare "optimal" as soon as ITS share
        // holds no eligible arc.  It is only valid inside the price -> all-gather -> pivot protocol below.
        h->err = "mcf_solve: handle was created with shard_count > 1; drive it with mcf_enqueue_price / mcf_enqueue_pivot";
        return MCF_E_STATE;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t m = h->im.m, n = h->im.n;
    if (max_pivots < 0) max_pivots = 20 * (m + n) > 100 ? 20 * (m + n) : 100;  // simplex.py:1470
    int rc = sync_ctx(h, h->stream);
    if (rc) return rc;
    h->external_driver = false;   // from here on the library enqueues the pivots itself
    const int64_t start = h->h_ctx->pivots;
    const int64_t final_cap = start + max_pivots;
    if (cb_interval <= 0) cb_interval = 100;
    int batch = h->opt.batch_pivots;
    if (h->mid) {  // a slot is a whole run of pivots there
        const int per = h->opt.rule == MCF_RULE_CANDIDATE_LIST ? mcf_minor_cap(h->price_blocks) + 1 : batch;
        batch = batch / per > 2 ? batch / per : 2;
        if (h->opt.rule != MCF_RULE_CANDIDATE_LIST) batch = 1;
    }
    if (!h->mid && !h->small && h->opt.rule == MCF_RULE_CANDIDATE_LIST) {
        // The sweep cadence inside a batch is fixed (one sweep per minor_cap + 1 slots, launch_pivot_triplet): a batch that is
        // not a whole number of such periods leaves the next batch's first sweep slot in front of a list that is still
        // live, the sweep is skipped and the slots up to the following sweep slot idle (1 M / 16 M: 33 pivots per 64-slot
        // graph).  Whole periods only: 64 -> 66 slots at period 33, 63 at period 9.
        const int period = mcf_minor_cap(h->price_blocks) + 1;
        const int periods = (batch + period / 2) / period;
        batch = (periods > 0 ? periods : 1) * period;
    }
    const bool graph = h->opt.use_graph && !h->opt.profile && !h->small && !(h->mid && batch == 1);
    if (graph) { rc = build_graph(h, batch); if (rc) return rc; }
    bool stop = false;
    while (!stop) {
        // inner cap: stop at the next progress point so the callback cadence is exact
        int64_t cap = final_cap;
        if (cb) {
            const int64_t done = h->h_ctx->pivots - start;
            const int64_t next_cb = start + (done / cb_interval + 1) * cb_interval;
            if (next_cb < cap) cap = next_cb;
        }
        // the persistent loops outside a graph take the cap as a kernel argument; the other paths re-arm with k_ctl
        const bool arm_in_kernel = h->small || (h->mid && !graph && !h->opt.profile);
        if (!arm_in_kernel) hipLaunchKernelGGL(k_ctl, dim3(1), dim3(1), 0, h->stream, h->d_ctx, cap, 1);
        h->ctx_current = false;
        // pivot until the device reports something other than "still running"
        for (;;) {
            const bool timed_loop = arm_in_kernel;  // one kernel = many pivots: its duration is the per-launch figure bench.py reports
            if (timed_loop) {
                if (!h->loop_ev[0]) { HIP_TRY(h, hipEventCreate(&h->loop_ev[0])); HIP_TRY(h, hipEventCreate(&h->loop_ev[1])); }
                HIP_TRY(h, hipEventRecord(h->loop_ev[0], h->stream));
            }
            if (h->small) {
                hipLaunchKernelGGL(k_solve_small, dim3(1), dim3(kSmallThreads), h->small_layout.total, h->stream, h->view,
                                   h->small_layout, h->opt.rule, h->d_cand, cap, h->h_ctx);
            }
            else {
                // eager launches need not run past the cap (a replayed graph has a fixed length: its surplus slots early-exit)
                const int64_t left = cap - h->h_ctx->pivots;
                const int slots = h->mid ? batch : (int)(left < 1 ? 1 : (left < batch ? left : batch));
                // a replayed graph has a fixed length: when fewer pivots than that are left (a small budget, the tail of a
                // progress interval) its surplus slots would idle through three early-exit kernels each -- launch just the slots
                // that are needed instead (bench.py --steps 20 on a 64-slot graph: 73 us per pivot, 41 us of them real)
                if (h->opt.profile) { rc = run_batch_profiled(h, slots); if (rc) return rc; }
                else if (graph && (h->mid || left >= batch)) HIP_TRY(h, hipGraphLaunch(h->graph_exec, h->stream));
                else for (int i = 0; i < slots; ++i) launch_pivot_triplet(h, h->stream, i, arm_in_kernel ? cap : (int64_t)-1);
            }
            if (timed_loop) HIP_TRY(h, hipEventRecord(h->loop_ev[1], h->stream));
            HIP_TRY(h, hipGetLastError());
            if (h->small) {   // the kernel wrote the host's pinned copy itself
                // (polling a pinned done word instead was tried: the wake-up saved here comes back in the caller's own
                //  device synchronisation, and a system-scope fence in front of the word cost 15 us of kernel time)
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                h->ctx_current = true;
            } else {
                rc = read_ctx(h, h->stream);
                if (rc) return rc;
            }
            if (timed_loop) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, h->loop_ev[0], h->loop_ev[1]) == hipSuccess) { h->stats.loop_ms += ms; h->stats.loop_launches += 1; }
            }
            h->stats.batches += 1;
            if (h->h_ctx->status != MCF_RUNNING) break;
            // Run shape: still worth it?  A run launch that ends after a pivot or two (its update was too large for one
            // workgroup) is a k_pivot with extra baggage, and a period then needs more launches than the shape holds.  Three
            // batches in a row with fewer than three pivots per run launch: back to one k_pivot per slot, for good.
            if (h->run_pairs > 0 && graph) {
                const int period = mcf_minor_cap(h->price_blocks) + 1;
                const int64_t made = h->h_ctx->pivots - h->run_seen;
                h->run_seen = h->h_ctx->pivots;
                const int64_t launches = (int64_t)(batch / period) * h->run_pairs;
                h->run_low = made < 3 * launches ? h->run_low + 1 : 0;
                if (h->run_low >= 3) {
                    h->run_pairs = 0;
                    h->stats.run_left_at = h->h_ctx->pivots;
                    (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr;
                    (void)hipGraphDestroy(h->graph); h->graph = nullptr;
                    h->graph_batch = 0;
                    rc = build_graph(h, batch);
                    if (rc) return rc;
                }
            }
            // Drop the resident reduced costs?  Their patch walks the adjacency of every node of the re-hung subtree: a few
            // arcs early in a solve, hundreds of thousands of random read-modify-writes per pivot once subtrees of thousands
            // of nodes move (1 M / 16 M, last third of the solve: 20+ us of a pivot).  Pricing from the potentials costs a
            // dearer sweep instead (gathers), which a candidate-list handle pays once per minor_cap + 1 pivots and a Devex
            // handle on one block.  Same keys, same candidates, same pivots; one way (until the next reset).
            if (h->rcached && h->rc_drop_subtree > 0 && !h->opt.profile) {
                const int64_t dp = h->h_ctx->pivots - h->sw_pivots, ds = h->h_ctx->subtree_nodes - h->sw_subtree;
                if (dp >= 4096) {
                    h->sw_pivots = h->h_ctx->pivots; h->sw_subtree = h->h_ctx->subtree_nodes;
                    if (ds >= dp * h->rc_drop_subtree) {
                        h->rc_dropped = true;
                        h->rcached = false;
                        h->overlap = false;   // (two-stream graphs price beside the permutation: only valid from resident values)
                        h->view.rcache = nullptr; h->view.vkey = nullptr; h->view.dirty = nullptr;
                        h->stats.rc_dropped_at = h->h_ctx->pivots;
                        if (graph) {
                            (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr;
                            (void)hipGraphDestroy(h->graph); h->graph = nullptr;
                            h->graph_batch = 0;
                            rc = build_graph(h, batch);
                            if (rc) return rc;
                        }
                    }
                }
            }
        }
        const int32_t st = h->h_ctx->status;
        if (st == MCF_PIVOT_LIMIT && h->h_ctx->pivots < final_cap) {
            if (cb) {
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (cb(user, h->h_ctx->pivots, final_cap, el) != 0) stop = true;
            }
            continue;  // re-arm with the next cap
        }
        stop = true;
    }
    if (h->h_ctx->status == MCF_PIVOT_LIMIT && !h->h_ctx->limit_checked) {
        // simplex.py:1678-1699: at the budget, price once more to tell optimal from iteration_limit
        // (the persistent loops have done that on the device: limit_checked)
        int64_t arc = -1, key = 0; int32_t dir = 0;
        rc = mcf_price_once(h, h->opt.rule == MCF_RULE_DEVEX_BLOCK ? MCF_RULE_DANTZIG_FULL : h->opt.rule, 0, m, &arc, &dir, &key);
        if (rc) return rc;
        if (arc < 0) {
            h->h_ctx->status = MCF_OPTIMAL;
            HIP_TRY(h, hipMemcpyAsync(&h->d_ctx->status, &h->h_ctx->status, sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
    }
    if (h->h_ctx->status == MCF_INTERNAL_ERROR) { h->err = "internal error: preorder permutation did not close"; return MCF_E_INTERNAL; }
    h->stats.solve_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    h->solved_once = true;
    return MCF_OK;
}

// Independent instances side by side: one launch per engine path, one persistent workgroup per handle
// (k_solve_small_batch: whole instance in LDS; k_solve_mid_batch: state in global memory, Dantzig / Devex).
int mcf_solve_batch(mcf_handle* const* handles, int32_t count, const int64_t* max_pivots, double* kernel_ms) {
    if (!handles || count <= 0) return MCF_E_BAD_ARG;
    mcf_handle* h0 = handles[0];
    if (!h0) return MCF_E_BAD_ARG;
    for (int32_t i = 0; i < count; ++i) {
        mcf_handle* h = handles[i];
        if (!h) return MCF_E_BAD_ARG;
        const bool loop = h->small || h->mid;
        if (!loop || h->shards != 1 || h->device != h0->device) {
            h0->err = "mcf_solve_batch: every handle must run as ONE persistent workgroup -- mcf_stats.pricing_mode 2 (LDS loop) or 3 "
                      "(persistent loop; mcf_options.mid_loop = 1 asks for it at any size) -- on one device";
            return MCF_E_STATE;
        }
        for (int32_t j = 0; j < i; ++j)
            if (handles[j] == h) { h0->err = "mcf_solve_batch: a handle appears twice"; return MCF_E_BAD_ARG; }
    }
    HIP_TRY(h0, hipSetDevice(h0->device));
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<SmallJob> small_jobs;
    std::vector<MidJob> mid_jobs;
    uint32_t lds = 0;
    for (int32_t i = 0; i < count; ++i) {
        mcf_handle* h = handles[i];
        int rc = sync_ctx(h, h->stream);   // (also drains whatever the handle's own stream still holds)
        if (rc) return rc;
        h->external_driver = false;
        const int64_t m = h->im.m, n = h->im.n;
        int64_t mp = max_pivots ? max_pivots[i] : -1;
        if (mp < 0) mp = 20 * (m + n) > 100 ? 20 * (m + n) : 100;  // simplex.py:1470
        const int64_t cap = h->h_ctx->pivots + mp;
        if (h->small) {
            SmallJob J;
            J.g = h->view; J.L = h->small_layout; J.rule = h->opt.rule; J.pad = 0; J.list = h->d_cand; J.cap = cap; J.host_ctx = h->h_ctx;
            small_jobs.push_back(J);
            if (h->small_layout.total > lds) lds = h->small_layout.total;
        } else {
            MidJob J;
            J.g = h->view; J.rule = h->opt.rule; J.nlist = h->price_blocks; J.list = h->d_cand; J.cap = cap; J.host_ctx = h->h_ctx;
            mid_jobs.push_back(J);
        }
    }
    // candidate-list jobs last (their own launch)
    std::stable_partition(mid_jobs.begin(), mid_jobs.end(), [](const MidJob& J) { return J.rule != MCF_RULE_CANDIDATE_LIST; });
    size_t n_listing = 0;
    for (const MidJob& J : mid_jobs) n_listing += J.rule == MCF_RULE_CANDIDATE_LIST ? 1 : 0;
    // job arrays and the two timing events come from a per-device pool that only ever grows (a call used to pay two
    // hipMalloc / hipFree pairs and two event creations); the pool's lock also serialises concurrent batches on a device
    struct BatchPool { void* small = nullptr; size_t small_bytes = 0; void* mid = nullptr; size_t mid_bytes = 0; hipEvent_t ev[2] = {nullptr, nullptr}; };
    static std::mutex pool_mu;
    static BatchPool pools[64];
    std::lock_guard<std::mutex> pool_lock(pool_mu);
    BatchPool& pool = pools[h0->device & 63];
    SmallJob* d_small = nullptr;
    MidJob* d_mid = nullptr;
    auto cleanup = [&]() {};   // (nothing is owned by the call any more)
    auto bail = [&](const char* what, hipError_t e) {
        h0->err = std::string(what) + ": " + hipGetErrorString(e);
        cleanup();
        return MCF_E_HIP;
    };
    hipError_t e;
    hipStream_t s = h0->stream;
    auto grow = [&](void** buf, size_t* have, size_t need) -> hipError_t {
        if (need <= *have) return hipSuccess;
        if (*buf) (void)hipFree(*buf);
        *buf = nullptr; *have = 0;
        const hipError_t ge = hipMalloc(buf, need * 2);
        if (ge == hipSuccess) *have = need * 2;
        return ge;
    };
    if (!small_jobs.empty()) {
        if ((e = grow(&pool.small, &pool.small_bytes, small_jobs.size() * sizeof(SmallJob))) != hipSuccess) return bail("hipMalloc jobs", e);
        d_small = static_cast<SmallJob*>(pool.small);
        if ((e = hipMemcpyAsync(d_small, small_jobs.data(), small_jobs.size() * sizeof(SmallJob), hipMemcpyHostToDevice, s)) != hipSuccess) return bail("hipMemcpy jobs", e);
    }
    if (!mid_jobs.empty()) {
        if ((e = grow(&pool.mid, &pool.mid_bytes, mid_jobs.size() * sizeof(MidJob))) != hipSuccess) return bail("hipMalloc jobs", e);
        d_mid = static_cast<MidJob*>(pool.mid);
        if ((e = hipMemcpyAsync(d_mid, mid_jobs.data(), mid_jobs.size() * sizeof(MidJob), hipMemcpyHostToDevice, s)) != hipSuccess) return bail("hipMemcpy jobs", e);
    }
    if (!pool.ev[0] && ((e = hipEventCreate(&pool.ev[0])) != hipSuccess || (e = hipEventCreate(&pool.ev[1])) != hipSuccess)) return bail("hipEventCreate", e);
    hipEvent_t* ev = pool.ev;
    if ((e = hipEventRecord(ev[0], s)) != hipSuccess) return bail("hipEventRecord", e);
    if (!small_jobs.empty())
        hipLaunchKernelGGL(k_solve_small_batch, dim3((unsigned)small_jobs.size()), dim3(kSmallThreads), lds, s, (const SmallJob*)d_small);
    {
        // Two narrow workgroups per CU pay when there are more instances than CUs and the per-pivot passes are short
        // (measured, 512 instances: 1 024 nodes 1.4x, 4 096 nodes 1.06x; with <= 256 instances a narrow workgroup only has
        // half the lanes: 0.75-0.9x).  MCF_BATCH_THREADS = 512 / 1024 forces either.
        const char* bt = std::getenv("MCF_BATCH_THREADS");
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h0->device);
        int32_t max_nodes = 0;
        for (const MidJob& J : mid_jobs) if (J.g.n_nodes > max_nodes) max_nodes = J.g.n_nodes;
        const bool wide = bt ? std::atoi(bt) == 1024 : !((int)mid_jobs.size() > cus && max_nodes <= 4096);
        // the candidate-list jobs sit behind the others in the array (sorted above) and get the kernel that sweeps for itself
        const unsigned n_plain = (unsigned)(mid_jobs.size() - n_listing), n_list = (unsigned)n_listing;
        const MidJob* plain = d_mid;
        const MidJob* lists = d_mid ? d_mid + n_plain : nullptr;
        if (n_plain && wide) hipLaunchKernelGGL((k_solve_mid_batch<1024, false>), dim3(n_plain), dim3(1024), 0, s, plain);
        else if (n_plain) hipLaunchKernelGGL((k_solve_mid_batch<512, false>), dim3(n_plain), dim3(512), 0, s, plain);
        if (n_list && wide) hipLaunchKernelGGL((k_solve_mid_batch<1024, true>), dim3(n_list), dim3(1024), 0, s, lists);
        else if (n_list) hipLaunchKernelGGL((k_solve_mid_batch<512, true>), dim3(n_list), dim3(512), 0, s, lists);
    }
    if ((e = hipGetLastError()) != hipSuccess) return bail("mcf_solve_batch launch", e);
    // (every workgroup writes its final control block straight into its handle's pinned host copy: nothing to read back)
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return bail("hipStreamSynchronize", e);
    // A persistent-loop launch runs at most 2^22 loop iterations per job: a job that is still RUNNING below its budget
    // (large mid_loop = 1 instances) goes round again, like mcf_solve's loop, instead of coming back as "iteration limit".
    for (int round = 0; round < 4096 && !mid_jobs.empty(); ++round) {
        std::vector<MidJob> again_plain, again_list;
        for (const MidJob& J : mid_jobs) {
            const McfCtx* hc = J.host_ctx;
            if (hc->status == MCF_RUNNING && hc->pivots < J.cap) (J.rule == MCF_RULE_CANDIDATE_LIST ? again_list : again_plain).push_back(J);
        }
        if (again_plain.empty() && again_list.empty()) break;
        std::vector<MidJob> again(again_plain);
        again.insert(again.end(), again_list.begin(), again_list.end());
        if ((e = hipMemcpyAsync(d_mid, again.data(), again.size() * sizeof(MidJob), hipMemcpyHostToDevice, s)) != hipSuccess) return bail("hipMemcpy jobs", e);
        if (!again_plain.empty()) hipLaunchKernelGGL((k_solve_mid_batch<1024, false>), dim3((unsigned)again_plain.size()), dim3(1024), 0, s, (const MidJob*)d_mid);
        if (!again_list.empty()) hipLaunchKernelGGL((k_solve_mid_batch<1024, true>), dim3((unsigned)again_list.size()), dim3(1024), 0, s, (const MidJob*)(d_mid + again_plain.size()));
        if ((e = hipGetLastError()) != hipSuccess) return bail("mcf_solve_batch relaunch", e);
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return bail("hipStreamSynchronize", e);
    }
    if ((e = hipEventRecord(ev[1], s)) != hipSuccess) return bail("hipEventRecord", e);
    if ((e = hipEventSynchronize(ev[1])) != hipSuccess) return bail("hipEventSynchronize", e);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
    if (kernel_ms) *kernel_ms = ms;
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int ret = MCF_OK;
    for (int32_t i = 0; i < count; ++i) {
        mcf_handle* h = handles[i];
        h->ctx_current = true;
        h->stats.batches += 1;
        h->stats.solve_seconds += secs / count;   // the batch's wall time, shared out
        h->solved_once = true;
        if (h->h_ctx->status == MCF_INTERNAL_ERROR) { h->err = "internal error: preorder permutation did not close"; ret = MCF_E_INTERNAL; }
    }
    cleanup();
    return ret;
}

int mcf_get_result(mcf_handle* h, int32_t* status, int64_t* objective_hi_lo, int64_t* flow, int64_t* potential,
                   int8_t* in_tree, mcf_stats* stats) {
    if (!h) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = sync_ctx(h, h->stream);
    if (rc) return rc;
    const McfHostImage& im = h->im;
    // the flow copy (16 B per arc) is only made when something derived from it is asked for
    const bool need_arcw = status || objective_hi_lo || flow;
    std::vector<McfArcW> arcw;
    McfHostResult r;
    r.artificial_flow = -1;
    if (need_arcw) {
        arcw.resize(im.arcw.size());
        HIP_TRY(h, hipMemcpy(arcw.data(), h->d_arcw, arcw.size() * sizeof(McfArcW), hipMemcpyDeviceToHost));
        mcf_extract(im, arcw, h->h_ctx->status, r);
    }
    if (status) *status = r.status;
    if (objective_hi_lo) {
        objective_hi_lo[0] = (int64_t)(r.objective >> 64);
        objective_hi_lo[1] = (int64_t)(uint64_t)r.objective;
    }
    if (flow) for (int64_t i = 0; i < im.m; ++i) flow[im.orig[i]] = arcw[i].flow;  // engine order -> caller's
    if (potential) {
        std::vector<int64_t> pi(im.n_nodes);
        HIP_TRY(h, hipMemcpy(pi.data(), h->d_pi, pi.size() * 8, hipMemcpyDeviceToHost));
        for (int32_t v = 0; v < im.n; ++v) potential[v] = pi[v] - pi[im.n];
    }
    if (in_tree) {
        std::vector<int8_t> st(im.m_pad);
        HIP_TRY(h, hipMemcpy(st.data(), h->d_state, st.size(), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < im.m; ++i) in_tree[im.orig[i]] = st[i] == 0;
    }
    if (stats) {
        const McfCtx& c = *h->h_ctx;
        h->stats.pivots = c.pivots; h->stats.degenerate = c.degenerate; h->stats.bound_flips = c.bound_flips;
        h->stats.arcs_priced = c.arcs_priced; h->stats.nodes_moved = c.nodes_moved;
        h->stats.subtree_nodes = c.subtree_nodes; h->stats.cycle_arcs = c.cycle_arcs;
        h->stats.unbounded_arc = c.unbounded_arc >= 0 ? im.orig[c.unbounded_arc] : -1;
        h->stats.artificial_flow = r.artificial_flow;
        h->stats.pricing_mode = h->small ? 2 : (h->mid ? 3 : (h->rcached ? 1 : 0));
        h->stats.sweep_variant = ((h->view.vkey && h->opt.rule != MCF_RULE_DEVEX_BLOCK) ? 1 : 0) | (h->view.vkey && h->nt_sweep ? 2 : 0) | (h->view.dirty ? 4 : 0);
        h->stats.tree_blocks = h->bpl ? h->bpl_shift : 0; h->stats.tree_rebuilds = c.rebuilds;
        h->stats.run_pairs = h->run_pairs;
        h->stats.cycle_scans = c.scans; h->stats.scan_rounds = c.scan_rounds; {
            h->stats.arcs_swept = c.arcs_priced;  // full sweeps read what they cover ...
            if (h->view.dirty) {                   // ... incremental ones count per pricing workgroup
                std::vector<int64_t> sw(kMaxPriceBlocks);
                HIP_TRY(h, hipMemcpy(sw.data(), h->d_swept, sw.size() * 8, hipMemcpyDeviceToHost));
                h->stats.arcs_swept = 0;
                for (int64_t x : sw) h->stats.arcs_swept += x;
            }
        }
        h->stats.unbounded_rc = 0;
        if (c.status == MCF_UNBOUNDED && c.unbounded_arc >= 0) {
            std::vector<int64_t> pi(im.n_nodes);
            HIP_TRY(h, hipMemcpy(pi.data(), h->d_pi, pi.size() * 8, hipMemcpyDeviceToHost));
            const int64_t a = c.unbounded_arc;
            const int64_t rcost = im.cost64[a] + pi[im.tail[a]] - pi[im.head[a]];
            h->stats.unbounded_rc = rcost < 0 ? rcost : -rcost;
        }
        *stats = h->stats;
    }
    return MCF_OK;
}

int mcf_price_once(mcf_handle* h, int32_t rule, int64_t start, int64_t end, int64_t* arc, int32_t* dir, int64_t* key) {
    if (!h || !arc) return MCF_E_BAD_ARG;
    if (rule == MCF_RULE_CANDIDATE_LIST) rule = MCF_RULE_DANTZIG_FULL;  // the list is built by a Dantzig sweep
    if (rule != MCF_RULE_DANTZIG_FULL && rule != MCF_RULE_DEVEX_BLOCK) return MCF_E_BAD_ARG;
    if (start < 0 || end > h->im.m || start > end) { h->err = "mcf_price_once: bad range"; return MCF_E_BAD_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    // price even when the solve has finished: temporarily view the control block as running
    McfView v = h->view;
    v.dirty = nullptr;  // a parity hook prices everything and leaves the flags alone
    v.candx = nullptr;  // ... and the live list's records
    if (rule == MCF_RULE_DEVEX_BLOCK) v.weight = h->d_weight;
    HIP_TRY(h, hipMemcpyAsync(h->h_ctx, h->d_ctx, sizeof(McfCtx), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int32_t saved = h->h_ctx->status;
    int32_t running = MCF_RUNNING;
    HIP_TRY(h, hipMemcpyAsync(&h->d_ctx->status, &running, 4, hipMemcpyHostToDevice, h->stream));
    // whole arc list (shard 0 of 1), restricted to ORIGINAL indices [start, end)
    {
        const dim3 grid(h->price_blocks), block(kPriceThreads);
        const int64_t z = 0, one = 1;
        if (h->rcached) {
            if (rule == MCF_RULE_DEVEX_BLOCK)
                hipLaunchKernelGGL((k_price_rc<MCF_RULE_DEVEX_BLOCK, true, false>), grid, block, 0, h->stream, v, z, one, 0, start, end, h->d_cand_aux, (int64_t*)nullptr, (const int32_t*)nullptr);
            else
                hipLaunchKernelGGL((k_price_rc<MCF_RULE_DANTZIG, true, false>), grid, block, 0, h->stream, v, z, one, 0, start, end, h->d_cand_aux, (int64_t*)nullptr, (const int32_t*)nullptr);
        } else {
            if (rule == MCF_RULE_DEVEX_BLOCK)
                hipLaunchKernelGGL((k_price<MCF_RULE_DEVEX_BLOCK, true>), grid, block, 0, h->stream, v, z, one, 0, start, end, h->d_cand_aux);
            else
                hipLaunchKernelGGL((k_price<MCF_RULE_DANTZIG, true>), grid, block, 0, h->stream, v, z, one, 0, start, end, h->d_cand_aux);
        }
    }
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kReduceThreads), 0, h->stream, h->d_cand_aux, h->price_blocks, h->d_one);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(&h->d_ctx->status, &saved, 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->h_one, h->d_one, sizeof(McfCand), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const bool found = h->h_one->key > 0 && h->h_one->arc >= 0;
    *arc = found ? ((h->h_one->arc >> 32) & (MCF_DIR_FLAG - 1)) : -1;  // caller's arc index (Devex ids carry the direction in bit 30)
    // Devex and the capacity-weighted variant report their merit (bit pattern of the double), the other variants the violation
    if (key) *key = found ? ((rule == MCF_RULE_DEVEX_BLOCK || h->view.key_mode == MCF_KEY_CAPACITY) ? h->h_one->key : (h->h_one->key & ~MCF_FWD_BIT)) : 0;
    if (dir) {
        *dir = 0;
        if (found) {
            int8_t st = 0;
            HIP_TRY(h, hipMemcpy(&st, h->d_state + (h->h_one->arc & 0xffffffff), 1, hipMemcpyDeviceToHost));
            *dir = st;
        }
    }
    return MCF_OK;
}

int mcf_enqueue_price(mcf_handle* h, void* stream, int64_t* cand_out_dev) {
    if (!h || !cand_out_dev) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int32_t rule = h->opt.rule;
    h->ctx_current = false;
    h->external_driver = true;
    launch_price(h, s, h->view, rule, rule != MCF_RULE_DANTZIG);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kReduceThreads), 0, s, h->d_cand, h->price_blocks,
                       reinterpret_cast<McfCand*>(cand_out_dev));
    HIP_TRY(h, hipGetLastError());
    return MCF_OK;
}

int mcf_enqueue_pivot(mcf_handle* h, void* stream, const int64_t* cands_dev, int32_t ncand) {
    if (!h || !cands_dev || ncand < 1) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    h->ctx_current = false;
    h->external_driver = true;
    launch_k_pivot(h, s, reinterpret_cast<const McfCand*>(cands_dev), ncand, h->opt.rule, 1);
    launch_apply(h, s);
    HIP_TRY(h, hipGetLastError());
    return MCF_OK;
}

int mcf_shard_info(mcf_handle* h, int32_t* list_len, int32_t* minor_cap) {
    if (!h) return MCF_E_BAD_ARG;
    if (list_len) *list_len = h->price_blocks;
    if (minor_cap) *minor_cap = mcf_minor_cap(h->price_blocks);
    return MCF_OK;
}

int mcf_enqueue_price_list(mcf_handle* h, void* stream, int64_t* cands_out_dev) {
    if (!h || !cands_out_dev) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int32_t rule = h->opt.rule;
    h->ctx_current = false;
    h->external_driver = true;
    // the pricing grid writes one candidate per workgroup straight into the caller's buffer (price_blocks entries);
    // candidate-list rule: a no-op while minor iterations are pending, so the buffer keeps the live list
    launch_price(h, s, h->view, rule, rule != MCF_RULE_DANTZIG, reinterpret_cast<McfCand*>(cands_out_dev));
    HIP_TRY(h, hipGetLastError());
    return MCF_OK;
}

int mcf_enqueue_pivots(mcf_handle* h, void* stream, const int64_t* cands_dev, int32_t ncand, int32_t count) {
    if (!h || !cands_dev || ncand < 1 || count < 1) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    h->ctx_current = false;
    h->external_driver = true;
    for (int i = 0; i < count; ++i) {  // slot 0 takes the fresh list, the others re-price it (minor iterations)
        launch_k_pivot(h, s, reinterpret_cast<const McfCand*>(cands_dev), ncand, h->opt.rule, i == 0 ? 1 : 0);
        launch_apply(h, s);
    }
    HIP_TRY(h, hipGetLastError());
    return MCF_OK;
}

int mcf_poll(mcf_handle* h, void* stream, int32_t* status_or_running, int64_t* pivots) {
    if (!h) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    const int rc = sync_ctx(h, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    if (h->h_ctx->status == MCF_INTERNAL_ERROR) { h->err = "internal error: preorder permutation did not close"; return MCF_E_INTERNAL; }
    if (status_or_running) {
        const int32_t st = h->h_ctx->status;
        // infeasibility (artificial flow left) is only resolved by mcf_get_result
        *status_or_running = st == MCF_RUNNING ? -1 : st == MCF_OPTIMAL ? MCF_ST_OPTIMAL
                             : st == MCF_UNBOUNDED ? MCF_ST_UNBOUNDED : MCF_ST_ITERATION_LIMIT;
    }
    if (pivots) *pivots = h->h_ctx->pivots;
    return MCF_OK;
}

int mcf_time_pricing(mcf_handle* h, int32_t rule, int32_t reps, double* ms_per_launch) {
    if (!h || !ms_per_launch || reps < 1) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    McfView v = h->view;
    v.dirty = nullptr;  // timed sweeps are full sweeps
    v.candx = nullptr;  // (and leave the live list's records alone)
    if (rule == MCF_RULE_DEVEX_BLOCK) v.weight = h->d_weight;
    HIP_TRY(h, hipMemcpyAsync(h->h_ctx, h->d_ctx, sizeof(McfCtx), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int32_t saved = h->h_ctx->status;
    int32_t running = MCF_RUNNING;
    HIP_TRY(h, hipMemcpyAsync(&h->d_ctx->status, &running, 4, hipMemcpyHostToDevice, h->stream));
    hipEvent_t e0, e1;
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
    auto once = [&]() { launch_price(h, h->stream, v, rule == MCF_RULE_CANDIDATE_LIST ? MCF_RULE_DANTZIG : rule, rule == MCF_RULE_DEVEX_BLOCK, h->d_cand_aux); };
    once();  // warm
    HIP_TRY(h, hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) once();
    HIP_TRY(h, hipEventRecord(e1, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&h->d_ctx->status, &saved, 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / reps;
    return MCF_OK;
}

int mcf_time_copy(int32_t device, int64_t bytes, int32_t reps, double* ms_per_copy) {
    if (!ms_per_copy || bytes < 16 || reps < 1) return MCF_E_BAD_ARG;
    if (usable_devices() <= 0) return MCF_E_NO_DEVICE;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return MCF_E_HIP;
    uint4 *a = nullptr, *b = nullptr;
    const int64_t n16 = bytes / 16;
    if (hipMalloc(reinterpret_cast<void**>(&a), n16 * 16) != hipSuccess) return MCF_E_ALLOC;
    if (hipMalloc(reinterpret_cast<void**>(&b), n16 * 16) != hipSuccess) { (void)hipFree(a); return MCF_E_ALLOC; }
    (void)hipMemset(a, 1, n16 * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, nullptr, a, b, n16);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, nullptr, a, b, n16);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    *ms_per_copy = (double)ms / reps;
    return hipGetLastError() == hipSuccess ? MCF_OK : MCF_E_HIP;
}

int mcf_get_tree(mcf_handle* h, int32_t* parent, int32_t* pred_arc, int32_t* size, int32_t* pos, int32_t* order,
                 int8_t* state, int64_t* potential_with_root, int32_t* depth, int32_t* psize) {
    if (!h) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = read_ctx(h, h->stream);
    if (rc) return rc;
    const McfHostImage& im = h->im;
    std::vector<McfNode> nodes(im.n_nodes);
    HIP_TRY(h, hipMemcpy(nodes.data(), h->d_node, nodes.size() * sizeof(McfNode), hipMemcpyDeviceToHost));
    for (int32_t v = 0; v < im.n_nodes; ++v) {
        if (parent) parent[v] = nodes[v].parent;
        if (pred_arc) {
            const int64_t a = nodes[v].pred < 0 ? -1 : nodes[v].pred >> 1;
            pred_arc[v] = a < 0 ? -1 : (a < im.m ? im.orig[a] : (int32_t)a);  // artificial arcs keep m + node
        }
        if (size) size[v] = nodes[v].size;
        if (depth) depth[v] = nodes[v].depth;
    }
    const int cur = h->h_ctx->cur ^ (h->h_ctx->pending_flip ? 1 : 0);
    if (h->bpl) {
        // blocked preorder list: flatten the blocks handed out so far into the logical arrays the dense layout would hold
        const int arena = h->h_ctx->arena ^ ((h->h_ctx->pending_flip && h->h_ctx->rebuild) ? 1 : 0);
        const int32_t nb = h->h_ctx->alloc_next;
        const size_t slots = (size_t)nb << h->bpl_shift;
        std::vector<int32_t> tok(slots), psz(slots), ext((size_t)nb);
        std::vector<McfBlkMeta> meta((size_t)nb);
        HIP_TRY(h, hipMemcpy(tok.data(), arena ? h->d_order1 : h->d_order0, slots * 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(psz.data(), arena ? h->d_psz1 : h->d_psz0, slots * 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(meta.data(), h->d_bmeta[cur], (size_t)nb * sizeof(McfBlkMeta), hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(ext.data(), h->d_bext[arena], (size_t)nb * 4, hipMemcpyDeviceToHost));
        std::vector<int32_t> ford((size_t)im.n_nodes, -1), fpos((size_t)im.n_nodes, -1), fpsz((size_t)im.n_nodes, -1);
        if (!mcf_bpl_flatten(im.n_nodes, h->bpl_shift, nb, tok.data(), psz.data(), meta.data(), ext.data(), ford.data(), fpos.data(), fpsz.data())) {
            h->err = "internal error: the blocked preorder list does not tile the positions";
            return MCF_E_INTERNAL;
        }
        for (int32_t v = 0; v < im.n_nodes; ++v) {
            if (order) order[v] = ford[v];
            if (pos) pos[v] = fpos[v];
            if (psize) psize[v] = fpsz[v];
        }
        order = nullptr; pos = nullptr; psize = nullptr;
    }
    if (order) HIP_TRY(h, hipMemcpy(order, cur ? h->d_order1 : h->d_order0, (size_t)im.n_nodes * 4, hipMemcpyDeviceToHost));
    if (pos) HIP_TRY(h, hipMemcpy(pos, cur ? h->d_pos1 : h->d_pos0, (size_t)im.n_nodes * 4, hipMemcpyDeviceToHost));
    if (psize) {
        if (h->d_psz0) HIP_TRY(h, hipMemcpy(psize, cur ? h->d_psz1 : h->d_psz0, (size_t)im.n_nodes * 4, hipMemcpyDeviceToHost));
        else for (int32_t v = 0; v < im.n_nodes; ++v) psize[v] = -1;  // not kept for this handle
    }
    if (state) {
        std::vector<int8_t> st(im.m_pad);
        HIP_TRY(h, hipMemcpy(st.data(), h->d_state, st.size(), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < im.m; ++i) state[im.orig[i]] = st[i];
    }
    if (potential_with_root) HIP_TRY(h, hipMemcpy(potential_with_root, h->d_pi, (size_t)im.n_nodes * 8, hipMemcpyDeviceToHost));
    return MCF_OK;
}

int mcf_get_reduced_costs(mcf_handle* h, int64_t* rc_out, int32_t* resident) {
    if (!h || !rc_out) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const McfHostImage& im = h->im;
    if (resident) *resident = h->rcached ? 1 : 0;
    if (h->rcached) {
        std::vector<int64_t> rc(im.m_pad);
        HIP_TRY(h, hipMemcpy(rc.data(), h->d_rcache, rc.size() * 8, hipMemcpyDeviceToHost));
        if (h->view.rc_partial) {  // a sharded handle keeps only its own shard exact: the other arcs from the potentials
            std::vector<int64_t> pi(im.n_nodes);
            HIP_TRY(h, hipMemcpy(pi.data(), h->d_pi, pi.size() * 8, hipMemcpyDeviceToHost));
            std::vector<int8_t> mine(im.m, 0);
            for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
                int64_t lo, hi;
                mcf_bucket_slice(im.bucket_off, x, h->shard, h->shards, 0, 1, &lo, &hi);
                for (int64_t e2 = lo; e2 < hi; ++e2) mine[e2] = 1;
            }
            for (int64_t i = 0; i < im.m; ++i) if (!mine[i]) rc[i] = im.cost64[i] + pi[im.tail[i]] - pi[im.head[i]];
            if (resident) *resident = 2;  // own shard resident
        }
        for (int64_t i = 0; i < im.m; ++i) rc_out[im.orig[i]] = rc[i];
    } else {
        std::vector<int64_t> pi(im.n_nodes);
        HIP_TRY(h, hipMemcpy(pi.data(), h->d_pi, pi.size() * 8, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < im.m; ++i) rc_out[im.orig[i]] = im.cost64[i] + pi[im.tail[i]] - pi[im.head[i]];
    }
    return MCF_OK;
}

int mcf_get_pricing_keys(mcf_handle* h, int32_t* keys_out, int32_t* present) {
    if (!h || !keys_out) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const McfHostImage& im = h->im;
    if (present) *present = h->d_vkey ? 1 : 0;
    if (!h->d_vkey) { for (int64_t i = 0; i < im.m; ++i) keys_out[i] = 0; return MCF_OK; }
    std::vector<int32_t> vk(im.m_pad);
    HIP_TRY(h, hipMemcpy(vk.data(), h->d_vkey, vk.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < im.m; ++i) keys_out[im.orig[i]] = vk[i];
    return MCF_OK;
}

int mcf_get_weights(mcf_handle* h, float* weight_out) {
    if (!h || !weight_out) return MCF_E_BAD_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const McfHostImage& im = h->im;
    std::vector<float> w(im.m_pad);
    HIP_TRY(h, hipMemcpy(w.data(), h->d_weight, w.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < im.m; ++i) weight_out[im.orig[i]] = w[i];
    return MCF_OK;
}

#ifdef MCF_STAMPS
int mcf_debug_stamps(mcf_handle* h, unsigned long long* out8) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out8, h->d_rec1, 64, hipMemcpyDeviceToHost));
    return MCF_OK;
}
// k_pivot phase stamps (24 slots, [23] = launches that pivoted); clear = 1 zeroes them afterwards
int mcf_debug_pivot_stamps(mcf_handle* h, unsigned long long* out24, int clear) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_pivot_stamps), 24 * 8));
    if (clear) { unsigned long long z[24] = {0}; HIP_TRY(h, hipMemcpyToSymbol(HIP_SYMBOL(g_pivot_stamps), z, 24 * 8)); }
    return MCF_OK;
}
#endif

// ---------------------------------------------------------------------------------------------
// native DIMACS reader (host code; benchmarks/parsers/dimacs.py:105-286 restated for flat arrays)
// ---------------------------------------------------------------------------------------------
extern "C++" {
namespace {
int dimacs_fail(char* err, int32_t err_len, const std::string& msg) {
    if (err && err_len > 0) { std::snprintf(err, (size_t)err_len, "%s", msg.c_str()); }
    return MCF_E_BAD_ARG;
}

// parse one integer token; "inf" and values >= 1e15 map to `inf_value` when allow_inf
bool dimacs_int(const char*& p, int64_t* out, bool allow_inf, int64_t inf_value) {
    while (*p == ' ' || *p == '\t') ++p;
    if (allow_inf && (p[0] == 'i' || p[0] == 'I') && (p[1] == 'n' || p[1] == 'N') && (p[2] == 'f' || p[2] == 'F')) {
        p += 3;
        *out = inf_value;
        return true;
    }
    char* end = nullptr;
    errno = 0;
    const long long v = std::strtoll(p, &end, 10);
    if (end == p || errno) return false;
    if (*end == '.' || *end == 'e' || *end == 'E') {  // "12.0" is fine, "12.5" is not integral
        char* fend = nullptr;
        const double d = std::strtod(p, &fend);
        if (fend == p || d != (double)(long long)d) return false;
        p = fend;
        *out = allow_inf && d >= 1e15 ? inf_value : (int64_t)d;
        return true;
    }
    p = end;
    *out = allow_inf && v >= 1000000000000000LL ? inf_value : (int64_t)v;
    return true;
}

template <typename OnProblem, typename OnNode, typename OnArc>
int dimacs_walk(const char* path, char* err, int32_t err_len, OnProblem on_p, OnNode on_n, OnArc on_a) {
    FILE* f = std::fopen(path, "r");
    if (!f) return dimacs_fail(err, err_len, std::string("DIMACS file not found: ") + path);
    std::vector<char> line(1 << 16);
    int64_t line_no = 0;
    bool seen_p = false;
    int rc = 0;
    while (rc == 0 && std::fgets(line.data(), (int)line.size(), f)) {
        ++line_no;
        const char* p = line.data();
        while (*p == ' ' || *p == '\t') ++p;
        if (*p == 0 || *p == '\n' || *p == '\r' || *p == 'c') continue;
        const char kind = *p++;
        const std::string where = "Line " + std::to_string(line_no) + ": ";
        if (kind == 'p') {
            if (seen_p) { rc = dimacs_fail(err, err_len, where + "Multiple problem descriptor lines found."); break; }
            while (*p == ' ' || *p == '\t') ++p;
            if (std::strncmp(p, "min", 3) != 0) { rc = dimacs_fail(err, err_len, where + "Only 'min' (minimum cost flow) problems supported."); break; }
            p += 3;
            int64_t n = 0, m = 0;
            if (!dimacs_int(p, &n, false, 0) || !dimacs_int(p, &m, false, 0)) { rc = dimacs_fail(err, err_len, where + "Expected 'p min <nodes> <arcs>'."); break; }
            if (n <= 0) { rc = dimacs_fail(err, err_len, where + "Number of nodes must be positive."); break; }
            if (m < 0) { rc = dimacs_fail(err, err_len, where + "Number of arcs cannot be negative."); break; }
            seen_p = true;
            rc = on_p(n, m, where);
        } else if (kind == 'n') {
            if (!seen_p) { rc = dimacs_fail(err, err_len, where + "Node descriptor before problem descriptor."); break; }
            int64_t id = 0, sup = 0;
            if (!dimacs_int(p, &id, false, 0) || !dimacs_int(p, &sup, false, 0)) { rc = dimacs_fail(err, err_len, where + "Expected 'n <node_id> <supply>' with integer data."); break; }
            rc = on_n(id, sup, where);
        } else if (kind == 'a') {
            if (!seen_p) { rc = dimacs_fail(err, err_len, where + "Arc descriptor before problem descriptor."); break; }
            int64_t v[5];
            int k = 0;
            // tail head [lower] cap cost : the capacity token is the second-to-last one
            const char* q = p;
            int tokens = 0;
            for (const char* t = p; *t;) {
                while (*t == ' ' || *t == '\t') ++t;
                if (*t == 0 || *t == '\n' || *t == '\r') break;
                ++tokens;
                while (*t && *t != ' ' && *t != '\t' && *t != '\n' && *t != '\r') ++t;
            }
            if (tokens != 4 && tokens != 5) { rc = dimacs_fail(err, err_len, where + "Invalid arc descriptor format."); break; }
            bool ok = true;
            for (k = 0; k < tokens && ok; ++k) ok = dimacs_int(q, &v[k], k == tokens - 2, -1);
            if (!ok) { rc = dimacs_fail(err, err_len, where + "Failed to parse arc line (integer data expected)."); break; }
            const int64_t lower = tokens == 5 ? v[2] : 0;
            int64_t cap = v[tokens - 2];
            if (cap < 0) cap = -1;
            rc = on_a(v[0], v[1], lower, cap, v[tokens - 1], where);
        } else {
            rc = dimacs_fail(err, err_len, where + "Unknown line type '" + std::string(1, kind) + "'.");
        }
    }
    std::fclose(f);
    if (rc == 0 && !seen_p) rc = dimacs_fail(err, err_len, "No problem descriptor found. DIMACS file must contain a 'p min <nodes> <arcs>' line.");
    return rc;
}
}  // namespace
}  // extern "C++"

int mcf_dimacs_scan(const char* path, int64_t* n_nodes, int64_t* n_arcs, char* err, int32_t err_len) {
    if (!path || !n_nodes || !n_arcs) return MCF_E_BAD_ARG;
    int64_t n = 0, m = 0, arcs = 0;
    const int rc = dimacs_walk(
        path, err, err_len, [&](int64_t nn, int64_t mm, const std::string&) { n = nn; m = mm; return 0; },
        [&](int64_t, int64_t, const std::string&) { return 0; },
        [&](int64_t, int64_t, int64_t, int64_t, int64_t, const std::string&) { ++arcs; return 0; });
    if (rc) return rc;
    if (arcs != m)
        return dimacs_fail(err, err_len, "Arc count mismatch: problem descriptor specifies " + std::to_string(m) + " arcs, but " +
                                             std::to_string(arcs) + " arc descriptors found.");
    *n_nodes = n;
    *n_arcs = m;
    return MCF_OK;
}

int mcf_dimacs_load(const char* path, int64_t n_nodes, int64_t n_arcs, int32_t* tail, int32_t* head, int64_t* lower,
                    int64_t* cap, int64_t* cost, int64_t* supply, char* err, int32_t err_len) {
    if (!path || !supply || (n_arcs > 0 && (!tail || !head || !lower || !cap || !cost))) return MCF_E_BAD_ARG;
    for (int64_t v = 0; v < n_nodes; ++v) supply[v] = 0;
    int64_t i = 0;
    return dimacs_walk(
        path, err, err_len,
        [&](int64_t nn, int64_t mm, const std::string& where) {
            return (nn == n_nodes && mm == n_arcs) ? 0 : dimacs_fail(err, err_len, where + "counts differ from mcf_dimacs_scan");
        },
        [&](int64_t id, int64_t sup, const std::string& where) {
            if (id < 1 || id > n_nodes) return dimacs_fail(err, err_len, where + "node id outside [1, " + std::to_string(n_nodes) + "]");
            supply[id - 1] = sup;
            return 0;
        },
        [&](int64_t t, int64_t h, int64_t lo, int64_t cp, int64_t c, const std::string& where) {
            if (t < 1 || t > n_nodes || h < 1 || h > n_nodes)
                return dimacs_fail(err, err_len, where + "Arc references node IDs outside the expected range [1, " + std::to_string(n_nodes) + "]");
            if (i >= n_arcs) return dimacs_fail(err, err_len, where + "more arc descriptors than the problem line announces");
            tail[i] = (int32_t)(t - 1); head[i] = (int32_t)(h - 1); lower[i] = lo; cap[i] = cp; cost[i] = c;
            ++i;
            return 0;
        });
}

void mcf_destroy(mcf_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_all(h);
    delete h;
}

}  // extern "C"
