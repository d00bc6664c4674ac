// oracle/emul_engine.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Host-only emulation of the HIP engine: the per-pivot logic comes from the very
// headers the kernels are compiled from (network_flow_solver_amd/csrc/mcf_core.h,
// mcf_host.h); the three kernels (pricing sweep, pivot, apply) are replaced by scalar
// loops.  Purpose: (1) let the CPU-only test-suite exercise the preorder-tree pivot
// algorithm against the reference-derived goldens without a GPU, (2) differential
// testing on the GPU box -- the kernels must reproduce this pivot sequence exactly, so
// any divergence is a parallelisation / memory-ordering bug, not an algorithm bug.
//
// The shipped library never links or loads this file.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../network_flow_solver_amd/csrc/mcf_host.h"

namespace {

struct Emul {
    McfHostImage im;
    std::vector<int32_t> order1, path1, path2, pos1, psz1, ppos1, ppos2, reach, chg;
    std::vector<McfNode> rec1, rec2;
    std::vector<McfSeg> seg;
    McfCtx ctx{};
    McfView view{};    // value-initialised: every optional pointer of the view starts out null
    int rule = 0;
    int key_mode = 0;           // bits 8-9 of the `rule` argument: MCF_KEY_* (mcf_options.key_mode / forward_first)
    std::vector<int8_t> prio;   // MCF_KEY_PRIORITY: preference bits per arc, engine order
    int price_blocks = 8;
    std::vector<McfCand> cand;  // candidate-list rule: one entry per (virtual) pricing workgroup
    McfDevex dx;                // Devex: granule table + touched-weight list
    int bpl_shift = 0, bpl_pool = 0;   // bits 16-19 / 20-31 of the `rule` argument: blocked preorder list (0 = dense array)
    McfBplImage bp;
};

// `rule` argument: bits 0-7 the pricing rule, 8-9 the key variant, 16-19 log2 of the block size of the blocked preorder
// list (0: dense array), 20-31 its spare blocks (0: auto, 1: none -- every pivot rewrites the whole list --, k: k - 1).
void decode_rule(Emul& e, int32_t rule) {
    e.rule = rule & 0xff;
    e.key_mode = (rule >> 8) & 3;
    e.bpl_shift = (rule >> 16) & 15;
    const int f = (rule >> 20) & 0xfff;
    e.bpl_pool = f == 0 ? 0 : (f == 1 ? -1 : f - 1);
}

void bind(Emul& e) {
    McfHostImage& im = e.im;
    e.order1 = im.order;
    e.pos1 = im.pos;
    e.path1.assign(im.n_nodes, 0);
    e.path2.assign(im.n_nodes, 0);
    e.ppos1.assign(2 * (size_t)im.n_nodes, 0);   // (blocked list: positions, then slots)
    e.ppos2.assign(2 * (size_t)im.n_nodes, 0);
    e.rec1.assign(im.n_nodes, McfNode{0, 0, 0, 0});
    e.rec2.assign(im.n_nodes, McfNode{0, 0, 0, 0});
    e.seg.assign(2 * (size_t)im.n_nodes + 2, McfSeg{0, 0, 0, 0});
    std::memset(&e.ctx, 0, sizeof e.ctx);
    e.ctx.unbounded_arc = -1;
    McfView& v = e.view;
    v.n_nodes = im.n_nodes;
    v.m = im.m;
    v.tail = im.tail.data();
    v.head = im.head.data();
    v.cost = im.cost.data();
    v.orig = im.orig.data();
    for (int x = 0; x <= MCF_NUM_BUCKETS; ++x) v.bucket_off[x] = im.bucket_off[x];
    v.state = im.state.data();
    v.weight = e.rule == MCF_RULE_DEVEX_BLOCK ? im.weight.data() : nullptr;
    v.dx = e.rule == MCF_RULE_DEVEX_BLOCK ? &e.dx : nullptr;
    mcf_devex_fill_granules(&e.dx, im.bucket_off);
    v.arcw = im.arcw.data();
    v.pi = im.pi.data();
    v.node = im.node.data();
    v.order[0] = im.order.data();
    v.order[1] = e.order1.data();
    v.path1 = e.path1.data();
    v.path2 = e.path2.data();
    v.ppos1 = e.ppos1.data();
    v.ppos2 = e.ppos2.data();
    v.rec1 = e.rec1.data();
    v.rec2 = e.rec2.data();
    v.seg = e.seg.data();
    v.ctx = &e.ctx;
    v.key_mode = e.key_mode;
    v.prio = e.prio.empty() ? nullptr : e.prio.data();
    v.rc_partial = 0;
    v.vkey = nullptr;   // (the compressed keys ride on the resident reduced costs, which the emulation does not keep)
    v.vk_bigm = im.big_m; v.vk_half = 1 << 28;
    v.rcache = nullptr;  // the emulation always prices by gathering potentials: an independent
    v.adj_off = nullptr; // check of the engine's resident reduced costs
    v.adj = nullptr;
    v.dirty = nullptr;  // the emulation always sweeps everything: an independent check of the incremental sweeps
    v.posbuf[0] = im.pos.data();
    v.posbuf[1] = e.pos1.data();
    const size_t np = ((size_t)im.n_nodes + MCF_REACH_BLOCK - 1) / MCF_REACH_BLOCK * MCF_REACH_BLOCK + 4;
    im.psize.resize(np, 0);  // the scan reads whole coarse blocks / groups of four positions
    e.psz1 = im.psize;
    v.psz[0] = im.psize.data();
    v.psz[1] = e.psz1.data();
    // coarse index over the sizes (see McfView::reach); the emulation keeps it exact
    e.reach.assign(np / MCF_REACH_BLOCK + 8, 0);
    e.chg.assign(im.n_nodes, 0);
    v.reach = e.reach.data();
    v.chg = e.chg.data();
    for (int32_t b = 0; b < (im.n_nodes + MCF_REACH_BLOCK - 1) / MCF_REACH_BLOCK; ++b) mcf_reach_reindex_block(v, v.psz[0], b);
    v.bmeta[0] = v.bmeta[1] = nullptr;
    v.bext[0] = v.bext[1] = nullptr;
    v.blk_shift = 0; v.blk_cap = 0; v.ncandx = 0; v.candx = nullptr;
    if (e.bpl_shift > 0) {
        // blocked preorder list: the same logical preorder in physical blocks (mcf_core.h); the arenas take the place of
        // the order / size arrays, loc[] that of the positions
        mcf_bpl_build(im, e.bpl_shift, e.bpl_pool, e.bp);
        for (int a = 0; a < 2; ++a) {
            v.order[a] = e.bp.tok[a].data();
            v.psz[a] = e.bp.psz[a].data();
            v.bmeta[a] = e.bp.meta[a].data();
            v.bext[a] = e.bp.ext[a].data();
        }
        v.posbuf[0] = e.bp.loc.data();
        v.posbuf[1] = nullptr;
        v.reach = nullptr;
        e.chg.assign(std::max((size_t)im.n_nodes, 2 * (size_t)e.bp.cap + 2), 0);
        v.chg = e.chg.data();
        v.blk_shift = e.bp.shift;
        v.blk_cap = e.bp.cap;
        e.ctx.arena = 0;
        e.ctx.alloc_next = e.bp.dense;
        e.ctx.dense_blocks = e.bp.dense;
    }
}

// scalar stand-in for the pricing kernel: shard r of G, Devex block = ctx.block_index of
// ctx.num_blocks (Dantzig: the single block 0 of 1); same arc set as k_price, same tie rule.
// For the candidate-list rule it also fills e.cand[] with the best arc of every (virtual) pricing
// workgroup, using the kernel's arc -> workgroup map.  Returns the number of arcs looked at.
int64_t price(Emul& e, int64_t r, int64_t G, int64_t* key, int64_t* arc) {
    const McfView& v = e.view;
    const bool devex = e.rule == MCF_RULE_DEVEX_BLOCK;
    const bool listing = e.rule == MCF_RULE_CANDIDATE_LIST;
    const int64_t nlb = e.price_blocks / MCF_NUM_BUCKETS;
    if (listing) e.cand.assign(e.price_blocks, McfCand{0, -1});
    int64_t bk = 0, ba = -1, priced = 0;
    for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
        int64_t lo, hi;
        if (devex) mcf_devex_slice(v.dx, x, r, G, (int32_t)v.ctx->block_index, v.ctx->block_granules, &lo, &hi);
        else mcf_bucket_slice(v.bucket_off, x, r, G, 0, 1, &lo, &hi);
        priced += hi - lo;
        const int64_t g_lo = lo >> 2;
        for (int64_t i = lo; i < hi; ++i) {
            if (!v.state[i]) continue;
            const int64_t viol = mcf_violation(v, i);
            if (viol <= 0) continue;
            int64_t kk = mcf_dantzig_key(v, i, viol, v.state[i]);
            if (e.rule == MCF_RULE_DEVEX_BLOCK) {
                const double merit = ((double)viol * (double)viol) / (double)v.weight[i];
                std::memcpy(&kk, &merit, 8);
            }
            const int64_t id = mcf_pack_arc(devex ? mcf_devex_tie_id(v.orig[i], v.state[i]) : v.orig[i], i);
            if (mcf_cand_better(kk, id, bk, ba)) { bk = kk; ba = id; }
            if (listing) {
                McfCand& c = e.cand[(((i >> 2) - g_lo) >> 8) % nlb * MCF_NUM_BUCKETS + x];
                if (mcf_cand_better(kk, id, c.key, c.arc)) { c.key = kk; c.arc = id; }
            }
        }
    }
    *key = bk;
    *arc = ba;
    return priced;
}

// candidate-list minor iteration: re-price the listed arcs only
int64_t price_minor(Emul& e, const McfCand* cands, int64_t ncand, int64_t* key, int64_t* arc) {
    int64_t bk = 0, ba = -1;
    for (int64_t i = 0; i < ncand; ++i) {
        const int64_t kk = mcf_minor_key(e.view, cands[i].arc);
        if (mcf_cand_better(kk, cands[i].arc, bk, ba)) { bk = kk; ba = cands[i].arc; }
    }
    *key = bk;
    *arc = ba;
    return ncand;
}

// the apply pass of one pivot (k_update's permutation half), then the coarse blocks it touched are re-indexed
void apply_all(Emul& e) {
    McfCtx& c = e.ctx;
    if (!c.apply) return;
    if (MCF_HAS_BPL(e.view)) { mcf_bpl_update_seq(e.view, c); return; }
    // the two ranges the apply kernel covers: this pivot's and the stale one
    for (int32_t j = c.lo; j < c.hi; ++j) mcf_apply_one(e.view, c, j);
    for (int32_t j = c.prev_lo; j < c.prev_hi; ++j)
        if (j < c.lo || j >= c.hi) mcf_apply_one(e.view, c, j);
    if (!e.view.reach) return;
    const int32_t* znew = c.cur ? e.view.psz[0] : e.view.psz[1];  // the copy the pass just wrote
    auto range = [&](int32_t lo, int32_t hi) {
        if (hi <= lo) return;
        for (int32_t b = lo >> MCF_REACH_SHIFT; b <= (hi - 1) >> MCF_REACH_SHIFT; ++b) mcf_reach_reindex_block(e.view, znew, b);
    };
    range(c.lo, c.hi);
    range(c.prev_lo, c.prev_hi);
    // shrunken subtrees elsewhere: their blocks too (block-wise like the kernels: a block that meets one of the two
    // ranges was re-indexed above, from the copy that holds its new arrangement)
    auto meets = [&](int32_t b, int32_t lo, int32_t hi) { return hi > lo && b >= (lo >> MCF_REACH_SHIFT) && b <= ((hi - 1) >> MCF_REACH_SHIFT); };
    for (int32_t t = 0; t < c.nchg; ++t) {
        const int32_t b = e.view.chg[t] >> MCF_REACH_SHIFT;
        if (meets(b, c.lo, c.hi) || meets(b, c.prev_lo, c.prev_hi)) continue;
        mcf_reach_reindex_block(e.view, znew, b);   // (outside both ranges the two copies agree)
    }
}

void init_blocks(Emul& e, int64_t block_size) {
    McfCtx& c = e.ctx;
    const int64_t m = e.im.m;
    mcf_init_block_state(&c, e.rule, m, block_size);
    if (std::getenv("MCF_DEVEX_CYCLIC")) c.devex_cyclic = std::atoi(std::getenv("MCF_DEVEX_CYCLIC"));
    if (std::getenv("MCF_DEVEX_NOTUNE")) c.auto_tune = 0;
    e.price_blocks = mcf_price_blocks(m, 1, 0);
    c.minor_cap = mcf_minor_cap(e.price_blocks);
}

}  // namespace

extern "C" {

// Returns 0 or a negative MCF_E_* code.  Outputs sized like mcf_get_result / mcf_get_tree.
int emul_solve(int32_t n, int64_t m, const int32_t* tail, const int32_t* head, const int64_t* cost,
               const int64_t* cap, const int64_t* supply, int32_t rule, int64_t block_size, int64_t max_pivots,
               int32_t* status, int64_t* objective_hi_lo, int64_t* flow, int64_t* potential, int8_t* in_tree,
               int64_t* stats /*[12]*/, int32_t* parent, int32_t* pred_arc, int32_t* size, int32_t* pos,
               int32_t* order, int64_t* trace_arcs, int64_t trace_cap, int32_t bucketed, int32_t* depth, int32_t* psize,
               int32_t climb_budget /* < 0: always climb */, int64_t* scan_stats /*[2] or null*/,
               const int8_t* warm_in_tree /* null: cold start */, const int8_t* warm_at_upper, int32_t* warm_applied,
               const int8_t* arc_priority /* MCF_KEY_PRIORITY: caller's order; else null */) {
    Emul e;
    decode_rule(e, rule);
    rule = e.rule;
    int err = 0;
    std::string msg = mcf_build_image(n, m, tail, head, cost, cap, supply, e.im, &err, bucketed != 0);
    if (err) { std::fprintf(stderr, "emul_solve: %s\n", msg.c_str()); return err; }
    if (e.key_mode == MCF_KEY_PRIORITY && arc_priority) {   // the caller's order -> engine order
        e.prio.assign((size_t)e.im.m_pad, 0);
        for (int64_t i = 0; i < e.im.m; ++i) e.prio[(size_t)i] = (int8_t)(arc_priority[e.im.orig[(size_t)i]] & 3);
    }
    if (warm_applied) *warm_applied = 0;
    if (warm_in_tree) {  // same host routine the HIP library's mcf_set_basis runs
        const std::string why = mcf_apply_basis(e.im, warm_in_tree, warm_at_upper);
        if (!why.empty()) mcf_init_cold_basis(e.im);
        else if (warm_applied) *warm_applied = 1;
    }
    bind(e);
    McfCtx& c = e.ctx;
    c.max_pivots = max_pivots < 0 ? (20 * (m + n) > 100 ? 20 * (m + n) : 100) : max_pivots;
    init_blocks(e, block_size);
    c.climb_budget = climb_budget < 0 ? INT32_MAX : climb_budget;
    c.climb_depth = 0;   // (the cycle search never changes the pivot sequence; the emulation scans whenever it is asked to)
    const auto t0 = std::chrono::steady_clock::now();
    int64_t ntrace = 0;
    while (c.status == MCF_RUNNING) {
        int64_t key, arc;
        const bool minor = rule == MCF_RULE_CANDIDATE_LIST && c.minor_left > 0;
        const int64_t priced = minor ? price_minor(e, e.cand.data(), (int64_t)e.cand.size(), &key, &arc)
                                     : price(e, 0, 1, &key, &arc);
        if (c.pivots < c.max_pivots) c.arcs_priced += priced;
        if (trace_arcs && ntrace < trace_cap) trace_arcs[ntrace++] = arc < 0 ? -1 : ((arc >> 32) & (MCF_DIR_FLAG - 1));
        mcf_pivot_seq(e.view, key, arc, rule);
        apply_all(e);
    }
    if (c.pending_flip) { c.cur ^= 1; c.pending_flip = 0; if (c.rebuild) { c.arena ^= 1; c.rebuild = 0; } }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    McfHostResult r;
    mcf_extract(e.im, e.im.arcw, c.status, r);
    if (c.status == MCF_INTERNAL_ERROR) return -7;
    *status = r.status;
    objective_hi_lo[0] = (int64_t)(r.objective >> 64);
    objective_hi_lo[1] = (int64_t)(uint64_t)r.objective;
    for (int64_t i = 0; i < m; ++i) {  // engine order -> caller's order
        if (flow) flow[e.im.orig[i]] = e.im.arcw[i].flow;
        if (in_tree) in_tree[e.im.orig[i]] = e.im.state[i] == 0;
    }
    if (potential) for (int32_t v = 0; v < n; ++v) potential[v] = e.im.pi[v] - e.im.pi[n];
    stats[0] = c.pivots; stats[1] = c.degenerate; stats[2] = c.bound_flips; stats[3] = c.arcs_priced;
    stats[4] = c.nodes_moved; stats[5] = c.subtree_nodes; stats[6] = c.cycle_arcs;
    stats[7] = c.unbounded_arc >= 0 ? e.im.orig[c.unbounded_arc] : -1;
    stats[8] = r.artificial_flow; stats[9] = (int64_t)(secs * 1e9);
    stats[10] = c.minor_pivots; stats[11] = c.major_sweeps;
    if (scan_stats) { scan_stats[0] = c.scans; scan_stats[1] = c.scan_rounds; }
    std::vector<int32_t> ford, fpos, fpsz;
    const int32_t* ord = e.view.order[c.cur];
    const int32_t* pcur = e.view.posbuf[c.cur];
    const int32_t* zcur = e.view.psz[c.cur];
    if (MCF_HAS_BPL(e.view)) {   // the logical preorder out of the blocks
        ford.assign((size_t)n + 1, -1); fpos.assign((size_t)n + 1, -1); fpsz.assign((size_t)n + 1, -1);
        if (!mcf_bpl_flatten(n + 1, e.view.blk_shift, c.alloc_next, e.view.order[c.arena], e.view.psz[c.arena], e.view.bmeta[c.cur],
                             e.view.bext[c.arena], ford.data(), fpos.data(), fpsz.data())) {
            std::fprintf(stderr, "emul_solve: the blocked preorder list does not tile [0, n]\n");
            return -7;
        }
        ord = ford.data(); pcur = fpos.data(); zcur = fpsz.data();
        if (scan_stats) scan_stats[1] = c.rebuilds;   // (what a scan round is has no meaning here: report the dense rewrites)
    }
    for (int32_t v = 0; v <= n; ++v) {
        if (parent) parent[v] = e.im.node[v].parent;
        if (pred_arc) {
            const int64_t a = e.im.node[v].pred < 0 ? -1 : e.im.node[v].pred >> 1;
            pred_arc[v] = a < 0 ? -1 : (a < m ? e.im.orig[a] : (int32_t)a);  // artificial arcs keep m + node
        }
        if (size) size[v] = e.im.node[v].size;
        if (pos) pos[v] = pcur[v];
        if (depth) depth[v] = e.im.node[v].depth;
        if (psize) psize[v] = zcur[v];
        if (order) order[v] = ord[v];
    }
    return 0;
}


// ---- step-wise API: lets the multi-process (gloo) tests drive one replica per rank exactly
// the way the HIP engine is driven per pivot (price shard -> all-gather -> pivot).
void* emul_create(int32_t n, int64_t m, const int32_t* tail, const int32_t* head, const int64_t* cost,
                  const int64_t* cap, const int64_t* supply, int32_t rule, int64_t block_size, int32_t bucketed,
                  const int8_t* arc_priority) {
    Emul* e = new Emul();
    decode_rule(*e, rule);
    int err = 0;
    std::string msg = mcf_build_image(n, m, tail, head, cost, cap, supply, e->im, &err, bucketed != 0);
    if (err) { std::fprintf(stderr, "emul_create: %s\n", msg.c_str()); delete e; return nullptr; }
    if (e->key_mode == MCF_KEY_PRIORITY && arc_priority) {   // the caller's order -> engine order
        e->prio.assign((size_t)e->im.m_pad, 0);
        for (int64_t i = 0; i < e->im.m; ++i) e->prio[(size_t)i] = (int8_t)(arc_priority[e->im.orig[(size_t)i]] & 3);
    }
    bind(*e);
    McfCtx& c = e->ctx;
    c.max_pivots = INT64_MAX;
    c.climb_budget = INT32_MAX;
    init_blocks(*e, block_size);
    return e;
}

// best candidate of shard `r` of `G` (and of the current Devex block).  Candidate-list rule: while
// minor iterations are pending the pricing launch is a no-op and key_arc keeps the last sweep's
// entry (exactly what the skipped k_price + k_reduce leave behind on the GPU).
void emul_price(void* h, int64_t r, int64_t G, int64_t* key_arc /*[2]*/) {
    Emul* e = static_cast<Emul*>(h);
    if (e->rule == MCF_RULE_CANDIDATE_LIST && e->ctx.minor_left > 0 && e->ctx.status == MCF_RUNNING) return;
    key_arc[0] = 0; key_arc[1] = -1;
    if (e->ctx.status == MCF_RUNNING) price(*e, r, G, &key_arc[0], &key_arc[1]);
}

// Candidate-list rule over G ranks (mcf_shard_info / mcf_enqueue_price_list): the pricing grid of a sharded handle is
// sized for m / G arcs, and a sweep leaves one candidate per (virtual) pricing workgroup.
void emul_set_shards(void* h, int64_t G) {
    Emul* e = static_cast<Emul*>(h);
    e->price_blocks = mcf_price_blocks(e->im.m, G, 0);
    e->ctx.minor_cap = mcf_minor_cap(e->price_blocks);
}
int32_t emul_list_len(void* h) { return static_cast<Emul*>(h)->price_blocks; }
int32_t emul_minor_cap(void* h) { return static_cast<Emul*>(h)->ctx.minor_cap; }

// out[2 * price_blocks] <- the shard's per-workgroup candidates; a no-op while minor iterations are pending
void emul_price_list(void* h, int64_t r, int64_t G, int64_t* out) {
    Emul* e = static_cast<Emul*>(h);
    if (e->rule == MCF_RULE_CANDIDATE_LIST && e->ctx.minor_left > 0 && e->ctx.status == MCF_RUNNING) return;
    if (e->ctx.status != MCF_RUNNING) return;
    int64_t key, arc;
    price(*e, r, G, &key, &arc);
    for (int i = 0; i < e->price_blocks; ++i) { out[2 * i] = e->cand[i].key; out[2 * i + 1] = e->cand[i].arc; }
}

// apply the best of `ncand` (key, arc) candidates, like k_pivot + k_apply
void emul_pivot(void* h, const int64_t* cands, int32_t ncand) {
    Emul* e = static_cast<Emul*>(h);
    McfCtx& c = e->ctx;
    if (c.status != MCF_RUNNING) return;
    const bool minor = e->rule == MCF_RULE_CANDIDATE_LIST && c.minor_left > 0;
    int64_t key = 0, arc = -1;
    for (int32_t i = 0; i < ncand; ++i) {
        const int64_t kk = minor ? mcf_minor_key(e->view, cands[2 * i + 1]) : cands[2 * i];
        if (mcf_cand_better(kk, cands[2 * i + 1], key, arc)) { key = kk; arc = cands[2 * i + 1]; }
    }
    mcf_pivot_seq(e->view, key, arc, e->rule);
    apply_all(*e);
}

// `count` pivot slots on one gathered list (mcf_enqueue_pivots): the first takes the sweep's keys unless minor iterations
// are pending, the others only re-price the list and idle once it is exhausted (k_pivot with have_sweep == 0)
void emul_pivots(void* h, const int64_t* cands, int32_t ncand, int32_t count) {
    Emul* e = static_cast<Emul*>(h);
    for (int32_t i = 0; i < count; ++i) {
        if (e->ctx.status != MCF_RUNNING) return;
        if (i > 0 && e->ctx.minor_left <= 0) continue;
        emul_pivot(h, cands, ncand);
    }
}

void emul_set_max_pivots(void* h, int64_t cap) {
    Emul* e = static_cast<Emul*>(h);
    e->ctx.max_pivots = cap;
    if (e->ctx.status == MCF_PIVOT_LIMIT && e->ctx.pivots < cap) e->ctx.status = MCF_RUNNING;
}

// status: -1 running, else MCF_ST_* numbering (infeasible resolved from the artificial flow)
void emul_poll(void* h, int32_t* status, int64_t* pivots, int64_t* objective_hi_lo, int64_t* flow) {
    Emul* e = static_cast<Emul*>(h);
    McfHostResult r;
    mcf_extract(e->im, e->im.arcw, e->ctx.status, r);
    *status = e->ctx.status == MCF_RUNNING ? -1 : r.status;
    *pivots = e->ctx.pivots;
    if (objective_hi_lo) { objective_hi_lo[0] = (int64_t)(r.objective >> 64); objective_hi_lo[1] = (int64_t)(uint64_t)r.objective; }
    if (flow) for (int64_t i = 0; i < e->im.m; ++i) flow[e->im.orig[i]] = e->im.arcw[i].flow;
}

void emul_destroy(void* h) { delete static_cast<Emul*>(h); }

// Incremental pricing: the arc -> pricing-workgroup map the marking passes use (mcf_price_block_of) against the
// sweep's own loop structure (k_price_rc: workgroup lb * 8 + x takes the groups g_lo + lb * 256 + lane + j * nlb * 256
// of this rank's share of bucket x).  Returns the number of arcs on which the two disagree (0 = consistent);
// arcs of other ranks must map to -1.
int64_t emul_check_block_map(int32_t n, int64_t m, const int32_t* tail, const int32_t* head, const int64_t* cost,
                             const int64_t* cap, const int64_t* supply, int32_t price_blocks, int64_t shard, int64_t shards) {
    Emul e;
    int err = 0;
    std::string msg = mcf_build_image(n, m, tail, head, cost, cap, supply, e.im, &err, true);
    if (err) return -1;
    bind(e);
    McfDirty d;
    std::memset(&d, 0, sizeof d);
    d.nlb = price_blocks / MCF_NUM_BUCKETS;
    for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
        int64_t lo, hi;
        mcf_bucket_slice(e.im.bucket_off, x, shard, shards, 0, 1, &lo, &hi);
        d.lo[x] = (int32_t)lo; d.hi[x] = (int32_t)hi;
    }
    std::vector<int32_t> truth(m, -1);
    const int64_t nlb = d.nlb;
    for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
        const int64_t lo = d.lo[x], hi = d.hi[x], g_lo = lo >> 2, g_hi = (hi + 3) >> 2;
        for (int64_t lb = 0; lb < nlb; ++lb)
            for (int64_t g0 = g_lo + lb * 256; g0 < g_hi; g0 += nlb * 256)
                for (int64_t g = g0; g < g0 + 256 && g < g_hi; ++g)
                    for (int k = 0; k < 4; ++k) {
                        const int64_t i = (g << 2) + k;
                        if (i >= lo && i < hi) truth[i] = (int32_t)(lb * MCF_NUM_BUCKETS + x);
                    }
    }
    int64_t bad = 0;
    for (int64_t i = 0; i < m; ++i) bad += mcf_price_block_of(e.view, &d, i) != truth[i];
    return bad;
}

}  // extern "C"
