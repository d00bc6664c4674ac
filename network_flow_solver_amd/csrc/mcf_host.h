// mcf_host.h -- host-side problem staging shared by the HIP engine (mcf_engine.hip)
// and the CPU emulation build used by the test-suite (oracle/emul_engine.cpp).
//
// Restates, in integer form, the set-up half of NetworkSimplex.__init__
// (/root/reference/src/network_solver/simplex.py):
//   _build_vectorized_arrays  :434-456  -> arc SoA (tail, head, cost, state) + walk records
//   penalty cost              :161-163  -> big-M for artificial arcs
//   _initialize_tree          :619-730  -> one artificial root arc per node, all basic
//   basis.rebuild             basis.py:82-122 -> parent/pred/size/pos/order + potentials
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "mcf_core.h"

struct McfHostImage {
    int32_t n = 0;          // real nodes
    int32_t n_nodes = 0;    // n + 1 (root = n)
    int64_t m = 0;          // real arcs
    int64_t m_pad = 0;      // m rounded up to a multiple of 1024 (padding arcs have state 0)
    int64_t big_m = 0;
    std::vector<int32_t> tail, head, cost;  // [m_pad]  ENGINE order (see mcf_build_image)
    std::vector<int32_t> orig;              // [m_pad]  engine arc index -> caller's arc index
    int64_t bucket_off[MCF_NUM_BUCKETS + 1] = {0};
    std::vector<int64_t> cost64;            // [m] exact costs for the objective (engine order)
    std::vector<int8_t> state;              // [m_pad]
    std::vector<float> weight;              // [m_pad]
    std::vector<McfArcW> arcw;              // [m + n]
    std::vector<int64_t> pi;                // [n_nodes]
    std::vector<McfNode> node;              // [n_nodes]
    std::vector<int32_t> order;             // [n_nodes]
    std::vector<int32_t> pos;               // [n_nodes] preorder position of every node (inverse of order)
    std::vector<int32_t> psize;             // [n_nodes] subtree size of the node at each POSITION
    std::vector<int64_t> supply;            // [n]
    // resident reduced costs + node->arc adjacency (mcf_build_rcache)
    std::vector<int64_t> rcache;            // [m_pad]
    std::vector<int64_t> adj_off;           // [n + 1]
    std::vector<int64_t> adj;               // [2m]
};

// The all-artificial start basis (_initialize_tree, simplex.py:619-730): every real arc non-basic at its
// lower bound, one artificial arc per node, all of them basic, carrying the node's supply to / from the root.
inline void mcf_init_cold_basis(McfHostImage& im) {
    const int32_t n = im.n, root = im.n;
    const int64_t m = im.m;
    for (int64_t e = 0; e < m; ++e) { im.state[e] = 1; im.arcw[e].flow = 0; }
    std::fill(im.weight.begin(), im.weight.end(), 1.0f);
    im.pi.assign(im.n_nodes, 0);
    im.node.assign(im.n_nodes, McfNode{-1, -1, 1, 0});
    im.order.assign(im.n_nodes, 0);
    im.pos.assign(im.n_nodes, 0);
    im.psize.assign(im.n_nodes, 1);
    im.psize[0] = im.n_nodes;  // the root sits at position 0
    im.node[root] = McfNode{-1, -1, im.n_nodes, 0};
    im.order[0] = root;
    im.pos[root] = 0;
    for (int32_t v = 0; v < n; ++v) {
        const int64_t a = m + v;
        const int64_t s = im.supply[v];
        // supply >= 0: arc v -> root carrying s (up arc); demand: root -> v carrying -s.
        // Zero-flow tree arcs point at the root, so the start tree is strongly feasible.
        const int32_t up = s >= 0 ? 1 : 0;
        im.arcw[a] = McfArcW{MCF_INF, s >= 0 ? s : -s};
        im.pi[v] = up ? -im.big_m : im.big_m;
        im.node[v] = McfNode{root, (int32_t)((a << 1) | up), 1, 1};  // depth 1: hangs off the root
        im.order[v + 1] = v;
        im.pos[v] = v + 1;
    }
}

// Warm start (simplex.py:740-1010 restated for the preorder-array tree): install the caller's basis.
//   in_tree[m]   (caller's arc order) the basic real arcs; they must form a forest;
//   at_upper[m]  (may be null) non-basic arcs that sit at their capacity instead of at zero.
// Every component of the forest that does not reach the root gets one artificial arc (the reference adds
// them per component, :826-873); tree flows follow from conservation (:905-1010) and must respect the
// bounds, else the basis is rejected ("" = applied; on rejection the image is back at the cold start, so
// the caller simply solves from there, as the reference does).  Beyond the reference: a basic arc that
// sits at a bound pointing the wrong way (zero flow away from the root / full towards it) would break
// the strongly feasible tree the pivot rule relies on; it is made non-basic at that bound and its
// subtree re-hung on the root by a zero-flow artificial arc.
inline std::string mcf_apply_basis(McfHostImage& im, const int8_t* in_tree, const int8_t* at_upper) {
    const int32_t n = im.n, root = im.n, N = im.n_nodes;
    const int64_t m = im.m;
    if (!in_tree) return "null basis";
    // --- forest check (union-find over the real nodes)
    std::vector<int32_t> uf(N);
    for (int32_t v = 0; v < N; ++v) uf[v] = v;
    auto find = [&](int32_t x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
    std::vector<int8_t> basic(m, 0);
    int64_t nbasic = 0;
    for (int64_t e = 0; e < m; ++e) {
        if (!in_tree[im.orig[e]]) continue;
        const int32_t a = find(im.tail[e]), b = find(im.head[e]);
        if (a == b) return "basis arcs contain a cycle";
        uf[a] = b;
        basic[e] = 1;
        ++nbasic;
    }
    if (nbasic == 0) return "empty basis";
    // --- non-basic flows and node balances
    std::vector<int64_t> flow(m, 0), bal(N, 0);
    for (int32_t v = 0; v < n; ++v) bal[v] = im.supply[v];
    for (int64_t e = 0; e < m; ++e) {
        if (basic[e] || !at_upper || !at_upper[im.orig[e]]) continue;
        const int64_t cp = im.arcw[e].cap;
        if (cp >= MCF_INF) return "uncapacitated arc marked as sitting at its capacity";
        flow[e] = cp;
        bal[im.tail[e]] -= cp;
        bal[im.head[e]] += cp;
    }
    // --- tree adjacency (CSR over the basic arcs)
    std::vector<int64_t> off((size_t)N + 1, 0);
    auto build_adj = [&](std::vector<int32_t>& adj) {
        std::fill(off.begin(), off.end(), 0);
        for (int64_t e = 0; e < m; ++e) if (basic[e]) { off[im.tail[e] + 1]++; off[im.head[e] + 1]++; }
        for (int32_t v = 0; v < N; ++v) off[v + 1] += off[v];
        adj.assign((size_t)off[N], 0);
        std::vector<int64_t> fill(off.begin(), off.end() - 1);
        for (int64_t e = 0; e < m; ++e) if (basic[e]) { adj[fill[im.tail[e]]++] = (int32_t)e; adj[fill[im.head[e]]++] = (int32_t)e; }
    };
    std::vector<int32_t> adj, order(N), parent(N), depth(N), stack;
    std::vector<int64_t> parc(N);     // arc to the parent (>= m: artificial), -1 for the root
    std::vector<int8_t> rep(N, 0);    // hangs on the root by its artificial arc
    // preorder DFS from the root over: root -> every representative, basic arcs below
    auto dfs = [&]() -> bool {
        std::vector<int8_t> seen(N, 0);
        int32_t cnt = 0;
        order[cnt++] = root; seen[root] = 1; parent[root] = -1; parc[root] = -1; depth[root] = 0;
        for (int32_t r = 0; r < n; ++r) {
            if (!rep[r]) continue;
            if (seen[r]) return false;
            seen[r] = 1; parent[r] = root; parc[r] = m + r; depth[r] = 1;
            stack.clear(); stack.push_back(r);
            while (!stack.empty()) {
                const int32_t u = stack.back(); stack.pop_back();
                order[cnt++] = u;
                for (int64_t p = off[u + 1] - 1; p >= off[u]; --p) {  // reversed push: children come out in adjacency order
                    const int32_t e = adj[p];
                    const int32_t w = im.tail[e] == u ? im.head[e] : im.tail[e];
                    if (seen[w]) continue;
                    seen[w] = 1; parent[w] = u; parc[w] = e; depth[w] = depth[u] + 1;
                    stack.push_back(w);
                }
            }
        }
        return cnt == N;
    };
    // representatives: the lowest node of every component
    {
        std::vector<int8_t> have(N, 0);
        for (int32_t v = 0; v < n; ++v) { const int32_t c = find(v); if (!have[c]) { have[c] = 1; rep[v] = 1; } }
    }
    build_adj(adj);
    if (!dfs()) return "basis does not span the nodes";
    // --- tree flows from conservation, children before parents; bounds checked
    std::vector<int64_t> art_flow(n, 0);
    std::vector<int8_t> art_up(n, 1);
    auto flows_ok = [&]() -> bool {
        std::vector<int64_t> b2(bal);
        for (int32_t k = N - 1; k >= 1; --k) {
            const int32_t v = order[k];
            const int64_t a = parc[v];
            const int64_t x = b2[v];  // surplus the subtree of v has to send up (negative: must receive)
            if (a >= m) {
                art_up[v] = x >= 0 ? 1 : 0;
                art_flow[v] = x >= 0 ? x : -x;
            } else {
                const bool up = im.tail[a] == v;
                const int64_t f = up ? x : -x;
                if (f < 0 || f > im.arcw[a].cap) return false;
                flow[a] = f;
            }
            b2[parent[v]] += x;
        }
        return b2[root] == 0;
    };
    if (!flows_ok()) return "basis incompatible with the current supplies / capacities";
    // --- where each component hangs.  A Basis names real arcs only, so the node its artificial arc sat on is not handed over;
    // the tree flows do not depend on it, but strong feasibility does: a basic arc sitting on a bound has to point the right way
    // relative to the root, and every such arc on the path between the old and the new hanging node turns round.  Count, for
    // every node v, the wrong-way arcs its component would have if it hung on v (moving the hanging node across one arc
    // changes that arc's contribution only: one preorder pass), and hang each component on its best node, lowest index among
    // equals.  A basis taken from a strongly feasible tree (every FlowResult.basis of this engine) then needs no repair below and
    // an optimal one is confirmed in zero pivots; before, the lowest node was used and an optimal 1M-node basis lost 100+
    // degenerate arcs to the repair, one pivot each to win back.
    {
        auto wrong = [&](int64_t a, bool up) {
            const int64_t cp = im.arcw[a].cap;
            return (up && cp < MCF_INF && flow[a] == cp) || (!up && flow[a] == 0) ? 1 : 0;
        };
        std::vector<int32_t> viol(N, 0), comp_wrong(N, 0);
        bool any = false;
        for (int32_t k = 1; k < N; ++k) {
            const int32_t v = order[k];
            const int64_t a = parc[v];
            if (a < m && wrong(a, im.tail[a] == v)) { comp_wrong[find(v)]++; any = true; }
        }
        if (any) {
            std::vector<int32_t> best(N, -1);
            for (int32_t k = 1; k < N; ++k) {          // preorder: a parent's count is final before its children's
                const int32_t v = order[k];
                const int64_t a = parc[v];
                if (a >= m) viol[v] = comp_wrong[find(v)];
                else {
                    const bool up = im.tail[a] == v;
                    viol[v] = viol[parent[v]] - wrong(a, up) + wrong(a, !up);
                }
                int32_t& b = best[find(v)];
                if (b < 0 || viol[v] < viol[b] || (viol[v] == viol[b] && v < b)) b = v;
            }
            bool moved = false;
            for (int32_t v = 0; v < n; ++v) {
                if (!rep[v]) continue;
                const int32_t b = best[find(v)];
                if (b != v && viol[b] < viol[v]) { rep[v] = 0; rep[b] = 1; moved = true; }
            }
            if (moved && (!dfs() || !flows_ok())) return "internal: re-hanging a basis component failed";
        }
    }
    // --- strong feasibility: drop wrong-way degenerate basic arcs, re-hang their subtrees on the root
    bool changed = false;
    for (int32_t k = 1; k < N; ++k) {
        const int32_t v = order[k];
        const int64_t a = parc[v];
        if (a >= m) continue;
        const bool up = im.tail[a] == v;
        const int64_t cp = im.arcw[a].cap;
        if ((up && cp < MCF_INF && flow[a] == cp) || (!up && flow[a] == 0)) {
            basic[a] = 0;      // stays at the bound it is sitting on (its flow is unchanged, so are all balances)
            rep[v] = 1;
            changed = true;
        }
    }
    if (changed) {
        build_adj(adj);
        // the dropped arcs' flows were tree flows so far: move them into the node balances
        std::vector<int64_t> b3(N, 0);
        for (int32_t v = 0; v < n; ++v) b3[v] = im.supply[v];
        for (int64_t e = 0; e < m; ++e) if (!basic[e] && flow[e] != 0) { b3[im.tail[e]] -= flow[e]; b3[im.head[e]] += flow[e]; }
        bal.swap(b3);
        if (!dfs() || !flows_ok()) return "internal: strong-feasibility repair failed";
    }
    // --- install: states, flows, records, preorder arrays, potentials
    for (int64_t e = 0; e < m; ++e) {
        im.arcw[e].flow = flow[e];
        const int64_t cp = im.arcw[e].cap;
        im.state[e] = basic[e] ? 0 : ((flow[e] != 0 && flow[e] == cp) ? -1 : 1);
    }
    std::fill(im.weight.begin(), im.weight.end(), 1.0f);
    for (int32_t v = 0; v < n; ++v) im.arcw[m + v] = McfArcW{MCF_INF, 0};
    std::vector<int32_t> size(N, 1);
    for (int32_t k = N - 1; k >= 1; --k) size[parent[order[k]]] += size[order[k]];
    im.pi[root] = 0;
    im.node[root] = McfNode{-1, -1, N, 0};
    for (int32_t k = 0; k < N; ++k) {
        const int32_t v = order[k];
        im.order[k] = v;
        im.pos[v] = k;
        im.psize[k] = size[v];
        if (v == root) continue;
        const int64_t a = parc[v];
        int32_t up;
        int64_t c;
        if (a >= m) { up = art_up[v]; c = im.big_m; im.arcw[a].flow = art_flow[v]; }
        else { up = im.tail[a] == v ? 1 : 0; c = im.cost[a]; }
        im.node[v] = McfNode{parent[v], (int32_t)((a << 1) | up), size[v], depth[v]};
        im.pi[v] = up ? im.pi[parent[v]] - c : im.pi[parent[v]] + c;
    }
    return "";
}

// resident reduced costs of the current image (values only; the adjacency does not depend on the basis)
inline void mcf_refresh_rcache(McfHostImage& im) {
    if (im.rcache.empty()) return;
    for (int64_t e = 0; e < im.m; ++e) im.rcache[e] = (int64_t)im.cost[e] + im.pi[im.tail[e]] - im.pi[im.head[e]];
}

// Validate the caller's arrays and build the start basis.  Returns "" or an error text.
//
// Arc layout ("engine order"): arcs are bucketed by the node range their HEAD falls in --
// MCF_NUM_BUCKETS = 8 ranges, one per XCD -- and sorted by tail inside a bucket (stable, so
// equal tails keep the caller's order).  The pricing kernel lets the workgroups of XCD x
// sweep bucket x: the random head gathers then stay inside 1/8 of the potential array, which
// fits that XCD's private 4 MiB L2 up to ~4M nodes, while the tail gathers run through pi
// almost sequentially (and coalesce).  `bucketed = false` keeps the caller's order and cuts
// it into 8 equal slices instead (used by tests to show results do not depend on the layout).
inline std::string mcf_build_image(int32_t n, int64_t m, const int32_t* tail, const int32_t* head,
                                   const int64_t* cost, const int64_t* cap, const int64_t* supply,
                                   McfHostImage& im, int* err_code, bool bucketed = true) {
    *err_code = -1;  // MCF_E_BAD_ARG
    if (n < 1) return "n must be >= 1";
    if (m < 0) return "m must be >= 0";
    if (m > 0 && (!tail || !head || !cost || !cap)) return "null arc array";
    if (!supply) return "null supply array";
    if ((int64_t)n + m >= ((int64_t)1 << 30)) { *err_code = -5; return "m + n must stay below 2^30"; }
    im.n = n;
    im.n_nodes = n + 1;
    im.m = m;
    im.m_pad = ((m + 1023) / 1024) * 1024;
    if (im.m_pad == 0) im.m_pad = 1024;
    im.tail.assign(im.m_pad, 0);
    im.head.assign(im.m_pad, 0);
    im.cost.assign(im.m_pad, 0);
    im.orig.assign(im.m_pad, 0);
    im.cost64.assign(m, 0);
    im.state.assign(im.m_pad, 0);
    im.weight.assign(im.m_pad, 1.0f);
    im.arcw.assign(m + n, McfArcW{0, 0});
    im.supply.assign(supply, supply + n);

    int64_t max_abs_cost = 0, total = 0;
    for (int32_t v = 0; v < n; ++v) total += supply[v];
    if (total != 0) return "supplies do not balance";
    for (int64_t i = 0; i < m; ++i) {
        if (tail[i] < 0 || tail[i] >= n || head[i] < 0 || head[i] >= n) return "arc end point out of range";
        if (tail[i] == head[i]) return "self-loop";
        const int64_t c = cost[i];
        if (c > INT32_MAX || c < -(int64_t)INT32_MAX) { *err_code = -5; return "|cost| must fit int32"; }
    }

    // engine order: stable counting sort by tail, then stable distribution into head buckets
    std::vector<int32_t> perm(m);
    if (bucketed && m > 0) {
        std::vector<int64_t> cnt((size_t)n + 1, 0);
        for (int64_t i = 0; i < m; ++i) cnt[tail[i] + 1]++;
        for (int32_t v = 0; v < n; ++v) cnt[v + 1] += cnt[v];
        std::vector<int32_t> by_tail(m);
        for (int64_t i = 0; i < m; ++i) by_tail[cnt[tail[i]]++] = (int32_t)i;
        const int64_t per = ((int64_t)n + MCF_NUM_BUCKETS - 1) / MCF_NUM_BUCKETS;
        int64_t bcnt[MCF_NUM_BUCKETS + 1] = {0};
        for (int64_t i = 0; i < m; ++i) bcnt[head[i] / per + 1]++;
        for (int x = 0; x < MCF_NUM_BUCKETS; ++x) bcnt[x + 1] += bcnt[x];
        for (int x = 0; x <= MCF_NUM_BUCKETS; ++x) im.bucket_off[x] = bcnt[x];
        for (int64_t j = 0; j < m; ++j) {
            const int32_t i = by_tail[j];
            perm[bcnt[head[i] / per]++] = i;
        }
    } else {
        for (int64_t i = 0; i < m; ++i) perm[i] = (int32_t)i;
        for (int x = 0; x <= MCF_NUM_BUCKETS; ++x) im.bucket_off[x] = m * x / MCF_NUM_BUCKETS;
    }

    for (int64_t e = 0; e < m; ++e) {
        const int64_t i = perm[e];
        const int64_t c = cost[i];
        int64_t cp = cap[i];
        if (cp < 0 || cp >= MCF_INF) cp = MCF_INF;
        im.tail[e] = tail[i];
        im.head[e] = head[i];
        im.cost[e] = (int32_t)c;
        im.orig[e] = (int32_t)i;
        im.cost64[e] = c;
        im.state[e] = 1;  // every real arc starts non-basic at its lower bound (flow 0)
        im.arcw[e] = McfArcW{cp, 0};
        const int64_t ac = c < 0 ? -c : c;
        if (ac > max_abs_cost) max_abs_cost = ac;
    }
    // big-M: any simple path costs < (n+1) * max|c|, so an artificial arc is never cheaper
    // than a real detour (the reference uses max|c| * (n_nodes+1) as its penalty, simplex.py:163)
    im.big_m = (max_abs_cost + 1) * ((int64_t)n + 2);
    if (im.big_m >= ((int64_t)1 << 44)) { *err_code = -5; return "max|cost| * n too large for the big-M start"; }

    mcf_init_cold_basis(im);
    *err_code = 0;
    return "";
}

// Resident reduced costs for the start basis and the CSR adjacency the update kernel walks.
// rc[e] = cost + pi[tail] - pi[head] (simplex.py:508-512), kept exact for every arc from here on:
// a basis swap shifts the potentials of the re-hung subtree T2 by sigma, so exactly the arcs
// with one end point in T2 change, by +sigma (tail inside) or -sigma (head inside).
// `shard` of `shards` > 1: the adjacency lists only the arcs of that rank's shard (its 1/shards share of every head
// bucket, mcf_bucket_slice), so the rank's patch pass does 1/shards of the work and keeps exactly the reduced costs
// it sweeps; McfView::rc_partial tells the pivot code not to trust the others.
inline void mcf_build_rcache(McfHostImage& im, int64_t shard = 0, int64_t shards = 1) {
    const int64_t m = im.m;
    const int32_t n = im.n;
    im.rcache.assign(im.m_pad, 0);
    for (int64_t e = 0; e < m; ++e) im.rcache[e] = (int64_t)im.cost[e] + im.pi[im.tail[e]] - im.pi[im.head[e]];
    std::vector<int8_t> mine;
    if (shards > 1) {
        mine.assign(m, 0);
        for (int x = 0; x < MCF_NUM_BUCKETS; ++x) {
            int64_t lo, hi;
            mcf_bucket_slice(im.bucket_off, x, shard, shards, 0, 1, &lo, &hi);
            for (int64_t e = lo; e < hi; ++e) mine[e] = 1;
        }
    }
    auto own = [&](int64_t e) { return shards <= 1 || mine[e]; };
    im.adj_off.assign((size_t)n + 1, 0);
    int64_t total = 0;
    for (int64_t e = 0; e < m; ++e) if (own(e)) { im.adj_off[im.tail[e] + 1]++; im.adj_off[im.head[e] + 1]++; total += 2; }
    for (int32_t v = 0; v < n; ++v) im.adj_off[v + 1] += im.adj_off[v];
    im.adj.assign((size_t)total, 0);
    std::vector<int64_t> fill(im.adj_off.begin(), im.adj_off.end() - 1);
    for (int64_t e = 0; e < m; ++e) {
        if (!own(e)) continue;
        const int64_t t = im.tail[e], h = im.head[e];
        im.adj[fill[t]++] = (h << 32) | (e << 1) | 1;
        im.adj[fill[h]++] = (t << 32) | (e << 1);
    }
}

// Pricing grid: 8 * k workgroups (one group of k per XCD head bucket).  Shared with the CPU
// emulation because the candidate-list rule keeps ONE candidate per pricing workgroup, so the
// arc -> workgroup map ((group - first group of the slice) / 256 mod k, see k_price) is part of
// the rule's definition.
inline int mcf_price_blocks(int64_t m, int64_t shards, int requested) {
    const int64_t per_pass = 256 * 4 * 2;  // lanes x arcs per group x groups in flight (sizing only)
    int64_t pb = requested > 0 ? (requested + 7) / 8 : (m / (shards > 0 ? shards : 1) / MCF_NUM_BUCKETS + per_pass - 1) / per_pass;
    if (pb < 1) pb = 1;
    if (pb > 2048 / MCF_NUM_BUCKETS) pb = 2048 / MCF_NUM_BUCKETS;
    return (int)pb * MCF_NUM_BUCKETS;
}

// minor pivots per full sweep for the candidate-list rule: the list has one entry per workgroup
inline int mcf_minor_cap(int price_blocks) {
    int r = price_blocks / 8;
    if (r < 3) r = 3;       // the reference's minor_iterations_per_candidate (simplex_pricing.py:400)
    if (r > 32) r = 32;
    return r;
}

// ---- blocked preorder list (mcf_core.h): host image of the arenas, built densely from the preorder arrays of `im`
// (slot = position), and the way back for introspection.
struct McfBplImage {
    int32_t shift = 0, cap = 0, dense = 0;      // log2 slots per block, blocks per arena, blocks of a dense list
    std::vector<int32_t> tok[2], psz[2];        // [cap << shift (+ pad)] per arena
    std::vector<int32_t> loc;                   // [n_nodes]
    std::vector<McfBlkMeta> meta[2];            // [cap + 2]
    std::vector<int32_t> ext[2];                // [cap]
};

// Block size by tree size and pool size.  The pivot kernel reads one {base, rrel} record per block handed out (small blocks: a
// long pass) and, per ancestor of the entering arc's end points, one whole block of sizes (large blocks: a lot of bytes through
// one CU): measured at 1 M nodes, whole solve, 64 / 128 / 256 slots: 108.8 / 105.8 / 107.6 s -- up to 8 192 dense blocks; `shift` / `pool` > 0 override (tests use tiny blocks and pools; pool < 0 = no spare
// blocks at all, i.e. a dense rewrite on every pivot).
inline void mcf_bpl_geometry(int32_t n_nodes, int32_t shift, int32_t pool, int32_t* shift_out, int32_t* cap_out, int32_t* dense_out) {
    int32_t sh = shift;
    if (sh <= 0) { sh = 6; while (sh < 10 && (n_nodes >> sh) > 8192) ++sh; }
    if (sh < 2) sh = 2;
    if (sh > 10) sh = 10;
    const int32_t dense = (int32_t)(((int64_t)n_nodes + (1 << sh) - 1) >> sh);
    int32_t extra = pool > 0 ? pool : (pool < 0 ? 0 : (dense * 3 / 2 > 256 ? dense * 3 / 2 : 256));
    *shift_out = sh; *cap_out = dense + extra; *dense_out = dense;
}

inline void mcf_bpl_build(const McfHostImage& im, int32_t shift, int32_t pool, McfBplImage& bp) {
    mcf_bpl_geometry(im.n_nodes, shift, pool, &bp.shift, &bp.cap, &bp.dense);
    const int32_t B = 1 << bp.shift, N = im.n_nodes;
    const size_t slots = ((size_t)bp.cap << bp.shift) + 4;
    for (int a = 0; a < 2; ++a) {
        bp.tok[a].assign(slots, 0);
        bp.psz[a].assign(slots, 0);
        bp.meta[a].assign((size_t)bp.cap + 2, McfBlkMeta{MCF_BLK_FREE, 0});
        bp.ext[a].assign((size_t)bp.cap, 0);
    }
    bp.loc.assign((size_t)N, 0);
    for (int32_t p = 0; p < N; ++p) {
        bp.tok[0][p] = im.order[p];
        bp.psz[0][p] = im.psize[p];
        bp.loc[im.order[p]] = p;   // arena 0
    }
    for (int32_t b = 0; b < bp.dense; ++b) {
        const int32_t lo = b << bp.shift, hi = lo + B < N ? lo + B : N;
        int32_t r = 0;
        for (int32_t p = lo; p < hi; ++p) if (p - lo + im.psize[p] > r) r = p - lo + im.psize[p];
        bp.meta[0][b] = bp.meta[1][b] = McfBlkMeta{lo, r};
        bp.ext[0][b] = mcf_ext_make(0, hi - lo);
    }
}

// Logical preorder arrays from the arenas: order[p], pos[node], psize[p] (any may be null).
// Returns false when the blocks do not tile [0, n_nodes) exactly (a corrupted list).
inline bool mcf_bpl_flatten(int32_t n_nodes, int32_t shift, int32_t nblocks, const int32_t* tok, const int32_t* psz,
                            const McfBlkMeta* meta, const int32_t* ext, int32_t* order, int32_t* pos, int32_t* psize) {
    std::vector<int8_t> seen((size_t)n_nodes, 0);
    int64_t count = 0;
    for (int32_t b = 0; b < nblocks; ++b) {
        if (meta[b].base == MCF_BLK_FREE) continue;
        const int32_t beg = mcf_ext_beg(ext[b]), end = mcf_ext_end(ext[b]);
        for (int32_t o = beg; o < end; ++o) {
            const int32_t p = meta[b].base + o, slot = (b << shift) + o;
            if (p < 0 || p >= n_nodes || seen[p]) return false;
            seen[p] = 1;
            ++count;
            if (order) order[p] = tok[slot];
            if (pos) pos[tok[slot]] = p;
            if (psize) psize[p] = psz[slot];
        }
    }
    return count == n_nodes;
}

struct McfHostResult {
    int32_t status = 0;        // MCF_ST_* numbering of include/mcf.h
    __int128 objective = 0;
    int64_t artificial_flow = 0;
};

// simplex.py:1573-1624 (infeasible when an artificial arc still carries flow) and
// simplex.py:1703-1728 (objective over the original costs).
inline void mcf_extract(const McfHostImage& im, const std::vector<McfArcW>& arcw, int32_t core_status,
                        McfHostResult& r) {
    r.artificial_flow = 0;
    for (int32_t v = 0; v < im.n; ++v) r.artificial_flow += arcw[im.m + v].flow;
    r.objective = 0;
    for (int64_t i = 0; i < im.m; ++i) r.objective += (__int128)arcw[i].flow * (__int128)im.cost64[i];
    if (core_status == MCF_UNBOUNDED) r.status = 3;
    else if (core_status == MCF_OPTIMAL) r.status = r.artificial_flow > 0 ? 1 : 0;
    else r.status = 2;
}
