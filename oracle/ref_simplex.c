/*
 * oracle/ref_simplex.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU restatement of the reference's network-simplex hot path
 * (jeffreyhorn/network_flow_solver, /root/reference/src/network_solver/).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the shipped HIP engine never calls into it.
 *
 * What is restated, and where it lives in the reference:
 *   arc build + lower-bound shift ........ simplex.py:392-432   -> build_arcs()
 *   penalty cost .......................... simplex.py:161-163   -> ref_solve()
 *   cost perturbation ..................... simplex.py:36-37,1431-1440 -> apply_cost_perturbation()
 *   artificial-root initial tree .......... simplex.py:619-730   -> initialize_tree()
 *   phase costs ........................... simplex.py:1162-1170 -> apply_phase_costs()
 *   tree rebuild (BFS parent/potential) ... basis.py:82-122      -> rebuild()
 *   cycle collection ...................... basis.py:178-241     -> collect_cycle()
 *   Dantzig pricing ....................... simplex_pricing.py:97-137  -> select_dantzig()
 *   vectorised block pricing .............. simplex.py:528-617   -> select_block_vectorized()
 *   Devex block search + deferred weights . simplex_pricing.py:271-292,310-372 -> select_devex()
 *   candidate list / adaptive pricing ..... simplex_pricing.py:375-639 -> select_candidate_list(), select_adaptive()
 *   block-size tuner ...................... simplex_adaptive.py:70-151 -> tuner_*()
 *   specialised entering rules ............ specialized_pivots.py:69-424, dispatch simplex.py:1061-1064 -> select_special()
 *   pivot (ratio test, flow update, swap) . simplex.py:1176-1425 -> pivot()
 *   pivot loop ............................ simplex.py:1109-1160 -> run_iterations()
 *   two-phase driver + result extraction .. simplex.py:1446-1765 -> ref_solve()
 *
 * Arithmetic is IEEE double throughout, in the reference's operation order,
 * so Dantzig pivots follow the reference's pivots one for one.
 *
 * Deliberate departures (the dense-basis machinery the build drops, SURVEY.md
 * section 2 rows 3-4): the Devex weight ||B^-1 a||^2 is taken as the exact
 * tree-path length between the arc's end points (what the reference's LU/
 * Forrest-Tomlin solve returns up to rounding); Forrest-Tomlin updates on a
 * tree basis always succeed in exact arithmetic, so only the "64 updates, then
 * refactorise + Devex reset" cadence (simplex.py:1370-1400) is kept; the
 * condition-number trigger (bound 2n << 1e12 for an incidence basis) never
 * fires.  The O(n*m) post-phase-1 conservation audit (simplex.py:1581-1598) is
 * restated in O(n+m).
 *
 * Node 0 is the artificial root; real nodes are 1..n (the caller has already
 * applied the reference's string-sorted node and arc order,
 * simplex.py:149,395).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PERTURB_EPS_BASE 1e-10 /* simplex.py:36 */
#define PERTURB_GROWTH 1.00001 /* simplex.py:37 */
#define DEVEX_WEIGHT_MIN 1e-12 /* simplex_pricing.py:45 */
#define DEVEX_WEIGHT_MAX 1e12  /* simplex_pricing.py:46 */

enum { ST_OPTIMAL = 0, ST_INFEASIBLE = 1, ST_ITERATION_LIMIT = 2, ST_UNBOUNDED = 3 };
enum { PR_DANTZIG = 0, PR_DEVEX = 1, PR_CANDIDATE_LIST = 2, PR_ADAPTIVE = 3 };
/* NetworkType values that install a specialised pivot strategy (specialized_pivots.py:452-527) */
enum { SP_NONE = 0, SP_TRANSPORTATION = 1, SP_ASSIGNMENT = 2, SP_BIPARTITE_MATCHING = 3, SP_MAX_FLOW = 4, SP_SHORTEST_PATH = 5 };

typedef struct {
    /* sizes */
    int n_nodes;     /* incl. root */
    int64_t m_real;  /* actual_arc_count */
    int64_t m_tot;   /* after artificial arcs were appended */
    double tol;
    /* ArcState SoA (simplex.py:40-59) */
    int32_t *tail, *head;
    double *cost, *upper, *flow, *shift;
    uint8_t *in_tree, *artificial;
    double *original_cost, *perturbed_cost;
    double *fwd_res, *bwd_res;     /* cached residuals, simplex.py:454-456 */
    double *vec_cost;              /* numpy arc_costs mirror, simplex.py:440 (never refreshed by phase costs) */
    double *supply;                /* node_supply after lower-bound adjustment */
    double penalty;
    /* TreeBasis (basis.py:40-44) */
    int32_t *parent, *parent_arc, *depth;
    int8_t *parent_dir;
    double *potential;
    /* tree bookkeeping for rebuild: list of tree arcs + CSR scratch */
    int32_t *tree_arcs;            /* n_nodes-1 entries */
    int32_t *tree_slot;            /* arc -> slot in tree_arcs, or -1 */
    int32_t *adj_off, *adj_arc, *queue;
    int32_t *cyc_arc;              /* cycle scratch */
    int8_t *cyc_sign;
    int32_t *tmp_nodes;
    /* counters */
    int64_t artificial_with_flow, degenerate_pivots, ft_updates_since_rebuild;
    int64_t arcs_priced;
    /* pricing state */
    int strategy;
    int use_vectorized;
    double *weights;               /* DevexPricing.weights */
    int64_t block_size, pricing_block, last_degenerate_arc;
    /* tuner (simplex_adaptive.py) */
    int auto_tune;
    int64_t tn_last_adapt, tn_degenerate, tn_total;
    /* candidate list (simplex_pricing.py:375-542) */
    int32_t *cand;
    int cand_len, cand_since_refresh, cand_minor;
    double *cand_merit;
    int32_t *cand_tmp;
    /* adaptive (simplex_pricing.py:545-639) */
    int ad_current, ad_failed;
    /* specialised pivot strategy (tried before the pricing strategy, simplex.py:1061-1064) */
    int special;
    const uint8_t *left_part;      /* bipartite matching: node is in the left partition (colour 0) */
    int sp_source;                 /* shortest path: the unit-supply node */
    uint8_t *sp_label;             /* shortest path: node has a finite distance label */
    int sp_label_init;
    /* unbounded diagnostics */
    int64_t unb_arc;
} Ref;

/* ---- simplex.py:1162-1170 ------------------------------------------------ */
static void apply_phase_costs(Ref *s, int phase) {
    if (phase == 1) {
        for (int64_t i = 0; i < s->m_real; ++i) s->cost[i] = s->perturbed_cost[i] - 1.0 - 1e-6 * (double)i;
    } else {
        for (int64_t i = 0; i < s->m_tot; ++i) s->cost[i] = s->perturbed_cost[i];
    }
}

/* ---- simplex.py:1431-1440 ------------------------------------------------ */
static void apply_cost_perturbation(Ref *s) {
    double factor = 1.0;
    for (int64_t i = 0; i < s->m_real; ++i) {
        double perturb = PERTURB_EPS_BASE * factor;
        s->perturbed_cost[i] = s->original_cost[i] + perturb;
        s->cost[i] = s->perturbed_cost[i];
        factor *= PERTURB_GROWTH;
    }
}

/* ---- basis.py:82-122: BFS from the root over the tree arcs ----------------
 * The reference rebuilds tree_adj by scanning every arc (simplex.py:1103-1107)
 * and BFSes; parent/potential depend only on the tree, not on the visiting
 * order, so the adjacency is built here from the tree-arc list instead. */
static int rebuild(Ref *s) {
    const int n = s->n_nodes;
    memset(s->adj_off, 0, sizeof(int32_t) * (size_t)(n + 1));
    for (int k = 0; k < n - 1; ++k) {
        int32_t a = s->tree_arcs[k];
        s->adj_off[s->tail[a] + 1]++;
        s->adj_off[s->head[a] + 1]++;
    }
    for (int v = 0; v < n; ++v) s->adj_off[v + 1] += s->adj_off[v];
    int32_t *fill = s->tmp_nodes;
    memcpy(fill, s->adj_off, sizeof(int32_t) * (size_t)n);
    for (int k = 0; k < n - 1; ++k) {
        int32_t a = s->tree_arcs[k];
        s->adj_arc[fill[s->tail[a]]++] = a;
        s->adj_arc[fill[s->head[a]]++] = a;
    }
    for (int v = 0; v < n; ++v) { s->parent[v] = -1; s->parent_arc[v] = -1; s->parent_dir[v] = 0; }
    s->parent[0] = 0;
    s->potential[0] = 0.0;
    int qh = 0, qt = 0, visited = 1;
    s->queue[qt++] = 0;
    while (qh < qt) {
        int node = s->queue[qh++];
        for (int32_t p = s->adj_off[node]; p < s->adj_off[node + 1]; ++p) {
            int32_t a = s->adj_arc[p];
            int nb = (s->tail[a] == node) ? s->head[a] : s->tail[a];
            if (s->parent[nb] != -1) continue;
            s->parent[nb] = node;
            s->parent_arc[nb] = a;
            if (s->tail[a] == node && s->head[a] == nb) {
                s->parent_dir[nb] = 1;
                s->potential[nb] = s->potential[node] + s->cost[a];
            } else {
                s->parent_dir[nb] = -1;
                s->potential[nb] = s->potential[node] - s->cost[a];
            }
            s->depth[nb] = s->depth[node] + 1;
            s->queue[qt++] = nb;
            ++visited;
        }
    }
    return visited == n ? 0 : -1;
}

/* ---- basis.py:178-241: path head -> tail in the tree, as (arc, sign) ------
 * sign = +1 when the arc points along the walk (parent->child in the
 * reference's BFS-from-head sense).  Uses parent/depth instead of a BFS. */
static int collect_cycle(Ref *s, int tail, int head) {
    if (tail == head) return 0;
    int len = 0, nt = 0;
    int u = head, v = tail;
    /* climb from `head`; walking child -> parent over arc with parent_dir d
     * traverses it against its tree orientation: sign = -d */
    while (s->depth[u] > s->depth[v]) {
        s->cyc_arc[len] = s->parent_arc[u]; s->cyc_sign[len] = (int8_t)(-s->parent_dir[u]); ++len;
        u = s->parent[u];
    }
    while (s->depth[v] > s->depth[u]) { s->tmp_nodes[nt++] = v; v = s->parent[v]; }
    while (u != v) {
        s->cyc_arc[len] = s->parent_arc[u]; s->cyc_sign[len] = (int8_t)(-s->parent_dir[u]); ++len;
        u = s->parent[u];
        s->tmp_nodes[nt++] = v; v = s->parent[v];
    }
    /* descend towards `tail`: parent -> child, sign = +d */
    for (int k = nt - 1; k >= 0; --k) {
        int c = s->tmp_nodes[k];
        s->cyc_arc[len] = s->parent_arc[c]; s->cyc_sign[len] = s->parent_dir[c]; ++len;
    }
    return len;
}

/* exact tree-path length = ||B^-1 a||^2 */
static int path_length(const Ref *s, int a, int b) {
    int len = 0;
    while (s->depth[a] > s->depth[b]) { a = s->parent[a]; ++len; }
    while (s->depth[b] > s->depth[a]) { b = s->parent[b]; ++len; }
    while (a != b) { a = s->parent[a]; b = s->parent[b]; len += 2; }
    return len;
}

/* ---- simplex_pricing.py:97-137 ------------------------------------------- */
static int select_dantzig_list(Ref *s, const int32_t *list, int64_t count, int allow_zero,
                               int64_t *out_arc, int *out_dir) {
    const double tol = s->tol;
    int have = 0;
    double best_rc = 0.0;
    for (int64_t k = 0; k < count; ++k) {
        int64_t idx = list ? list[k] : k;
        if (list && idx >= s->m_tot) continue;
        if (s->in_tree[idx] || s->artificial[idx]) continue;
        double rc = s->cost[idx] + s->potential[s->tail[idx]] - s->potential[s->head[idx]];
        double fr = s->fwd_res[idx], br = s->bwd_res[idx];
        if (fr > tol && rc < -tol) {
            if (!have || rc < best_rc) { have = 1; *out_arc = idx; *out_dir = 1; best_rc = rc; }
        } else if (br > tol && rc > tol) {
            if (!have || -rc < best_rc) { have = 1; *out_arc = idx; *out_dir = -1; best_rc = -rc; }
        } else if (allow_zero && fr > tol && fabs(rc) <= tol && !have) {
            have = 1; *out_arc = idx; *out_dir = 1;
        } else if (allow_zero && br > tol && fabs(rc) <= tol && !have) {
            have = 1; *out_arc = idx; *out_dir = -1;
        }
    }
    s->arcs_priced += count;
    return have;
}

/* ---- simplex.py:528-617 ---------------------------------------------------
 * rc uses the numpy arc_costs mirror (vec_cost), which _apply_phase_costs
 * never refreshes: kept as is. */
static int select_block_vectorized(Ref *s, int64_t start, int64_t end, int allow_zero, int64_t excluded,
                                   int64_t *out_arc, int *out_dir, double *out_merit) {
    if (start >= end) return 0;
    const double tol = s->tol;
    /* the reference computes rc for every arc, then slices (simplex.py:555) */
    s->arcs_priced += s->m_tot;
    int any_eligible = 0;
    double bf = -INFINITY, bb = -INFINITY;
    int64_t bfi = start, bbi = start;
    for (int64_t i = start; i < end; ++i) {
        if (s->in_tree[i] || s->artificial[i] || i == excluded) continue;
        any_eligible = 1;
        double rc = s->vec_cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
        if (s->fwd_res[i] > tol && rc < -tol) {
            double merit = (rc * rc) / s->weights[i];
            if (merit > bf) { bf = merit; bfi = i; } /* np.argmax: first maximum */
        }
        if (s->bwd_res[i] > tol && rc > tol) {
            double merit = (rc * rc) / s->weights[i];
            if (merit > bb) { bb = merit; bbi = i; }
        }
    }
    if (!any_eligible) return 0;
    if (bf > bb) {
        if (bf > -INFINITY) { *out_arc = bfi; *out_dir = 1; *out_merit = bf; return 1; }
    } else {
        if (bb > -INFINITY) { *out_arc = bbi; *out_dir = -1; *out_merit = bb; return 1; }
    }
    if (allow_zero) {
        for (int64_t i = start; i < end; ++i) {
            if (s->in_tree[i] || s->artificial[i] || i == excluded) continue;
            double rc = s->vec_cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
            if (s->fwd_res[i] > tol && fabs(rc) <= tol) { *out_arc = i; *out_dir = 1; *out_merit = 0.0; return 1; }
        }
        for (int64_t i = start; i < end; ++i) {
            if (s->in_tree[i] || s->artificial[i] || i == excluded) continue;
            double rc = s->vec_cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
            if (s->bwd_res[i] > tol && fabs(rc) <= tol) { *out_arc = i; *out_dir = -1; *out_merit = 0.0; return 1; }
        }
    }
    return 0;
}

/* ---- simplex_pricing.py:271-292 (weight of the selected arc only) --------- */
static void devex_update_weight(Ref *s, int64_t idx) {
    double w = (double)path_length(s, s->tail[idx], s->head[idx]);
    if (!isfinite(w) || w <= DEVEX_WEIGHT_MIN) w = DEVEX_WEIGHT_MIN;
    else if (w > DEVEX_WEIGHT_MAX) w = DEVEX_WEIGHT_MAX;
    s->weights[idx] = w;
}

/* ---- simplex_pricing.py:294-308 ------------------------------------------ */
static int devex_is_better(double merit, int64_t idx, double best_merit, int have, int64_t best_idx, double tol) {
    int better = merit > best_merit + tol;
    int tie = !better && fabs(merit - best_merit) <= tol;
    return better || (tie && (!have || idx < best_idx));
}

/* ---- simplex_pricing.py:187-269 (loop) and 310-357 (vectorised) ----------- */
static int select_devex(Ref *s, int allow_zero, int64_t *out_arc, int *out_dir) {
    const double tol = s->tol;
    int64_t bs = s->block_size;
    int64_t block_count = (s->m_real + bs - 1) / bs;
    if (block_count < 1) block_count = 1;
    for (int64_t b = 0; b < block_count; ++b) {
        int64_t start = s->pricing_block * bs;
        if (start >= s->m_real) { s->pricing_block = 0; start = 0; }
        int64_t end = start + bs < s->m_real ? start + bs : s->m_real;
        if (s->use_vectorized) {
            double merit = 0.0;
            if (select_block_vectorized(s, start, end, allow_zero, s->last_degenerate_arc, out_arc, out_dir, &merit)) {
                s->last_degenerate_arc = -1;
                if (merit > 0) devex_update_weight(s, *out_arc);
                return 1;
            }
        } else {
            int have = 0, have_zero = 0;
            double best_merit = -INFINITY;
            int64_t best = -1, zero_arc = -1;
            int best_dir = 0, zero_dir = 0;
            s->arcs_priced += end - start;
            for (int64_t i = start; i < end; ++i) {
                if (s->in_tree[i] || s->artificial[i]) continue;
                double rc = s->cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
                double fr = s->fwd_res[i], br = s->bwd_res[i];
                if (fr > tol && rc < -tol) {
                    double w = s->weights[i] > DEVEX_WEIGHT_MIN ? s->weights[i] : DEVEX_WEIGHT_MIN;
                    double merit = (rc * rc) / w;
                    if (devex_is_better(merit, i, best_merit, have, best, tol)) { best_merit = merit; best = i; best_dir = 1; have = 1; }
                    continue;
                }
                if (br > tol && rc > tol) {
                    double w = s->weights[i] > DEVEX_WEIGHT_MIN ? s->weights[i] : DEVEX_WEIGHT_MIN;
                    double merit = (rc * rc) / w;
                    if (devex_is_better(merit, i, best_merit, have, best, tol)) { best_merit = merit; best = i; best_dir = -1; have = 1; }
                    continue;
                }
                if (allow_zero && !have_zero && fr > tol && fabs(rc) <= tol) { have_zero = 1; zero_arc = i; zero_dir = 1; }
                else if (allow_zero && !have_zero && br > tol && fabs(rc) <= tol) { have_zero = 1; zero_arc = i; zero_dir = -1; }
            }
            if (have) { devex_update_weight(s, best); *out_arc = best; *out_dir = best_dir; return 1; }
            if (allow_zero && have_zero) {
                s->pricing_block = (s->pricing_block + 1) % block_count;
                *out_arc = zero_arc; *out_dir = zero_dir; return 1;
            }
        }
        s->pricing_block = (s->pricing_block + 1) % block_count;
    }
    return 0;
}

/* ---- simplex_pricing.py:502-536: full scan, keep the 100 largest |rc| ------
 * Python sorts (merit, idx) tuples descending: ties -> larger index first. */
static int cand_cmp(const void *pa, const void *pb, void *ctx) {
    const double *merit = (const double *)ctx;
    int32_t a = *(const int32_t *)pa, b = *(const int32_t *)pb;
    if (merit[a] > merit[b]) return -1;
    if (merit[a] < merit[b]) return 1;
    return (a > b) ? -1 : (a < b);
}
static void refresh_candidate_list(Ref *s) {
    const double tol = s->tol;
    int64_t cnt = 0;
    for (int64_t i = 0; i < s->m_real; ++i) {
        if (s->in_tree[i] || s->artificial[i]) continue;
        double rc = s->cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
        double merit = 0.0;
        if ((s->fwd_res[i] > tol && rc < -tol) || (s->bwd_res[i] > tol && rc > tol)) merit = fabs(rc);
        if (merit > tol) { s->cand_merit[i] = merit; s->cand_tmp[cnt++] = (int32_t)i; }
    }
    s->arcs_priced += s->m_real;
    qsort_r(s->cand_tmp, (size_t)cnt, sizeof(int32_t), cand_cmp, s->cand_merit);
    s->cand_len = cnt < 100 ? (int)cnt : 100;
    memcpy(s->cand, s->cand_tmp, sizeof(int32_t) * (size_t)s->cand_len);
}

/* ---- simplex_pricing.py:419-456 ------------------------------------------ */
static int select_candidate_list(Ref *s, int allow_zero, int64_t *out_arc, int *out_dir) {
    if (s->cand_len > 0 && s->cand_minor < 3) {
        if (select_dantzig_list(s, s->cand, s->cand_len, allow_zero, out_arc, out_dir)) { s->cand_minor++; return 1; }
    }
    s->cand_since_refresh++;
    s->cand_minor = 0;
    if (s->cand_since_refresh >= 10 || s->cand_len == 0) { refresh_candidate_list(s); s->cand_since_refresh = 0; }
    if (select_dantzig_list(s, s->cand, s->cand_len, allow_zero, out_arc, out_dir)) return 1;
    if (s->cand_since_refresh > 0) {
        refresh_candidate_list(s);
        s->cand_since_refresh = 0;
        return select_dantzig_list(s, s->cand, s->cand_len, allow_zero, out_arc, out_dir);
    }
    return 0;
}

static void pricing_reset(Ref *s, int which) { /* PricingStrategy.reset() */
    if (which == PR_DEVEX) {
        for (int64_t i = 0; i < s->m_tot; ++i) s->weights[i] = 1.0;
        s->pricing_block = 0;
    } else if (which == PR_CANDIDATE_LIST) {
        s->cand_len = 0; s->cand_since_refresh = 0; s->cand_minor = 0;
    }
}

/* ---- simplex_pricing.py:589-631 ------------------------------------------ */
static int select_adaptive(Ref *s, int allow_zero, int64_t *out_arc, int *out_dir) {
    int found;
    if (s->ad_current == PR_CANDIDATE_LIST) found = select_candidate_list(s, allow_zero, out_arc, out_dir);
    else if (s->ad_current == PR_DEVEX) found = select_devex(s, allow_zero, out_arc, out_dir);
    else found = select_dantzig_list(s, NULL, s->m_real, allow_zero, out_arc, out_dir);
    if (!found) s->ad_failed++; else s->ad_failed = 0;
    if (s->ad_failed >= 5) {
        if (s->ad_current == PR_CANDIDATE_LIST) s->ad_current = PR_DEVEX;
        else if (s->ad_current == PR_DEVEX) s->ad_current = PR_DANTZIG;
        else { s->ad_current = PR_CANDIDATE_LIST; pricing_reset(s, PR_CANDIDATE_LIST); }
        s->ad_failed = 0;
    }
    return found;
}


/* ---- specialized_pivots.py: every strategy is one O(m_tot) scan of rc = cost + pi[tail] - pi[head] with its own
 * acceptance / merit test; a strategy that finds nothing hands over to the pricing strategy.
 *   TransportationPivotStrategy.find_entering_arc_row_scan   :69-117   most negative rc, both directions
 *   AssignmentPivotStrategy.find_entering_arc_min_cost       :191-223  forward only, better by more than tol
 *   BipartiteMatchingPivotStrategy.find_augmenting_path      :251-287  first arc out of an unmatched left node
 *   MaxFlowPivotStrategy.find_entering_arc                   :312-355  merit residual * |rc|
 *   ShortestPathPivotStrategy.find_entering_arc              :383-448  forward arcs only from labelled nodes */
static int select_special(Ref *s, int64_t *out_arc, int *out_dir) {
    const double tol = s->tol;
    int have = 0;
    s->arcs_priced += s->m_tot;
    if (s->special == SP_TRANSPORTATION) {
        double best_rc = 0.0;
        for (int64_t i = 0; i < s->m_tot; ++i) {
            if (s->in_tree[i] || s->artificial[i]) continue;
            double rc = s->cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
            if (s->fwd_res[i] > tol && rc < -tol && rc < best_rc) { best_rc = rc; *out_arc = i; *out_dir = 1; have = 1; }
            if (s->bwd_res[i] > tol && rc > tol && -rc < best_rc) { best_rc = -rc; *out_arc = i; *out_dir = -1; have = 1; }
        }
        return have;
    }
    if (s->special == SP_ASSIGNMENT) {
        double best_rc = 0.0;
        for (int64_t i = 0; i < s->m_tot; ++i) {
            if (s->in_tree[i] || s->artificial[i]) continue;
            double rc = s->cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
            if (s->fwd_res[i] > tol && rc < best_rc - tol) { best_rc = rc; *out_arc = i; *out_dir = 1; have = 1; }
        }
        return have;
    }
    if (s->special == SP_MAX_FLOW) {
        double best = -INFINITY;
        for (int64_t i = 0; i < s->m_tot; ++i) {
            if (s->in_tree[i] || s->artificial[i]) continue;
            double rc = s->cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
            if (s->fwd_res[i] > tol && rc < -tol) {
                double merit = s->fwd_res[i] * fabs(rc);
                if (merit > best) { best = merit; *out_arc = i; *out_dir = 1; have = 1; }
            }
            if (s->bwd_res[i] > tol && rc > tol) {
                double merit = s->bwd_res[i] * fabs(rc);
                if (merit > best) { best = merit; *out_arc = i; *out_dir = -1; have = 1; }
            }
        }
        return have;
    }
    if (s->special == SP_SHORTEST_PATH) {
        if (!s->sp_label_init) { /* _initialize_distance_labels :450-471: BFS over the real arcs from the source */
            s->sp_label_init = 1;
            int qh = 0, qt = 0;
            s->sp_label[s->sp_source] = 1; s->queue[qt++] = s->sp_source;
            while (qh < qt) {
                int node = s->queue[qh++];
                for (int64_t i = 0; i < s->m_tot; ++i) {
                    if (s->artificial[i]) continue;
                    if (s->tail[i] == node && !s->sp_label[s->head[i]]) { s->sp_label[s->head[i]] = 1; s->queue[qt++] = s->head[i]; }
                }
            }
        }
        double best_rc = 0.0;
        for (int64_t i = 0; i < s->m_tot; ++i) {
            if (s->in_tree[i] || s->artificial[i]) continue;
            double rc = s->cost[i] + s->potential[s->tail[i]] - s->potential[s->head[i]];
            if (s->fwd_res[i] > tol && rc < -tol) {
                /* only the finiteness of a label is ever tested: labelled = reachable from the source */
                if (s->sp_label[s->tail[i]] && rc < best_rc - tol) {
                    best_rc = rc; *out_arc = i; *out_dir = 1; have = 1;
                    s->sp_label[s->head[i]] = 1;
                }
            }
            if (s->bwd_res[i] > tol && rc > tol && -rc < best_rc - tol) { best_rc = -rc; *out_arc = i; *out_dir = -1; have = 1; }
        }
        return have;
    }
    if (s->special == SP_BIPARTITE_MATCHING) {
        /* unmatched left nodes: unit supply and no basic arc carrying flow out of them.  The reference iterates Python
         * sets of node indices; small ints iterate in increasing order, which is what is restated here. */
        for (int v = 1; v < s->n_nodes; ++v) {
            if (!s->left_part[v] || fabs(s->supply[v] - 1.0) > tol) continue;
            int has_flow = 0;
            for (int64_t i = 0; i < s->m_tot && !has_flow; ++i)
                if (s->in_tree[i] && s->flow[i] > tol && s->tail[i] == v) has_flow = 1;
            if (has_flow) continue;
            for (int64_t i = 0; i < s->m_tot; ++i)
                if (s->tail[i] == v && !s->in_tree[i] && s->fwd_res[i] > tol) { *out_arc = i; *out_dir = 1; return 1; }
        }
        return 0;
    }
    return 0;
}

/* ---- simplex.py:1058-1075 ------------------------------------------------ */
static int find_entering_arc(Ref *s, int allow_zero, int64_t *out_arc, int *out_dir) {
    if (s->special != SP_NONE && select_special(s, out_arc, out_dir)) return 1;
    switch (s->strategy) {
    case PR_DANTZIG: return select_dantzig_list(s, NULL, s->m_real, allow_zero, out_arc, out_dir);
    case PR_DEVEX: return select_devex(s, allow_zero, out_arc, out_dir);
    case PR_CANDIDATE_LIST: return select_candidate_list(s, allow_zero, out_arc, out_dir);
    default: return select_adaptive(s, allow_zero, out_arc, out_dir);
    }
}

/* ---- simplex.py:1763-1769: _reset_devex_weights -> pricing_strategy.reset() */
static void reset_devex_weights(Ref *s) {
    if (s->strategy == PR_DEVEX) pricing_reset(s, PR_DEVEX);
    else if (s->strategy == PR_CANDIDATE_LIST) pricing_reset(s, PR_CANDIDATE_LIST);
    else if (s->strategy == PR_ADAPTIVE) { /* AdaptivePricing.reset(), simplex_pricing.py:633-639 */
        pricing_reset(s, PR_DEVEX); pricing_reset(s, PR_CANDIDATE_LIST);
        s->ad_current = PR_CANDIDATE_LIST; s->ad_failed = 0;
    }
}

/* ---- simplex_adaptive.py:98-151 ------------------------------------------ */
static void tuner_adapt(Ref *s, int64_t iteration) {
    if (!s->auto_tune) return;
    if (iteration - s->tn_last_adapt < 50) return;
    if (s->tn_total < 10) return;
    double ratio = (double)s->tn_degenerate / (double)s->tn_total;
    if (ratio > 0.30) {
        int64_t nb = (int64_t)((double)s->block_size * 1.5);
        s->block_size = nb < s->m_real ? nb : s->m_real;
    } else if (ratio < 0.10) {
        int64_t nb = (int64_t)((double)s->block_size * 0.75);
        s->block_size = nb > 10 ? nb : 10;
    }
    s->tn_degenerate = 0; s->tn_total = 0; s->tn_last_adapt = iteration;
}

static void set_flow(Ref *s, int64_t idx, double f) {
    s->flow[idx] = f;
    s->fwd_res[idx] = isinf(s->upper[idx]) ? INFINITY : s->upper[idx] - f;
    s->bwd_res[idx] = f - 0.0;
}

/* ---- simplex.py:1176-1425 ------------------------------------------------- */
static int pivot(Ref *s, int64_t arc_idx, int direction) {
    const double tol = s->tol;
    int tail = direction == 1 ? s->tail[arc_idx] : s->head[arc_idx];
    int head = direction == 1 ? s->head[arc_idx] : s->tail[arc_idx];
    int len = collect_cycle(s, tail, head);
    s->cyc_arc[len] = (int32_t)arc_idx; s->cyc_sign[len] = (int8_t)direction; ++len;

    double theta = INFINITY, best_residual = -INFINITY;
    int64_t leaving = arc_idx;
    for (int k = 0; k < len; ++k) { /* simplex.py:1201-1229 */
        int64_t idx = s->cyc_arc[k];
        double residual = s->cyc_sign[k] == 1 ? s->fwd_res[idx] : s->bwd_res[idx];
        if (residual < theta - tol) {
            theta = residual; leaving = idx; best_residual = residual;
        } else if (fabs(residual - theta) <= tol) {
            if (residual > best_residual + tol || (fabs(residual - best_residual) <= tol && idx < leaving)) {
                leaving = idx; best_residual = residual;
            }
        }
    }
    if (isinf(theta)) { s->unb_arc = arc_idx; return -1; } /* simplex.py:1231-1246 */
    if (theta < 0.0) theta = 0.0;
    if (theta <= tol) s->degenerate_pivots++;

    for (int k = 0; k < len; ++k) { /* simplex.py:1255-1283 */
        int64_t idx = s->cyc_arc[k];
        double old = s->flow[idx];
        int had = s->artificial[idx] && old > tol;
        double f = old + (double)s->cyc_sign[k] * theta;
        if (f < 0.0 - tol) f = 0.0;
        if (!isinf(s->upper[idx]) && f > s->upper[idx] + tol) f = s->upper[idx];
        if (s->artificial[idx]) {
            int has = f > tol;
            if (had && !has) s->artificial_with_flow--;
            else if (!had && has) s->artificial_with_flow++;
        }
        set_flow(s, idx, f);
    }
    s->in_tree[arc_idx] = 1;
    int is_degenerate = (leaving == arc_idx) || (fabs(theta) < tol);
    s->tn_total++; if (is_degenerate) s->tn_degenerate++; /* record_pivot */

    if (leaving == arc_idx) { /* simplex.py:1320-1334 */
        s->in_tree[arc_idx] = 0;
        if (s->use_vectorized && s->strategy == PR_DEVEX) s->last_degenerate_arc = arc_idx;
        return 0;
    }
    s->in_tree[leaving] = 0;
    int slot = s->tree_slot[leaving];
    s->tree_slot[leaving] = -1;
    s->tree_arcs[slot] = (int32_t)arc_idx;
    s->tree_slot[arc_idx] = slot;

    int force_rebuild = s->ft_updates_since_rebuild >= 64; /* simplex.py:1370-1373, ft_update_limit=64 */
    if (rebuild(s) != 0) return -2;
    if (force_rebuild) { s->ft_updates_since_rebuild = 0; reset_devex_weights(s); }
    else s->ft_updates_since_rebuild++;
    return 0;
}

/* ---- simplex.py:1109-1160 ------------------------------------------------ */
static int64_t run_iterations(Ref *s, int64_t max_iterations, int allow_zero, int phase_one, int64_t offset, int *err) {
    int64_t iterations = 0;
    while (iterations < max_iterations) {
        int64_t arc; int dir;
        if (!find_entering_arc(s, allow_zero, &arc, &dir)) break;
        int rc = pivot(s, arc, dir);
        if (rc != 0) { *err = rc; return iterations; }
        iterations++;
        tuner_adapt(s, offset + iterations);
        if (phase_one && s->artificial_with_flow == 0) break;
    }
    return iterations;
}

static void *xcalloc(size_t n, size_t sz) { return calloc(n ? n : 1, sz); }

static void ref_free(Ref *s) {
    free(s->sp_label);
    free(s->tail); free(s->head); free(s->cost); free(s->upper); free(s->flow); free(s->shift);
    free(s->in_tree); free(s->artificial); free(s->original_cost); free(s->perturbed_cost);
    free(s->fwd_res); free(s->bwd_res); free(s->vec_cost); free(s->supply);
    free(s->parent); free(s->parent_arc); free(s->depth); free(s->parent_dir); free(s->potential);
    free(s->tree_arcs); free(s->tree_slot); free(s->adj_off); free(s->adj_arc); free(s->queue);
    free(s->cyc_arc); free(s->cyc_sign); free(s->tmp_nodes); free(s->weights);
    free(s->cand); free(s->cand_merit); free(s->cand_tmp);
}

/*
 * Solve one instance the way NetworkSimplex(problem, options).solve() does.
 *
 * Inputs use reference-internal numbering: nodes 1..n (0 = root); arcs in the
 * reference's sorted order.  cap[i] = +inf for "capacity None".
 * strategy: 0 dantzig, 1 devex, 2 candidate_list, 3 adaptive.
 * block_size <= 0 means "auto" (simplex_adaptive.py:70-96).
 * max_iterations < 0 means the default max(100, 20*len(arcs)) (simplex.py:1470).
 * pivot_budget >= 0 stops after that many pivots in total and reports what was
 * done so far (used by bench.py's bounded cpu_baseline sample).
 *
 * Outputs: flow_out[m] = flow + shift per input arc; potential_out[n+1];
 * stats[0]=iterations, [1]=degenerate pivots, [2]=arcs priced, [3]=unbounded arc.
 * Returns 0, or <0 on internal error.
 */
int ref_solve(int n, int64_t m, const int32_t *tail, const int32_t *head, const double *cost,
              const double *cap, const double *lower, const double *supply_in, double tol,
              int strategy, int use_vectorized, int64_t block_size, int64_t max_iterations,
              int64_t pivot_budget, int special, const uint8_t *left_part /* [n+1] or NULL */,
              int *status_out, double *objective_out, double *flow_out, double *potential_out,
              uint8_t *in_tree_out, int64_t *stats) {
    Ref S; memset(&S, 0, sizeof S);
    Ref *s = &S;
    const int N = n + 1;
    const int64_t M = m + n; /* one artificial arc per real node */
    s->n_nodes = N; s->m_real = m; s->m_tot = M; s->tol = tol;
    s->strategy = strategy; s->use_vectorized = use_vectorized; s->last_degenerate_arc = -1; s->unb_arc = -1;
    s->special = special; s->left_part = left_part; s->sp_label = xcalloc((size_t)N, 1);
    s->tail = xcalloc((size_t)M, 4); s->head = xcalloc((size_t)M, 4);
    s->cost = xcalloc((size_t)M, 8); s->upper = xcalloc((size_t)M, 8); s->flow = xcalloc((size_t)M, 8);
    s->shift = xcalloc((size_t)M, 8); s->in_tree = xcalloc((size_t)M, 1); s->artificial = xcalloc((size_t)M, 1);
    s->original_cost = xcalloc((size_t)M, 8); s->perturbed_cost = xcalloc((size_t)M, 8);
    s->fwd_res = xcalloc((size_t)M, 8); s->bwd_res = xcalloc((size_t)M, 8); s->vec_cost = xcalloc((size_t)M, 8);
    s->supply = xcalloc((size_t)N, 8);
    s->parent = xcalloc((size_t)N, 4); s->parent_arc = xcalloc((size_t)N, 4); s->depth = xcalloc((size_t)N, 4);
    s->parent_dir = xcalloc((size_t)N, 1); s->potential = xcalloc((size_t)N, 8);
    s->tree_arcs = xcalloc((size_t)N, 4); s->tree_slot = xcalloc((size_t)M, 4);
    s->adj_off = xcalloc((size_t)N + 1, 4); s->adj_arc = xcalloc((size_t)2 * N, 4); s->queue = xcalloc((size_t)N, 4);
    s->cyc_arc = xcalloc((size_t)N + 1, 4); s->cyc_sign = xcalloc((size_t)N + 1, 1); s->tmp_nodes = xcalloc((size_t)N + 1, 4);
    s->weights = xcalloc((size_t)M, 8);
    s->cand = xcalloc(128, 4); s->cand_merit = xcalloc((size_t)M, 8); s->cand_tmp = xcalloc((size_t)M, 4);

    /* _initial_supplies + _build_arcs (simplex.py:376-432) */
    for (int v = 1; v < N; ++v) s->supply[v] = supply_in[v - 1];
    double max_cost = 0.0; int have_cost = 0;
    for (int64_t i = 0; i < m; ++i) {
        s->tail[i] = tail[i]; s->head[i] = head[i];
        double lo = lower ? lower[i] : 0.0;
        double up;
        if (isinf(cap[i])) up = INFINITY;
        else { up = cap[i] - lo; if (up < 0.0) up = 0.0; }
        if (lo != 0.0) { s->supply[tail[i]] -= lo; s->supply[head[i]] += lo; }
        s->cost[i] = cost[i]; s->upper[i] = up; s->shift[i] = lo;
        double ac = fabs(cost[i]);
        if (!have_cost || ac > max_cost) { max_cost = ac; have_cost = 1; }
    }
    if (!have_cost) max_cost = 1.0;                       /* max(..., default=1.0) */
    s->penalty = max_cost * (double)(N + 1);              /* simplex.py:161-163 */

    /* block size (simplex.py:194-211, simplex_adaptive.py:70-96): len(self.arcs) is still m here */
    if (block_size <= 0) {
        s->auto_tune = 1;
        int64_t bs = m < 1000 ? m / 4 : (m < 10000 ? m / 8 : m / 16);
        s->block_size = bs > 1 ? bs : 1;
    } else s->block_size = block_size;
    s->ad_current = PR_CANDIDATE_LIST;

    for (int64_t i = 0; i < m; ++i) { s->original_cost[i] = s->cost[i]; s->perturbed_cost[i] = s->cost[i]; }
    apply_cost_perturbation(s);

    /* _initialize_tree (simplex.py:619-730) */
    for (int64_t i = 0; i < M; ++i) s->tree_slot[i] = -1;
    int slot = 0;
    for (int v = 1; v < N; ++v) {
        int64_t a = m + (v - 1);
        double sup = s->supply[v];
        s->artificial[a] = 1; s->in_tree[a] = 1;
        s->cost[a] = s->penalty; s->original_cost[a] = s->penalty; s->perturbed_cost[a] = s->penalty;
        if (fabs(sup) <= tol) { s->tail[a] = 0; s->head[a] = v; s->upper[a] = INFINITY; s->flow[a] = 0.0; }
        else if (sup > 0) { s->tail[a] = v; s->head[a] = 0; s->upper[a] = sup; s->flow[a] = sup; s->artificial_with_flow++; }
        else { s->tail[a] = 0; s->head[a] = v; s->upper[a] = -sup; s->flow[a] = -sup; s->artificial_with_flow++; }
        s->tree_arcs[slot] = (int32_t)a; s->tree_slot[a] = slot; ++slot;
    }
    for (int64_t i = 0; i < M; ++i) {
        s->weights[i] = 1.0;
        s->vec_cost[i] = s->cost[i]; /* _build_vectorized_arrays after perturbation, simplex.py:250-253 */
        set_flow(s, i, s->flow[i]);
    }

    /* select_pivot_strategy (specialized_pivots.py:452-527): some types need a source / sink or the partitions */
    if (s->special == SP_BIPARTITE_MATCHING && !s->left_part) s->special = SP_NONE;
    if (s->special == SP_MAX_FLOW) {
        int src = -1, snk = -1;
        for (int v = 1; v < N; ++v) {
            if (s->supply[v] > tol && src < 0) src = v;
            else if (s->supply[v] < -tol && snk < 0) snk = v;
        }
        if (src < 0 || snk < 0) s->special = SP_NONE;
    }
    if (s->special == SP_SHORTEST_PATH) {
        int src = -1, snk = -1;
        for (int v = 1; v < N; ++v) {
            if (fabs(s->supply[v] - 1.0) <= tol && src < 0) src = v;
            else if (fabs(s->supply[v] + 1.0) <= tol && snk < 0) snk = v;
        }
        if (src < 0 || snk < 0) s->special = SP_NONE; else s->sp_source = src;
    }

    if (max_iterations < 0) max_iterations = 20 * M > 100 ? 20 * M : 100; /* simplex.py:1466-1470 */
    if (pivot_budget >= 0 && pivot_budget < max_iterations) max_iterations = pivot_budget;

    int err = 0, status = ST_OPTIMAL;
    int64_t total = 0;
    /* Phase 1 (simplex.py:1534-1553) */
    apply_phase_costs(s, 1);
    if (rebuild(s) != 0) { ref_free(s); return -2; }
    total += run_iterations(s, max_iterations, 1, 1, 0, &err);
    if (err == -1) { status = ST_UNBOUNDED; goto finish; }
    if (err) { ref_free(s); return err; }

    /* infeasibility + conservation audit (simplex.py:1573-1624), O(n+m) */
    {
        int infeasible = s->artificial_with_flow > 0;
        double *net = s->cand_merit; /* scratch, >= N doubles whenever M >= N-1; sized below */
        double *netbuf = (M >= N) ? net : (double *)xcalloc((size_t)N, 8);
        for (int v = 0; v < N; ++v) netbuf[v] = s->supply[v];
        for (int64_t i = 0; i < M; ++i) {
            if (s->tail[i] != 0) netbuf[s->tail[i]] -= s->flow[i];
            if (s->head[i] != 0) netbuf[s->head[i]] += s->flow[i];
        }
        for (int v = 1; v < N; ++v) if (fabs(netbuf[v]) > tol) infeasible = 1;
        if (netbuf != net) free(netbuf);
        if (infeasible) {
            status = total >= max_iterations ? ST_ITERATION_LIMIT : ST_INFEASIBLE;
            *status_out = status; *objective_out = 0.0;
            for (int64_t i = 0; i < m; ++i) flow_out[i] = 0.0;
            for (int v = 0; v < N; ++v) potential_out[v] = 0.0;
            if (in_tree_out) memset(in_tree_out, 0, (size_t)m);
            stats[0] = total; stats[1] = s->degenerate_pivots; stats[2] = s->arcs_priced; stats[3] = -1;
            stats[4] = 1; /* flows/duals empty, like FlowResult(objective=0.0, flows={}, duals={}) */
            ref_free(s);
            return 0;
        }
    }
    /* Phase 2 (simplex.py:1626-1643) */
    {
        int64_t remaining = max_iterations - total; if (remaining < 0) remaining = 0;
        apply_phase_costs(s, 2);
        if (rebuild(s) != 0) { ref_free(s); return -2; }
        total += run_iterations(s, remaining, 0, 0, total, &err);
        if (err == -1) { status = ST_UNBOUNDED; goto finish; }
        if (err) { ref_free(s); return err; }
    }
    /* status at the budget (simplex.py:1676-1701) */
    if (total >= max_iterations) {
        int64_t arc; int dir;
        status = find_entering_arc(s, 0, &arc, &dir) ? ST_ITERATION_LIMIT : ST_OPTIMAL;
    } else status = ST_OPTIMAL;

finish:
    *status_out = status;
    stats[0] = total; stats[1] = s->degenerate_pivots; stats[2] = s->arcs_priced; stats[3] = s->unb_arc; stats[4] = 0;
    {
        /* result extraction (simplex.py:1703-1728): objective over original costs */
        double objective = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            double fv = s->flow[i] + s->shift[i];
            flow_out[i] = fv;
            objective += fv * s->original_cost[i];
        }
        *objective_out = objective;
        for (int v = 0; v < N; ++v) potential_out[v] = s->potential[v];
        if (in_tree_out) for (int64_t i = 0; i < m; ++i) in_tree_out[i] = s->in_tree[i];
    }
    ref_free(s);
    return 0;
}

/* One full-scan Dantzig pricing pass over caller-provided state; used by the
 * kernel-level parity tests (simplex_pricing.py:97-137 on integer data).
 * state: +1 at lower bound, -1 at upper bound, 0 basic.  Returns the arc
 * index or -1; *dir_out = +1/-1. */
int64_t ref_price_dantzig(int64_t m, const int32_t *tail, const int32_t *head, const double *cost,
                          const double *potential, const double *fwd_res, const double *bwd_res,
                          const uint8_t *in_tree, double tol, int allow_zero, int *dir_out) {
    Ref S; memset(&S, 0, sizeof S);
    S.m_real = m; S.m_tot = m; S.tol = tol;
    S.tail = (int32_t *)tail; S.head = (int32_t *)head; S.cost = (double *)cost;
    S.potential = (double *)potential; S.fwd_res = (double *)fwd_res; S.bwd_res = (double *)bwd_res;
    S.in_tree = (uint8_t *)in_tree;
    uint8_t *art = xcalloc((size_t)m, 1);
    S.artificial = art;
    int64_t arc = -1; int dir = 0;
    int found = select_dantzig_list(&S, NULL, m, allow_zero, &arc, &dir);
    free(art);
    *dir_out = dir;
    return found ? arc : -1;
}

/* One vectorised block-selection pass over caller-provided state (simplex.py:528-617 via
 * select_block_vectorized above): eligibility, merit rc^2 / w, first maximum per direction,
 * forward wins only when strictly greater.  Used by the kernel-level Devex parity test.
 * Returns the arc index or -1; *dir_out = +1/-1; *merit_out = the winning merit. */
int64_t ref_price_block(int64_t m, const int32_t *tail, const int32_t *head, const double *cost,
                        const double *potential, const double *fwd_res, const double *bwd_res,
                        const uint8_t *in_tree, const double *weights, int64_t start, int64_t end,
                        double tol, int allow_zero, int64_t excluded, int *dir_out, double *merit_out) {
    Ref S; memset(&S, 0, sizeof S);
    S.m_real = m; S.m_tot = m; S.tol = tol;
    S.tail = (int32_t *)tail; S.head = (int32_t *)head; S.vec_cost = (double *)cost;
    S.potential = (double *)potential; S.fwd_res = (double *)fwd_res; S.bwd_res = (double *)bwd_res;
    S.in_tree = (uint8_t *)in_tree; S.weights = (double *)weights;
    uint8_t *art = xcalloc((size_t)m, 1);
    S.artificial = art;
    int64_t arc = -1; int dir = 0; double merit = 0.0;
    int found = select_block_vectorized(&S, start, end, allow_zero, excluded, &arc, &dir, &merit);
    free(art);
    *dir_out = dir; *merit_out = merit;
    return found ? arc : -1;
}
