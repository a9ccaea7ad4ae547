/*
 * oracle/incremental_oracle.c -- the "incremental" CPU checker (SURVEY.md section 7 step 4,
 * BASELINE.md section 3 B2): one pre-order walk of the tree per read instead of one ancestor
 * walk per (read, node).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under wepp_amd/ may include, link or call this file.
 * It exists because the faithful restatement (mapper2_oracle.c, O(path^2) per (read, node) like
 * the reference) needs seconds per read at 16 M nodes; this one needs ~0.1 s and is what the
 * full-size GPU tests compare against.  It is trusted only as far as tests/test_incremental.py
 * shows it equal to o_place_sample (scores of EVERY node, eligibility, winner, counts, the list of
 * optimal nodes) on >= 10 000 random (tree, sample) pairs.  PARITY UNPINNED, like the rest of
 * oracle/ (see mapper2_oracle.c).
 *
 * What it restates (file:line relative to /root/reference/src), in closed form (SURVEY.md
 * Appendix A.1-A.5):
 *   genotype distance   usher_mapper.cpp:293-389 (sample vs ancestral set), :394-446 (ancestral
 *                       set vs sample)
 *   own mutations       usher_mapper.cpp:191-265 (the merge with the sample: common / unique,
 *                       masked mutation breaks the loop :198-201), root :266-271
 *   ancestor walk       usher_mapper.cpp:276-287 (most recent non-masked mutation per position)
 *   eligibility         usher_mapper.cpp:455-456
 *   argmin / tie-break  usher_mapper.cpp:457-499, initial state usher_common.cpp:364-381
 *
 * Closed form used.  For an allele state x at position p (x = 0: no mutation on the path):
 *     cost(x, p) = p listed by the sample ? (missing ? 0 : (a_p & (x ? x : ref_p)) == 0)
 *                                         : (x != 0 && x != tree_ref(p))
 * D(n) = D(parent) + sum over the non-masked mutations m of n of cost(m.mut) - cost(state of the
 * parent genotype at m.position); D(-) of the empty genotype = number of non-missing entries whose
 * allele excludes their own reference base.  score(n) = D(parent) + the same sum restricted to the
 * mutations of n the sample "shares" (:215, :245) up to the first masked one; score(root) = D(root)
 * + 1 per masked root mutation with ref != mut (:429-437 with the masked guard of :402).
 * The state of the parent genotype at every mutation is read-independent and computed once per
 * tree (inc_tree_build), with a genotype array and an undo log -- not taken from Mutation::par_nuc.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_tree.h"

typedef struct {
    int n;
    int64_t m;
    int max_pos;
    /* pre-order (depth_first_expansion) arrays */
    uint32_t *moff;      /* [n+1] */
    int32_t *mpos;       /* [m] position (<0 masked) */
    uint8_t *mref, *mmut, *mpar;   /* [m] ref, mut, state of the parent genotype (0 = none) */
    uint32_t *depth;     /* [n] root = 0 */
    uint8_t *leaf;       /* [n] */
    uint32_t *bfs_j;     /* [n] BFS index of the node */
    int64_t *num_leaves; /* [n] */
    int32_t *node_id;    /* [n] caller id */
    uint32_t max_depth;
    /* the same, packed for the walks: one word per node (depth:16 | mutations:15 | leaf:1) and one
     * per mutation (position:20, 0xFFFFF = masked | ref:4 | mut:4 | parent state:4) */
    uint32_t *hdr;       /* [n] */
    uint32_t *mw;        /* [m] */
} inc_tree;

void inc_tree_free(inc_tree *t) {
    if (!t) return;
    free(t->moff); free(t->mpos); free(t->mref); free(t->mmut); free(t->mpar); free(t->depth);
    free(t->leaf); free(t->bfs_j); free(t->num_leaves); free(t->node_id); free(t->hdr); free(t->mw); free(t);
}

inc_tree *inc_tree_build(const otree *T) {
    inc_tree *t = (inc_tree *)calloc(1, sizeof(inc_tree));
    const int n = T->n;
    t->n = n;
    int64_t m = 0;
    int max_pos = 0;
    for (int i = 0; i < n; i++) {
        m += T->nodes[i].nmuts;
        for (int k = 0; k < T->nodes[i].nmuts; k++)
            if (T->nodes[i].muts[k].position > max_pos) max_pos = T->nodes[i].muts[k].position;
    }
    t->m = m;
    t->max_pos = max_pos;
    t->moff = (uint32_t *)calloc((size_t)n + 1, 4);
    t->mpos = (int32_t *)calloc((size_t)(m ? m : 1), 4);
    t->mref = (uint8_t *)calloc((size_t)(m ? m : 1), 1);
    t->mmut = (uint8_t *)calloc((size_t)(m ? m : 1), 1);
    t->mpar = (uint8_t *)calloc((size_t)(m ? m : 1), 1);
    t->depth = (uint32_t *)calloc((size_t)n, 4);
    t->leaf = (uint8_t *)calloc((size_t)n, 1);
    t->bfs_j = (uint32_t *)calloc((size_t)n, 4);
    t->num_leaves = (int64_t *)calloc((size_t)n, 8);
    t->node_id = (int32_t *)calloc((size_t)n, 4);
    uint32_t *bfs_of_id = (uint32_t *)calloc((size_t)n, 4);
    for (int k = 0; k < n; k++) bfs_of_id[T->bfs[k]->id] = (uint32_t)k;
    /* genotype of the current root path (most recent non-masked mutation per position) */
    uint8_t *geno = (uint8_t *)calloc((size_t)max_pos + 2, 1);
    typedef struct { int32_t pos; uint8_t old; } undo_t;
    undo_t *undo = (undo_t *)calloc((size_t)(m ? m : 1), sizeof(undo_t));
    size_t *undo_at = (size_t *)calloc((size_t)n + 1, sizeof(size_t));   /* undo log size below the open node of depth d - 1 */
    size_t nu = 0;
    uint32_t w = 0;
    for (int i = 0; i < n; i++) {
        const onode *nd = T->dfs[i];
        uint32_t d = 0;
        if (nd->parent) d = t->depth[nd->parent->dfs_idx] + 1;
        /* leave the subtrees that ended before this node */
        while (nu > undo_at[d]) { nu--; geno[undo[nu].pos] = undo[nu].old; }
        t->depth[i] = d;
        if (d > t->max_depth) t->max_depth = d;
        t->leaf[i] = (uint8_t)n_is_leaf(nd);
        t->bfs_j[i] = bfs_of_id[nd->id];
        t->num_leaves[i] = nd->num_leaves;
        t->node_id[i] = nd->id;
        t->moff[i] = w;
        for (int k = 0; k < nd->nmuts; k++, w++) {
            const omut *mu = &nd->muts[k];
            t->mpos[w] = mu->position;
            t->mref[w] = (uint8_t)mu->ref_nuc;
            t->mmut[w] = (uint8_t)mu->mut_nuc;
            t->mpar[w] = mu->position >= 0 ? geno[mu->position] : 0;
        }
        /* the genotype the children see: every non-masked mutation of this node (:276-287) */
        for (int k = 0; k < nd->nmuts; k++) {
            const omut *mu = &nd->muts[k];
            if (mu->position < 0) continue;
            undo[nu].pos = mu->position;
            undo[nu].old = geno[mu->position];
            nu++;
            geno[mu->position] = (uint8_t)mu->mut_nuc;
        }
        undo_at[d + 1] = nu;      /* what a node at depth d + 1 restores before it looks at the genotype */
    }
    t->moff[n] = w;
    t->hdr = (uint32_t *)calloc((size_t)n, 4);
    t->mw = (uint32_t *)calloc((size_t)(m ? m : 1), 4);
    for (int i = 0; i < n; i++)
        t->hdr[i] = (t->depth[i] & 0xFFFFu) | (((t->moff[i + 1] - t->moff[i]) & 0x7FFFu) << 16) | ((uint32_t)t->leaf[i] << 31);
    for (int64_t k = 0; k < m; k++)
        t->mw[k] = (t->mpos[k] < 0 ? 0xFFFFFu : (uint32_t)t->mpos[k]) | ((uint32_t)(t->mref[k] & 15) << 20) |
                   ((uint32_t)(t->mmut[k] & 15) << 24) | ((uint32_t)(t->mpar[k] & 15) << 28);
    free(bfs_of_id); free(geno); free(undo); free(undo_at);
    return t;
}

typedef struct {
    int32_t best_set_difference;
    uint32_t num_best;
    uint32_t best_j;
    int32_t best_node_id;
    uint32_t best_node_has_unique;
} inc_result;

static inline int cost_listed(uint8_t a, uint8_t sref, uint8_t missing, uint8_t x) {
    if (missing) return 0;
    return (a & (x ? x : sref)) == 0;
}

/* one sample as the walks see it: its entries and sidx[p] = 1 + index of the entry at position p
 * (0 = not listed) */
typedef struct {
    int nS;
    const int32_t *pos;
    const uint8_t *ref, *mut, *missing;
    const uint16_t *sidx;
    int max_pos;
} inc_sample;

/* Node i (pre-order) for one sample whose parent genotype costs dpar: the node's own value `score`,
 * the cost `dn` of its genotype (what its children start from), eligibility, has_unique. */
static inline void node_eval(const inc_tree *t, int i, const inc_sample *S, int dpar, int *score_out, int *dn_out,
                             int *elig_out, int *hu_out) {
    const uint32_t h = t->hdr[i];
    const int is_root = (h & 0xFFFFu) == 0, leaf = (int)(h >> 31);
    int score = dpar, dn = dpar;
    int has_unique = 0, node_num_mut = 0, num_common = 0, stopped = 0;
    const uint32_t w0 = t->moff[i], w1 = t->moff[i + 1];
    for (uint32_t w = w0; w < w1; w++) {
        const uint32_t x = t->mw[w];
        const uint32_t p = x & 0xFFFFFu;
        const uint8_t ref = (x >> 20) & 15, mut = (x >> 24) & 15, par = (uint8_t)(x >> 28);
        if (p == 0xFFFFFu) {
            if (is_root) { if (ref != mut) score += 1; }                       /* :402, :429-437 */
            else if (!stopped) { node_num_mut++; has_unique = 1; stopped = 1; }   /* :198-201 */
            continue;
        }
        const int si = ((int)p <= S->max_pos) ? (int)S->sidx[p] - 1 : -1;
        int c_mut, c_par, common;
        if (si >= 0) {
            const uint8_t a = S->mut[si], sref = S->ref[si], miss = S->missing[si];
            c_mut = cost_listed(a, sref, miss, mut);
            c_par = cost_listed(a, sref, miss, par);
            /* :210-216: a missing base shares the mutation but leaves the genotype alone: its cost is 0
             * either way */
            common = miss ? 1 : ((a & mut) != 0);
        } else {
            c_mut = mut != ref;
            c_par = par != 0 && par != ref;
            common = mut == ref;                                               /* :245 */
        }
        dn += c_mut - c_par;
        if (is_root) { score += c_mut - c_par; continue; }                     /* :266-271 */
        if (stopped) continue;                                                 /* behind a masked mutation */
        node_num_mut++;
        if (common) { num_common++; score += c_mut - c_par; }
        else has_unique = 1;
    }
    *score_out = score;
    *dn_out = dn;
    *hu_out = has_unique;
    *elig_out = is_root || (has_unique && !leaf && num_common > 0 && node_num_mut != num_common) ||
                (leaf && num_common > 0) || (!has_unique && !leaf && node_num_mut == num_common);   /* :455-456 */
}

typedef struct {
    int best, best_hu;
    int64_t best_leaves;
    uint32_t best_j, num_best;
    int32_t best_id;
} inc_best;

static inline void best_init(inc_best *b) {
    b->best = 0x7fffffff; b->best_hu = 0; b->best_leaves = -1; b->best_j = 0; b->num_best = 0; b->best_id = -1;
}
/* an eligible node with `score` <= the current best (:466-498; the order of arrival does not matter) */
static inline void best_take(inc_best *b, const inc_tree *t, int i, int score, int has_unique) {
    if (score < b->best) {
        b->best = score; b->num_best = 1; b->best_leaves = t->num_leaves[i]; b->best_j = t->bfs_j[i];
        b->best_id = t->node_id[i]; b->best_hu = has_unique;
    } else if (score == b->best) {
        b->num_best++;
        if (t->num_leaves[i] > b->best_leaves || (t->num_leaves[i] == b->best_leaves && t->bfs_j[i] > b->best_j)) {   /* :484-487 */
            b->best_leaves = t->num_leaves[i]; b->best_j = t->bfs_j[i]; b->best_id = t->node_id[i]; b->best_hu = has_unique;
        }
    }
}
static inline void best_out(const inc_best *b, inc_result *out) {
    out->best_set_difference = b->best;
    out->num_best = b->num_best;
    out->best_j = b->best_j;
    out->best_node_id = b->best_id;
    out->best_node_has_unique = (uint32_t)b->best_hu;
}

static int d_empty_of(int nS, const uint8_t *s_ref, const uint8_t *s_mut, const uint8_t *s_missing) {
    int d = 0;
    for (int i = 0; i < nS; i++)
        if (!s_missing[i]) d += (s_mut[i] & s_ref[i]) == 0;
    return d;
}

/* One sample, one walk.  node_scores (optional, [n], BFS order): the -p mode's values incl. the +1
 * of ineligible nodes (:500-505).  best_vec (optional, capacity n): BFS indices of the optimal
 * nodes, in pre-order. */
int inc_place_sample(const inc_tree *t, int nS, const int32_t *s_pos, const uint8_t *s_ref, const uint8_t *s_mut,
                     const uint8_t *s_missing, int32_t *node_scores, inc_result *out, uint32_t *best_vec) {
    if (nS >= 65535) return 1;
    uint16_t *sidx = (uint16_t *)calloc((size_t)t->max_pos + 2, 2);
    int *dstack = (int *)calloc((size_t)t->max_depth + 2, sizeof(int));
    if (!sidx || !dstack) { free(sidx); free(dstack); return 2; }
    for (int i = 0; i < nS; i++)
        if (s_pos[i] >= 0 && s_pos[i] <= t->max_pos) sidx[s_pos[i]] = (uint16_t)(i + 1);
    const inc_sample S = {nS, s_pos, s_ref, s_mut, s_missing, sidx, t->max_pos};
    const int d_empty = d_empty_of(nS, s_ref, s_mut, s_missing);
    inc_best b;
    best_init(&b);
    uint32_t nbv = 0;
    for (int i = 0; i < t->n; i++) {
        const uint32_t d = t->hdr[i] & 0xFFFFu;
        int score, dn, elig, hu;
        node_eval(t, i, &S, d ? dstack[d - 1] : d_empty, &score, &dn, &elig, &hu);
        dstack[d] = dn;
        if (node_scores) node_scores[t->bfs_j[i]] = elig ? score : score + 1;
        if (!elig || score > b.best) continue;
        if (best_vec) {
            if (score < b.best) nbv = 0;
            best_vec[nbv++] = t->bfs_j[i];
        }
        best_take(&b, t, i, score, hu);
    }
    best_out(&b, out);
    free(sidx); free(dstack);
    return 0;
}

/* ---- GROUP reads per walk: the tree is streamed once for the group ------------------------ *
 * A node none of whose mutations is listed by a read of the group looks the same to all of them:
 * its static deltas are added to every read's parent cost in one (vectorisable) loop and the rare
 * candidates (score <= best) are taken one by one.  A node with a listed mutation goes through
 * node_eval read by read.  Same arithmetic as inc_place_sample, node for node. */
#define INC_GROUP 16

typedef struct {
    const inc_tree *t;
    const uint32_t *read_off;
    const int32_t *r_pos;
    const uint8_t *r_ref, *r_mut, *r_missing;
    inc_result *out;
    uint32_t n_reads;
    volatile uint32_t *next;
    int rc;
} inc_job;

static void *inc_worker(void *p) {
    inc_job *jb = (inc_job *)p;
    const inc_tree *t = jb->t;
    const size_t np = (size_t)t->max_pos + 2;
    uint8_t *any = (uint8_t *)calloc(np, 1);
    uint16_t *sidx = (uint16_t *)calloc(np * INC_GROUP, 2);
    int32_t *dstack = (int32_t *)calloc(((size_t)t->max_depth + 2) * INC_GROUP, sizeof(int32_t));
    if (!any || !sidx || !dstack) { jb->rc = 2; free(any); free(sidx); free(dstack); return NULL; }
    for (;;) {
        const uint32_t g = __sync_fetch_and_add(jb->next, 1u);
        const uint32_t r0 = g * INC_GROUP;
        if (r0 >= jb->n_reads) break;
        const int nb = (int)((jb->n_reads - r0 < INC_GROUP) ? jb->n_reads - r0 : INC_GROUP);
        inc_sample S[INC_GROUP];
        int32_t d_empty[INC_GROUP];
        int32_t best[INC_GROUP];
        inc_best bst[INC_GROUP];
        int bad = 0;
        for (int b = 0; b < INC_GROUP; b++) {
            const uint32_t lo = b < nb ? jb->read_off[r0 + b] : 0, hi = b < nb ? jb->read_off[r0 + b + 1] : 0;
            if (hi - lo >= 65535) bad = 1;
            S[b].nS = (int)(hi - lo);
            S[b].pos = jb->r_pos + lo; S[b].ref = jb->r_ref + lo; S[b].mut = jb->r_mut + lo; S[b].missing = jb->r_missing + lo;
            S[b].sidx = sidx + np * (size_t)b;
            S[b].max_pos = t->max_pos;
            best_init(&bst[b]);
            best[b] = 0x7fffffff;
        }
        if (bad) { jb->rc = 1; break; }
        for (int b = 0; b < nb; b++) {
            for (int i = 0; i < S[b].nS; i++) {
                const int32_t ps = S[b].pos[i];
                if (ps >= 0 && ps <= t->max_pos) { sidx[np * (size_t)b + ps] = (uint16_t)(i + 1); any[ps] = 1; }
            }
            d_empty[b] = d_empty_of(S[b].nS, S[b].ref, S[b].mut, S[b].missing);
        }
        for (int b = nb; b < INC_GROUP; b++) d_empty[b] = 0;
        for (int i = 0; i < t->n; i++) {
            const uint32_t h = t->hdr[i];
            const uint32_t d = h & 0xFFFFu;
            const int is_root = d == 0, leaf = (int)(h >> 31);
            const int32_t *dpar = d ? dstack + (size_t)(d - 1) * INC_GROUP : d_empty;
            int32_t *dcur = dstack + (size_t)d * INC_GROUP;
            /* static part: what the node is to a read that lists none of its positions */
            int sdn = 0, sscore = 0, shu = 0, snum = 0, scom = 0, stopped = 0, listed = 0;
            const uint32_t w0 = t->moff[i], w1 = t->moff[i + 1];
            for (uint32_t w = w0; w < w1; w++) {
                const uint32_t x = t->mw[w];
                const uint32_t ps = x & 0xFFFFFu;
                const uint8_t ref = (x >> 20) & 15, mut = (x >> 24) & 15, par = (uint8_t)(x >> 28);
                if (ps == 0xFFFFFu) {
                    if (is_root) { if (ref != mut) sscore += 1; }
                    else if (!stopped) { snum++; shu = 1; stopped = 1; }
                    continue;
                }
                if ((int)ps <= t->max_pos && any[ps]) listed = 1;
                const int dlt = (mut != ref) - (par != 0 && par != ref);
                sdn += dlt;
                if (is_root) { sscore += dlt; continue; }
                if (stopped) continue;
                snum++;
                if (mut == ref) { scom++; sscore += dlt; }
                else shu = 1;
            }
            if (!listed) {
                const int elig = is_root || (shu && !leaf && scom > 0 && snum != scom) || (leaf && scom > 0) ||
                                 (!shu && !leaf && snum == scom);
                int cand = 0;
                for (int b = 0; b < INC_GROUP; b++) {
                    const int32_t dp = dpar[b];
                    dcur[b] = dp + sdn;
                    cand |= (dp + sscore <= best[b]);
                }
                if (elig && cand) {
                    for (int b = 0; b < nb; b++) {
                        const int sc = dpar[b] + sscore;
                        if (sc <= best[b]) { best_take(&bst[b], t, i, sc, shu); best[b] = bst[b].best; }
                    }
                }
            } else {
                for (int b = 0; b < nb; b++) {
                    int sc, dn, elig, hu;
                    node_eval(t, i, &S[b], dpar[b], &sc, &dn, &elig, &hu);
                    dcur[b] = dn;
                    if (elig && sc <= best[b]) { best_take(&bst[b], t, i, sc, hu); best[b] = bst[b].best; }
                }
                for (int b = nb; b < INC_GROUP; b++) dcur[b] = 0;
            }
        }
        for (int b = 0; b < nb; b++) {
            best_out(&bst[b], &jb->out[r0 + b]);
            for (int i = 0; i < S[b].nS; i++) {
                const int32_t ps = S[b].pos[i];
                if (ps >= 0 && ps <= t->max_pos) { sidx[np * (size_t)b + ps] = 0; any[ps] = 0; }
            }
        }
    }
    free(any); free(sidx); free(dstack);
    return NULL;
}

/* reads CSR like oracle_place_batch; the threads take groups of INC_GROUP consecutive reads from a
 * shared counter */
int inc_place_batch(const inc_tree *t, uint32_t n_reads, const uint32_t *read_off, const int32_t *r_pos,
                    const uint8_t *r_ref, const uint8_t *r_mut, const uint8_t *r_missing, inc_result *out,
                    int nthreads) {
    if (nthreads < 1) nthreads = 1;
    const uint32_t ngroups = (n_reads + INC_GROUP - 1) / INC_GROUP;
    if ((uint32_t)nthreads > ngroups) nthreads = ngroups ? (int)ngroups : 1;
    volatile uint32_t next = 0;
    inc_job *jobs = (inc_job *)calloc((size_t)nthreads, sizeof(inc_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; i++) {
        jobs[i].t = t; jobs[i].read_off = read_off; jobs[i].r_pos = r_pos; jobs[i].r_ref = r_ref;
        jobs[i].r_mut = r_mut; jobs[i].r_missing = r_missing; jobs[i].out = out; jobs[i].n_reads = n_reads;
        jobs[i].next = &next;
    }
    if (nthreads == 1) inc_worker(&jobs[0]);
    else {
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, inc_worker, &jobs[i]);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    }
    int rc = 0;
    for (int i = 0; i < nthreads; i++) if (jobs[i].rc) rc = jobs[i].rc;
    free(th); free(jobs);
    return rc;
}
