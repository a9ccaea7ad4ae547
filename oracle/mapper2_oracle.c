/*
 * oracle/mapper2_oracle.c -- CPU restatement of the reference placement path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under wepp_amd/ may include, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (TurakhiaLab/WEPP) ships no tests, golden
 * vectors or fixtures for this path, and its sources
 * cannot be compiled in this image without writing stand-ins for TBB, Boost
 * and protobuf headers (usher_graph.hpp:10 includes <tbb/queuing_rw_mutex.h>,
 * mutation_annotated_tree.hpp pulls tbb + boost + parsimony.pb.h), which the
 * build rules forbid.  The only reference-derived known answer available is
 * the 7-node smoke result recorded in SURVEY.md Appendix B (checked by
 * tests/test_oracle.py).  Everything else in this file is a line-by-line
 * restatement of the cited reference code.
 *
 * What is restated (file:line relative to /root/reference/src):
 *   o_mapper2_body()      usher_mapper.cpp:168-506  (mapper2_body; the excess / imputed
 *                         mutation vectors of compute_vecs=true are filled when the
 *                         caller provides them)
 *   o_place_sample()      usher_common.cpp:339-446  (per-sample loop: BFS
 *                         expansion, initial best state :364-381, pass 1
 *                         :386-411, pass 2 :413-446)
 *   o_bfs()/o_dfs()       mutation_annotated_tree.cpp:1115-1163
 *   o_get_num_leaves()    mutation_annotated_tree.cpp:839-852 (memoised; the
 *                         tree is immutable on this path so the value is the
 *                         same as the reference's recursive recount)
 *   is_leaf/is_root       mutation_annotated_tree.cpp:684-690
 *   Mutation              mutation_annotated_tree.hpp:44-78
 *
 * Nucleotides are the reference's int8 one-hot masks (A=1,C=2,G=4,T=8, IUPAC =
 * OR, N=15).  position < 0 means masked (mutation_annotated_tree.hpp:68-70).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "oracle_tree.h"

/* ---- tree construction ------------------------------------------------ */

/* parent[i] = id of parent or -1 for the (single) root.  Children of a node
 * are stored in ascending id order (the order create_node(.., par, ..) would
 * have pushed them, mutation_annotated_tree.cpp:865-878).  Mutations of node i
 * are mut_*[mut_off[i] .. mut_off[i+1]) in stored order. */
otree *oracle_tree_build(int n, const int32_t *parent, const uint32_t *mut_off,
                         const int32_t *mut_pos, const uint8_t *mut_ref,
                         const uint8_t *mut_par, const uint8_t *mut_mut) {
    if (n <= 0) return NULL;
    otree *t = (otree *)calloc(1, sizeof(otree));
    t->n = n;
    t->nodes = (onode *)calloc((size_t)n, sizeof(onode));
    uint32_t m = mut_off[n];
    t->mut_pool = (omut *)calloc(m ? m : 1, sizeof(omut));
    t->child_pool = (onode **)calloc((size_t)n, sizeof(onode *));
    int *cnt = (int *)calloc((size_t)n, sizeof(int));
    int nroots = 0;
    for (int i = 0; i < n; i++) {
        if (parent[i] < 0) { nroots++; t->root = &t->nodes[i]; }
        else if (parent[i] >= n || parent[i] == i) { nroots = -1000000; }
        else cnt[parent[i]]++;
    }
    if (nroots != 1) { free(cnt); free(t->child_pool); free(t->mut_pool); free(t->nodes); free(t); return NULL; }
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        onode *nd = &t->nodes[i];
        nd->id = i;
        nd->children = t->child_pool + off;
        off += (size_t)cnt[i];
        nd->nchildren = 0;
        nd->num_leaves = -1;
        nd->muts = t->mut_pool + mut_off[i];
        nd->nmuts = (int)(mut_off[i + 1] - mut_off[i]);
        for (uint32_t k = mut_off[i]; k < mut_off[i + 1]; k++) {
            omut *mm = &t->mut_pool[k];
            mm->position = mut_pos[k];
            mm->ref_nuc = (int8_t)mut_ref[k];
            mm->par_nuc = (int8_t)mut_par[k];
            mm->mut_nuc = (int8_t)mut_mut[k];
            mm->is_missing = 0;
        }
    }
    for (int i = 0; i < n; i++) {
        if (parent[i] >= 0) {
            onode *p = &t->nodes[parent[i]];
            t->nodes[i].parent = p;
            p->children[p->nchildren++] = &t->nodes[i];
        }
    }
    free(cnt);
    /* breadth_first_expansion, mutation_annotated_tree.cpp:1115-1141 */
    t->bfs = (onode **)malloc(sizeof(onode *) * (size_t)n);
    {
        size_t head = 0, tail = 0;
        t->bfs[tail++] = t->root;
        while (head < tail) {
            onode *c = t->bfs[head++];
            for (int k = 0; k < c->nchildren; k++) {
                if (tail >= (size_t)n) { tail = (size_t)n + 1; break; }
                t->bfs[tail++] = c->children[k];
            }
            if (tail > (size_t)n) break;
        }
        if (tail != (size_t)n) { /* cycle / disconnected */
            free(t->bfs); free(t->child_pool); free(t->mut_pool); free(t->nodes); free(t);
            return NULL;
        }
    }
    /* depth_first_expansion (pre-order), mutation_annotated_tree.cpp:1143-1163 */
    t->dfs = (onode **)malloc(sizeof(onode *) * (size_t)n);
    {
        onode **stack = (onode **)malloc(sizeof(onode *) * (size_t)n);
        size_t sp = 0, cnt2 = 0;
        stack[sp++] = t->root;
        while (sp) {
            onode *c = stack[--sp];
            c->dfs_idx = (int)cnt2;
            t->dfs[cnt2++] = c;
            for (int k = c->nchildren - 1; k >= 0; k--) stack[sp++] = c->children[k];
        }
        free(stack);
    }
    /* get_num_leaves memo, bottom-up over reversed BFS (same value as the
     * recursion at mutation_annotated_tree.cpp:839-852). */
    for (int i = n - 1; i >= 0; i--) {
        onode *c = t->bfs[i];
        if (n_is_leaf(c)) c->num_leaves = 1;
        else {
            int64_t s = 0;
            for (int k = 0; k < c->nchildren; k++) s += c->children[k]->num_leaves;
            c->num_leaves = s;
        }
    }
    return t;
}

void oracle_tree_free(otree *t) {
    if (!t) return;
    free(t->dfs); free(t->bfs); free(t->child_pool); free(t->mut_pool); free(t->nodes); free(t);
}

int oracle_tree_size(const otree *t) { return t->n; }
/* out[k] = caller id of bfs[k] */
void oracle_tree_bfs_ids(const otree *t, int32_t *out) { for (int i = 0; i < t->n; i++) out[i] = t->bfs[i]->id; }
void oracle_tree_dfs_ids(const otree *t, int32_t *out) { for (int i = 0; i < t->n; i++) out[i] = t->dfs[i]->id; }
void oracle_tree_num_leaves(const otree *t, int64_t *out) { for (int i = 0; i < t->n; i++) out[i] = t->nodes[i].num_leaves; }

/* ---- mapper2_input, usher_graph.hpp:74-102 ---------------------------- */
typedef struct {
    const otree *T;
    onode *node;
    const omut *missing_sample_mutations;
    int n_missing_sample_mutations;
    int *best_set_difference;
    int *set_difference;
    size_t *best_node_num_leaves;
    size_t j;
    size_t *best_j;
    size_t distance;            /* ctor: 0, best_distance = &distance */
    size_t *best_distance;
    size_t *num_best;
    onode **best_node;
    uint8_t *node_has_unique;
    struct jvec *best_j_vec;    /* std::vector<size_t>* */
    int *has_unique;
    omut *imputed_mutations;    /* compute_vecs: std::vector<MAT::Mutation>* (capacity >= |S|), or NULL */
    int *n_imputed;
    omut *excess_mutations;     /* compute_vecs: std::vector<MAT::Mutation>* (capacity >= |S| + path), or NULL */
    int *n_excess;
} o_mapper2_input;

typedef struct { omut *v; int n, cap; int *pos; } anc_vec;
struct jvec { size_t *v; size_t n, cap; };

static void jvec_push(struct jvec *a, size_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 16;
        a->v = (size_t *)realloc(a->v, sizeof(size_t) * a->cap);
    }
    a->v[a->n++] = x;
}

static void anc_push(anc_vec *a, const omut *m) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 64;
        a->v = (omut *)realloc(a->v, sizeof(omut) * (size_t)a->cap);
        a->pos = (int *)realloc(a->pos, sizeof(int) * (size_t)a->cap);
    }
    a->v[a->n] = *m;
    a->pos[a->n] = m->position;
    a->n++;
}

static int cmp_omut(const void *a, const void *b) {
    int pa = ((const omut *)a)->position, pb = ((const omut *)b)->position;
    return (pa > pb) - (pa < pb);
}

/* usher_mapper.cpp:168-506.  `scratch` only recycles the two std::vectors the
 * reference allocates per call (:179-180). */
static void o_mapper2_body(o_mapper2_input *input, int compute_parsimony_scores, anc_vec *scratch) {
    int set_difference = 0;                                    /* :173 */
    int best_set_difference = *input->best_set_difference;     /* :177 (snapshot) */
    anc_vec *anc = scratch;
    anc->n = 0;                                                /* :179-180 */
    int has_unique = 0;                                        /* :184 */
    int node_num_mut = 0;
    int num_common_mut = 0;
    const omut *S = input->missing_sample_mutations;
    const int nS = input->n_missing_sample_mutations;

    if (!n_is_root(input->node)) {                             /* :191 */
        int start_index = 0;
        for (int i1 = 0; i1 < input->node->nmuts; i1++) {      /* :193 */
            omut m1 = input->node->muts[i1];
            node_num_mut++;
            int8_t anc_nuc = m1.mut_nuc;
            if (m_is_masked(&m1)) {                            /* :198-201 */
                has_unique = 1;
                break;
            }
            int found = 0, found_pos = 0;
            for (int k = start_index; k < nS; k++) {           /* :205 */
                omut m2 = S[k];
                start_index = k;
                if (m1.position == m2.position) {
                    found_pos = 1;
                    if (m2.is_missing) {                       /* :210-212 */
                        found = 1;
                        num_common_mut++;
                    } else {
                        int8_t nuc = m2.mut_nuc;
                        if ((nuc & anc_nuc) != 0) {            /* :215 */
                            omut m;
                            m.position = m1.position;
                            m.ref_nuc = m1.ref_nuc;
                            m.par_nuc = m1.par_nuc;
                            m.mut_nuc = anc_nuc;
                            m.is_missing = 0;
                            anc_push(anc, &m);                 /* :223-224 */
                            if (input->imputed_mutations != NULL && input->excess_mutations)      /* :226-228 */
                                input->excess_mutations[(*input->n_excess)++] = m;
                            found = 1;
                            num_common_mut++;
                            break;
                        }
                    }
                }
                if (m1.position < m2.position) break;          /* :240-242 */
            }
            if (!found) {                                      /* :244 */
                if (!found_pos && (anc_nuc == m1.ref_nuc)) {
                    omut m;
                    m.position = m1.position;
                    m.ref_nuc = m1.ref_nuc;
                    m.par_nuc = m1.par_nuc;
                    m.mut_nuc = anc_nuc;
                    m.is_missing = 0;
                    anc_push(anc, &m);                         /* :253-254 */
                    if (input->imputed_mutations != NULL && input->excess_mutations)              /* :256-258 */
                        input->excess_mutations[(*input->n_excess)++] = m;
                    num_common_mut++;
                } else {
                    has_unique = 1;
                }
            }
        }
    } else {
        for (int i = 0; i < input->node->nmuts; i++)           /* :267-270 */
            anc_push(anc, &input->node->muts[i]);
    }

    {                                                          /* :276-287 */
        onode *n = input->node;
        while (n->parent != NULL) {
            n = n->parent;
            for (int i = 0; i < n->nmuts; i++) {
                const omut *m = &n->muts[i];
                if (m_is_masked(m)) continue;
                int seen = 0;
                for (int q = 0; q < anc->n; q++)               /* std::find :281 */
                    if (anc->pos[q] == m->position) { seen = 1; break; }
                if (!seen) anc_push(anc, m);
            }
        }
    }

    /* :290 std::sort by position (order of equal positions is unspecified in
     * the reference and score-neutral). */
    qsort(anc->v, (size_t)anc->n, sizeof(omut), cmp_omut);

    for (int i1 = 0; i1 < nS; i1++) {                          /* :293 */
        omut m1 = S[i1];
        if (m1.is_missing) continue;                           /* :295 */
        int found_pos = 0, found = 0, has_ref = 0;
        int8_t anc_nuc = m1.ref_nuc;
        if ((m1.mut_nuc & m1.ref_nuc) != 0) has_ref = 1;
        for (int k = 0; k < anc->n; k++) {                     /* :307 (start_index is 0) */
            omut m2 = anc->v[k];
            if (m_is_masked(&m2)) continue;
            if (m1.position == m2.position) {
                found_pos = 1;
                anc_nuc = m2.mut_nuc;
                if ((m1.mut_nuc & anc_nuc) != 0) found = 1;
                break;
            }
        }
        const int compute_vecs = input->imputed_mutations != NULL;
        if (found) {
            if (compute_vecs && ((m1.mut_nuc & (m1.mut_nuc - 1)) != 0)) {           /* :327-335 */
                omut m;
                m.position = m1.position; m.ref_nuc = m1.ref_nuc; m.par_nuc = anc_nuc; m.mut_nuc = anc_nuc; m.is_missing = 0;
                input->imputed_mutations[(*input->n_imputed)++] = m;
            }
        } else if (!found_pos && has_ref) {
            if (compute_vecs && ((m1.mut_nuc & (m1.mut_nuc - 1)) != 0)) {           /* :343-351 */
                omut m;
                m.position = m1.position; m.ref_nuc = m1.ref_nuc; m.par_nuc = anc_nuc; m.mut_nuc = m1.ref_nuc; m.is_missing = 0;
                input->imputed_mutations[(*input->n_imputed)++] = m;
            }
        } else {
            omut m;                                            /* :357-388 */
            m.position = m1.position;
            m.ref_nuc = m1.ref_nuc;
            m.par_nuc = anc_nuc;
            m.mut_nuc = 0;  /* the reference leaves it unset when mut_nuc has no bit in 0..3 */
            if (has_ref) {
                m.mut_nuc = m1.ref_nuc;
            } else {
                for (int j = 0; j < 4; j++) {
                    if (((1 << j) & m1.mut_nuc) != 0) { m.mut_nuc = (int8_t)(1 << j); break; }
                }
            }
            if (compute_vecs && ((m1.mut_nuc & (m1.mut_nuc - 1)) != 0))            /* :376-378 */
                input->imputed_mutations[(*input->n_imputed)++] = m;
            if (m.mut_nuc != m.par_nuc) {                      /* :379 */
                if (compute_vecs && input->excess_mutations) input->excess_mutations[(*input->n_excess)++] = m;   /* :380-382 */
                set_difference += 1;
                if (!compute_parsimony_scores && (set_difference > best_set_difference)) return;
            }
        }
    }

    for (int i1 = 0; i1 < anc->n; i1++) {                      /* :394 */
        omut m1 = anc->v[i1];
        int found = 0, found_pos = 0;
        int8_t anc_nuc = m1.mut_nuc;
        for (int k = 0; k < nS; k++) {                         /* :399 */
            if (m_is_masked(&m1)) break;                       /* :402 */
            omut m2 = S[k];
            if (m1.position == m2.position) {
                found_pos = 1;
                if (m2.is_missing) { found = 1; break; }       /* :410-413 */
                if ((m2.mut_nuc & anc_nuc) != 0) found = 1;
            }
        }
        if (found) {
        } else if (!found_pos && !m_is_masked(&m1) && (anc_nuc == m1.ref_nuc)) {
        } else if (found_pos && !found) {
        } else {
            omut m;                                            /* :429-444 */
            m.position = m1.position;
            m.ref_nuc = m1.ref_nuc;
            m.par_nuc = anc_nuc;
            m.mut_nuc = m1.ref_nuc;
            if (m.mut_nuc != m.par_nuc) {
                set_difference += 1;
                if (!compute_parsimony_scores && (set_difference > best_set_difference)) return;
                if (input->imputed_mutations != NULL && input->excess_mutations)                      /* :440-442 */
                    input->excess_mutations[(*input->n_excess)++] = m;
            }
        }
    }

    if (compute_parsimony_scores) *input->set_difference = set_difference;   /* :449-451 */

    if (n_is_root(input->node) ||
        ((has_unique && !n_is_leaf(input->node) && (num_common_mut > 0) && (node_num_mut != num_common_mut)) ||
         (n_is_leaf(input->node) && (num_common_mut > 0)) ||
         (!has_unique && !n_is_leaf(input->node) && (node_num_mut == num_common_mut)))) {   /* :455-456 */
        if (set_difference > *input->best_set_difference) return;            /* :458-461 */
        size_t num_leaves = (size_t)input->node->num_leaves;                 /* :465 */
        if (set_difference < *input->best_set_difference) {                  /* :466-476 */
            *input->best_set_difference = set_difference;
            *input->best_node = input->node;
            *input->best_node_num_leaves = num_leaves;
            *input->best_j = input->j;
            *input->num_best = 1;
            *input->has_unique = has_unique;
            *input->best_distance = input->distance;
            input->node_has_unique[input->j] = (uint8_t)has_unique;
            input->best_j_vec->n = 0;                                        /* clear() */
            jvec_push(input->best_j_vec, input->j);
        } else if (set_difference == *input->best_set_difference) {          /* :477-498 */
            if (((input->distance == *input->best_distance) &&
                 ((num_leaves > *input->best_node_num_leaves) ||
                  ((num_leaves == *input->best_node_num_leaves) && (*input->best_j < input->j)))) ||
                (input->distance < *input->best_distance)) {
                *input->best_set_difference = set_difference;
                *input->best_node = input->node;
                *input->best_node_num_leaves = num_leaves;
                *input->best_j = input->j;
                *input->has_unique = has_unique;
                *input->best_distance = input->distance;
            }
            *input->num_best += 1;
            input->node_has_unique[input->j] = (uint8_t)has_unique;
            jvec_push(input->best_j_vec, input->j);
        }
    } else if (compute_parsimony_scores) {
        *input->set_difference = set_difference + 1;                         /* :500-505 */
    }
}

/* ---- per-sample loop, usher_common.cpp:339-446 ------------------------- */
typedef struct {
    int32_t best_set_difference;   /* parsimony score of the placement   */
    uint32_t num_best;             /* number of parsimony-optimal nodes  */
    uint32_t best_j;               /* BFS index of the chosen node       */
    int32_t best_node_id;          /* caller id of the chosen node       */
    uint32_t best_node_has_unique; /* usher_common.cpp:374, :401         */
} oracle_result;

typedef struct {
    int best_set_difference;
    size_t best_node_num_leaves, best_j, num_best;
    onode *best_node;
    int has_unique;
    struct jvec best_j_vec;
    uint8_t *node_has_unique;
} o_best_state;

static void fill_input(o_mapper2_input *inp, const otree *T, o_best_state *st, const omut *S, int nS,
                       size_t k, int *node_set_difference) {
    inp->T = T;                                        /* usher_common.cpp:389-407 */
    inp->node = T->bfs[k];
    inp->missing_sample_mutations = S;
    inp->n_missing_sample_mutations = nS;
    inp->best_node_num_leaves = &st->best_node_num_leaves;
    inp->best_set_difference = &st->best_set_difference;
    inp->best_node = &st->best_node;
    inp->best_j = &st->best_j;
    inp->num_best = &st->num_best;
    inp->j = k;
    inp->has_unique = &st->has_unique;
    inp->set_difference = node_set_difference ? &node_set_difference[k] : NULL;
    inp->best_j_vec = &st->best_j_vec;
    inp->node_has_unique = st->node_has_unique;
    inp->distance = 0;                                 /* usher_graph.hpp:98-101 */
    inp->best_distance = &inp->distance;
    inp->imputed_mutations = NULL;
    inp->n_imputed = NULL;
    inp->excess_mutations = NULL;
    inp->n_excess = NULL;
}

/* One sample against the whole tree.  print_parsimony_scores != 0 reproduces
 * the -p mode (per-node scores in BFS order into node_set_difference[n], no
 * second pass, usher_common.cpp:409,413).  best_j_vec_out (optional, capacity
 * n) receives the BFS indices of all optimal nodes. */
int oracle_place_sample(const otree *T, int nS, const int32_t *s_pos, const uint8_t *s_ref,
                        const uint8_t *s_mut, const uint8_t *s_missing, int print_parsimony_scores,
                        int32_t *node_set_difference, oracle_result *out, uint32_t *best_j_vec_out) {
    size_t total_nodes = (size_t)T->n;
    omut *S = (omut *)calloc((size_t)(nS ? nS : 1), sizeof(omut));
    for (int i = 0; i < nS; i++) {
        S[i].position = s_pos[i];
        S[i].ref_nuc = (int8_t)s_ref[i];
        S[i].par_nuc = (int8_t)s_ref[i];               /* mutation_annotated_tree.cpp:2092 */
        S[i].mut_nuc = (int8_t)s_mut[i];
        S[i].is_missing = s_missing[i];
    }
    o_best_state st;
    st.best_node_num_leaves = 0;                       /* :364 */
    st.best_set_difference = nS + T->root->nmuts + 1;  /* :371 */
    st.best_j = 0;                                     /* :373 */
    st.has_unique = 0;                                 /* :374 */
    st.node_has_unique = (uint8_t *)calloc(total_nodes, 1);            /* :376 */
    st.best_j_vec.v = NULL; st.best_j_vec.n = 0; st.best_j_vec.cap = 0;
    jvec_push(&st.best_j_vec, 0);                      /* :378 */
    st.num_best = 1;                                   /* :380 */
    st.best_node = T->root;                            /* :381 */
    int *nsd = NULL;
    if (print_parsimony_scores) nsd = (int *)calloc(total_nodes, sizeof(int));   /* :360-362 */
    anc_vec scratch = {0, 0, 0, 0};

    for (size_t k = 0; k < total_nodes; k++) {         /* :386-411 */
        o_mapper2_input inp;
        fill_input(&inp, T, &st, S, nS, k, nsd);
        o_mapper2_body(&inp, print_parsimony_scores, &scratch);
    }
    if (!print_parsimony_scores) {                     /* :413-446 */
        st.best_set_difference += 1;
        size_t ntmp = st.best_j_vec.n;
        size_t *tmp_vec = (size_t *)malloc(sizeof(size_t) * (ntmp ? ntmp : 1));
        memcpy(tmp_vec, st.best_j_vec.v, sizeof(size_t) * ntmp);
        st.num_best = 0;
        st.best_j_vec.n = 0;
        for (size_t l = 0; l < ntmp; l++) {
            size_t k = tmp_vec[l];
            o_mapper2_input inp;
            fill_input(&inp, T, &st, S, nS, k, NULL);
            o_mapper2_body(&inp, 0, &scratch);
        }
        free(tmp_vec);
    }
    out->best_set_difference = st.best_set_difference;
    out->num_best = (uint32_t)st.num_best;
    out->best_j = (uint32_t)st.best_j;
    out->best_node_id = st.best_node->id;
    out->best_node_has_unique = (uint32_t)st.has_unique;
    if (best_j_vec_out)
        for (size_t i = 0; i < st.best_j_vec.n; i++) best_j_vec_out[i] = (uint32_t)st.best_j_vec.v[i];
    if (nsd) {
        if (node_set_difference)
            for (size_t k = 0; k < total_nodes; k++) node_set_difference[k] = nsd[k];
        free(nsd);
    }
    free(scratch.v); free(scratch.pos);
    free(st.best_j_vec.v); free(st.node_has_unique); free(S);
    return 0;
}

/* ---- batch driver (reads CSR) ------------------------------------------ *
 * Reads: read_off[R+1] into r_pos/r_ref/r_mut/r_missing.  Threads split the
 * READS (each read still runs the serial loop above), which is the same
 * total work as the reference's tbb::parallel_for over nodes
 * (usher_common.cpp:386) without its shared-state lock.  nthreads<=1 = serial. */
typedef struct {
    const otree *T;
    const uint32_t *read_off;
    const int32_t *r_pos;
    const uint8_t *r_ref, *r_mut, *r_missing;
    oracle_result *out;
    uint32_t lo, hi;
} batch_job;

static void *batch_worker(void *p) {
    batch_job *jb = (batch_job *)p;
    for (uint32_t r = jb->lo; r < jb->hi; r++) {
        uint32_t a = jb->read_off[r], b = jb->read_off[r + 1];
        oracle_place_sample(jb->T, (int)(b - a), jb->r_pos + a, jb->r_ref + a, jb->r_mut + a,
                            jb->r_missing + a, 0, NULL, &jb->out[r], NULL);
    }
    return NULL;
}

int oracle_place_batch(const otree *T, uint32_t n_reads, const uint32_t *read_off, const int32_t *r_pos,
                       const uint8_t *r_ref, const uint8_t *r_mut, const uint8_t *r_missing,
                       oracle_result *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n_reads) nthreads = n_reads ? (int)n_reads : 1;
    batch_job *jobs = (batch_job *)calloc((size_t)nthreads, sizeof(batch_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) {
        jobs[t].T = T; jobs[t].read_off = read_off; jobs[t].r_pos = r_pos;
        jobs[t].r_ref = r_ref; jobs[t].r_mut = r_mut; jobs[t].r_missing = r_missing; jobs[t].out = out;
        jobs[t].lo = (uint32_t)((uint64_t)n_reads * (uint64_t)t / (uint64_t)nthreads);
        jobs[t].hi = (uint32_t)((uint64_t)n_reads * (uint64_t)(t + 1) / (uint64_t)nthreads);
    }
    if (nthreads == 1) batch_worker(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    free(th); free(jobs);
    return 0;
}

/* ---- node-parallel driver: the reference's parallelisation ---------------- *
 * tbb::parallel_for over the node range (usher_common.cpp:386-411): every
 * thread owns a contiguous range of BFS indices and runs pass 1 for sample
 * after sample with a PRIVATE best state; the private states are merged with
 * the same rule mapper2_body applies under its lock (usher_mapper.cpp:466-498),
 * which gives the result of the serial loop (the result is independent of the
 * evaluation order).  Pass 2 (usher_common.cpp:413-446)
 * then runs serially over the merged best_j_vec, as in the reference.  Used as
 * the timed CPU baseline of bench.py. */
typedef struct {
    const otree *T;
    const uint32_t *read_off;
    const omut *S_all;
    uint32_t n_reads;
    size_t k_lo, k_hi;
    o_best_state *st;        /* [n_reads] private states of this thread */
} np_job;

static void np_state_init(o_best_state *st, const otree *T, int nS, uint8_t *nhu) {
    st->best_node_num_leaves = 0;
    st->best_set_difference = nS + T->root->nmuts + 1;
    st->best_j = 0;
    st->has_unique = 0;
    st->node_has_unique = nhu;
    st->best_j_vec.v = NULL; st->best_j_vec.n = 0; st->best_j_vec.cap = 0;
    jvec_push(&st->best_j_vec, 0);
    st->num_best = 1;
    st->best_node = T->root;
}

static void *np_worker(void *p) {
    np_job *jb = (np_job *)p;
    anc_vec scratch = {0, 0, 0, 0};
    uint8_t *nhu = (uint8_t *)calloc((size_t)jb->T->n, 1);
    for (uint32_t r = 0; r < jb->n_reads; r++) {
        const omut *S = jb->S_all + jb->read_off[r];
        int nS = (int)(jb->read_off[r + 1] - jb->read_off[r]);
        o_best_state *st = &jb->st[r];
        np_state_init(st, jb->T, nS, nhu);
        for (size_t k = jb->k_lo; k < jb->k_hi; k++) {
            o_mapper2_input inp;
            fill_input(&inp, jb->T, st, S, nS, k, NULL);
            o_mapper2_body(&inp, 0, &scratch);
        }
        st->node_has_unique = NULL;
    }
    free(nhu);
    free(scratch.v); free(scratch.pos);
    return NULL;
}

int oracle_place_batch_nodepar(const otree *T, uint32_t n_reads, const uint32_t *read_off, const int32_t *r_pos,
                               const uint8_t *r_ref, const uint8_t *r_mut, const uint8_t *r_missing,
                               oracle_result *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > T->n) nthreads = T->n;
    uint32_t nw = read_off[n_reads];
    omut *S_all = (omut *)calloc(nw ? nw : 1, sizeof(omut));
    for (uint32_t i = 0; i < nw; i++) {
        S_all[i].position = r_pos[i];
        S_all[i].ref_nuc = (int8_t)r_ref[i];
        S_all[i].par_nuc = (int8_t)r_ref[i];
        S_all[i].mut_nuc = (int8_t)r_mut[i];
        S_all[i].is_missing = r_missing[i];
    }
    np_job *jobs = (np_job *)calloc((size_t)nthreads, sizeof(np_job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) {
        jobs[t].T = T; jobs[t].read_off = read_off; jobs[t].S_all = S_all; jobs[t].n_reads = n_reads;
        jobs[t].k_lo = (size_t)((uint64_t)T->n * (uint64_t)t / (uint64_t)nthreads);
        jobs[t].k_hi = (size_t)((uint64_t)T->n * (uint64_t)(t + 1) / (uint64_t)nthreads);
        jobs[t].st = (o_best_state *)calloc(n_reads ? n_reads : 1, sizeof(o_best_state));
    }
    if (nthreads == 1) np_worker(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, np_worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    anc_vec scratch = {0, 0, 0, 0};
    uint8_t *nhu = (uint8_t *)calloc((size_t)T->n, 1);
    for (uint32_t r = 0; r < n_reads; r++) {
        const omut *S = S_all + read_off[r];
        int nS = (int)(read_off[r + 1] - read_off[r]);
        /* merge the private states (usher_mapper.cpp:466-498) */
        o_best_state m;
        np_state_init(&m, T, nS, nhu);
        for (int t = 0; t < nthreads; t++) {
            o_best_state *s = &jobs[t].st[r];
            /* a thread whose range never beat the initial value keeps the
             * initial (root, j=0) placeholder, which must not be merged */
            int is_initial = (s->best_set_difference == nS + T->root->nmuts + 1);
            if (!is_initial) {
                if (s->best_set_difference < m.best_set_difference) {
                    m.best_set_difference = s->best_set_difference;
                    m.best_node = s->best_node; m.best_node_num_leaves = s->best_node_num_leaves;
                    m.best_j = s->best_j; m.num_best = s->num_best; m.has_unique = s->has_unique;
                    m.best_j_vec.n = 0;
                    for (size_t i = 0; i < s->best_j_vec.n; i++) jvec_push(&m.best_j_vec, s->best_j_vec.v[i]);
                } else if (s->best_set_difference == m.best_set_difference) {
                    if ((s->best_node_num_leaves > m.best_node_num_leaves) ||
                        ((s->best_node_num_leaves == m.best_node_num_leaves) && (m.best_j < s->best_j))) {
                        m.best_node = s->best_node; m.best_node_num_leaves = s->best_node_num_leaves;
                        m.best_j = s->best_j; m.has_unique = s->has_unique;
                    }
                    m.num_best += s->num_best;
                    for (size_t i = 0; i < s->best_j_vec.n; i++) jvec_push(&m.best_j_vec, s->best_j_vec.v[i]);
                }
            }
            free(s->best_j_vec.v);
        }
        /* pass 2, usher_common.cpp:413-446 */
        m.best_set_difference += 1;
        size_t ntmp = m.best_j_vec.n;
        size_t *tmp_vec = (size_t *)malloc(sizeof(size_t) * (ntmp ? ntmp : 1));
        memcpy(tmp_vec, m.best_j_vec.v, sizeof(size_t) * ntmp);
        m.num_best = 0;
        m.best_j_vec.n = 0;
        for (size_t l = 0; l < ntmp; l++) {
            o_mapper2_input inp;
            fill_input(&inp, T, &m, S, nS, tmp_vec[l], NULL);
            o_mapper2_body(&inp, 0, &scratch);
        }
        free(tmp_vec);
        out[r].best_set_difference = m.best_set_difference;
        out[r].num_best = (uint32_t)m.num_best;
        out[r].best_j = (uint32_t)m.best_j;
        out[r].best_node_id = m.best_node->id;
        out[r].best_node_has_unique = (uint32_t)m.has_unique;
        free(m.best_j_vec.v);
    }
    free(nhu);
    free(scratch.v); free(scratch.pos);
    for (int t = 0; t < nthreads; t++) free(jobs[t].st);
    free(th); free(jobs); free(S_all);
    return 0;
}

/* node_imputed_mutations[best_j] as pass 2 fills it (usher_common.cpp:423-446 call
 * mapper2_body with compute_vecs = true): the imputed mutations of sample S at
 * the node with BFS index j.  out_pos / out_nuc need capacity nS; returns the count. */
int oracle_imputed_at_node(const otree *T, int nS, const int32_t *s_pos, const uint8_t *s_ref,
                           const uint8_t *s_mut, const uint8_t *s_missing, uint32_t j,
                           int32_t *out_pos, uint8_t *out_nuc) {
    omut *S = (omut *)calloc((size_t)(nS ? nS : 1), sizeof(omut));
    for (int i = 0; i < nS; i++) {
        S[i].position = s_pos[i]; S[i].ref_nuc = (int8_t)s_ref[i]; S[i].par_nuc = (int8_t)s_ref[i];
        S[i].mut_nuc = (int8_t)s_mut[i]; S[i].is_missing = s_missing[i];
    }
    o_best_state st;
    uint8_t *nhu = (uint8_t *)calloc((size_t)T->n, 1);
    np_state_init(&st, T, nS, nhu);
    st.best_set_difference = 0x3fffffff;       /* never return early: the node is an optimal one in pass 2 */
    omut *imp = (omut *)calloc((size_t)(nS ? nS : 1), sizeof(omut));
    int n_imp = 0;
    anc_vec scratch = {0, 0, 0, 0};
    o_mapper2_input inp;
    fill_input(&inp, T, &st, S, nS, j, NULL);
    inp.imputed_mutations = imp;
    inp.n_imputed = &n_imp;
    o_mapper2_body(&inp, 0, &scratch);
    for (int i = 0; i < n_imp; i++) { out_pos[i] = imp[i].position; out_nuc[i] = (uint8_t)imp[i].mut_nuc; }
    free(scratch.v); free(scratch.pos); free(imp); free(nhu); free(st.best_j_vec.v); free(S);
    return n_imp;
}

/* node_excess_mutations[j] with compute_vecs = true (usher_common.cpp:393,411 in -p mode,
 * :431,446 in pass 2), in the order mapper2_body appends: first the node's own mutations that
 * the sample shares (usher_mapper.cpp:223-228, :253-258 -- not parsimony-increasing, but that is
 * where the reference puts them), then the sample's own alleles the genotype does not offer
 * (:357-388, in sample order), then the back-mutations to the reference (:394-446, in position
 * order of the ancestral set).  Output capacity: nS + number of mutations on the root path. */
int oracle_excess_at_node(const otree *T, int nS, const int32_t *s_pos, const uint8_t *s_ref,
                          const uint8_t *s_mut, const uint8_t *s_missing, uint32_t j, int capacity,
                          int32_t *out_pos, uint8_t *out_ref, uint8_t *out_par, uint8_t *out_mut) {
    omut *S = (omut *)calloc((size_t)(nS ? nS : 1), sizeof(omut));
    for (int i = 0; i < nS; i++) {
        S[i].position = s_pos[i]; S[i].ref_nuc = (int8_t)s_ref[i]; S[i].par_nuc = (int8_t)s_ref[i];
        S[i].mut_nuc = (int8_t)s_mut[i]; S[i].is_missing = s_missing[i];
    }
    o_best_state st;
    uint8_t *nhu = (uint8_t *)calloc((size_t)T->n, 1);
    np_state_init(&st, T, nS, nhu);
    st.best_set_difference = 0x3fffffff;
    omut *imp = (omut *)calloc((size_t)(nS ? nS : 1), sizeof(omut));
    omut *exc = (omut *)calloc((size_t)(capacity > 0 ? capacity : 1), sizeof(omut));
    int n_imp = 0, n_exc = 0;
    anc_vec scratch = {0, 0, 0, 0};
    o_mapper2_input inp;
    fill_input(&inp, T, &st, S, nS, j, NULL);
    inp.imputed_mutations = imp;
    inp.n_imputed = &n_imp;
    inp.excess_mutations = exc;
    inp.n_excess = &n_exc;
    o_mapper2_body(&inp, 0, &scratch);
    for (int i = 0; i < n_exc; i++) {
        out_pos[i] = exc[i].position; out_ref[i] = (uint8_t)exc[i].ref_nuc; out_par[i] = (uint8_t)exc[i].par_nuc;
        out_mut[i] = (uint8_t)exc[i].mut_nuc;
    }
    free(scratch.v); free(scratch.pos); free(imp); free(exc); free(nhu); free(st.best_j_vec.v); free(S);
    return n_exc;
}

/* ======================================================================== *
 * mapper_body: per-site Fitch-Sankoff used when a MAT is built from a tree   *
 * and a VCF (usher_mapper.cpp:7-162, called from read_vcf with              *
 * create_new_mat, mutation_annotated_tree.cpp:1964).  One call = one VCF row.*
 * Restated for the tree samples only (variants of samples that are not in   *
 * the tree are appended to Missing_Sample lists at :64-83, which is not a   *
 * computation).                                                            *
 *   var_node[i]  caller id of a tree node named in the row                  *
 *   var_nuc[i]   its allele mask (:50, one bit per possible base)           *
 * Output: one (node id, par_nuc mask, mut_nuc mask) per mutation the site    *
 * adds (:145-156), in BFS order of the nodes.  Returns their number.         *
 * ======================================================================== */
int oracle_mapper_body(const otree *T, uint8_t ref_nuc, int n_var, const int32_t *var_node,
                       const uint8_t *var_nuc, int32_t *out_node, uint8_t *out_par, uint8_t *out_mut) {
    const int num_nodes = T->n;
    int *scores = (int *)calloc((size_t)num_nodes * 4, sizeof(int));        /* :22-32 */
    int8_t *states = (int8_t *)calloc((size_t)num_nodes, 1);
    int *bfs_idx = (int *)malloc(sizeof(int) * (size_t)num_nodes);          /* bfs_idx[caller id] */
    for (int i = 0; i < num_nodes; i++) bfs_idx[T->bfs[i]->id] = i;
    int ref_nuc_id = -1;                                                    /* MAT::get_nt, mat.cpp:142-162 */
    for (int j = 0; j < 4; j++) if (ref_nuc == (1 << j)) ref_nuc_id = j;
    if (ref_nuc_id < 0) { free(scores); free(states); free(bfs_idx); return -1; }
    for (int i = 0; i < num_nodes; i++) {                                   /* :36-45 leaves */
        if (!n_is_leaf(T->bfs[i])) continue;
        for (int j = 0; j < 4; j++) if (j != ref_nuc_id) scores[i * 4 + j] = num_nodes;
    }
    for (int v = 0; v < n_var; v++) {                                       /* :48-63 */
        int idx = bfs_idx[var_node[v]];
        for (int j = 0; j < 4; j++) {
            scores[idx * 4 + j] = num_nodes;
            if (((1 << j) & var_nuc[v]) != 0) scores[idx * 4 + j] = 0;
        }
    }
    for (int i = num_nodes - 1; i >= 0; i--) {                              /* :87-112 forward pass */
        onode *node = T->bfs[i];
        if (n_is_leaf(node)) continue;
        for (int c = 0; c < node->nchildren; c++) {
            int c_idx = bfs_idx[node->children[c]->id];
            for (int j = 0; j < 4; j++) {
                int min_s = num_nodes + 1;
                for (int k = 0; k < 4; k++) {
                    int c_s = (k == j) ? scores[c_idx * 4 + k] : scores[c_idx * 4 + k] + 1;
                    if (c_s < min_s) min_s = c_s;
                }
                scores[i * 4 + j] += min_s;
            }
        }
    }
    int n_out = 0;
    for (int i = 0; i < num_nodes; i++) {                                   /* :115-157 backward pass */
        onode *node = T->bfs[i];
        int8_t par_state = node->parent ? states[bfs_idx[node->parent->id]] : (int8_t)ref_nuc_id;
        int8_t state = par_state;
        int min_s = scores[i * 4 + par_state];
        for (int j = 0; j < 4; j++)
            if (scores[i * 4 + j] < min_s) { min_s = scores[i * 4 + j]; state = (int8_t)j; }
        if (state != par_state && scores[i * 4 + par_state] == min_s) state = par_state;
        states[i] = state;
        if (state != par_state) {
            out_node[n_out] = node->id;
            out_par[n_out] = (uint8_t)(1 << par_state);
            out_mut[n_out] = (uint8_t)(1 << state);
            n_out++;
        }
    }
    free(scores); free(states); free(bfs_idx);
    return n_out;
}
