/*
 * oracle/epp_oracle.c -- CPU restatement of WEPP's own read placement: the windowed
 * parsimony of every read against every haplotype of the (condensed) tree, the set of
 * equally parsimonious placements (EPPs) and the haplotype scores accumulated from them.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under wepp_amd/ may include, link or call this file;
 * only tests/ and the cpu_baseline leg of the measurement tools use it, as the checker.
 *
 * PARITY UNPINNED: the reference ships no tests or golden vectors for this path and its
 * sources do not compile in this image (TBB / Boost / protobuf headers are absent), see
 * mapper2_oracle.c.  Every function below restates the cited reference lines literally,
 * including the range trees the reference uses to skip work, and haplotype::
 * mutation_distance -- the reference's second, independent formulation of the same
 * distance -- is restated too so that the tests can check one against the other.
 *
 * What is restated (file:line relative to /root/reference/src/WEPP):
 *   hap_from_mat()          arena.cpp:3-55      arena::from_mat (pre-order arena, muts, stack_muts)
 *   build_range_tree()      arena.cpp:68-93     arena::build_range_tree
 *   build_range_trees()     arena.cpp:95-152    window choice + true_read_counts
 *   find_range_tree_for()   arena.cpp:154-169
 *   single_read_tree()      initial_filter.cpp:41-105 (recursion) and :112-136 (driver)
 *   oracle_epp_map()        initial_filter.cpp:140-239  wepp_filter::cartesian_map
 *                           (scores, per-bin read counts, divergence, max_parismony,
 *                           parsimony_multiplicity, epp_positions_cache) without the final sort
 *   oracle_epp_distance()   haplotype.hpp:123-173  haplotype::mutation_distance(comp, min, max)
 *   node_score              initial_filter.hpp:54-57
 *   constants               config.hpp:9,13,14,18
 */
#include <limits.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_tree.h"

#define NUM_RANGE_BINS 50               /* config.hpp:13 */
#define NUM_RANGE_TREES 25              /* config.hpp:14 */
#define MAX_CACHED_EPP_SIZE 2048        /* config.hpp:9  */
#define READ_DIST_FACTOR_THRESHOLD (0.5 / 100) /* config.hpp:18 */
#define NUC_N 15

typedef struct {
    int start, end, degree;
    const omut *muts;
    int nmuts;
} oread;

typedef struct ohap {
    struct ohap *parent;
    const onode *src;       /* condensed_source */
    omut *stack_muts;       /* genotype: non-reference positions, sorted */
    int n_stack;
    const omut *muts;       /* the node's own mutations, sorted */
    int n_muts;
    struct ohap **children;
    int nchildren;
    int mapped;
    double score;
    int mapped_read_counts[NUM_RANGE_BINS];
} ohap;

typedef struct {
    ohap *root;             /* first haplotype with a mutation in the window */
    int *sources;           /* arena indices */
    int nsources, csources;
    int *children;          /* indices into ranged_nodes */
    int nchildren, cchildren;
} omulti;

typedef struct {
    const otree *T;
    ohap *nodes;            /* arena, pre-order */
    int n;
    ohap **child_pool;
    omulti *ranged;
    int nranged, cranged;
    /* ranged_root_map: (start, end) -> index of the window's root, kept sorted */
    int (*win)[3];
    int nwin;
    int true_read_counts[NUM_RANGE_BINS];
    int genome_size;
} oarena;

static int cmp_omut_pos(const void *a, const void *b) {
    const omut *x = (const omut *)a, *y = (const omut *)b;
    return (x->position > y->position) - (x->position < y->position);
}

/* arena.cpp:3-55, iterative pre-order (children in stored order) */
static void hap_from_mat(oarena *A) {
    const otree *T = A->T;
    A->n = T->n;
    A->nodes = (ohap *)calloc((size_t)T->n, sizeof(ohap));
    A->child_pool = (ohap **)calloc((size_t)T->n, sizeof(ohap *));
    size_t coff = 0;
    for (int d = 0; d < T->n; d++) {
        const onode *node = T->dfs[d];
        ohap *ret = &A->nodes[d];
        ohap *parent = node->parent ? &A->nodes[node->parent->dfs_idx] : NULL;
        ret->parent = parent;
        ret->src = node;
        ret->children = A->child_pool + coff;
        coff += (size_t)node->nchildren;
        if (parent) parent->children[parent->nchildren++] = ret;
        int cap = (parent ? parent->n_stack : 0) + node->nmuts;
        ret->stack_muts = (omut *)malloc(sizeof(omut) * (size_t)(cap ? cap : 1));
        ret->n_stack = 0;
        if (parent) {                                               /* :17-37 */
            for (int i = 0; i < parent->n_stack; i++) {
                int valid = 1;
                for (int k = 0; k < node->nmuts; k++)
                    if (node->muts[k].position == parent->stack_muts[i].position) { valid = 0; break; }
                if (valid) ret->stack_muts[ret->n_stack++] = parent->stack_muts[i];
            }
        }
        for (int k = 0; k < node->nmuts; k++)                        /* :39-45 */
            if (node->muts[k].ref_nuc != node->muts[k].mut_nuc) ret->stack_muts[ret->n_stack++] = node->muts[k];
        omut *own = (omut *)malloc(sizeof(omut) * (size_t)(node->nmuts ? node->nmuts : 1));
        memcpy(own, node->muts, sizeof(omut) * (size_t)node->nmuts);
        qsort(ret->stack_muts, (size_t)ret->n_stack, sizeof(omut), cmp_omut_pos);   /* :47-48 */
        qsort(own, (size_t)node->nmuts, sizeof(omut), cmp_omut_pos);
        ret->muts = own;
        ret->n_muts = node->nmuts;
    }
}

/* haplotype.hpp:51-58: lower_bound over the node's own mutations */
static int has_mutations_in_range(const ohap *h, int start, int end) {
    int lo = 0, hi = h->n_muts;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (h->muts[mid].position < start) lo = mid + 1; else hi = mid;
    }
    return lo != h->n_muts && h->muts[lo].position <= end;
}

static int ranged_new(oarena *A) {
    if (A->nranged == A->cranged) {
        A->cranged = A->cranged ? A->cranged * 2 : 1024;
        A->ranged = (omulti *)realloc(A->ranged, sizeof(omulti) * (size_t)A->cranged);
    }
    memset(&A->ranged[A->nranged], 0, sizeof(omulti));
    return A->nranged++;
}
static void ipush(int **v, int *n, int *c, int x) {
    if (*n == *c) { *c = *c ? *c * 2 : 4; *v = (int *)realloc(*v, sizeof(int) * (size_t)*c); }
    (*v)[(*n)++] = x;
}

/* arena.cpp:68-93, recursion replaced by an explicit stack of (haplotype, ranged parent) */
static int build_range_tree(oarena *A, int start, int end) {
    typedef struct { ohap *h; int parent; } frame;
    frame *st = (frame *)malloc(sizeof(frame) * (size_t)(A->n + 1));
    int sp = 0, root_ret = -1;
    st[sp].h = &A->nodes[0]; st[sp].parent = -1; sp++;
    while (sp) {
        frame f = st[--sp];
        int ret;
        if (f.parent == -1 || has_mutations_in_range(f.h, start, end)) {
            ret = ranged_new(A);
            A->ranged[ret].root = f.h;
            ipush(&A->ranged[ret].sources, &A->ranged[ret].nsources, &A->ranged[ret].csources, (int)(f.h - A->nodes));
            if (f.parent != -1)
                ipush(&A->ranged[f.parent].children, &A->ranged[f.parent].nchildren, &A->ranged[f.parent].cchildren, ret);
            else root_ret = ret;
        } else {
            ret = f.parent;
            ipush(&A->ranged[ret].sources, &A->ranged[ret].nsources, &A->ranged[ret].csources, (int)(f.h - A->nodes));
        }
        /* children pushed in reverse so that they are visited in stored order */
        for (int k = f.h->nchildren - 1; k >= 0; k--) { st[sp].h = f.h->children[k]; st[sp].parent = ret; sp++; }
    }
    free(st);
    return root_ret;
}

static int cmp_range(const void *a, const void *b) {
    const int *x = (const int *)a, *y = (const int *)b;
    if (x[0] != y[0]) return (x[0] > y[0]) - (x[0] < y[0]);
    return (x[1] > y[1]) - (x[1] < y[1]);
}

/* arena.cpp:95-152 */
static void build_range_trees(oarena *A, const oread *reads, int n_reads) {
    int (*rr)[2] = (int (*)[2])malloc(sizeof(int[2]) * (size_t)(n_reads ? n_reads : 1));
    for (int i = 0; i < n_reads; i++) { rr[i][0] = reads[i].start; rr[i][1] = reads[i].end; }
    qsort(rr, (size_t)n_reads, sizeof(int[2]), cmp_range);
    int num = n_reads < NUM_RANGE_TREES ? n_reads : NUM_RANGE_TREES;
    A->win = (int (*)[3])malloc(sizeof(int[3]) * (size_t)(num ? num : 1));
    A->nwin = 0;
    for (int i = 0; i < num; i++) {
        int s = (int)((size_t)n_reads * (size_t)i / (size_t)num);
        int e = (int)((size_t)n_reads * (size_t)(i + 1) / (size_t)num);
        int read_start = rr[s][0], read_end = 0;
        for (int j = s; j < e; j++) if (rr[j][1] > read_end) read_end = rr[j][1];
        int found = 0;
        for (int k = 0; k < A->nwin; k++) if (A->win[k][0] == read_start && A->win[k][1] == read_end) found = 1;
        if (!found) {
            A->win[A->nwin][0] = read_start; A->win[A->nwin][1] = read_end;
            A->win[A->nwin][2] = build_range_tree(A, read_start, read_end);
            A->nwin++;
        }
    }
    /* std::map order */
    for (int i = 1; i < A->nwin; i++)
        for (int j = i; j > 0 && cmp_range(A->win[j - 1], A->win[j]) > 0; j--) {
            int t0 = A->win[j][0], t1 = A->win[j][1], t2 = A->win[j][2];
            memcpy(A->win[j], A->win[j - 1], sizeof(int[3]));
            A->win[j - 1][0] = t0; A->win[j - 1][1] = t1; A->win[j - 1][2] = t2;
        }
    free(rr);
    memset(A->true_read_counts, 0, sizeof(A->true_read_counts));
    int bin = A->genome_size / NUM_RANGE_BINS;                      /* :140 */
    for (int i = 0; i < n_reads; i++) {
        int b = reads[i].start / bin;
        if (b > NUM_RANGE_BINS - 1) b = NUM_RANGE_BINS - 1;
        A->true_read_counts[b] += reads[i].degree;
    }
}

/* arena.cpp:154-169: upper_bound({start, INT_MAX}) then walk back to a window that holds the read */
static omulti *find_range_tree_for(oarena *A, const oread *read) {
    int it = 0;
    while (it < A->nwin && A->win[it][0] <= read->start) it++;      /* first key > (start, INT_MAX) */
    do {
        it--;
        if (it < 0) return NULL;
        if (read->start >= A->win[it][0] && read->end <= A->win[it][1]) return &A->ranged[A->win[it][2]];
    } while (it != 0);
    return NULL;
}

/* lower_bound over the read's mutations by position (MAT::Mutation::operator<, mat.hpp:51-53) */
static uint8_t read_nuc_at(const oread *read, const omut *j) {
    int lo = 0, hi = read->nmuts;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (read->muts[mid].position < j->position) lo = mid + 1; else hi = mid;
    }
    return (lo == read->nmuts || read->muts[lo].position != j->position) ? (uint8_t)j->ref_nuc : (uint8_t)read->muts[lo].mut_nuc;
}

typedef struct { int *v; int n, c; } ivec;

/* initial_filter.cpp:41-105.  The recursion carries the parent's list of mismatching
 * positions; here an explicit stack of (node, depth) keeps one list per depth. */
static void single_read_tree_rec(oarena *A, const ivec *root_locs, omulti *root, const oread *read, ivec *max_nodes,
                                 int *max_val) {
    typedef struct { int node; int depth; } frame;
    int fcap = 1024, sp = 0, lcap = 64;
    frame *st = (frame *)malloc(sizeof(frame) * (size_t)fcap);
    ivec *locs = (ivec *)calloc((size_t)lcap, sizeof(ivec));
    /* locs[0] = the caller's parent_locations */
    locs[0].v = (int *)malloc(sizeof(int) * (size_t)(root_locs->n ? root_locs->n : 1));
    memcpy(locs[0].v, root_locs->v, sizeof(int) * (size_t)root_locs->n);
    locs[0].n = locs[0].c = root_locs->n;
    st[sp].node = (int)(root - A->ranged); st[sp].depth = 1; sp++;
    while (sp) {
        frame f = st[--sp];
        if (f.depth >= lcap) {
            locs = (ivec *)realloc(locs, sizeof(ivec) * (size_t)(lcap * 2));
            memset(locs + lcap, 0, sizeof(ivec) * (size_t)lcap);
            lcap *= 2;
        }
        omulti *curr = &A->ranged[f.node];
        const ivec *parent_locations = &locs[f.depth - 1];
        ivec *my = &locs[f.depth];
        my->n = 0;
        const omut *cm = curr->root->muts;
        int ncm = curr->root->n_muts;
        int i = 0, j;
        {   /* :54-56 lower_bound by read.start */
            int lo = 0, hi = ncm;
            while (lo < hi) { int mid = (lo + hi) / 2; if (cm[mid].position < read->start) lo = mid + 1; else hi = mid; }
            j = lo;
        }
        while (i < parent_locations->n || (j != ncm && cm[j].position <= read->end)) {      /* :60 */
            int parent_first = i < parent_locations->n &&
                               (j == ncm || cm[j].position > read->end || parent_locations->v[i] < cm[j].position);
            int us_first = (j != ncm && cm[j].position <= read->end) &&
                           (i == parent_locations->n || cm[j].position < parent_locations->v[i]);
            if (us_first) {                                                                  /* :64-72 */
                uint8_t rn = read_nuc_at(read, &cm[j]);
                if (rn != NUC_N && rn != (uint8_t)cm[j].mut_nuc) ipush(&my->v, &my->n, &my->c, cm[j].position);
                ++j;
            } else if (parent_first) {                                                       /* :73-77 */
                ipush(&my->v, &my->n, &my->c, parent_locations->v[i]);
                ++i;
            } else {                                                                         /* :78-87 */
                uint8_t rn = read_nuc_at(read, &cm[j]);
                if (rn != NUC_N && rn != (uint8_t)cm[j].mut_nuc) ipush(&my->v, &my->n, &my->c, cm[j].position);
                ++j; ++i;
            }
        }
        int parsimony = my->n;                                                               /* :90-99 */
        if (parsimony < *max_val) { *max_val = parsimony; max_nodes->n = 0; ipush(&max_nodes->v, &max_nodes->n, &max_nodes->c, f.node); }
        else if (parsimony == *max_val) ipush(&max_nodes->v, &max_nodes->n, &max_nodes->c, f.node);
        if (sp + curr->nchildren > fcap) {
            while (sp + curr->nchildren > fcap) fcap *= 2;
            st = (frame *)realloc(st, sizeof(frame) * (size_t)fcap);
        }
        for (int k = curr->nchildren - 1; k >= 0; k--) { st[sp].node = curr->children[k]; st[sp].depth = f.depth + 1; sp++; }
    }
    for (int k = 0; k < lcap; k++) free(locs[k].v);
    free(locs);
    free(st);
}

/* initial_filter.cpp:112-136 */
static int single_read_tree(oarena *A, const oread *read, ivec *max_indices, int *max_val) {
    ivec max_nodes = {0, 0, 0}, root_mutations = {0, 0, 0};
    omulti *root = find_range_tree_for(A, read);
    if (!root) return -1;                                           /* assert(0) in the reference */
    for (int k = 0; k < read->nmuts; k++)
        if ((uint8_t)read->muts[k].mut_nuc != NUC_N) ipush(&root_mutations.v, &root_mutations.n, &root_mutations.c, read->muts[k].position);
    single_read_tree_rec(A, &root_mutations, root, read, &max_nodes, max_val);
    for (int k = 0; k < max_nodes.n; k++) {
        omulti *rnode = &A->ranged[max_nodes.v[k]];
        for (int s = 0; s < rnode->nsources; s++)
            if (!A->nodes[rnode->sources[s]].mapped) ipush(&max_indices->v, &max_indices->n, &max_indices->c, rnode->sources[s]);
    }
    free(max_nodes.v);
    free(root_mutations.v);
    return 0;
}

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

static oarena *arena_new(const otree *T, int genome_size) {
    oarena *A = (oarena *)calloc(1, sizeof(oarena));
    A->T = T;
    A->genome_size = genome_size;
    hap_from_mat(A);
    return A;
}
static void arena_free(oarena *A) {
    for (int i = 0; i < A->n; i++) { free(A->nodes[i].stack_muts); free((void *)A->nodes[i].muts); }
    for (int i = 0; i < A->nranged; i++) { free(A->ranged[i].sources); free(A->ranged[i].children); }
    free(A->ranged); free(A->win); free(A->child_pool); free(A->nodes); free(A);
}

static void fill_reads(oread *reads, omut *pool, uint32_t n_reads, const uint32_t *read_off, const int32_t *r_pos,
                       const uint8_t *r_ref, const uint8_t *r_mut, const int32_t *r_start, const int32_t *r_end,
                       const int32_t *r_degree) {
    for (uint32_t k = 0; k < read_off[n_reads]; k++) {
        pool[k].position = r_pos[k]; pool[k].ref_nuc = (int8_t)r_ref[k]; pool[k].par_nuc = (int8_t)r_ref[k];
        pool[k].mut_nuc = (int8_t)r_mut[k]; pool[k].is_missing = r_mut[k] == NUC_N;
    }
    for (uint32_t r = 0; r < n_reads; r++) {
        reads[r].start = r_start[r]; reads[r].end = r_end[r]; reads[r].degree = r_degree[r];
        reads[r].muts = pool + read_off[r]; reads[r].nmuts = (int)(read_off[r + 1] - read_off[r]);
    }
}

/*
 * wepp_filter::cartesian_map, initial_filter.cpp:140-239 (serial; the reference's per-thread
 * partial sums only change the order of the floating-point additions).
 * Haplotype indices are arena indices = pre-order (depth_first_expansion) indices.
 * Outputs: max_parsimony[R], multiplicity[R]; epp_off[R+1]/epp_nodes: the sorted EPP list of
 * every read whose multiplicity is <= MAX_CACHED_EPP_SIZE (others stay empty, :205-210);
 * hap_score[N], hap_counts[N*50], hap_divergence[N] (:224-233).  node_mapped may be NULL.
 * Returns 0, -1 when a read fits no window (assert in the reference), -2 on bad arguments.
 */
/* the per-read part of the loop below (single_read_tree only reads the arena) for reads [lo, hi), strided over threads */
typedef struct { oarena *A; const oread *reads; uint32_t lo, hi; int tid, nthreads; ivec *idx; int *val; int rc; } epp_job;
static void *epp_worker(void *p) {
    epp_job *j = (epp_job *)p;
    for (uint32_t r = j->lo + (uint32_t)j->tid; r < j->hi; r += (uint32_t)j->nthreads) {
        ivec *v = &j->idx[r - j->lo];
        v->n = 0;
        j->val[r - j->lo] = INT32_MAX;
        if (single_read_tree(j->A, &j->reads[r], v, &j->val[r - j->lo]) != 0) j->rc = -1;
    }
    return NULL;
}

/* nthreads > 1 (test infrastructure at full size: a 16 M-node arena places ~4 reads/s on one core): the reads of a
 * block of `nthreads` are walked in parallel, their results folded in in read order -- the same additions in the same
 * order as the serial loop. */
int oracle_epp_map_mt(const otree *T, int genome_size, uint32_t n_reads, const uint32_t *read_off, const int32_t *r_pos,
                      const uint8_t *r_ref, const uint8_t *r_mut, const int32_t *r_start, const int32_t *r_end,
                      const int32_t *r_degree, const uint8_t *node_mapped, int32_t *max_parsimony,
                      uint32_t *multiplicity, uint64_t *epp_off, uint32_t *epp_nodes, uint64_t epp_capacity,
                      double *hap_score, int32_t *hap_counts, double *hap_divergence, int nthreads) {
    if (genome_size < NUM_RANGE_BINS) return -2;
    if (nthreads < 1) nthreads = 1;
    oarena *A = arena_new(T, genome_size);
    oread *reads = (oread *)calloc(n_reads ? n_reads : 1, sizeof(oread));
    omut *pool = (omut *)calloc(read_off[n_reads] ? read_off[n_reads] : 1, sizeof(omut));
    fill_reads(reads, pool, n_reads, read_off, r_pos, r_ref, r_mut, r_start, r_end, r_degree);
    if (node_mapped) for (int i = 0; i < A->n; i++) A->nodes[i].mapped = node_mapped[i];
    build_range_trees(A, reads, (int)n_reads);
    int bin_size = genome_size / NUM_RANGE_BINS;                    /* :148 */
    int rc = 0;
    uint64_t ecur = 0;
    epp_off[0] = 0;
    ivec *blk_idx = (ivec *)calloc((size_t)nthreads, sizeof(ivec));
    int *blk_val = (int *)calloc((size_t)nthreads, sizeof(int));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    epp_job *jobs = (epp_job *)calloc((size_t)nthreads, sizeof(epp_job));
    for (uint32_t r = 0; r < n_reads && rc == 0; r++) {
        if (r % (uint32_t)nthreads == 0) {
            const uint32_t hi = r + (uint32_t)nthreads < n_reads ? r + (uint32_t)nthreads : n_reads;
            for (int t = 0; t < nthreads; t++) jobs[t] = (epp_job){A, reads, r, hi, t, nthreads, blk_idx, blk_val, 0};
            if (nthreads == 1) epp_worker(&jobs[0]);
            else {
                for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, epp_worker, &jobs[t]);
                for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
            }
            for (int t = 0; t < nthreads; t++) if (jobs[t].rc) rc = -1;
            if (rc) break;
        }
        ivec max_indices = blk_idx[r % (uint32_t)nthreads];
        const int max_val = blk_val[r % (uint32_t)nthreads];
        double delta = (double)reads[r].degree / ((1 + max_val) * max_indices.n);   /* initial_filter.hpp:54-57 */
        int bucket = reads[r].start / bin_size;
        if (bucket > NUM_RANGE_BINS - 1) bucket = NUM_RANGE_BINS - 1;
        for (int k = 0; k < max_indices.n; k++) {                    /* :172-180 */
            ohap *h = &A->nodes[max_indices.v[k]];
            h->score += delta;
            h->mapped_read_counts[bucket] += reads[r].degree;
        }
        max_parsimony[r] = max_val;                                  /* :203-204 */
        multiplicity[r] = (uint32_t)max_indices.n;
        if (max_indices.n <= MAX_CACHED_EPP_SIZE) {                  /* :205-210 */
            qsort(max_indices.v, (size_t)max_indices.n, sizeof(int), cmp_int);
            for (int k = 0; k < max_indices.n; k++) {
                if (ecur >= epp_capacity) { rc = -2; break; }
                epp_nodes[ecur++] = (uint32_t)max_indices.v[k];
            }
        }
        epp_off[r + 1] = ecur;
    }
    for (int i = 0; i < A->n && rc == 0; i++) {                      /* :224-233 */
        int divergence = 0, bins_active = 0;
        for (int j = 0; j < NUM_RANGE_BINS; j++) {
            if (A->true_read_counts[j]) bins_active += 1;
            double proportion = (double)A->nodes[i].mapped_read_counts[j] / A->true_read_counts[j];
            if (proportion > READ_DIST_FACTOR_THRESHOLD) divergence += 1;
        }
        hap_score[i] = A->nodes[i].score;
        if (hap_divergence) hap_divergence[i] = (double)divergence / bins_active;
        if (hap_counts) memcpy(hap_counts + (size_t)i * NUM_RANGE_BINS, A->nodes[i].mapped_read_counts, sizeof(int) * NUM_RANGE_BINS);
    }
    for (int t = 0; t < nthreads; t++) free(blk_idx[t].v);
    free(blk_idx); free(blk_val); free(th); free(jobs); free(pool); free(reads);
    arena_free(A);
    return rc;
}

int oracle_epp_map(const otree *T, int genome_size, uint32_t n_reads, const uint32_t *read_off, const int32_t *r_pos,
                   const uint8_t *r_ref, const uint8_t *r_mut, const int32_t *r_start, const int32_t *r_end,
                   const int32_t *r_degree, const uint8_t *node_mapped, int32_t *max_parsimony,
                   uint32_t *multiplicity, uint64_t *epp_off, uint32_t *epp_nodes, uint64_t epp_capacity,
                   double *hap_score, int32_t *hap_counts, double *hap_divergence) {
    return oracle_epp_map_mt(T, genome_size, n_reads, read_off, r_pos, r_ref, r_mut, r_start, r_end, r_degree, node_mapped,
                             max_parsimony, multiplicity, epp_off, epp_nodes, epp_capacity, hap_score, hap_counts, hap_divergence, 1);
}

/* haplotype.hpp:123-173: mutation_distance(comp = read.mutations, min_pos = start, max_pos = end) */
static int mutation_distance(const ohap *h, const omut *comp, int ncomp, int min_pos, int max_pos) {
    int muts = 0;
    const int unknown_nuc = 15;
    int i, last_i, j = 0;
    { int lo = 0, hi = h->n_stack; while (lo < hi) { int mid = (lo + hi) / 2; if (h->stack_muts[mid].position < min_pos) lo = mid + 1; else hi = mid; } i = lo; }
    { int lo = 0, hi = h->n_stack; while (lo < hi) { int mid = (lo + hi) / 2; if (!(max_pos < h->stack_muts[mid].position)) lo = mid + 1; else hi = mid; } last_i = lo; }
    while (i < last_i || j < ncomp) {
        if (i == last_i) { if (comp[j].mut_nuc != unknown_nuc) ++muts; ++j; }
        else if (h->stack_muts[i].position < min_pos) ++i;
        else if (h->stack_muts[i].position > max_pos) return muts;
        else if (j == ncomp) { ++muts; ++i; }
        else if (h->stack_muts[i].position < comp[j].position) { ++muts; ++i; }
        else if (h->stack_muts[i].position > comp[j].position) { if (comp[j].mut_nuc != unknown_nuc) ++muts; ++j; }
        else if (h->stack_muts[i].position == comp[j].position && h->stack_muts[i].mut_nuc != comp[j].mut_nuc &&
                 comp[j].mut_nuc != unknown_nuc) { ++muts; ++i; ++j; }
        else { ++i; ++j; }
    }
    return muts;
}

/* distance of ONE read to every haplotype (arena order) through haplotype::mutation_distance */
int oracle_epp_distance(const otree *T, int nS, const int32_t *s_pos, const uint8_t *s_ref, const uint8_t *s_mut,
                        int start, int end, int32_t *out) {
    oarena *A = arena_new(T, NUM_RANGE_BINS);
    omut *S = (omut *)calloc((size_t)(nS ? nS : 1), sizeof(omut));
    for (int k = 0; k < nS; k++) { S[k].position = s_pos[k]; S[k].ref_nuc = (int8_t)s_ref[k]; S[k].mut_nuc = (int8_t)s_mut[k]; }
    for (int i = 0; i < A->n; i++) out[i] = mutation_distance(&A->nodes[i], S, nS, start, end);
    free(S);
    arena_free(A);
    return 0;
}
