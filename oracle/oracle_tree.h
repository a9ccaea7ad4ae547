/*
 * oracle/oracle_tree.h -- the pointer tree shared by the oracle's translation units.
 * TEST INFRASTRUCTURE ONLY (see mapper2_oracle.c).
 */
#ifndef WEPP_ORACLE_TREE_H
#define WEPP_ORACLE_TREE_H
#include <stdint.h>

typedef struct {
    int position;
    int8_t ref_nuc, par_nuc, mut_nuc;
    uint8_t is_missing;
} omut;

typedef struct onode {
    struct onode *parent;
    struct onode **children;
    int nchildren;
    omut *muts;
    int nmuts;
    int id;              /* caller's node id */
    int64_t num_leaves;  /* memo, -1 = not computed */
    int dfs_idx;
} onode;

typedef struct {
    int n;
    onode *nodes;   /* indexed by caller id */
    onode *root;
    onode **bfs;    /* breadth_first_expansion() */
    onode **dfs;    /* depth_first_expansion()   */
    omut *mut_pool;
    onode **child_pool;
} otree;

static inline int m_is_masked(const omut *m) { return m->position < 0; }
static inline int n_is_leaf(const onode *n) { return n->nchildren == 0; }
static inline int n_is_root(const onode *n) { return n->parent == NULL; }

otree *oracle_tree_build(int n, const int32_t *parent, const uint32_t *mut_off, const int32_t *mut_pos,
                         const uint8_t *mut_ref, const uint8_t *mut_par, const uint8_t *mut_mut);
void oracle_tree_free(otree *t);

#endif
