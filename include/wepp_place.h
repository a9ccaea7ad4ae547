/*
 * wepp_place.h -- C-ABI of the MI355X read-placement engine.
 *
 * Drop-in boundary for the reference's per-sample node loop.  The reference
 * (TurakhiaLab/WEPP, vendored UShER sources) has no FFI layer; its seam is the
 * C++ call
 *     void mapper2_body(mapper2_input&, bool, bool)          src/usher_graph.hpp:104
 * invoked from the tbb::parallel_for bodies of
 *     int usher_common(...)                                  src/usher_common.hpp:18-22
 *                                                            src/usher_common.cpp:386-411 (pass 1)
 *                                                            src/usher_common.cpp:413-446 (pass 2)
 * This library replaces BOTH passes for a whole batch of samples/reads with one
 * call (wepp_place_batch); INTEGRATION.md shows the few lines a maintainer adds
 * to usher_common.cpp to bind it.
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every host buffer;
 *  - every function returns 0 on success or a WEPP_E* code; the message is
 *    available from wepp_last_error() (thread-local).  Nothing here calls
 *    exit() or throws across the boundary (the reference's MAT layer exit(1)s,
 *    src/mutation_annotated_tree.cpp:474,514,533);
 *  - nucleotides are the reference's 4-bit one-hot/IUPAC masks A=1 C=2 G=4 T=8,
 *    N=15 (src/mutation_annotated_tree.cpp:19-74);
 *  - a handle is bound to one device and is not re-entrant: its workspace, routing
 *    counters and staging buffers belong to one call at a time, so the calls on one
 *    handle must be serial AND, for wepp_place_batch_device, ordered on ONE stream
 *    (or separated by a synchronisation).  Different handles -- on different GPUs or
 *    on the same one -- may be used concurrently from different host threads
 *    (tests/test_gpu_parity.py::test_two_handles_two_host_threads).
 */
#ifndef WEPP_PLACE_H
#define WEPP_PLACE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WEPP_OK          0
#define WEPP_EINVAL      1   /* malformed argument / precondition violated      */
#define WEPP_ENOMEM      2   /* host or device allocation failed                */
#define WEPP_EDEVICE     3   /* HIP runtime error (no device, launch failure..) */
#define WEPP_ELIMIT      4   /* input exceeds a documented limit                */

/* ---- packed 32-bit words ---------------------------------------------- *
 * Sample/read entry = one element of Missing_Sample::mutations
 * (src/usher_graph.hpp:34-54, filled by src/mutation_annotated_tree.cpp:2086-2127):
 *     bits  0..19  position (0 .. 2^20-2)
 *     bits 20..23  ref_nuc mask
 *     bits 24..27  mut_nuc mask (15 for N)
 *     bit  28      is_missing
 */
#define WEPP_MAX_POSITION 0xFFFFEu
static inline uint32_t wepp_pack_read_word(uint32_t position, uint32_t ref_nuc, uint32_t mut_nuc,
                                           uint32_t is_missing) {
    return (position & 0xFFFFFu) | ((ref_nuc & 15u) << 20) | ((mut_nuc & 15u) << 24) |
           ((is_missing & 1u) << 28);
}

/* ---- the tree as the caller hands it over ------------------------------ *
 * A pointer-free description of MAT::Tree (src/mutation_annotated_tree.hpp:80-152):
 * node ids 0..n_nodes-1 in any order, parent[i] = id of the parent or -1 for
 * the single root; the children of a node are ordered by ascending id (the
 * order Tree::create_node pushes them, src/mutation_annotated_tree.cpp:865-878).
 * Node i owns mutations mut_*[mut_off[i] .. mut_off[i+1]), sorted by position
 * (the loader guarantees it, src/mutation_annotated_tree.cpp:591-594), ref_nuc
 * one-hot (src/mutation_annotated_tree.cpp:572) and the same at every mutation
 * of a position;
 * mut_pos < 0 = masked mutation (src/mutation_annotated_tree.hpp:68-70).
 * mut_par (Mutation::par_nuc) may be NULL: the scorer never reads it
 * (src/usher_mapper.cpp:168-506 only copies it), the flattener recomputes the
 * true parent allele. */
typedef struct {
    uint32_t n_nodes;
    const int32_t *parent;     /* [n_nodes]                */
    const uint32_t *mut_off;   /* [n_nodes + 1]            */
    const int32_t *mut_pos;    /* [mut_off[n_nodes]]       */
    const uint8_t *mut_ref;    /* ref_nuc masks            */
    const uint8_t *mut_par;    /* par_nuc masks or NULL    */
    const uint8_t *mut_mut;    /* mut_nuc masks            */
} wepp_tree_desc;

typedef struct wepp_mat wepp_mat_t;   /* flattened MAT resident in one GPU's HBM */

typedef struct {
    uint64_t n_nodes;          /* N                                             */
    uint64_t n_mutations;      /* M: non-masked mutation words                  */
    uint64_t n_masked;         /* masked mutations (kept as node flags only)    */
    uint64_t n_events;         /* E: enter + exit events of the sweep stream    */
    uint64_t n_blocks;         /* sweep blocks (<=64 nodes, <=128 events each)  */
    uint64_t n_leaves;
    uint32_t max_depth;        /* root = 0                                      */
    uint32_t max_position;     /* largest mutated position                      */
    uint64_t stream_bytes;     /* bytes one sweep of the whole-tree stream reads */
    uint64_t device_bytes;     /* total HBM held by the handle                  */
    /* sweep streams: crowns {static score <= tau} closed under ancestors, in
     * increasing tau; the last one is the whole tree (tau = INT32_MAX) */
    uint32_t n_streams;
    int32_t  stream_tau[16];
    uint64_t stream_nodes[16];
    uint64_t stream_bytes_of[16];
    /* window crowns (see wepp_mat_last_crowns): how many were built, their nodes in all */
    uint32_t n_window_crowns;
    uint64_t window_crown_nodes;
    /* window streams (plan class WEPP_PLAN_WIN: tiles of reads with many entries inside one genome window): how many
     * of them are the window's CANDIDATE crown -- the nodes n with out_w(n) - in_w(n) <= score of the root for an
     * empty read, the only ones a read confined to the window can be placed on whatever it lists -- rather than the
     * whole tree as the window sees it, and the elements of all window streams */
    uint32_t n_window_streams;
    uint32_t n_window_streams_crown;
    uint64_t window_stream_nodes;
    /* genome windows: WIN_SIZE positions every WIN_STRIDE (flatmat.hpp; 2560 / 1024 in the product build), at most 32
     * of them: positions from 32 * stride on lie in no window -- window_uncovered_positions of the tree's mutated
     * positions -- and reads there take the tree-wide streams (a SARS-CoV-2 or RSV genome has none) */
    uint32_t window_size, window_stride, window_uncovered_positions;
    /* seed signatures (wepp_mat_set_use_seeds): chunks of the whole-tree stream, blocks per chunk, bytes of the
     * (position x chunk) nibble table; 0 when none was built (genomes beyond 2^18 positions, tables beyond 1 GB) */
    uint32_t seed_chunks, seed_chunk_blocks;
    uint64_t seed_sig_bytes;
} wepp_mat_stats;

/* Per-read result flags (out parameter `flags`). */
#define WEPP_FLAG_HAS_UNIQUE 1u  /* best_node_has_unique, src/usher_common.cpp:374,401 */

/* Build the flat MAT from `tree` and upload it to HIP device `device`.
 * Replaces: MAT::Tree::breadth_first_expansion() per sample
 * (src/usher_common.cpp:339), Tree::get_num_leaves() per tie
 * (src/usher_mapper.cpp:465) and the ancestor walk of every mapper2_body call
 * (src/usher_mapper.cpp:276-287) by a one-time precomputation. */
int wepp_mat_create(const wepp_tree_desc *tree, int device, wepp_mat_t **out);
int wepp_mat_destroy(wepp_mat_t *mat);
int wepp_mat_get_stats(const wepp_mat_t *mat, wepp_mat_stats *out);
/* bfs_ids[k] = caller node id of bfs[k] (breadth_first_expansion order,
 * src/mutation_annotated_tree.cpp:1115-1141); buffer of n_nodes entries. */
int wepp_mat_bfs_order(const wepp_mat_t *mat, uint32_t *bfs_ids);

/* Place a batch of samples/reads: for read r with entries
 * read_word[read_off[r] .. read_off[r+1]) (sorted by position, positions
 * unique -- the precondition of the merge at src/usher_mapper.cpp:205-243)
 * compute what the two passes at src/usher_common.cpp:386-446 leave in
 *     best_j               -> best_bfs_j[r]   (index into the BFS order)
 *     best_set_difference  -> score[r]
 *     num_best             -> num_best[r]
 *     best_node_has_unique -> flags[r] & WEPP_FLAG_HAS_UNIQUE
 * Host buffers in, host buffers out (H2D / D2H inside); synchronous: the
 * results are in the caller's buffers on return.  Batches of 65 536 reads and
 * more run as a pipeline of 2-4 sub-batches (contiguous ranges of the reads):
 * host worker threads the handle starts at the first such call and keeps
 * (wepp_amd/csrc/host_pool.hpp) check and stage the next sub-batch while the
 * previous one's words go up, the one before runs its kernels and the results
 * of the one before that come down and are moved out.  Buffers the caller has
 * pinned (hipHostMalloc / hipHostRegister) are the source / target of the DMA
 * as they stand: no staging pass.  Results never depend on the split.  Any
 * output pointer may be NULL.  per_node_scores, when non-NULL, receives n_reads * n_nodes
 * int32 values: the -p mode's node_set_difference[k] in BFS order
 * (src/usher_common.cpp:403-409, +1 for ineligible nodes src/usher_mapper.cpp:500-505). */
int wepp_place_batch(wepp_mat_t *mat, const uint32_t *read_off, const uint32_t *read_word,
                     uint32_t n_reads, uint32_t *best_bfs_j, int32_t *score, uint32_t *num_best,
                     uint32_t *flags, int32_t *per_node_scores);

/* Imputed mutations of the chosen placement: what pass 2 leaves in
 * node_imputed_mutations[best_j] (src/usher_mapper.cpp:323-378) and
 * usher_common prints as column 4 of placement_stats.tsv
 * (src/usher_common.cpp:764-781).  Every ambiguous (more than one bit set),
 * non-missing entry of a read yields exactly one imputed mutation, in entry
 * order; imp_off[r] .. imp_off[r+1] index imp_pos / imp_nuc (nucleotide masks,
 * one bit set).  best_bfs_j is the placement returned by wepp_place_batch.
 * capacity = size of imp_pos / imp_nuc; WEPP_ELIMIT if too small (the needed
 * size is the number of ambiguous non-missing read words). */
int wepp_imputed_mutations(wepp_mat_t *mat, const uint32_t *read_off, const uint32_t *read_word,
                           uint32_t n_reads, const uint32_t *best_bfs_j, uint32_t *imp_off,
                           int32_t *imp_pos, uint8_t *imp_nuc, uint64_t capacity);

/* best_j_vec: the BFS indices of ALL optimal nodes of every read -- what pass 1 collects for pass 2
 * (src/usher_common.cpp:376-381, filled at src/usher_mapper.cpp:475-476,497; pass 2 loops over it at
 * src/usher_common.cpp:413-446).  `score` / `num_best` are the outputs of wepp_place_batch for the same reads on
 * this handle; best_off[n_reads + 1] receives the CSR over the reads (best_off[r + 1] - best_off[r] = num_best[r]),
 * best_nodes[best_off[r] .. best_off[r + 1]) the optimal nodes of read r in ascending BFS index (the reference's
 * vector is unordered).  capacity = size of best_nodes; WEPP_ELIMIT (best_off filled: best_off[n_reads] is the
 * needed size) when it is too small; WEPP_EINVAL when score / num_best are not this read's placement.
 * Cost: every node of a read's own stream (the crown its bound admits, wepp_mat_stats::stream_nodes) is evaluated
 * once -- a second pass for batches of samples, not for every read of a sequencing run. */
int wepp_best_nodes(wepp_mat_t *mat, const uint32_t *read_off, const uint32_t *read_word, uint32_t n_reads,
                    const int32_t *score, const uint32_t *num_best, uint64_t *best_off, uint32_t *best_nodes,
                    uint64_t capacity);

/* ---- excess mutations of (sample, node) pairs --------------------------------------- *
 * node_excess_mutations[j] as mapper2_body appends it with compute_vecs for sample
 * pair_read[i] at the node with BFS index pair_bfs_j[i]: first the node's own mutations the
 * sample shares (src/usher_mapper.cpp:223-228, :253-258), then the sample's own alleles the
 * node's genotype does not offer (:357-388, sample order), then back-mutations to the
 * reference (:394-446, position order).  usher prints the first `score` entries for the
 * optimal nodes in the last column of parsimony-scores.tsv (src/usher_common.cpp:555-574).
 * exc_off[n_pairs + 1] is the CSR over the pairs; entry q is
 * MAT::Mutation{position exc_pos[q], ref_nuc exc_ref[q], par_nuc exc_par[q], mut_nuc exc_mut[q]};
 * par_nuc of a node's own mutation is the true parent allele (what Mutation::par_nuc holds
 * in a consistent MAT).  WEPP_ELIMIT (exc_off filled with the needed sizes) when capacity is
 * too small.  Not listed: masked root mutations that carry non-zero nucleotides (the .pb
 * loader zeroes them, src/mutation_annotated_tree.cpp:566-571). */
int wepp_excess_mutations(wepp_mat_t *mat, const uint32_t *read_off, const uint32_t *read_word, uint32_t n_reads,
                          uint32_t n_pairs, const uint32_t *pair_read, const uint32_t *pair_bfs_j,
                          uint64_t *exc_off, int32_t *exc_pos, uint8_t *exc_ref, uint8_t *exc_par, uint8_t *exc_mut,
                          uint64_t capacity);

/* Same computation with every buffer already resident on the handle's device
 * (device pointers); work is enqueued on `hip_stream` (a hipStream_t, NULL =
 * the default stream).  n_read_words = read_off[n_reads].  Used when the caller
 * keeps reads in HBM.
 * Synchronisation: the call BLOCKS the host until the two routing kernels (~20 us per
 * 1 M reads, plus whatever precedes them on the stream) have finished -- the sweep
 * grids are sized from their counters --, then enqueues the sweeps and returns
 * WITHOUT waiting for them: the results are valid once `hip_stream` has reached the
 * point of return.  Work the caller wants overlapped with the routing (the next
 * batch's H2D) goes on another stream. */
int wepp_place_batch_device(wepp_mat_t *mat, const uint32_t *d_read_off, const uint32_t *d_read_word,
                            uint32_t n_reads, uint64_t n_read_words, uint32_t *d_best_bfs_j,
                            int32_t *d_score, uint32_t *d_num_best, uint32_t *d_flags,
                            void *hip_stream);

/* Tuning knob: reads that share one sweep of the event stream (1..64,
 * default 64).  Affects speed only, never results. */
int wepp_mat_set_tile_reads(wepp_mat_t *mat, uint32_t reads_per_tile);
/* Work skipping on (default) / off: when off every read sweeps the whole-tree
 * stream.  Affects speed only, never results. */
int wepp_mat_set_use_crowns(wepp_mat_t *mat, int enable);
/* Per-read walks on (default) / off: when on, a read that lists at most 16 positions visits only the
 * events of those positions (position index + range queries over the stream) instead of sweeping the
 * whole stream with 63 other reads; when off every read is placed by a sweep.  Affects speed only,
 * never results.  WEPP_ELIMIT when enabling on a tree with a stream of 2^25 nodes or more (the walk's interval
 * stack packs a subtree end into 25 bits; such a tree is placed by sweeps). */
int wepp_mat_set_use_walk(wepp_mat_t *mat, int enable);

/* Seeded placement of whole-genome samples on (default) / off.  A sample that lists more positions than a walk takes
 * and fits no genome window -- what read_vcf makes of a consensus genome, src/mutation_annotated_tree.cpp:2033-2130 --
 * is placed by a workgroup of its own: for every chunk (about a thousand nodes) of the whole-tree stream it counts how
 * many of the sample's reference-excluding alleles the chunk's signature -- the alleles carried by the chunk's nodes
 * and the ancestors of its first node -- could serve; a node of the chunk scores at least (such entries) - (count),
 * so only the chunks whose bound reaches the best score found so far are evaluated, node by node, exactly
 * (wepp_amd/csrc/seed_kernels.hip).  Off: such samples share tile sweeps of the whole-tree stream.  Affects speed
 * only, never results. */
int wepp_mat_set_use_seeds(wepp_mat_t *mat, int enable);
/* Diagnostic: samples seeded by the handle since wepp_mat_timing_reset(), the chunks they evaluated, the chunks they
 * could have evaluated (samples x wepp_mat_stats::seed_chunks), the most chunks one sample evaluated, and the samples
 * by chunks evaluated (histogram8: <= 1, <= 4, <= 16, <= 64, <= 256, <= 1024, <= 4096, more).  Any pointer may be
 * NULL.  Synchronises the device. */
int wepp_mat_last_seeds(wepp_mat_t *mat, uint64_t *samples, uint64_t *chunks_evaluated, uint64_t *chunks_total,
                        uint64_t *most_per_sample, uint64_t *histogram8);

/* Tuning knob: sub-batches wepp_place_batch cuts a batch of 65 536 reads or more into (1..8; 0 = default:
 * 4 from 262 144 reads, 2 below).  Affects speed only, never results. */
int wepp_mat_set_pipeline(wepp_mat_t *mat, uint32_t sub_batches);

/* Timing of the dominant kernel (k_sweep), measured with HIP events recorded on
 * the launch stream around the sweep (+ the 10-80 us finalize) launches of every
 * placement call -- they run concurrently on internal side streams forked from
 * and joined into the launch stream -- since the
 * handle was created or wepp_mat_timing_reset() was called (the most recent 64
 * calls are kept).  Blocks until those launches have finished.
 * mean_sweep_ms = mean duration of the sweep / walk launches of one call; passes =
 * event-stream sweeps of a call (one per tile) plus the 64-read waves of its walks, averaged like the bytes;
 * algorithmic_bytes = bytes the sweeps read (sum over tiles of their stream's size) plus what the walks' lanes
 * asked memory for, counted by the kernel (a 32-byte index entry per node event, the sparse-table bytes and
 * range-query aggregates actually requested, list heads, read words, the start states of the jobs, results),
 * averaged over the calls since the reset. */
int wepp_mat_timing_reset(wepp_mat_t *mat);
int wepp_mat_last_timing(wepp_mat_t *mat, float *mean_sweep_ms, uint32_t *n_calls, uint64_t *passes,
                         uint64_t *algorithmic_bytes);

/* Diagnostic: tiers[r] = index of the sweep stream (wepp_mat_stats::stream_tau) read r of
 * the handle's most recent placement call was routed to; n_reads must be that call's.
 * Synchronises the device.  Lets a test reach every stream with the oracle. */
int wepp_mat_last_tiers(wepp_mat_t *mat, uint8_t *tiers, uint32_t n_reads);

/* Diagnostic: the full plan of every read of the handle's most recent placement call: plan_class[r] = how it was
 * placed (WEPP_PLAN_*), plan_stream[r] = on which stream (index into wepp_mat_stats::stream_tau; for
 * WEPP_PLAN_WIN the index of the genome window).  n_reads must be that call's.  Synchronises the device.  Lets a
 * test reach every (class, stream) pair with the oracle. */
#define WEPP_PLAN_WALK8   0   /* per-read walk, up to 8 listed positions                  */
#define WEPP_PLAN_WALK16  1   /* per-read walk, up to 16                                   */
#define WEPP_PLAN_SWEEP   2   /* sweep of the stream with up to 63 other reads             */
#define WEPP_PLAN_WALKC8  3   /* walk cut into jobs (many events), up to 8 positions       */
#define WEPP_PLAN_WALKC16 4   /* walk cut into jobs, up to 16 positions                    */
#define WEPP_PLAN_WIN     5   /* tile sweep of a genome window's stream (reads with more than 32 entries inside one window) */
#define WEPP_PLAN_SEED    6   /* whole-genome sample: chunk signatures, then exact evaluation of the chunks left (plan_stream 0) */
int wepp_mat_last_plans(wepp_mat_t *mat, uint8_t *plan_class, uint8_t *plan_stream, uint32_t n_reads);
/* plan_stream == WEPP_WINDOW_CROWN_SLOT with a class other than WEPP_PLAN_WIN: the read walked (or, class
 * WEPP_PLAN_SWEEP, swept alone) a WINDOW CROWN -- of the nodes a read confined to its genome window can be placed on at
 * all (at least as many of the path's mutations inside the window as outside), those whose score can be as low as the
 * read's root score (positions outside the window cost every such read the same): far fewer than the tree-wide crown
 * of root score + entries.  wepp_mat_last_crowns tells which for the reads that walked: window[r] = index of the genome
 * window, crown[r] = index of the crown among the window's (increasing bound; the last one admits any root score);
 * 255 / 255 for a read placed otherwise.  (For class WEPP_PLAN_WIN plan_stream is the window's index, which may be 15.) */
#define WEPP_WINDOW_CROWN_SLOT 15
int wepp_mat_last_crowns(wepp_mat_t *mat, uint8_t *window, uint8_t *crown, uint32_t n_reads);

/* Diagnostic: reads of the handle's most recent placement call that walked their own events (k_walk;
 * the others were placed by sweeps of their stream), and the loop iterations (one per event, interval end or
 * range query) all walks of the handle ran since wepp_mat_timing_reset().  Synchronises the device. */
int wepp_mat_last_walk(wepp_mat_t *mat, uint64_t *reads_walked, uint64_t *walk_iterations);

const char *wepp_last_error(void);

/* ---- synthetic workload generators (host only; no GPU needed) ---------- *
 * The reference bundles no MAT or reads; every config is
 * generated deterministically from seeds with a splitmix64 PRNG. */
typedef struct wepp_gen_tree wepp_gen_tree_t;
typedef struct wepp_gen_reads wepp_gen_reads_t;

typedef struct {
    uint64_t seed;
    uint32_t n_nodes;
    uint32_t genome_len;        /* positions 1..genome_len                      */
    double   p_recent_parent;   /* prob. parent is one of the last 8 nodes      */
    double   zipf_s;            /* site-weight exponent (homoplasy skew)        */
    double   p_back_mutation;   /* prob. a mutation at a non-ref site reverts   */
    double   p_ambiguous;       /* prob. a node mutation gets a 2-bit mut_nuc   */
    double   p_masked_node;     /* prob. a non-root node gets a masked mutation */
    uint32_t root_mutations;    /* mutations placed on the root                 */
    /* tree shape (0 = the default shape: uniform attachment, median root path ~20 mutations at 16 M nodes) */
    uint32_t depth_choices;     /* parent = the DEEPEST of this many uniformly drawn earlier nodes: 2 -> paths of ~2x
                                 * the mutations, 16 -> ~5x (real SARS-CoV-2 paths carry 60-100+)              */
    double   p_hub;             /* prob. the parent is one of the tree's hubs instead: polytomies ...           */
    uint32_t n_hubs;            /* ... the first n_hubs nodes (0: n_nodes / 4096 + 1), each ~p_hub * n / n_hubs children */
} wepp_gen_tree_params;

/* shape of a generated tree: mutations on the root path of its leaves, depth, largest polytomy */
typedef struct {
    uint32_t n_nodes, n_leaves, max_depth, max_children;
    uint32_t path_mutations_median, path_mutations_p95, path_mutations_max;
    double   path_mutations_mean, mutations_per_node;
} wepp_gen_tree_shape;

typedef struct {
    uint64_t seed;
    uint32_t n_reads;
    uint32_t read_len;          /* 150 (ARTIC) or ~1200 (midnight)              */
    uint32_t amplicon_len;      /* 400 (ARTIC-like) or 1200 (midnight-like)     */
    uint32_t amplicon_step;     /* tiling step between amplicon starts          */
    double   p_substitution;    /* per-base sequencing error                    */
    double   p_n;               /* per-base N                                   */
    double   p_iupac;           /* prob. an error is an ambiguity code instead  */
} wepp_gen_reads_params;

int wepp_gen_tree_create(const wepp_gen_tree_params *p, wepp_gen_tree_t **out);
int wepp_gen_tree_desc(const wepp_gen_tree_t *t, wepp_tree_desc *out);  /* borrowed pointers */
int wepp_gen_tree_get_shape(const wepp_gen_tree_t *t, wepp_gen_tree_shape *out);
int wepp_gen_tree_destroy(wepp_gen_tree_t *t);

int wepp_gen_reads_create(const wepp_gen_tree_t *t, const wepp_gen_reads_params *p, wepp_gen_reads_t **out);
/* borrowed pointers: read_off[n_reads+1], read_word[read_off[n_reads]] */
int wepp_gen_reads_get(const wepp_gen_reads_t *r, uint32_t *n_reads, const uint32_t **read_off,
                       const uint32_t **read_word);
/* borrowed pointers: the 1-based inclusive genome window [start, end] every read covers
 * (raw_read::start / end for wepp_epp_map) */
int wepp_gen_reads_windows(const wepp_gen_reads_t *r, const int32_t **start, const int32_t **end);
int wepp_gen_reads_destroy(wepp_gen_reads_t *r);

/* ---- per-site Fitch-Sankoff: building a MAT from a tree and a VCF ---------- *
 * Replaces mapper_body::operator() (src/usher_mapper.cpp:7-162), which read_vcf
 * runs once per VCF row when create_new_mat is set
 * (src/mutation_annotated_tree.cpp:1907-2031).  Row s has reference base
 * site_ref[s] (one-hot mask) and names the tree samples
 * var_node[var_off[s] .. var_off[s+1]) (caller node ids) whose allele masks
 * var_nuc[..] differ from the reference or are ambiguous; every other leaf
 * carries the reference base.  The mutation lists of `tree` are ignored.
 * Output: the mutations mapper_body would add (:145-156), as (row, node id,
 * par_nuc mask, mut_nuc mask), rows in order and BFS order inside a row.
 * *n_out receives their number; WEPP_ELIMIT (with *n_out set) when capacity is
 * too small.  Limits: tree depth <= 140, < 2^28 nodes. */
int wepp_fitch_sites(const wepp_tree_desc *tree, int device, uint32_t n_sites, const uint8_t *site_ref,
                     const uint32_t *var_off, const uint32_t *var_node, const uint8_t *var_nuc,
                     uint64_t capacity, uint64_t *n_out, uint32_t *out_site, uint32_t *out_node,
                     uint8_t *out_par, uint8_t *out_mut);

/* The same with the tree-dependent work done once: read_vcf runs mapper_body row after row on ONE tree
 * (src/mutation_annotated_tree.cpp:1962-2031); a plan keeps the flattened topology, the level tables and the
 * decision-table memory on the device, so that repeated calls (batches of VCF rows) only pay for their rows.
 * wepp_fitch_sites == create + run + destroy.  A plan is bound to one device and is not re-entrant. */
typedef struct wepp_fitch_plan wepp_fitch_plan_t;
int wepp_fitch_plan_create(const wepp_tree_desc *tree, int device, wepp_fitch_plan_t **out);
int wepp_fitch_plan_run(wepp_fitch_plan_t *plan, uint32_t n_sites, const uint8_t *site_ref, const uint32_t *var_off,
                        const uint32_t *var_node, const uint8_t *var_nuc, uint64_t capacity, uint64_t *n_out,
                        uint32_t *out_site, uint32_t *out_node, uint8_t *out_par, uint8_t *out_mut);
int wepp_fitch_plan_destroy(wepp_fitch_plan_t *plan);
/* wall time of the calling thread's last wepp_fitch_plan_run by phase (ms): host preparation of the rows, uploads,
 * kernels, sort + decode + copy-out of the mutations */
int wepp_fitch_last_timing(double *prep_ms, double *upload_ms, double *kernels_ms, double *output_ms);

/* ---- WEPP's own read placement: EPP sets and haplotype scores -------------- *
 * Replaces wepp_filter::cartesian_map (src/WEPP/initial_filter.cpp:140-239) with
 * single_read_tree (:41-136), the range trees it walks (src/WEPP/arena.cpp:68-169)
 * and the per-read distance haplotype::mutation_distance (src/WEPP/haplotype.hpp:
 * 123-173).  `mat` is created from the CONDENSED tree (create_condensed_tree,
 * src/WEPP/util.cpp:79-133): one node per haplotype; haplotypes are addressed by
 * their arena index = pre-order index (arena::from_mat, arena.cpp:3-55), which
 * wepp_mat_dfs_order maps to caller node ids.
 * A read is a raw_read (src/WEPP/read.hpp:8-14): genome window [start, end]
 * (1-based, inclusive), multiplicity `degree`, and one packed word
 * (wepp_pack_read_word, mut_nuc 15 = N) per position where it differs from the
 * reference, sorted by position.
 * Outputs (all host buffers):
 *   max_parsimony[R]   wepp_filter::max_parismony: smallest windowed distance (:203)
 *   multiplicity[R]    parsimony_multiplicity: number of haplotypes attaining it (:204)
 *   epp_off/epp_nodes  epp_positions_cache: ascending arena indices of those
 *                      haplotypes for every read with multiplicity <= max_cached_epp
 *                      (MAX_CACHED_EPP_SIZE = 2048, config.hpp:9), CSR over the reads;
 *                      both may be NULL
 *   hap_score[N]       haplotype::score = sum over reads of
 *                      degree / ((1 + parsimony) * multiplicity) (initial_filter.hpp:54-57);
 *                      accumulated in 64-bit fixed point: run-to-run identical, exact zeros,
 *                      absolute error <= (reads mapped to the haplotype) * 2^-(42 - log2(sum of degrees))
 *   hap_read_counts[N*50]  haplotype::mapped_read_counts (bin = min(start / (genome_size / 50), 49)); may be NULL
 *   hap_divergence[N]  haplotype::dist_divergence (:224-233); may be NULL
 * The final sort of the haplotypes by score (:236) is left to the caller.
 * Preconditions (WEPP_EINVAL otherwise): listed alleles differ from the reference base,
 * positions sorted and unique, 1 <= start <= end, genome_size >= 50. */
#define WEPP_NUM_RANGE_BINS 50
#define WEPP_MAX_CACHED_EPP_SIZE 2048
typedef struct {
    uint32_t n_reads;
    const uint32_t *read_off;   /* [n_reads + 1] */
    const uint32_t *read_word;  /* [read_off[n_reads]] */
    const int32_t *start;       /* [n_reads] raw_read::start */
    const int32_t *end;         /* [n_reads] raw_read::end   */
    const int32_t *degree;      /* [n_reads] raw_read::degree */
} wepp_epp_reads;
typedef struct {
    int32_t *max_parsimony;
    uint32_t *multiplicity;
    uint64_t *epp_off;          /* [n_reads + 1] or NULL */
    uint32_t *epp_nodes;        /* [epp_capacity] or NULL */
    uint64_t epp_capacity;
    double *hap_score;
    int32_t *hap_read_counts;   /* node-major [N][50] or NULL */
    double *hap_divergence;     /* or NULL */
} wepp_epp_out;
int wepp_epp_map(wepp_mat_t *mat, const wepp_epp_reads *reads, uint32_t genome_size, uint32_t max_cached_epp,
                 wepp_epp_out *out);
/* The size of the EPP lists is only known once the map has run (the sum of the multiplicities <= max_cached_epp).
 * When epp_nodes is too small (epp_capacity < epp_off[n_reads]; epp_capacity = 0 with epp_nodes = NULL asks for
 * exactly that) wepp_epp_map still delivers EVERY other output, keeps the lists on the handle and returns
 * WEPP_ELIMIT; the caller sizes its buffer from epp_off[n_reads] and collects them here -- the map never runs twice.
 * The pending lists belong to the handle's most recent wepp_epp_map and are dropped by the fetch. */
int wepp_epp_fetch_lists(wepp_mat_t *mat, uint32_t *epp_nodes, uint64_t capacity);
/* caller node id of the haplotype with arena (pre-order) index k, for k = 0 .. n_nodes-1 */
int wepp_mat_dfs_order(const wepp_mat_t *mat, uint32_t *ids);
/* device time of the calling thread's last wepp_epp_map, by phase (HIP events), and its work:
 * events_swept = sum over tiles of the window-stream events one tile walks per pass */
int wepp_epp_last_timing(double *select_ms, double *sweep1_ms, double *sweep2_ms, double *finish_ms,
                         uint64_t *events_swept, uint64_t *stream_events, uint32_t *groups, uint32_t *jobs);

/* ---- host-side introspection of the flattened MAT (no GPU needed) -------- *
 * Lets the CPU test-suite check the flattener (orders, parent alleles, per-node
 * constants, event stream) against the oracle.  `name` is one of: node_woff,
 * words, nkey, nstat, rank2dfs, dfs2bfs, bfs2id, dfs2id, parent_dfs, dfs_end,
 * num_leaves, blk_node0, blk_eoff, blk_sum, ev_word, ev_meta, ev_lb, cp_off, cp_word, seed_sig (the signature
 * table: rows of wepp_mat_stats::seed_chunks nibbles, 8 to a dword, padded to 16 bytes).
 * Stream fields (nkey, nstat, blk_*, ev_*, cp_*) take an optional "<i>:" prefix
 * selecting sweep stream i (default: the whole-tree stream).
 * The returned pointer is borrowed from the handle; *elem_bytes is the element
 * size and *count the number of elements. */
typedef struct wepp_flat wepp_flat_t;
int wepp_flat_create(const wepp_tree_desc *tree, wepp_flat_t **out);
int wepp_flat_get(const wepp_flat_t *flat, const char *name, const void **data, uint64_t *count,
                  uint32_t *elem_bytes);
int wepp_flat_scalars(const wepp_flat_t *flat, wepp_mat_stats *stats, uint32_t *cp_stride);
int wepp_flat_destroy(wepp_flat_t *flat);
/* The device half of wepp_mat_create on its own: uploads an image built by wepp_flat_create to HIP device `device`.
 * The image is only read, so ONE flatten serves every GPU of a node (and several handles on one GPU): the
 * multi-GPU host loop calls wepp_flat_create once and wepp_mat_upload from each device's host thread, where the
 * reference re-expands the tree per sample (src/usher_common.cpp:339).  wepp_mat_create == flatten + upload.
 * The image may be destroyed as soon as the uploads have returned. */
int wepp_mat_upload(const wepp_flat_t *flat, int device, wepp_mat_t **out);
/* The flat image as a file: one flatten per NODE when the ranks are processes of their own (one per GPU under
 * torch.distributed.run, MPI, ...): rank 0 calls wepp_flat_create + wepp_flat_save (to /dev/shm), the other ranks
 * wepp_flat_load + wepp_mat_upload -- seconds instead of a flatten each.  A cache bound to this build of the library
 * (the header pins its layout constants; WEPP_EINVAL otherwise), not an exchange format.  The file is written under
 * a temporary name and renamed: a reader never sees a partial image. */
int wepp_flat_save(const wepp_flat_t *flat, const char *path);
int wepp_flat_load(const char *path, wepp_flat_t **out);
/* Diagnostic: full flattens (wepp_mat_create, wepp_flat_create) this process has run. */
uint64_t wepp_debug_flatten_count(void);

#ifdef __cplusplus
}
#endif
#endif /* WEPP_PLACE_H */
