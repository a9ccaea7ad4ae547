// usher_place.hpp -- the placement half of `usher_common` on the GPU.
//
// Mirrors /root/reference/src/usher_common.hpp:18-22 for the --no-add path
// (the tree is not modified, usher_common.cpp:649): same argument names and
// meaning for the ones that matter here, same files written into `outdir`
// (placement_stats.tsv, and parsimony-scores.tsv with print_parsimony_scores),
// same stderr lines, same return convention (0 ok, 1 error after printing).
// Everything between is one wepp_place_batch call (include/wepp_place.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mat.hpp"

struct usher_place_result {       // what the reference keeps in locals per sample
    int best_set_difference;      // usher_common.cpp:371
    size_t num_best;              // :380
    size_t best_j;                // :373  (index into breadth_first_expansion())
    MAT::Node* best_node;         // :381
    bool best_node_has_unique;    // :374
};

// Wall time of the phases of the calling process's last usher_place_samples (seconds): tree -> flat description,
// the ONE flatten of the node (wepp_flat_create), the slowest device thread's upload (wepp_mat_upload) and its
// placement + pass-2 calls, the writers.  `flattens` = full flattens the call ran (1, whatever the device count).
struct usher_place_timing {
    double describe_s = 0, flatten_s = 0, upload_s = 0, place_s = 0, write_s = 0;
    uint64_t flattens = 0;
};
extern usher_place_timing usher_last_timing;

// Places every sample of `missing_samples` (in order) on `T`.
//   outdir                   directory for the TSV files ("" = write none)
//   max_uncertainty / max_parsimony   thresholds of the warnings at :453-466
//   print_parsimony_scores   the -p mode (:328-336, :403-409, :555-574)
//   low_confidence_samples   receives the samples with >1 optimal placement (:453-456)
//   results                  optional, one entry per placed sample, in output order
//   device                   HIP device index
//   sort_before_placement_1/2/3, reverse_sort   the sample orderings of usher_common.cpp:140-155
//                            (3: by number of ambiguous bases, reorders missing_samples) and
//                            :184-298 (1: by score then number of optimal placements, 2: the
//                            other way round); with --no-add they only change the row order
int usher_place_samples(std::string outdir, uint32_t max_uncertainty, uint32_t max_parsimony,
                        bool print_parsimony_scores, std::vector<Missing_Sample>& missing_samples,
                        std::vector<std::string>& low_confidence_samples, MAT::Tree* T,
                        std::vector<usher_place_result>* results = nullptr, int device = 0,
                        bool sort_before_placement_1 = false, bool sort_before_placement_2 = false,
                        bool sort_before_placement_3 = false, bool reverse_sort = false);
// The same over several GPUs of one node: one host thread and one handle per entry of `devices`
// (HIP device indices; an index may repeat), device g placing the contiguous range of samples
// [R*g/G, R*(g+1)/G) -- the per-sample work of usher_common.cpp:386-446 is independent between
// samples, so the ranges never exchange anything and the rows are written in the usual order.
int usher_place_samples(std::string outdir, uint32_t max_uncertainty, uint32_t max_parsimony,
                        bool print_parsimony_scores, std::vector<Missing_Sample>& missing_samples,
                        std::vector<std::string>& low_confidence_samples, MAT::Tree* T,
                        const std::vector<int>& devices, std::vector<usher_place_result>* results = nullptr,
                        bool sort_before_placement_1 = false, bool sort_before_placement_2 = false,
                        bool sort_before_placement_3 = false, bool reverse_sort = false);
