// rnamc_internal.h — declarations shared by the host and device translation
// units of librnamc.so.  Not part of the public ABI.
#ifndef RNAMC_INTERNAL_H
#define RNAMC_INTERNAL_H

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rnamc.h"

namespace rnamc {

void set_last_error(const std::string& msg);
bool is_canonical(int a, int b);

// DP matrices of one sequence inside the workspace.  Every matrix is a packed
// upper triangle of n(n+1)/2 f32 (padded to a multiple of 64 floats):
//  - "diag-major": cell (i,j) at  d*n - d(d-1)/2 + i  with d = j-i.  A lane that
//    owns row i of a diagonal sweep then reads consecutive addresses next to its
//    neighbours' for every k of the reference's inner loops.
//  - "row-major":  cell (r,c) at  r*n - r(r-1)/2 + (c-r).
enum Mat : int {
  M_QB = 0,   // sums_close                          diag-major
  M_QA = 1,   // sums_accessible                     diag-major
  M_MBC = 2,  // multibranch_close_scores            diag-major
  M_Z = 3,    // sums_external                       diag-major
  M_Q1D = 4,  // sums_1ormore_basepairs              diag-major
  M_Q1C = 5,  // sums_1ormore_basepairs              column-major, shifted one row (see col_off)
  M_ZRE = 6,  // sums_rightmost_basepairs_external   diag-major (inside pass)
  M_PM = 6,   // probs_multibranch                   column-major (outside pass, same slot)
  M_QM = 7,   // sums_multibranch                    diag-major (inside pass)
  M_PM2 = 7,  // probs_multibranch2                  column-major (outside pass, same slot)
  M_W = 8,    // (P + mbclose) - Qb of a pair        diag-major (outside pass)
  M_ZRM = 9,  // sums_rightmost_basepairs_multibranch diag-major (CONTRAfold, inside pass)
  M_P = 9,    // log basepair_probs                  diag-major (outside pass, same slot)
  M_PQ = 10,  // {log basepair_prob, sums_close} of a FINISHED pair, interleaved (float2, two
              // slots), diag-major: the enclosing-pair probes read both with one 8-byte gather
  M_COUNT = 12
};

struct SeqDesc {
  uint32_t n;
  uint32_t tri_pad;   // padded floats per matrix
  uint64_t seq_off;   // offset of the first base in the bases buffer
  uint64_t ws_off;    // float offset of this sequence's matrices in the workspace
  uint64_t out_off;   // float offset of this sequence's bpp triangle in the output
  uint32_t batch_idx; // position in the caller's batch (for log_partition)
  uint32_t pk_words;  // 32-bit words of the 2-bit packed sequence copy
  uint64_t pk_off;    // float offset of that copy in the workspace
  uint64_t cidx_off;  // float offset of the u16 lists of canonical cells (one per diagonal)
  uint64_t ccnt_off;  // float offset of the u32 list lengths (one per diagonal)
  uint64_t c64_off;   // float offset of the u32 table: canonical cells before position 64*w
                      // of diagonal D at [w * (n + 64) + D]
};

// Traceback of the gamma-centroid fold (src/centroid_fold.rs:64-102): exact float equality
// tests in the order left-skip, right-skip, pair, first bifurcation k.  M(r, c) reads the filled
// matrix (0 below and on the main diagonal), prob(i, j) the bpp entry (negative = absent).
template <class MatFn, class ProbFn>
inline uint32_t centroid_traceback(uint32_t n, float centroid_threshold, MatFn&& M, ProbFn&& prob,
                                   uint32_t* pairs_out, uint32_t max_pairs) {
  uint32_t np = 0;
  std::vector<std::pair<uint32_t, uint32_t>> stack;
  stack.emplace_back(0u, n - 1);
  while (!stack.empty()) {
    auto [i, j] = stack.back();
    stack.pop_back();
    if (j <= i) continue;
    const float best = M(i, j);
    if (best == 0.f) continue;
    if (best == M(i + 1, j)) {
      stack.emplace_back(i + 1, j);
    } else if (best == M(i, j - 1)) {
      stack.emplace_back(i, j - 1);
    } else if (prob(i, j) >= -0.5f &&
               best == M(i + 1, j - 1) + centroid_threshold * prob(i, j) - 1.f) {
      stack.emplace_back(i + 1, j - 1);
      if (pairs_out && np < max_pairs) {
        pairs_out[2 * np] = i;
        pairs_out[2 * np + 1] = j;
      }
      np++;
    } else {
      for (uint32_t k = i + 1; k < j; k++) {
        if (best == M(i, k) + M(k + 1, j)) {
          stack.emplace_back(i, k);
          stack.emplace_back(k + 1, j);
          break;
        }
      }
    }
  }
  return np;
}

}  // namespace rnamc

#endif
