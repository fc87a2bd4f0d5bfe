// rna_algos/durbin_algo.hpp — C++ host-side mirror of the reference crate's `durbin_algo`
// module over the C ABI of librnamc.so (include/rnamc.h).  Header-only, C++17.
// Reference items mirrored:
//   src/durbin_algo.rs:4-14    pub struct AlignScores { ... }
//   src/durbin_algo.rs:25-58   impl AlignScores { new, transfer }
//   src/durbin_algo.rs:73-77   pub fn durbin_algo(seq_pair, align_scores) -> ProbMat
//   src/utils.rs:83,122        ProbMat, PSEUDO_BASE
// Sequences carry PSEUDO_BASE at both ends, as the reference's callers build them
// (tests/tests.rs:53-55).
#ifndef RNA_ALGOS_DURBIN_ALGO_HPP
#define RNA_ALGOS_DURBIN_ALGO_HPP

#include <utility>
#include <vector>

#include "mccaskill_algo.hpp"

namespace rna_algos {

constexpr Base PSEUDO_BASE = 4;  // U + 1, src/utils.rs:122
using Probs = std::vector<Prob>;
using ProbMat = std::vector<Probs>;
using SeqPair = std::pair<const Seq*, const Seq*>;

// AlignScores: the C struct has the reference's fields under their own names
struct AlignScores : rnamc_align_scores {
  explicit AlignScores(Score init_val = 0.f) { check(rnamc_align_scores_new(init_val, this)); }
  static AlignScores new_(Score init_val) { return AlignScores(init_val); }
  void transfer() { check(rnamc_align_scores_transfer(this)); }
};

// every pair of `pairs` (indices into `seqs`) in one device batch: what
// src/bin/durbin_algo.rs:55-75 does with one pool task per pair
inline std::vector<ProbMat> durbin_algo_batch(const Context& ctx, const std::vector<Seq>& seqs,
                                              const std::vector<std::pair<size_t, size_t>>& pairs,
                                              const AlignScores& align_scores) {
  std::vector<uint8_t> bases;
  std::vector<uint64_t> offsets{0}, out_offsets{0};
  for (const Seq& s : seqs) {
    bases.insert(bases.end(), s.begin(), s.end());
    offsets.push_back(bases.size());
  }
  std::vector<uint32_t> pa, pb;
  for (const auto& p : pairs) {
    pa.push_back(static_cast<uint32_t>(p.first));
    pb.push_back(static_cast<uint32_t>(p.second));
    out_offsets.push_back(out_offsets.back() + seqs[p.first].size() * seqs[p.second].size());
  }
  std::vector<float> flat(out_offsets.back() ? out_offsets.back() : 1);
  check(rnamc_durbin_batch(ctx.get(), &align_scores, static_cast<uint32_t>(seqs.size()), bases.data(),
                           offsets.data(), static_cast<uint32_t>(pairs.size()), pa.data(), pb.data(),
                           flat.data(), out_offsets.data()));
  std::vector<ProbMat> out;
  for (size_t x = 0; x < pairs.size(); x++) {
    const size_t n1 = seqs[pairs[x].first].size(), n2 = seqs[pairs[x].second].size();
    ProbMat m(n1, Probs(n2));
    for (size_t i = 0; i < n1; i++)
      for (size_t j = 0; j < n2; j++) m[i][j] = flat[out_offsets[x] + i * n2 + j];
    out.push_back(std::move(m));
  }
  return out;
}

// durbin_algo, src/durbin_algo.rs:73-77
inline ProbMat durbin_algo(const Context& ctx, const SeqPair& seq_pair, const AlignScores& align_scores) {
  return durbin_algo_batch(ctx, {*seq_pair.first, *seq_pair.second}, {{0, 1}}, align_scores)[0];
}

}  // namespace rna_algos

#endif
