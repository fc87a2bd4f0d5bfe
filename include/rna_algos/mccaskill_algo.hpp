// rna_algos/mccaskill_algo.hpp — C++ host-side mirror of the reference crate's
// `mccaskill_algo`, `utils` (the part this path uses) and `centroid_fold` modules
// over the C ABI of librnamc.so (include/rnamc.h).  Header-only, C++17.
//
// The reference is a Rust library; its toolchain is absent from this build
// environment, so the host layer above the C ABI is written in C++ with the
// reference's names, argument meaning and error behaviour (a non-zero status
// throws where the reference panics).  Reference items mirrored:
//   src/mccaskill_algo.rs:247-255  pub fn mccaskill_algo<T>(seq, uses_contra_model,
//                                  allows_short_hairpins, fold_score_sets)
//                                  -> (SparseProbMat<T>, FoldScores<T>)
//   src/mccaskill_algo.rs:24-211   impl FoldScoreSets { new, accumulate, transfer }
//   src/utils.rs:45-89,562-577     PosPair, SparseProbMat, Prob, Seq, bytes2seq
//   src/centroid_fold.rs:4-7,25    CentroidFold<T>, centroid_fold<T>
#ifndef RNA_ALGOS_MCCASKILL_ALGO_HPP
#define RNA_ALGOS_MCCASKILL_ALGO_HPP

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../rnamc.h"

namespace rna_algos {

using Prob = float;
using Score = float;
using Base = uint8_t;               // codes A,C,G,U = 0,1,2,3 (usize in the reference)
using Seq = std::vector<Base>;
template <class T>
using PosPair = std::pair<T, T>;
template <class T>
using PosQuad = std::tuple<T, T, T, T>;

struct PosPairHash {
  template <class T>
  size_t operator()(const std::pair<T, T>& p) const noexcept {
    return (static_cast<size_t>(p.first) << 32) ^ static_cast<size_t>(p.second);
  }
};
template <class T>
using SparseProbMat = std::unordered_map<PosPair<T>, Prob, PosPairHash>;
template <class T>
using SparseScoreMat = std::unordered_map<PosPair<T>, Score, PosPairHash>;

constexpr Prob EPSILON = 0.001f;
constexpr Prob PROB_BOUND_LOWER = -EPSILON;
constexpr Prob PROB_BOUND_UPPER = 1.f + EPSILON;

struct RnamcError : std::runtime_error {
  int status;
  RnamcError(int st) : std::runtime_error(std::string("rnamc: ") + rnamc_strerror(st) + " (" +
                                          rnamc_last_error() + ")"), status(st) {}
};
inline void check(int st) {
  if (st != RNAMC_OK) throw RnamcError(st);  // the reference panics
}

// bytes2seq, src/utils.rs:562-577
inline Seq bytes2seq(const std::string& x) {
  Seq y(x.size());
  check(rnamc_bytes2seq(reinterpret_cast<const uint8_t*>(x.data()), x.size(), y.data()));
  return y;
}

// FoldScoreSets, src/utils.rs:91-119.  `contra` carries the fields of the reference
// struct under their own names; `turner` the constants the reference reads from
// rna-ss-params directly.
struct FoldScoreSets {
  std::unique_ptr<rnamc_params> p{new rnamc_params};
  explicit FoldScoreSets(Score init_val = 0.f) { check(rnamc_params_new(init_val, p.get())); }
  static FoldScoreSets new_(Score init_val) { return FoldScoreSets(init_val); }
  static FoldScoreSets synthetic(uint64_t seed) {
    FoldScoreSets f;
    check(rnamc_params_synthetic(seed, f.p.get()));
    return f;
  }
  static FoldScoreSets load(const std::string& path) {
    FoldScoreSets f;
    check(rnamc_params_load(path.c_str(), f.p.get()));
    return f;
  }
  void accumulate() { check(rnamc_fold_score_sets_accumulate(&p->contra)); }
  // transfer(): the "compiled" tables come from `src` (a table file or the synthetic set)
  void transfer(const FoldScoreSets& src) {
    p->turner = src.p->turner;
    p->table_id = src.p->table_id;
    check(rnamc_fold_score_sets_transfer(&p->contra, &src.p->contra));
  }
  rnamc_fold_score_sets& contra() { return p->contra; }
  rnamc_turner_scores& turner() { return p->turner; }
};

// FoldScores<T>, src/mccaskill_algo.rs:13-19 — side products no in-crate caller reads, so
// mccaskill_algo leaves them empty unless asked (with_fold_scores); fold_scores<T> fills
// them through rnamc_fold_scores.  twoloop_scores is keyed by quad_key(i,j,k,l).
inline uint64_t quad_key(uint64_t i, uint64_t j, uint64_t k, uint64_t l) {
  return (i << 48) | (j << 32) | (k << 16) | l;
}
template <class T>
struct FoldScores {
  SparseScoreMat<T> hairpin_scores;
  std::unordered_map<uint64_t, Score> twoloop_scores;
  SparseScoreMat<T> multibranch_close_scores;
  SparseScoreMat<T> accessible_scores;
};

// One device context per FoldScoreSets (tables uploaded once).
class Context {
 public:
  explicit Context(const FoldScoreSets& f, int device = -1) {
    check(rnamc_ctx_create(f.p.get(), device, 0, &ctx_));
  }
  ~Context() { rnamc_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  rnamc_ctx* get() const { return ctx_; }

 private:
  rnamc_ctx* ctx_ = nullptr;
};

template <class T>
SparseProbMat<T> unpack(const float* packed, uint32_t n) {
  SparseProbMat<T> m;
  size_t x = 0;
  for (uint32_t d = 0; d < n; d++)
    for (uint32_t i = 0; i + d < n; i++, x++)
      if (packed[x] >= -0.5f) m.emplace(PosPair<T>(static_cast<T>(i), static_cast<T>(i + d)), packed[x]);
  return m;
}

// the four maps the reference's inside pass fills (src/mccaskill_algo.rs:302-304, 320,
// 333-338 / 407-409, 431, 457-462)
template <class T>
FoldScores<T> fold_scores(const Context& ctx, const Seq& seq, bool uses_contra_model,
                          bool allows_short_hairpins) {
  const uint32_t n = static_cast<uint32_t>(seq.size());
  const uint64_t len = rnamc_bpp_len(n);
  std::vector<float> hp(len ? len : 1), mb(hp.size()), ac(hp.size());
  uint64_t count = 0;
  check(rnamc_fold_scores(ctx.get(), seq.data(), n, uses_contra_model, allows_short_hairpins,
                          hp.data(), mb.data(), ac.data(), nullptr, 0, &count));
  std::vector<rnamc_twoloop_score> tl(count ? count : 1);
  check(rnamc_fold_scores(ctx.get(), seq.data(), n, uses_contra_model, allows_short_hairpins,
                          nullptr, nullptr, nullptr, tl.data(), count, &count));
  FoldScores<T> out;
  uint64_t x = 0;
  for (uint32_t d = 0; d < n; d++)
    for (uint32_t i = 0; i + d < n; i++, x++) {
      const PosPair<T> key{static_cast<T>(i), static_cast<T>(i + d)};
      if (hp[x] == hp[x]) out.hairpin_scores.emplace(key, hp[x]);
      if (mb[x] == mb[x]) out.multibranch_close_scores.emplace(key, mb[x]);
      if (ac[x] == ac[x]) out.accessible_scores.emplace(key, ac[x]);
    }
  out.twoloop_scores.reserve(count);
  for (uint64_t e = 0; e < count; e++)
    out.twoloop_scores.emplace(quad_key(tl[e].i, tl[e].j, tl[e].k, tl[e].l), tl[e].score);
  return out;
}

// FoldSums<T>, src/mccaskill_algo.rs:3-11: the value of the reference's first stage
// (get_fold_sums / get_fold_sums_contra, 282 / 380), through rnamc_fold_sums (inside sweep alone)
using SumMat = std::vector<std::vector<Score>>;
template <class T>
struct FoldSums {
  SumMat sums_external;
  SumMat sums_rightmost_basepairs_external;
  SumMat sums_rightmost_basepairs_multibranch;
  SparseScoreMat<T> sums_close;
  SparseScoreMat<T> sums_accessible;
  SumMat sums_multibranch;
  SumMat sums_1ormore_basepairs;
};
template <class T>
FoldSums<T> get_fold_sums(const Context& ctx, const Seq& seq, bool uses_contra_model,
                          bool allows_short_hairpins) {
  const uint32_t n = static_cast<uint32_t>(seq.size());
  std::vector<std::vector<float>> m(7, std::vector<float>(static_cast<size_t>(n) * n + 1));
  check(rnamc_fold_sums(ctx.get(), seq.data(), n, uses_contra_model, allows_short_hairpins,
                        m[0].data(), m[1].data(), m[2].data(), m[3].data(), m[4].data(), m[5].data(),
                        m[6].data()));
  auto rows = [n](const std::vector<float>& v) {
    SumMat out(n);
    for (uint32_t i = 0; i < n; i++) out[i].assign(v.begin() + static_cast<size_t>(i) * n, v.begin() + static_cast<size_t>(i + 1) * n);
    return out;
  };
  auto sparse = [n](const std::vector<float>& v) {
    SparseScoreMat<T> out;  // finite sums only, as the reference inserts them (332-338 / 456-462)
    for (uint32_t i = 0; i < n; i++)
      for (uint32_t j = i; j < n; j++) {
        const float x = v[static_cast<size_t>(i) * n + j];
        if (x - x == 0.f) out.emplace(PosPair<T>(static_cast<T>(i), static_cast<T>(j)), x);
      }
    return out;
  };
  return {rows(m[0]), rows(m[1]), rows(m[2]), sparse(m[3]), sparse(m[4]), rows(m[5]), rows(m[6])};
}

// mccaskill_algo, src/mccaskill_algo.rs:247-280
template <class T>
std::pair<SparseProbMat<T>, FoldScores<T>> mccaskill_algo(const Context& ctx, const Seq& seq,
                                                          bool uses_contra_model,
                                                          bool allows_short_hairpins,
                                                          bool with_fold_scores = false) {
  const uint32_t n = static_cast<uint32_t>(seq.size());
  const uint64_t offsets[2] = {0, n};
  const uint64_t out_offsets[2] = {0, rnamc_bpp_len(n)};
  std::vector<float> packed(rnamc_bpp_len(n) ? rnamc_bpp_len(n) : 1);
  float logz = 0.f;
  check(rnamc_bpp_batch(ctx.get(), 1, seq.data(), offsets, uses_contra_model, allows_short_hairpins,
                        packed.data(), out_offsets, &logz));
  return {unpack<T>(packed.data(), n),
          with_fold_scores ? fold_scores<T>(ctx, seq, uses_contra_model, allows_short_hairpins)
                           : FoldScores<T>()};
}

// the whole FASTA at once (what src/bin/mccaskill_algo.rs:64-93 does on a thread pool)
template <class T>
std::vector<SparseProbMat<T>> mccaskill_algo_batch(const Context& ctx, const std::vector<Seq>& seqs,
                                                   bool uses_contra_model,
                                                   bool allows_short_hairpins) {
  std::vector<uint64_t> off(seqs.size() + 1, 0), ooff(seqs.size() + 1, 0);
  for (size_t s = 0; s < seqs.size(); s++) {
    off[s + 1] = off[s] + seqs[s].size();
    ooff[s + 1] = ooff[s] + rnamc_bpp_len(static_cast<uint32_t>(seqs[s].size()));
  }
  std::vector<Base> bases(off.back());
  for (size_t s = 0; s < seqs.size(); s++) std::copy(seqs[s].begin(), seqs[s].end(), bases.begin() + off[s]);
  std::vector<float> packed(ooff.back() ? ooff.back() : 1);
  check(rnamc_bpp_batch(ctx.get(), static_cast<uint32_t>(seqs.size()), bases.data(), off.data(),
                        uses_contra_model, allows_short_hairpins, packed.data(), ooff.data(), nullptr));
  std::vector<SparseProbMat<T>> out;
  for (size_t s = 0; s < seqs.size(); s++)
    out.push_back(unpack<T>(packed.data() + ooff[s], static_cast<uint32_t>(seqs[s].size())));
  return out;
}

// CentroidFold<T>, src/centroid_fold.rs:4-7
template <class T>
struct CentroidFold {
  std::vector<PosPair<T>> basepair_pos_pairs;
  Score expect_accuracy = 0.f;
};

// centroid_fold, src/centroid_fold.rs:25-105
template <class T>
CentroidFold<T> centroid_fold(const SparseProbMat<T>& basepair_probs, size_t seq_len,
                              Prob centroid_threshold) {
  const uint32_t n = static_cast<uint32_t>(seq_len);
  std::vector<float> packed(rnamc_bpp_len(n), -1.0f);
  for (const auto& kv : basepair_probs)
    packed[rnamc_bpp_index(n, static_cast<uint32_t>(kv.first.first), static_cast<uint32_t>(kv.first.second))] = kv.second;
  std::vector<uint32_t> pairs(2 * (n / 2 + 1));
  uint32_t np = 0;
  CentroidFold<T> f;
  check(rnamc_centroid_fold(packed.data(), n, centroid_threshold, pairs.data(), n / 2 + 1, &np,
                            &f.expect_accuracy));
  for (uint32_t x = 0; x < np; x++)
    f.basepair_pos_pairs.emplace_back(static_cast<T>(pairs[2 * x]), static_cast<T>(pairs[2 * x + 1]));
  return f;
}

}  // namespace rna_algos

#endif
