// The reference's own integration tests (tests/tests.rs:7-43 test_mccaskill_algo and 45-80
// test_durbin_algo) written against the C++ host mirror: 6 tRNAs, both models, every bpp value
// in [-0.001, 1.001); all 15 pairs, every match probability in the same range.
// Exit 0 = pass, 77 = no GPU (the library refuses to run), anything else = failure.
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "rna_algos/durbin_algo.hpp"
#include "rna_algos/mccaskill_algo.hpp"

using namespace rna_algos;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  std::vector<Seq> seqs;
  std::string line, cur;
  while (std::getline(in, line)) {
    if (!line.empty() && line[0] == '>') {
      if (!cur.empty()) seqs.push_back(bytes2seq(cur));
      cur.clear();
    } else {
      cur += line;
    }
  }
  if (!cur.empty()) seqs.push_back(bytes2seq(cur));
  if (seqs.size() != 6) return 3;
  FoldScoreSets fold_scores = FoldScoreSets::new_(0.f);
  fold_scores.transfer(FoldScoreSets::synthetic(1));
  try {
    Context ctx(fold_scores);
    const bool allows_short_hairpins = false;
    for (const Seq& z : seqs) {
      for (bool uses_contra_model : {false, true}) {
        auto a = mccaskill_algo<uint8_t>(ctx, z, uses_contra_model, allows_short_hairpins).first;
        if (a.empty()) return 4;
        for (const auto& kv : a)
          if (!(kv.second >= PROB_BOUND_LOWER && kv.second < PROB_BOUND_UPPER)) return 5;
        auto f = centroid_fold<uint8_t>(a, z.size(), 4.0f);
        for (const auto& pr : f.basepair_pos_pairs)
          if (a.find(pr) == a.end()) return 6;
      }
    }
    {  // FoldScores<T>: every pair with a probability has its two per-pair scores
      auto r = mccaskill_algo<uint8_t>(ctx, seqs[0], false, false, true);
      const auto& fs = r.second;
      if (fs.multibranch_close_scores.size() != r.first.size()) return 8;
      if (fs.accessible_scores.size() != r.first.size()) return 8;
      for (const auto& kv : r.first)
        if (fs.accessible_scores.find(kv.first) == fs.accessible_scores.end()) return 9;
      if (fs.hairpin_scores.size() < r.first.size() || fs.twoloop_scores.empty()) return 10;
    }
    {  // test_durbin_algo, tests/tests.rs:45-80
      std::vector<Seq> padded;
      for (const Seq& z : seqs) {
        Seq y = z;
        y.insert(y.begin(), PSEUDO_BASE);
        y.push_back(PSEUDO_BASE);
        padded.push_back(y);
      }
      AlignScores align_scores = AlignScores::new_(0.f);
      align_scores.transfer();
      std::vector<std::pair<size_t, size_t>> pairs;
      for (size_t i = 0; i < padded.size(); i++)
        for (size_t j = i + 1; j < padded.size(); j++) pairs.emplace_back(i, j);
      auto mats = durbin_algo_batch(ctx, padded, pairs, align_scores);
      if (mats.size() != 15) return 11;
      for (const ProbMat& m : mats)
        for (const Probs& row : m)
          for (Prob x : row)
            if (!(x >= PROB_BOUND_LOWER && x < PROB_BOUND_UPPER)) return 12;
      ProbMat one = durbin_algo(ctx, SeqPair(&padded[0], &padded[1]), align_scores);
      if (one != mats[0]) return 13;
    }
    try {  // a byte outside ACGU: the reference panics
      bytes2seq("ACGT");
      return 7;
    } catch (const RnamcError&) {
    }
  } catch (const RnamcError& e) {
    if (e.status == RNAMC_ERR_NO_DEVICE) {
      std::printf("no GPU: %s\n", e.what());
      return 77;
    }
    std::printf("error: %s\n", e.what());
    return 1;
  }
  std::printf("cpp host mirror ok\n");
  return 0;
}
