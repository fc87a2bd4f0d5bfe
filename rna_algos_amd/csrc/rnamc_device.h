// rnamc_device.h — host<->kernel contract of librnamc.so (internal).
#ifndef RNAMC_DEVICE_H
#define RNAMC_DEVICE_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "rnamc_internal.h"

namespace rnamc {

// One lock-step group of sequences as the kernels see it (passed by value).
struct DeviceBatch {
  const SeqDesc* seqs;      // descriptors of this group, longest sequence first
  const uint8_t* bases;     // base codes of the whole batch
  float* workspace;         // DP matrices
  float* out;               // packed bpp triangles of the whole batch
  float* log_partition;     // may be null
  const rnamc_params* params;
  const float* hp_init;     // Turner hairpin initiation by loop length (host-built)
  int allows_short_hairpins;
  int order_inside, order_outside;  // order in which a launch's role blocks are dispatched
};

void launch_init(const DeviceBatch& b, uint32_t nseq, uint32_t max_n, hipStream_t st);
// folds of diagonal d (do_sums) and closing-pair block of diagonal d+1 (do_pair)
void launch_inside(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                   uint32_t block, bool do_sums, bool do_pair, hipStream_t st);
// Two-diagonal schedule: folds of diagonals d and d+1 (pair blocks of d and d+1 done; for
// CONTRAfold also launch_inside_zr2 of the same d) beside the early part of the pair
// blocks of d+2 and d+3 (do_head)
void launch_inside2(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                    uint32_t block, bool do_sums, bool do_head, hipStream_t st);
// CONTRAfold: sums_rightmost_basepairs_{external,multibranch} of diagonals d and d+1
void launch_inside_zr2(const DeviceBatch& b, uint32_t d, uint32_t max_n, uint32_t nseq,
                       uint32_t block, hipStream_t st);
// last (multibranch) term of the parked pair blocks of diagonals d0 .. d0+nd-1
void launch_pair_tail(const DeviceBatch& b, bool contra, uint32_t d0, uint32_t nd, uint32_t max_n,
                      uint32_t nseq, uint32_t block, hipStream_t st);
// true when the folds of diagonal d run in the latency form (launch too small to fill the chip)
bool inside_is_split(uint32_t d, uint32_t max_n, uint32_t nseq);
// roles: 7 = all three roles in one kernel; 5 = probs_multibranch + pair head; 2 = pair tail;
// 4 = pair head alone
void launch_outside(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                    uint32_t block, bool do_mb, bool do_tail, bool do_head, int roles,
                    hipStream_t st);
// Latency forms for groups too small to fill the chip (rnamc_latency.h): one wave per fold
// chain.  A group that uses them uses them on EVERY diagonal (they keep W dense, and complete
// sums_1ormore_basepairs of diagonal d-1 in the launch of diagonal d).
// pair_d != 0: the closing-pair blocks of diagonal pair_d in the same launch; head: the 2-loop
// half of the pair probabilities of diagonal d - 1 in the same launch
void launch_inside_lat(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                       int form, bool do_combine, uint32_t pair_d, hipStream_t st);
void launch_outside_lat(const DeviceBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                        bool do_mb, bool do_tail, bool head, hipStream_t st);
// closing-pair block (inside) / 2-loop half of the pair probabilities (outside) of diagonal d
void launch_pair_lat(const DeviceBatch& b, bool contra, bool outside, uint32_t d, uint32_t max_n,
                     uint32_t nseq, hipStream_t st);
// Durbin pair-HMM (src/durbin_algo.rs:79-242): one pair of sequences
struct DurbinPair {
  uint32_t n1, n2;   // lengths including the two pseudo bases
  uint64_t a_off, b_off;  // offsets of the two sequences in the bases buffer
  uint64_t ws_off;   // float offset of this pair's six n1 x n2 matrices in the workspace
  uint64_t out_off;  // float offset of this pair's ProbMat in the output
};
void launch_durbin(const DurbinPair* d_pairs, uint32_t n_pairs, uint32_t max_cells,
                   const uint8_t* d_bases, float* ws, float* d_out, const rnamc_align_scores& sc,
                   hipStream_t st);
void launch_finalize(const DeviceBatch& b, uint32_t nseq, uint32_t max_n, uint32_t dmin_out,
                     hipStream_t st);

// gamma-centroid fold (rnamc_centroid.hip): per threshold g two dense n x n matrices of msz
// floats at m + g * 2 * msz (row-major, then column-major), row stride ld, zero-initialised
struct CentroidBatch {
  const float* bpp;     // packed diagonal-major triangle, absent pairs negative (device)
  float* m;
  const float* gammas;  // device
  uint32_t n, ld;
  uint64_t msz;
};
void launch_centroid(const CentroidBatch& a, uint32_t d, uint32_t n_gammas, hipStream_t st);

// ---- tree-order summation mode (rnamc_tree.hip) ----
// Dense n x n matrices with row stride ld (>= n + 32, a multiple of 32 floats), msz floats
// each; "row" = [i * ld + j], "col" = [j * ld + i].  The outside sweep reuses four slots.
enum TreeMat : int {
  T_QB = 0,   // sums_close                                   row
  T_QA = 1,   // sums_accessible                              row
  T_Q1R = 2,  // sums_1ormore_basepairs                       row
  T_Q1C = 3,  // sums_1ormore_basepairs                       col
  T_ZRE = 4,  // sums_rightmost_basepairs_external            col | outside: W = (P + mbclose) - Qb, row
  T_ZRM = 5,  // sums_rightmost_basepairs_multibranch         col | outside: R = Pm (+) Pm2, col
  T_QM = 6,   // sums_multibranch                             row | outside: probs_multibranch2, row
  T_U = 7,    // column prefix of Zr_mb (first fold of L_c)   col | outside: column prefix of Pm, col
  T_HP = 8,    // hairpin score of the pair (static)                        row
  T_MBC = 9,   // multibranch_close score; -inf = (i,j) may not pair (static) row
  T_X4 = 10,   // four planes (one per 2-loop class), row: inside QbX4 = sums_close + IN4[class],
               // outside PX4 = (log bpp - sums_close) + CS4[class]: a probe reads the plane of its class
  T_ACCS = 14, // accessible score (static)                                  row
  T_CS4 = 15,  // float4 per cell, row: the pair as CLOSING pair of a generic 2-loop, by class (static)
  T_IN4 = 19,  // float4 per cell, row: the pair as ENCLOSED pair (static)
  T_NEAR4 = 23,  // float4 per cell, row: scores of the explicit small 2-loops this pair CLOSES, slots 0..3 of
                 // Special<>::slot: enclosed pair (i+1,j-1), (i+1,j-2), (i+2,j-1), (i+2,j-2) (static)
  T_NEAR8 = 27,  // float4 per cell, row: slots 4..6 (Turner: 1x2, 2x1, 2x2; .w unused) (static)
  // lane-per-cell sweeps (rnamc_tree_lane.h), DIAGONAL-major: cell (i, j) at [(j - i) * ld + i], so that
  // the 64 consecutive rows a wave holds read and write consecutive floats.  In those sweeps the four
  // T_X4 planes are diagonal-major too (their only readers are the sweeps' own 2-loop sums).
  T_QB_D = 31,   // sums_close
  T_Q1_D = 32,   // sums_1ormore_basepairs
  T_ZRM_D = 33,  // sums_rightmost_basepairs_multibranch | outside: R
  T_W_D = 34,    // outside: W
  T_LIST = 35,   // u32 per entry: row d holds the rows i of the cells (i, i + d) that may pair, in order; the
                 // count in the row's last entry (k_tlane_list; the 2-loop sums run a lane per LISTED cell)
  T_GEN_D = 36,  // float2 per cell over two slots: {max, sum} of a listed cell's generic 2-loop terms (k_tlane_gen,
                 // three diagonals a launch; as ONE f32 logarithm it cost an ulp of a ~300-nat value per cell and
                 // diagonal: 2.6e-4 in the probabilities at n = 257, measured)
  T_COUNT = 38,
  // ... and in those sweeps these slots hold diagonal-major data as well:
  T_ZRE_D = T_QB,  // sums_rightmost_basepairs_external (the row-major sums_close has no reader there)
  T_QA_D = T_W_D,  // sums_accessible until the band is spread into T_QA (inside sweep)
  T_P2_D = T_QB     // outside sweep: float2 per cell over the slots T_QB and T_QA (both free once the inside sweep
                    // and its sums_external are through): {max, sum} of the pair's enclosing 2-loop terms
  // T_QM, T_U: sums_multibranch / the column prefix (inside), probs_multibranch2 / the column prefix of
  // probs_multibranch (outside); the statics (T_HP .. T_NEAR8): cell (i, j) at [(j - i) * ld + i] too
};
// Length-dependent part of a generic 2-loop score per probe slot (rnamc_tree.hip, probe_slot),
// derived from rnamc_params on the host (rnamc_api.cpp, build_tree_tabs).  Model index 0 Turner,
// 1 CONTRAfold.
struct TreeTabs {
  float len[2][512];
  // lane-per-cell sweeps: the generic slots by class, inside a class in order of a + b (a prefix of the
  // class's list is what a cell of a given span can enclose): a | (a + b) << 8, the slot's length term;
  // class c's list starts at gstart[c] (a multiple of 8; lists padded to multiples of 8 with their
  // last slot), gcount[c][s] of its slots have a + b <= s
  uint32_t gslot[2][544];
  float glen[2][544];
  uint32_t gstart[2][4];
  uint32_t gcount[2][4][32];
  // ... and class 3 (both unpaired stretches >= 2 bases: three quarters of the slots) once more in runs: on a
  // level a + b = s its slots are CONSECUTIVE a, i.e. consecutive floats of one diagonal row, so four of them
  // are one 16-byte load.  Group g: first a | s << 8, the four length terms (-inf beyond the level's run);
  // g4count[s] of the groups have a + b <= s; the list is padded with eight empty groups.
  uint32_t g4slot[2][128];
  float g4len[2][128][4];
  uint32_t g4count[2][32];
  // ... and classes 0 and 1 (bulges, 1 x many) by LEVEL a + b = s: their slots are (0, s), (s, 0), (1, s - 1),
  // (s - 1, 1) — the row's first and last two positions — so a level is four loads whose offsets are the level's
  // row plus 0, s, 1, s - 1: no slot list at all.  elen[s] = the four length terms (-inf: not a generic slot).
  float elen[2][32][4];
};
struct TreeSeq {
  uint32_t n, ld;
  uint64_t msz;       // floats per matrix
  uint64_t seq_off;   // first base in the bases buffer
  uint64_t ws_off;    // float offset of the first matrix; after the T_COUNT matrices come the
                      // vectors Z(0,.) and Z(.,n-1), n + 64 floats each
  uint64_t out_off;   // float offset of the packed bpp triangle in the output
  uint64_t pk_off;    // float offset of the 2-bit packed copy of the bases (pk_words words)
  uint32_t pk_words;
  uint32_t batch_idx;
  uint64_t mid_off;   // float offset of the banded mid-field ring: three products x `ring`
                      // diagonals x (n + 64 rounded up to 64) cells x {max, sum} (rnamc_tree.hip)
};
struct TreeBatch {
  const TreeSeq* seqs;  // descriptors of the group (device memory)
  TreeSeq one;          // the descriptor itself when the group is a single sequence (use_one):
  uint32_t use_one;     // saves the dependent descriptor load in front of every launch
  const uint8_t* bases;
  float* workspace;
  float* out;
  float* log_partition;
  const rnamc_params* params;
  const TreeTabs* tabs;
  const float* hp_init;
  int allows_short_hairpins;
  int debug;  // timing experiments (builds with -DRNAMC_DEBUG_KNOBS only; 0 otherwise)
  uint32_t ring;  // diagonals the mid-field ring holds (twice the band width; 0: no banding)
  uint32_t lane;  // != 0: both sweeps run lane-per-cell (rnamc_tree_lane.h: statics and sweep matrices diagonal-major)
};
// Launch-shape policy of the tree-order sweep, per context (rnamc_ctx_set "tree_waves",
// "tree_short", "tree_ahead_waves", "tree_mid_wgs"): passed to every launch, no process globals.
struct TreePolicy {
  uint64_t waves = 5120;            // waves a launch may hold at once (5 per SIMD at ~88 VGPRs)
  uint32_t short_terms = 256;       // sums up to this many terms take one wave per cell
  uint64_t ahead_waves = 16384;    // waves up to which the ahead role takes one wave per CELL (a lone sequence;
                                    // beyond — batches — one per row: 8 % faster there, profiles/r04_tree_on_batches.txt)
  uint32_t xcd_rows = 0;            // != 0: workgroup ids of a sweep launch are dealt so that runs of 8 workgroups
                                    // (32 rows of a sequence) share an XCD, rotated by sequence (see xcd_chunk)
  uint32_t mid_wgs = 0;             // workgroups of a k_tree_mid launch; 0: by the longest sequence (256 below
                                    // 6 144 nt, 512 below 12 288, 1 024 beyond: from ~8 000 nt on the mid-field
                                    // products, not the launch chain, set the sweep's time: profiles/r04_tree_long.txt)
  uint32_t mid_mx = 1;              // != 0: the mid-field products on the matrix cores (k_tree_mid_mx, rnamc_tree_mx.h:
                                    // factors 2^(x log2 e - E) per operand element, v_mfma_f32_32x32x2_f32 per term)
};
// what = 0: everything before the inside sweep; 1: the four reused slots before the outside sweep
void launch_tree_init(const TreeBatch& b, uint32_t nseq, uint32_t max_n, bool contra, int what,
                      hipStream_t st);
// tpc_knob: threads per cell (64, 256, 1024), anything else = chosen by the diagonal's cells;
// two: diagonals d and d+1 in one launch (inside: d then d+1; outside: d+1 then d)
// thr: banded mid-field threshold of the launch's band (0: none; the launch's sums run whole)
// use_far: the far parts of this launch's 2-loop blocks were written by the previous launch's
// ahead role; nd0, nd_count: the diagonals of the NEXT launch, whose far parts this launch's extra
// workgroups write (nd_count = 0: none).  Needs the far ring of a banded workspace (ring != 0).
void launch_tree_inside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                        int64_t tpc_knob, bool two, uint32_t thr, bool use_far, uint32_t nd0,
                        uint32_t nd_count, const TreePolicy& pol, hipStream_t st);
void launch_tree_outside(const TreeBatch& b, bool contra, uint32_t d, uint32_t max_n, uint32_t nseq,
                         int64_t tpc_knob, bool two, uint32_t thr, bool use_far, uint32_t nd0,
                         uint32_t nd_count, const TreePolicy& pol, hipStream_t st);
// Mid-field of the cubic products for the cells of diagonals [dlo, dhi] (one band), threshold thr:
// inside (outside = false) sums_multibranch, outside probs_multibranch and the Q1 x R part of L_e
void launch_tree_mid(const TreeBatch& b, bool outside, uint32_t dlo, uint32_t dhi, uint32_t thr,
                     uint32_t max_n, uint32_t nseq, const TreePolicy& pol, hipStream_t st);
// sums_external's first row and last column of a banded sweep, diagonals [dlo, dhi] (in order)
void launch_tree_ext(const TreeBatch& b, bool contra, uint32_t dlo, uint32_t dhi, uint32_t max_n,
                     uint32_t nseq, hipStream_t st);
// Does `side` run beside `main`?  A bounded spin (~120 us of s_memtime ticks, every wave exits) on
// `side`, a trivial kernel on `main` behind it in submission order, events around the latter: if
// the trivial kernel only ends when the spin does, the two streams share a hardware queue.
// Returns 1 (concurrent), 2 (serialised), 0 on any HIP error.  Synchronises both streams.
int tree_side_stream_probe(hipStream_t main, hipStream_t side);
// per-cell statics (hairpin / multibranch-close / accessible scores, 2-loop sides), once per group
void launch_tree_static(const TreeBatch& b, bool contra, uint32_t nseq, uint32_t max_n, hipStream_t st);
// lane-per-cell sweeps of a batch (rnamc_tree_lane.h): one diagonal per launch, a lane per cell
// d: the diagonal whose cells take their sums and recurrences (a lane per row); d_b: the diagonal whose LISTED
// cells (the ones that may pair) take their 2-loop sums in the same launch — one diagonal ahead of d in the
// sweep's direction (either >= max_n: that role is absent)
void launch_tlane_inside(const TreeBatch& b, bool contra, uint32_t d, uint32_t d_b, uint32_t max_n, uint32_t nseq,
                         uint32_t thr, hipStream_t st);
void launch_tlane_outside(const TreeBatch& b, bool contra, uint32_t d, uint32_t d_b, uint32_t max_n, uint32_t nseq,
                          uint32_t thr, hipStream_t st);
// generic 2-loop sums of the listed cells of `count` (<= 3) diagonals from g0 on (inside upwards, outside downwards)
void launch_tlane_gen(const TreeBatch& b, bool contra, bool outside, uint32_t g0, uint32_t count, uint32_t max_n,
                      uint32_t nseq, hipStream_t st);
// the rows of the cells that may pair, diagonal by diagonal (after launch_tree_static)
void launch_tlane_list(const TreeBatch& b, uint32_t max_n, uint32_t nseq, hipStream_t st);
// the finished band [dlo, dhi] into the row- / column-major copies k_tree_mid and k_tree_ext read
void launch_tlane_spread(const TreeBatch& b, bool outside, uint32_t dlo, uint32_t dhi, uint32_t max_n, uint32_t nseq,
                         hipStream_t st);
void launch_tree_finalize(const TreeBatch& b, uint32_t nseq, uint32_t max_n, hipStream_t st);

}  // namespace rnamc

#endif
