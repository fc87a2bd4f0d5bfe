// rnamc_host.cpp — host-side plumbing of librnamc.so: status strings, sequence
// encoding, parameter-set construction (FoldScoreSets::{new,accumulate,transfer}),
// the seeded synthetic table generator, table-file I/O and the gamma-centroid
// fold that consumes the GPU bpp matrices.  No device code here.
//
// Reference interfaces mirrored (paths relative to the reference tree):
//   src/mccaskill_algo.rs:24-211  impl FoldScoreSets {new, accumulate, transfer}
//   src/utils.rs:562-577          bytes2seq
//   src/centroid_fold.rs:25-105   centroid_fold
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/rnamc.h"
#include "rnamc_internal.h"

namespace rnamc {

thread_local std::string g_last_error;

void set_last_error(const std::string& msg) { g_last_error = msg; }

bool is_canonical(int a, int b) {
  // AU | CG | GC | GU | UA | UG  (src/utils.rs:162-164)
  return (a == RNAMC_A && b == RNAMC_U) || (a == RNAMC_C && b == RNAMC_G) ||
         (a == RNAMC_G && b == RNAMC_C) || (a == RNAMC_G && b == RNAMC_U) ||
         (a == RNAMC_U && b == RNAMC_A) || (a == RNAMC_U && b == RNAMC_G);
}

namespace {

struct SplitMix64 {
  uint64_t state;
  explicit SplitMix64(uint64_t seed) : state(seed) {}
  uint64_t next() {
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  // uniform in [lo, hi), 24 random bits
  float uni(float lo, float hi) {
    float u = static_cast<float>(next() >> 40) * (1.0f / 16777216.0f);
    return lo + (hi - lo) * u;
  }
};

template <size_t N>
void fill_uni(SplitMix64& g, float (&a)[N], float lo, float hi) {
  for (size_t x = 0; x < N; x++) a[x] = g.uni(lo, hi);
}

void fill_uni_ptr(SplitMix64& g, float* a, size_t n, float lo, float hi) {
  for (size_t x = 0; x < n; x++) a[x] = g.uni(lo, hi);
}

void fill_all(float* base, size_t count, float v) {
  for (size_t x = 0; x < count; x++) base[x] = v;
}

// Every f32 array of rnamc_params, by name, for host-language mirrors.
struct FieldDesc {
  const char* name;
  size_t offset;
  size_t count;
};

#define FLD_T(field) \
  { "turner." #field, offsetof(rnamc_params, turner.field), sizeof(rnamc_turner_scores::field) / sizeof(float) }
#define FLD_C(field) \
  { "contra." #field, offsetof(rnamc_params, contra.field), sizeof(rnamc_fold_score_sets::field) / sizeof(float) }

const FieldDesc kFields[] = {
    FLD_T(hairpin_scores_init),
    FLD_T(terminal_mismatch_scores_hairpin),
    FLD_T(stack_scores),
    FLD_T(bulge_scores_init),
    FLD_T(interior_scores_init),
    FLD_T(interior_scores_1x1),
    FLD_T(interior_scores_1x2),
    FLD_T(interior_scores_2x2),
    FLD_T(terminal_mismatch_scores_1xmany),
    FLD_T(terminal_mismatch_scores_2x3),
    FLD_T(terminal_mismatch_scores_interior),
    FLD_T(terminal_mismatch_scores_multibranch),
    FLD_T(dangling_scores_5prime),
    FLD_T(dangling_scores_3prime),
    FLD_T(helix_augu_end_penalty),
    FLD_T(coeff_hairpin_len_extrapolation),
    FLD_T(ninio_coeff),
    FLD_T(ninio_max),
    FLD_T(init_multibranch_base),
    FLD_T(coeff_num_branches),
    FLD_T(special_hairpin_scores),
    FLD_C(hairpin_scores_len),
    FLD_C(bulge_scores_len),
    FLD_C(interior_scores_len),
    FLD_C(interior_scores_symmetric),
    FLD_C(interior_scores_asymmetric),
    FLD_C(stack_scores),
    FLD_C(terminal_mismatch_scores),
    FLD_C(dangling_scores_left),
    FLD_C(dangling_scores_right),
    FLD_C(helix_close_scores),
    FLD_C(basepair_scores),
    FLD_C(interior_scores_explicit),
    FLD_C(bulge_scores_0x1),
    FLD_C(interior_scores_1x1),
    FLD_C(multibranch_score_base),
    FLD_C(multibranch_score_basepair),
    FLD_C(multibranch_score_unpair),
    FLD_C(external_score_basepair),
    FLD_C(external_score_unpair),
    FLD_C(hairpin_scores_len_cumulative),
    FLD_C(bulge_scores_len_cumulative),
    FLD_C(interior_scores_len_cumulative),
    FLD_C(interior_scores_symmetric_cumulative),
    FLD_C(interior_scores_asymmetric_cumulative),
};
#undef FLD_T
#undef FLD_C

const char kMagic[8] = {'R', 'N', 'A', 'M', 'C', 'T', 'B', 'L'};

}  // namespace
}  // namespace rnamc

using namespace rnamc;

extern "C" {

uint32_t rnamc_abi_version(void) { return RNAMC_ABI_VERSION; }

size_t rnamc_params_sizeof(void) { return sizeof(rnamc_params); }

const char* rnamc_strerror(int status) {
  switch (status) {
    case RNAMC_OK: return "ok";
    case RNAMC_ERR_INVALID_ARG: return "invalid argument";
    case RNAMC_ERR_INVALID_BASE: return "sequence holds a byte outside ACGUacgu";
    case RNAMC_ERR_EMPTY_SEQ: return "empty sequence";
    case RNAMC_ERR_SEQ_TOO_LONG: return "sequence longer than 65535";
    case RNAMC_ERR_NO_DEVICE: return "no usable HIP device";
    case RNAMC_ERR_OOM: return "out of memory";
    case RNAMC_ERR_HIP: return "HIP runtime error";
    case RNAMC_ERR_IO: return "table file I/O error";
    case RNAMC_ERR_FORMAT: return "table file format error";
    default: return "unknown status";
  }
}

const char* rnamc_last_error(void) { return g_last_error.c_str(); }

int rnamc_bytes2seq(const uint8_t* ascii, uint64_t n, uint8_t* codes) {
  if ((!ascii || !codes) && n) return RNAMC_ERR_INVALID_ARG;
  for (uint64_t x = 0; x < n; x++) {
    switch (ascii[x]) {
      case 'a': case 'A': codes[x] = RNAMC_A; break;
      case 'c': case 'C': codes[x] = RNAMC_C; break;
      case 'g': case 'G': codes[x] = RNAMC_G; break;
      case 'u': case 'U': codes[x] = RNAMC_U; break;
      default: return RNAMC_ERR_INVALID_BASE;
    }
  }
  return RNAMC_OK;
}

uint64_t rnamc_bpp_len(uint32_t n) { return static_cast<uint64_t>(n) * (n + 1ull) / 2ull; }

uint64_t rnamc_bpp_index(uint32_t n, uint32_t i, uint32_t j) {
  uint64_t d = j - i;
  return d * n - d * (d - 1) / 2 + i;
}

int rnamc_fold_score_sets_new(float init_val, rnamc_fold_score_sets* out) {
  if (!out) return RNAMC_ERR_INVALID_ARG;
  fill_all(reinterpret_cast<float*>(out), sizeof(*out) / sizeof(float), init_val);
  return RNAMC_OK;
}

int rnamc_fold_score_sets_accumulate(rnamc_fold_score_sets* f) {
  if (!f) return RNAMC_ERR_INVALID_ARG;
  // Five running f32 sums, each started at 0.0 (src/mccaskill_algo.rs:60-86).
  float sum = 0.f;
  for (int i = 0; i < RNAMC_MAX_LOOP_LEN + 1; i++) {
    sum += f->hairpin_scores_len[i];
    f->hairpin_scores_len_cumulative[i] = sum;
  }
  sum = 0.f;
  for (int i = 0; i < RNAMC_MAX_LOOP_LEN; i++) {
    sum += f->bulge_scores_len[i];
    f->bulge_scores_len_cumulative[i] = sum;
  }
  sum = 0.f;
  for (int i = 0; i < RNAMC_MAX_LOOP_LEN - 1; i++) {
    sum += f->interior_scores_len[i];
    f->interior_scores_len_cumulative[i] = sum;
  }
  sum = 0.f;
  for (int i = 0; i < RNAMC_MAX_INTERIOR_SYMMETRIC; i++) {
    sum += f->interior_scores_symmetric[i];
    f->interior_scores_symmetric_cumulative[i] = sum;
  }
  sum = 0.f;
  for (int i = 0; i < RNAMC_MAX_INTERIOR_ASYMMETRIC; i++) {
    sum += f->interior_scores_asymmetric[i];
    f->interior_scores_asymmetric_cumulative[i] = sum;
  }
  return RNAMC_OK;
}

int rnamc_fold_score_sets_transfer(rnamc_fold_score_sets* dst, const rnamc_fold_score_sets* src) {
  if (!dst || !src) return RNAMC_ERR_INVALID_ARG;
  std::memcpy(dst->hairpin_scores_len, src->hairpin_scores_len, sizeof(dst->hairpin_scores_len));
  std::memcpy(dst->bulge_scores_len, src->bulge_scores_len, sizeof(dst->bulge_scores_len));
  std::memcpy(dst->interior_scores_len, src->interior_scores_len, sizeof(dst->interior_scores_len));
  std::memcpy(dst->interior_scores_symmetric, src->interior_scores_symmetric,
              sizeof(dst->interior_scores_symmetric));
  std::memcpy(dst->interior_scores_asymmetric, src->interior_scores_asymmetric,
              sizeof(dst->interior_scores_asymmetric));
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      if (!is_canonical(i, j)) continue;  // closing pair must be canonical
      for (int k = 0; k < 4; k++)
        for (int l = 0; l < 4; l++) {
          dst->terminal_mismatch_scores[i][j][k][l] = src->terminal_mismatch_scores[i][j][k][l];
          if (is_canonical(k, l)) dst->stack_scores[i][j][k][l] = src->stack_scores[i][j][k][l];
        }
      for (int k = 0; k < 4; k++) {
        dst->dangling_scores_left[i][j][k] = src->dangling_scores_left[i][j][k];
        dst->dangling_scores_right[i][j][k] = src->dangling_scores_right[i][j][k];
      }
      dst->helix_close_scores[i][j] = src->helix_close_scores[i][j];
      dst->basepair_scores[i][j] = src->basepair_scores[i][j];
    }
  std::memcpy(dst->interior_scores_explicit, src->interior_scores_explicit,
              sizeof(dst->interior_scores_explicit));
  std::memcpy(dst->bulge_scores_0x1, src->bulge_scores_0x1, sizeof(dst->bulge_scores_0x1));
  std::memcpy(dst->interior_scores_1x1, src->interior_scores_1x1, sizeof(dst->interior_scores_1x1));
  dst->multibranch_score_base = src->multibranch_score_base;
  dst->multibranch_score_basepair = src->multibranch_score_basepair;
  dst->multibranch_score_unpair = src->multibranch_score_unpair;
  dst->external_score_basepair = src->external_score_basepair;
  dst->external_score_unpair = src->external_score_unpair;
  return rnamc_fold_score_sets_accumulate(dst);
}

int rnamc_params_new(float init_val, rnamc_params* out) {
  if (!out) return RNAMC_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  out->abi_version = RNAMC_ABI_VERSION;
  out->struct_bytes = static_cast<uint32_t>(sizeof(rnamc_params));
  out->table_id = 0;
  for (const FieldDesc& fd : kFields)
    fill_all(reinterpret_cast<float*>(reinterpret_cast<char*>(out) + fd.offset), fd.count, init_val);
  out->turner.num_special_hairpins = 0;
  out->turner.min_hairpin_len = 3;
  out->turner.max_hairpin_len_extrapolation = 9;
  out->turner.min_hairpin_len_extrapolation = 10;
  return RNAMC_OK;
}

int rnamc_params_synthetic(uint64_t seed, rnamc_params* out) {
  if (!out) return RNAMC_ERR_INVALID_ARG;
  rnamc_params_new(0.f, out);
  out->table_id = 0x53594E5400000000ull ^ seed;  // "SYNT" ^ seed
  SplitMix64 g(seed);
  rnamc_turner_scores& t = out->turner;
  // Turner-like magnitudes in units of -dG/kT: helices favourable, loops costly.
  for (int len = 0; len <= RNAMC_MAX_LOOP_LEN; len++) {
    t.hairpin_scores_init[len] = -(5.0f + 0.15f * len) + g.uni(-0.5f, 0.5f);
    t.bulge_scores_init[len] = -(5.5f + 0.12f * len) + g.uni(-0.3f, 0.3f);
    t.interior_scores_init[len] = -(1.5f + 0.10f * len) + g.uni(-0.3f, 0.3f);
  }
  fill_uni_ptr(g, &t.terminal_mismatch_scores_hairpin[0][0][0][0], 256, -0.5f, 2.0f);
  for (int a = 0; a < 4; a++)
    for (int b = 0; b < 4; b++)
      for (int c = 0; c < 4; c++)
        for (int d = 0; d < 4; d++)
          t.stack_scores[a][b][c][d] =
              (is_canonical(a, b) && is_canonical(c, d)) ? g.uni(1.0f, 5.0f) : g.uni(-1.0f, 1.0f);
  fill_uni_ptr(g, &t.interior_scores_1x1[0][0][0][0][0][0], 4096, -3.0f, 1.5f);
  fill_uni_ptr(g, &t.interior_scores_1x2[0][0][0][0][0][0][0], 16384, -5.0f, -1.0f);
  fill_uni_ptr(g, &t.interior_scores_2x2[0][0][0][0][0][0][0][0], 65536, -4.0f, 2.0f);
  fill_uni_ptr(g, &t.terminal_mismatch_scores_1xmany[0][0][0][0], 256, -0.5f, 1.0f);
  fill_uni_ptr(g, &t.terminal_mismatch_scores_2x3[0][0][0][0], 256, -0.5f, 1.2f);
  fill_uni_ptr(g, &t.terminal_mismatch_scores_interior[0][0][0][0], 256, -0.5f, 1.3f);
  fill_uni_ptr(g, &t.terminal_mismatch_scores_multibranch[0][0][0][0], 256, -0.3f, 1.3f);
  fill_uni_ptr(g, &t.dangling_scores_5prime[0][0][0], 64, 0.0f, 0.8f);
  fill_uni_ptr(g, &t.dangling_scores_3prime[0][0][0], 64, 0.0f, 1.3f);
  t.helix_augu_end_penalty = -0.73f + g.uni(-0.05f, 0.05f);
  t.coeff_hairpin_len_extrapolation = -1.75f;
  t.ninio_coeff = -0.97f + g.uni(-0.05f, 0.05f);
  t.ninio_max = -4.87f;
  t.init_multibranch_base = -6.0f + g.uni(-0.5f, 0.5f);
  t.coeff_num_branches = 0.5f + g.uni(-0.2f, 0.2f);
  // special hairpins: closing pair canonical so that they can occur
  static const int kSpecLens[3] = {5, 6, 8};
  static const int kPairs[6][2] = {{0, 3}, {1, 2}, {2, 1}, {2, 3}, {3, 0}, {3, 2}};
  t.num_special_hairpins = 24;
  for (uint32_t x = 0; x < t.num_special_hairpins; x++) {
    int len = kSpecLens[g.next() % 3];
    const int* pr = kPairs[g.next() % 6];
    t.special_hairpin_lens[x] = static_cast<uint8_t>(len);
    for (int y = 0; y < len; y++) t.special_hairpin_seqs[x][y] = static_cast<uint8_t>(g.next() & 3);
    t.special_hairpin_seqs[x][0] = static_cast<uint8_t>(pr[0]);
    t.special_hairpin_seqs[x][len - 1] = static_cast<uint8_t>(pr[1]);
    t.special_hairpin_scores[x] = g.uni(-4.0f, 1.0f);
  }

  // CONTRAfold-like raw ("compiled") tables, then FoldScoreSets::new(0.).transfer().
  rnamc_fold_score_sets raw;
  rnamc_fold_score_sets_new(0.f, &raw);
  for (int x = 0; x <= RNAMC_MAX_LOOP_LEN; x++)
    raw.hairpin_scores_len[x] = (x < 3) ? g.uni(-2.5f, -1.5f) : g.uni(-0.4f, 0.2f);
  for (int x = 0; x < RNAMC_MAX_LOOP_LEN; x++)
    raw.bulge_scores_len[x] = (x == 0) ? g.uni(-3.0f, -2.0f) : g.uni(-0.3f, 0.1f);
  for (int x = 0; x < RNAMC_MAX_LOOP_LEN - 1; x++)
    raw.interior_scores_len[x] = (x == 0) ? g.uni(-1.5f, -0.5f) : g.uni(-0.3f, 0.1f);
  fill_uni(g, raw.interior_scores_symmetric, -0.2f, 0.3f);
  fill_uni(g, raw.interior_scores_asymmetric, -0.5f, 0.0f);
  fill_uni_ptr(g, &raw.stack_scores[0][0][0][0], 256, 0.3f, 2.0f);
  fill_uni_ptr(g, &raw.terminal_mismatch_scores[0][0][0][0], 256, -0.5f, 0.8f);
  fill_uni_ptr(g, &raw.dangling_scores_left[0][0][0], 64, -0.2f, 0.5f);
  fill_uni_ptr(g, &raw.dangling_scores_right[0][0][0], 64, -0.2f, 0.5f);
  fill_uni_ptr(g, &raw.helix_close_scores[0][0], 16, -0.6f, 0.4f);
  fill_uni_ptr(g, &raw.basepair_scores[0][0], 16, 0.2f, 1.5f);
  fill_uni_ptr(g, &raw.interior_scores_explicit[0][0], 16, -0.5f, 0.5f);
  fill_uni(g, raw.bulge_scores_0x1, -0.3f, 0.3f);
  fill_uni_ptr(g, &raw.interior_scores_1x1[0][0], 16, -0.4f, 0.6f);
  raw.multibranch_score_base = g.uni(-3.5f, -2.5f);
  raw.multibranch_score_basepair = g.uni(-0.8f, -0.2f);
  raw.multibranch_score_unpair = g.uni(-0.15f, -0.05f);
  raw.external_score_basepair = g.uni(-0.2f, 0.1f);
  raw.external_score_unpair = g.uni(-0.08f, -0.02f);
  rnamc_fold_score_sets_new(0.f, &out->contra);
  return rnamc_fold_score_sets_transfer(&out->contra, &raw);
}

int rnamc_params_save(const rnamc_params* p, const char* path) {
  if (!p || !path) return RNAMC_ERR_INVALID_ARG;
  FILE* f = std::fopen(path, "wb");
  if (!f) return RNAMC_ERR_IO;
  uint32_t hdr[2] = {RNAMC_ABI_VERSION, static_cast<uint32_t>(sizeof(rnamc_params))};
  bool ok = std::fwrite(kMagic, 1, 8, f) == 8 && std::fwrite(hdr, 4, 2, f) == 2 &&
            std::fwrite(p, sizeof(*p), 1, f) == 1;
  ok = (std::fclose(f) == 0) && ok;
  return ok ? RNAMC_OK : RNAMC_ERR_IO;
}

int rnamc_params_load(const char* path, rnamc_params* out) {
  if (!out || !path) return RNAMC_ERR_INVALID_ARG;
  FILE* f = std::fopen(path, "rb");
  if (!f) return RNAMC_ERR_IO;
  char magic[8];
  uint32_t hdr[2];
  int st = RNAMC_OK;
  if (std::fread(magic, 1, 8, f) != 8 || std::fread(hdr, 4, 2, f) != 2) {
    st = RNAMC_ERR_FORMAT;
  } else if (std::memcmp(magic, kMagic, 8) != 0 || hdr[0] != RNAMC_ABI_VERSION ||
             hdr[1] != sizeof(rnamc_params)) {
    st = RNAMC_ERR_FORMAT;
  } else if (std::fread(out, sizeof(*out), 1, f) != 1) {
    st = RNAMC_ERR_FORMAT;
  } else if (out->abi_version != RNAMC_ABI_VERSION || out->struct_bytes != sizeof(rnamc_params) ||
             out->turner.num_special_hairpins > RNAMC_MAX_SPECIAL_HAIRPINS) {
    st = RNAMC_ERR_FORMAT;
  }
  std::fclose(f);
  return st;
}

int rnamc_params_set_special_hairpins(rnamc_params* p, uint32_t n, const uint8_t* seqs,
                                      const uint8_t* lens, const float* scores) {
  if (!p || n > RNAMC_MAX_SPECIAL_HAIRPINS || (n && (!seqs || !lens || !scores)))
    return RNAMC_ERR_INVALID_ARG;
  for (uint32_t x = 0; x < n; x++) {
    if (lens[x] == 0 || lens[x] > RNAMC_MAX_SPECIAL_HAIRPIN_LEN) return RNAMC_ERR_INVALID_ARG;
    for (uint32_t y = 0; y < lens[x]; y++)
      if (seqs[x * RNAMC_MAX_SPECIAL_HAIRPIN_LEN + y] > 3) return RNAMC_ERR_INVALID_BASE;
  }
  rnamc_turner_scores& t = p->turner;
  std::memset(t.special_hairpin_seqs, 0, sizeof(t.special_hairpin_seqs));
  std::memset(t.special_hairpin_lens, 0, sizeof(t.special_hairpin_lens));
  for (uint32_t x = 0; x < RNAMC_MAX_SPECIAL_HAIRPINS; x++) t.special_hairpin_scores[x] = 0.f;
  for (uint32_t x = 0; x < n; x++) {
    std::memcpy(t.special_hairpin_seqs[x], seqs + x * RNAMC_MAX_SPECIAL_HAIRPIN_LEN, lens[x]);
    t.special_hairpin_lens[x] = lens[x];
    t.special_hairpin_scores[x] = scores[x];
  }
  t.num_special_hairpins = n;
  return RNAMC_OK;
}

int rnamc_params_set_hairpin_limits(rnamc_params* p, uint32_t min_hairpin_len,
                                    uint32_t max_hairpin_len_extrapolation,
                                    uint32_t min_hairpin_len_extrapolation) {
  if (!p || min_hairpin_len_extrapolation < 2 ||
      max_hairpin_len_extrapolation > RNAMC_MAX_LOOP_LEN ||
      min_hairpin_len_extrapolation - 1 > RNAMC_MAX_LOOP_LEN)
    return RNAMC_ERR_INVALID_ARG;
  p->turner.min_hairpin_len = min_hairpin_len;
  p->turner.max_hairpin_len_extrapolation = max_hairpin_len_extrapolation;
  p->turner.min_hairpin_len_extrapolation = min_hairpin_len_extrapolation;
  return RNAMC_OK;
}

int rnamc_params_field(uint32_t idx, const char** name, uint64_t* byte_offset, uint64_t* count) {
  if (idx >= sizeof(kFields) / sizeof(kFields[0])) return RNAMC_ERR_INVALID_ARG;
  if (name) *name = kFields[idx].name;
  if (byte_offset) *byte_offset = kFields[idx].offset;
  if (count) *count = kFields[idx].count;
  return RNAMC_OK;
}

// gamma-centroid fold (src/centroid_fold.rs:25-105).  The recurrence is a
// Nussinov-style max-plus DP over a 0-initialised n*n matrix with strict '>'
// updates in the order left-skip, right-skip, pair, bifurcation; the traceback
// compares floats for exact equality in that same order, so identical bpp bits
// give identical structures.
int rnamc_centroid_fold(const float* bpp_packed, uint32_t n, float centroid_threshold,
                        uint32_t* pairs_out, uint32_t max_pairs, uint32_t* n_pairs,
                        float* expect_accuracy) {
  if (!bpp_packed || !n_pairs || n == 0) return RNAMC_ERR_INVALID_ARG;
  if (n > RNAMC_MAX_SEQ_LEN) return RNAMC_ERR_SEQ_TOO_LONG;
  std::vector<float> acc;
  try {
    acc.assign(static_cast<size_t>(n) * n, 0.f);
  } catch (...) {
    return RNAMC_ERR_OOM;
  }
  auto M = [&](size_t r, size_t c) -> float& { return acc[r * n + c]; };
  auto prob = [&](uint32_t i, uint32_t j) { return bpp_packed[rnamc_bpp_index(n, i, j)]; };
  for (uint32_t span = 2; span <= n; span++) {
    const float* diag = bpp_packed + rnamc_bpp_index(n, 0, span - 1);
    for (uint32_t i = 0; i + span <= n; i++) {
      uint32_t j = i + span - 1;
      float best = M(i + 1, j);
      float cand = M(i, j - 1);
      if (cand > best) best = cand;
      float pr = diag[i];
      if (pr >= -0.5f) {  // present in the SparseProbMat
        cand = M(i + 1, j - 1) + centroid_threshold * pr - 1.f;
        if (cand > best) best = cand;
      }
      for (uint32_t k = i + 1; k < j; k++) {
        cand = M(i, k) + M(k + 1, j);
        if (cand > best) best = cand;
      }
      M(i, j) = best;
    }
  }
  const uint32_t np = rnamc::centroid_traceback(
      n, centroid_threshold, [&](size_t r, size_t c) { return acc[r * n + c]; }, prob, pairs_out, max_pairs);
  *n_pairs = np;
  if (expect_accuracy) *expect_accuracy = M(0, n - 1);
  return RNAMC_OK;
}

// AlignScores::new (src/durbin_algo.rs:26-40)
int rnamc_align_scores_new(float init_val, rnamc_align_scores* out) {
  if (!out) return RNAMC_ERR_INVALID_ARG;
  out->match2match_score = out->match2insert_score = init_val;
  out->insert_extend_score = out->insert_switch_score = init_val;
  out->init_match_score = out->init_insert_score = init_val;
  for (int x = 0; x < RNAMC_NUM_BASES; x++) {
    out->insert_scores[x] = init_val;
    for (int y = 0; y < RNAMC_NUM_BASES; y++) out->match_scores[x][y] = init_val;
  }
  return RNAMC_OK;
}

// AlignScores::transfer (src/durbin_algo.rs:42-57).  The values are the generated CONTRAlign
// constants of the reference's src/compiled_align_scores.rs:2-19 (data, f32 like `Prob`).
int rnamc_align_scores_transfer(rnamc_align_scores* s) {
  if (!s) return RNAMC_ERR_INVALID_ARG;
  static const float kMatch[4][4] = {
      {0.5256508867f, -0.40906402f, -0.2502759109f, -0.3252306723f},
      {-0.40906402f, 0.6665219366f, -0.3289391181f, -0.1326088918f},
      {-0.2502759109f, -0.3289391181f, 0.6684676551f, -0.3565888168f},
      {-0.3252306723f, -0.1326088918f, -0.3565888168f, 0.459052045f},
  };
  static const float kInsert[4] = {-0.002521927159f, -0.08313891561f, -0.07443970653f,
                                   -0.01290054598f};
  s->match2match_score = 2.50575671f;
  s->match2insert_score = 0.1970448791f;
  s->insert_extend_score = 1.014026583f;
  s->insert_switch_score = -7.346968782f;
  s->init_match_score = 0.3959924457f;
  s->init_insert_score = -0.3488104904f;
  for (int x = 0; x < 4; x++) {
    s->insert_scores[x] = kInsert[x];
    for (int y = 0; y < 4; y++) s->match_scores[x][y] = kMatch[x][y];
  }
  return RNAMC_OK;
}

}  // extern "C"
