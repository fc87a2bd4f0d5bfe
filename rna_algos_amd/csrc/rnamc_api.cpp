// rnamc_api.cpp — device context and batch orchestration behind the C ABI.
//
// A batch of independent sequences (the reference runs one thread-pool task per
// record: src/bin/mccaskill_algo.rs:64-93) is sorted by length and cut into
// lock-step groups; each group sweeps the anti-diagonals of all its sequences
// together, one kernel launch per diagonal and pass, on one HIP stream.  The DP
// state of a whole group lives in HBM (ten packed triangles per sequence); the
// 288 GB of an MI355X hold thousands of sequences at once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <atomic>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "rnamc_device.h"
#include "rnamc_scoring.h"

using namespace rnamc;

#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
      (void)hipGetLastError(); /* reported: a later call must not find it again */         \
      return (_e == hipErrorOutOfMemory) ? RNAMC_ERR_OOM : RNAMC_ERR_HIP;                  \
    }                                                                                      \
  } while (0)

struct rnamc_ctx {
  int device = 0;
  rnamc_params host_params;
  rnamc_params* d_params = nullptr;
  float* d_hp_init = nullptr;
  std::vector<float> h_hp_init;  // host copy, for rnamc_fold_scores
  uint32_t hp_init_len = 0;
  float* d_ws = nullptr;
  uint64_t ws_floats = 0;
  SeqDesc* d_seqs = nullptr;
  uint64_t seqs_cap = 0;
  TreeSeq* d_tseqs = nullptr;  // descriptors of the tree-order mode
  uint64_t tseqs_cap = 0;
  std::vector<TreeSeq> h_tseqs;  // (host copy: source of an async upload, must outlive it)
  TreeTabs* d_tree_tabs = nullptr;  // 2-loop tables of the tree-order mode (built from the params)
  bool tree_tabs_valid = false;
  // host-buffer entry: device staging of bases / result / log partition (grow-only)
  // The result is staged per lock-step GROUP in two alternating device buffers: group g's
  // D2H (copy stream, pinned bounce chunks, a host thread) runs while group g+1 sweeps.
  uint8_t* st_bases = nullptr;
  float* st_out[2] = {nullptr, nullptr};
  float* st_logz = nullptr;
  uint64_t st_bases_cap = 0, st_out_cap[2] = {0, 0}, st_logz_cap = 0;
  hipStream_t copy_stream = nullptr;
  float* pinned[2] = {nullptr, nullptr};  // bounce chunks (hipHostMalloc)
  hipEvent_t pinned_ev[2] = {nullptr, nullptr};
  hipEvent_t group_done[2] = {nullptr, nullptr};
  std::vector<uint64_t> group_out_floats;  // per group, when the output is staged group-local
  hipStream_t own_stream = nullptr;
  hipStream_t aux_stream = nullptr;           // pair tail of large outside launches
  std::vector<hipEvent_t> ev_a, ev_b;         // per-diagonal completion, ring of 16
  std::recursive_mutex mu;  // rnamc_fold_scores re-enters rnamc_bpp_batch
  // knobs
  // 0: every logsumexp fold in the reference's order (the parity gate); 1: order-free sums
  // (rnamc_tree.hip), not bit-comparable with the reference
  int64_t summation_mode = 0;
  int64_t tree_tpc = 0;  // tree mode: threads per cell (64 / 256 / 1024), 0 = by diagonal size
  int64_t tree_two = 1;  // tree mode: two diagonals per launch
  // tree mode: width of a band of diagonals whose products take their mid-field from k_tree_mid
  // (a multiple of 32, at most 128; 0: every launch walks its sums whole)
  int64_t tree_band = 64;
  // tree mode: lane-per-cell sweeps (rnamc_tree_lane.h) — 0 never, 1 for batches (a call of at least
  // tree_lane_min_nt nucleotides whose sweeps are banded), 2 always
  int64_t tree_lane = 1;
  int64_t tree_lane_min_nt = 65536;
  // lane-per-cell sweeps: a band's mid-field kernel runs in front of the band on the sweep's stream
  // (threshold = the band's first / last diagonal) instead of a band ahead beside it
  int64_t tree_mid_sync = 1;
  // lane-per-cell sweeps with the mid-field in front of its band: the band's width (the in-band terms are
  // the lanes' own loops: narrower bands, fewer of them; the matrix-core mid-field takes the rest)
  int64_t tree_lane_band = 32;
  // lane-per-cell sweeps: diagonals whose generic 2-loop sums share a launch (k_tlane_gen), 1 .. 3
  int64_t tree_gen_batch = 3;
  // tree mode, banded sweeps: the far part of a launch's 2-loop blocks is summed by extra
  // workgroups of the previous launch (rnamc_tree.hip, Ahead)
  int64_t tree_ahead = 1;
  TreePolicy tree_pol;  // launch shapes of the tree-order sweep ("tree_waves", "tree_short", ...)
  hipStream_t bulk_stream = nullptr;  // k_tree_mid, beside the sweep (lowest priority)
  // tree mode, batch form: every other group of a call sweeps on a second stream with its own half of the
  // workspace, side stream and event rings — two groups side by side fill each other's launch gaps (the
  // per-diagonal launches cost ~8 us whatever they hold); "tree_dual" 0 switches it off
  int64_t tree_dual = 1;
  hipStream_t dual_stream = nullptr, bulk_stream2 = nullptr;
  std::vector<hipEvent_t> ev_a2, ev_b2;
  hipEvent_t ev_dual = nullptr;
  // does bulk_stream run beside the stream of the last banded call?  (probed once per stream:
  // tree_side_stream_probe; 0 unknown, 1 yes, 2 no -> unbanded sweeps on that stream)
  hipStream_t side_probed_for = nullptr;
  bool side_probed = false;
  int side_verdict = 0;
  int64_t tree_side_force = 0;  // knob "tree_side_stream": 0 probe, 1 take it as concurrent, 2 as serialised
  int64_t tree_debug = 0;  // (RNAMC_DEBUG_KNOBS builds: bit 0 no 2-loops, 1 no products, 2 empty kernels)
  int64_t group_max_seqs = 8192;
  int64_t group_max_nt = 2ll << 20;  // a group holds ~2M nucleotides (or 64 GB of DP state)
  int64_t group_ws_bytes = 64ll << 30;
  bool group_ws_user = false;  // the knob was set: the tree-order batch form takes it as given
  int64_t block_threads = 256;
  int64_t profile = 0;
  // dispatch order of the role blocks of a launch (measured: pair-probability chains first,
  // probs_multibranch last is 2.5 % faster than the reverse; the inside order does not matter)
  int64_t order_inside = 0, order_outside = 1;
  int64_t dual_outside = 1;   // large outside launches: pair tail as its own kernel/stream
  uint64_t dual_min_cells = 256 * 1024;
  int64_t dual_max_diag = 1 << 30;  // ... while the diagonal has at most this many cells
  bool inside_only = false;  // set by rnamc_fold_scores around its own batch call
  int64_t fuse_inside = 1;  // Turner: fold two diagonals per launch where launches are large
  // latency forms (rnamc_latency.h) for groups that cannot fill the chip: 0 never, 1 when the
  // group's longest diagonal holds at most lat_max_cells cells over all its sequences (half of
  // that under CONTRAfold), 2 always
  int64_t latency_mode = 1;
  int64_t lat_max_cells = 32768;
  // inside folds of such a group: lat_inside != 0 takes the eight-chains-per-wave form (8-lane
  // speculative logsumexp) on the diagonals whose launches need at most lat_e_waves waves (beyond
  // ~2 waves per SIMD that form is issue-bound and loses: profiles/r02_latency_forms.txt), the
  // three-lanes-per-cell form elsewhere
  int64_t lat_inside = 2;
  int64_t lat_e_waves = 2048;
  // debug: probs_multibranch and the pair-probability chains as two launches (timing splits)
  int64_t lat_split = 0;
  // one launch per diagonal in a latency-form group (chains + 2-loop blocks), no second stream
  int64_t lat_merge = 1;
  // CONTRAfold, eight-chains form: the sums_rightmost_basepairs folds run one launch ahead
  int64_t lat_zr_ahead = 1;
  int64_t lat_pairs = 1;   // its 2-loop blocks run one wave per listed cell (both sweeps)
  // role mask of timing experiments (bit0 folds, 1 pair block, 2 mb, 3 pair probs); settable
  // only in builds with -DRNAMC_DEBUG_KNOBS (make DEBUG_KNOBS=1), constant 15 otherwise
  int64_t debug_roles = 15;
  // bookkeeping of the last call
  rnamc_batch_stats stats{};
  std::vector<SeqDesc> descs;       // all groups, group-major
  std::vector<uint32_t> group_begin;  // prefix into descs
  std::vector<hipEvent_t> events;
  std::vector<hipEvent_t> kev;        // per-launch event pairs of the outside kernels (profiling)
  std::vector<uint8_t> kev_class;     // 0 main, 1 tail, 2 / 3 small; one per pair
  // rnamc_fold_scores: sums_close key set of the last sequence it swept, so that the usual
  // "count, allocate, fill" pair of calls runs the device sweep once
  std::vector<uint8_t> fs_bases;
  std::vector<float> fs_qb;
  int fs_contra = -1, fs_short = -1;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

uint64_t tri_pad_of(uint32_t n) {
  // + 64 floats: kernels read up to one wave past a diagonal's end (values masked).
  // Column-major slots pad every column to 16 floats: 16 (m+1)(8m + r) floats for n columns.
  uint64_t t = static_cast<uint64_t>(n) * (n + 1ull) / 2ull;
  const uint64_t m = n >> 4, r = n & 15ull;
  t = std::max<uint64_t>(t, 16ull * (m + 1ull) * (8ull * m + r));
  return ((t + 63ull) & ~63ull) + 64ull;
}

int ensure_hp_init(rnamc_ctx* c, uint32_t max_n) {
  if (c->hp_init_len >= max_n + 1 && c->d_hp_init) return RNAMC_OK;
  const rnamc_turner_scores& t = c->host_params.turner;
  uint32_t len = std::max<uint32_t>(max_n + 1, 64);
  std::vector<float> hp(len);
  const uint32_t m = t.min_hairpin_len_extrapolation - 1;
  for (uint32_t l = 0; l < len; l++) {
    if (l <= t.max_hairpin_len_extrapolation) {
      hp[l] = t.hairpin_scores_init[std::min<uint32_t>(l, RNAMC_MAX_LOOP_LEN)];
    } else {
      // HAIRPIN_SCORES_INIT[MIN-1] + COEFF * ln(len / (MIN-1)), all f32 (src/utils.rs:181-183)
      const float ratio = static_cast<float>(l) / static_cast<float>(m);
      const float lg = logf(ratio);
      const float scaled = t.coeff_hairpin_len_extrapolation * lg;
      hp[l] = t.hairpin_scores_init[m] + scaled;
    }
  }
  if (c->d_hp_init) HIPCHK(hipFree(c->d_hp_init));
  c->d_hp_init = nullptr;
  c->hp_init_len = 0;
  HIPCHK(hipMalloc(&c->d_hp_init, sizeof(float) * len));
  HIPCHK(hipMemcpy(c->d_hp_init, hp.data(), sizeof(float) * len, hipMemcpyHostToDevice));
  c->hp_init_len = len;
  c->h_hp_init = std::move(hp);
  return RNAMC_OK;
}

int ensure_ws(rnamc_ctx* c, uint64_t floats) {
  if (c->ws_floats >= floats) return RNAMC_OK;
  if (c->d_ws) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipFree(c->d_ws));
    c->d_ws = nullptr;
    c->ws_floats = 0;
  }
  HIPCHK(hipMalloc(&c->d_ws, floats * sizeof(float)));
  c->ws_floats = floats;
  return RNAMC_OK;
}

// Per-group hooks of the host-buffer entry: where group g's result goes, and what happens
// once its work is enqueued.  With hooks the output offsets are group-local (packed in group
// order), without them the caller's out_offsets address one device buffer.
struct GroupHooks {
  std::function<int(size_t g, float** out_base)> before;
  std::function<int(size_t g, uint32_t first_desc, uint32_t n_desc)> after;
};

// Core: everything device-resident, work enqueued on `st`.
int run_batch(rnamc_ctx* c, uint32_t n_seqs, const uint8_t* d_bases, const uint64_t* offsets,
              bool contra, bool allows_short, float* d_out, const uint64_t* out_offsets,
              float* d_logz, hipStream_t st, const GroupHooks* hooks = nullptr) {
  c->stats = rnamc_batch_stats{};
  c->kev_class.clear();
  c->descs.clear();
  c->group_begin.clear();
  c->group_out_floats.clear();
  if (n_seqs == 0) return RNAMC_OK;
  uint32_t max_n = 0;
  for (uint32_t s = 0; s < n_seqs; s++) {
    if (offsets[s + 1] < offsets[s]) return RNAMC_ERR_INVALID_ARG;
    const uint64_t n = offsets[s + 1] - offsets[s];
    if (n == 0) return RNAMC_ERR_EMPTY_SEQ;
    if (n > RNAMC_MAX_SEQ_LEN) return RNAMC_ERR_SEQ_TOO_LONG;
    max_n = std::max<uint32_t>(max_n, static_cast<uint32_t>(n));
  }
  int rc = ensure_hp_init(c, max_n);
  if (rc) return rc;

  // longest first: within a group the sequences active on diagonal d are a prefix
  std::vector<uint32_t> order(n_seqs);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    return (offsets[a + 1] - offsets[a]) > (offsets[b + 1] - offsets[b]);
  });

  const uint64_t ws_cap_floats = static_cast<uint64_t>(std::max<int64_t>(c->group_ws_bytes, 1)) / 4;
  uint64_t max_group_floats = 0;
  {
    uint64_t cur = 0, cur_nt = 0, cur_out = 0;
    uint32_t cnt = 0;
    for (uint32_t x = 0; x < n_seqs; x++) {
      const uint32_t s = order[x];
      const uint32_t n = static_cast<uint32_t>(offsets[s + 1] - offsets[s]);
      const uint64_t pk_words = ((static_cast<uint64_t>(n) + 160) / 16 + 4 + 63) & ~63ull;
      const uint64_t cidx_words = (tri_pad_of(n) + 1) / 2;  // u16 per cell, in 4-byte units
      const uint64_t ccnt_words = (static_cast<uint64_t>(n) + 63) & ~63ull;
      const uint64_t c64_words =
          ((static_cast<uint64_t>(n) + 63) / 64 * (static_cast<uint64_t>(n) + 64) + 63) & ~63ull;
      const uint64_t need =
          tri_pad_of(n) * M_COUNT + pk_words + cidx_words + ccnt_words + c64_words;
      // a launch should carry enough cells to fill the chip: short sequences go into
      // larger groups (bounded by nucleotides, sequences and workspace bytes)
      if (cnt > 0 && (cnt >= static_cast<uint32_t>(c->group_max_seqs) ||
                      cur_nt + n > static_cast<uint64_t>(c->group_max_nt) ||
                      cur + need > ws_cap_floats)) {
        max_group_floats = std::max(max_group_floats, cur);
        c->group_out_floats.push_back(cur_out);
        cur = 0;
        cur_nt = 0;
        cur_out = 0;
        cnt = 0;
      }
      if (cnt == 0) c->group_begin.push_back(static_cast<uint32_t>(c->descs.size()));
      SeqDesc sd{};
      sd.n = n;
      sd.tri_pad = static_cast<uint32_t>(tri_pad_of(n));
      sd.seq_off = offsets[s];
      sd.ws_off = cur;
      sd.out_off = hooks ? cur_out : out_offsets[s];
      sd.batch_idx = s;
      sd.pk_words = static_cast<uint32_t>(pk_words);
      sd.pk_off = cur + tri_pad_of(n) * M_COUNT;
      sd.cidx_off = sd.pk_off + pk_words;
      sd.ccnt_off = sd.cidx_off + cidx_words;
      sd.c64_off = sd.ccnt_off + ccnt_words;
      c->descs.push_back(sd);
      cur += need;
      cur_nt += n;
      cur_out += rnamc_bpp_len(n);
      cnt++;
    }
    max_group_floats = std::max(max_group_floats, cur);
    c->group_out_floats.push_back(cur_out);
    c->group_begin.push_back(static_cast<uint32_t>(c->descs.size()));
  }
  rc = ensure_ws(c, max_group_floats);
  if (rc) return rc;
  if (c->seqs_cap < c->descs.size()) {
    if (c->d_seqs) {
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipFree(c->d_seqs));
      c->d_seqs = nullptr;
      c->seqs_cap = 0;
    }
    const uint64_t cap = std::max<uint64_t>(c->descs.size(), 1024);
    HIPCHK(hipMalloc(&c->d_seqs, cap * sizeof(SeqDesc)));
    c->seqs_cap = cap;
  }
  HIPCHK(hipMemcpyAsync(c->d_seqs, c->descs.data(), c->descs.size() * sizeof(SeqDesc),
                        hipMemcpyHostToDevice, st));

  const size_t n_groups = c->group_begin.size() - 1;
  const bool prof = c->profile != 0;
  if (prof) {
    const size_t need = n_groups * 4;
    while (c->events.size() < need) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      c->events.push_back(e);
    }
  }
  const uint32_t block = static_cast<uint32_t>(c->block_threads);
  const uint32_t dmin_in = contra ? 0u : (RNAMC_MIN_SPAN_HAIRPIN_CLOSE - 1);
  const uint32_t dmin_out = (contra && allows_short) ? 1u : (RNAMC_MIN_SPAN_HAIRPIN_CLOSE - 1);

  for (size_t g = 0; g < n_groups; g++) {
    const uint32_t gb = c->group_begin[g], ge = c->group_begin[g + 1];
    const uint32_t nseq = ge - gb;
    const uint32_t gmax = c->descs[gb].n;
    DeviceBatch b{};
    b.seqs = c->d_seqs + gb;
    b.bases = d_bases;
    b.workspace = c->d_ws;
    b.out = d_out;
    if (hooks) {
      rc = hooks->before(g, &b.out);
      if (rc) return rc;
    }
    b.log_partition = d_logz;
    b.params = c->d_params;
    b.hp_init = c->d_hp_init;
    b.allows_short_hairpins = allows_short ? 1 : 0;
    b.order_inside = static_cast<int>(c->order_inside);
    b.order_outside = static_cast<int>(c->order_outside);
    // sequences with n > d form a prefix of the group
    auto active = [&](uint32_t d) {
      uint32_t lo = 0, hi = nseq;  // first index with n <= d
      while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (c->descs[gb + mid].n > d) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 0], st));
    launch_init(b, nseq, gmax, st);
    c->stats.launches_other++;
    // Inside sweep.  Dependencies: the closing-pair block of diagonal D is a left fold whose
    // early part (hairpin, 2-loops) needs sums_close of diagonals <= D-2 and whose last term
    // needs the folds of diagonal D-2; the folds of diagonal D need the pair blocks of
    // diagonals <= D (<= D+1 for the second cell of a two-diagonal launch).
    //   one-diagonal launch d : folds(d) beside the whole pair block of d+1
    //   two-diagonal launch d : folds(d, d+1) beside the early part of pair blocks d+2, d+3;
    //                           their last term follows in a small launch of its own
    // `pairs_done`: pair blocks complete up to here; `heads_done`: early part parked.
    const bool do_sums = (c->debug_roles & 1) != 0, do_pair = (c->debug_roles & 2) != 0;
    int64_t pairs_done = static_cast<int64_t>(dmin_in) - 1;  // nothing pairs below dmin_in
    int64_t heads_done = pairs_done;
    const uint32_t ring = static_cast<uint32_t>(c->ev_a.size());
    auto need_pairs = [&](int64_t upto) {  // complete the pair blocks of diagonals <= upto
      upto = std::min<int64_t>(upto, static_cast<int64_t>(gmax) - 1);
      while (pairs_done < upto) {
        const uint32_t D = static_cast<uint32_t>(pairs_done + 1);
        if (static_cast<int64_t>(D) <= heads_done) {
          const uint32_t nd = static_cast<uint32_t>(std::min<int64_t>(heads_done, upto)) - D + 1;
          if (do_pair) {
            launch_pair_tail(b, contra, D, nd, gmax, active(D), block, st);
            c->stats.launches_inside++;
          }
          pairs_done = D + nd - 1;
        } else {
          if (D >= 1 && do_pair) {
            launch_inside(b, contra, D - 1, gmax, active(D), block, false, true, st);
            c->stats.launches_inside++;
          }
          pairs_done = D;
          heads_done = std::max(heads_done, pairs_done);
        }
      }
    };
    // Latency-form group: folds(d) on `st` (k_inside_lat) beside the pair block of d+1 on
    // aux_stream; folds(d) need the pair block of d (aux, step before), the pair block of
    // d+1 needs the folds of d-1 (st, step before).
    // (CONTRAfold's chains hold more general steps: its crossover against the batch forms lies
    // at half the cells, profiles/r02_latency_forms.txt)
    const bool lat = c->latency_mode == 2 ||
                     (c->latency_mode == 1 &&
                      static_cast<uint64_t>(nseq) * gmax <=
                          static_cast<uint64_t>(contra ? c->lat_max_cells / 2 : c->lat_max_cells));
    const bool lat_in = lat && (c->lat_inside != 0 || c->lat_pairs != 0);
    if (lat_in) {
      bool have_a = false, have_b = false, combine_due = false, zr_parked = false;
      for (uint32_t d = dmin_in; d < gmax; d++) {
        need_pairs(d);  // (only the first diagonal finds work here)
        const bool pair_next = heads_done < static_cast<int64_t>(d) + 1 && d + 1 < gmax;
        const uint32_t pv = (d + ring - 1) % ring, cu = d % ring;
        // The eight-chains form completes sums_1ormore_basepairs of diagonal d-1 in the launch
        // of diagonal d (sequences that end at d-1 included).
        const uint64_t chains = 3ull * (gmax - d) * active(d);
        // (CONTRAfold, eight-chains form: two more chain kinds per cell, see lat_zr_ahead)
        const uint64_t kinds_e = (contra && c->lat_zr_ahead != 0) ? 5 : 3;
        // (CONTRAfold: three times the waves — its three-lanes form folds two chains per cell
        // one after the other, measured crossover in profiles/r02_latency_forms.txt)
        const uint64_t e_waves = static_cast<uint64_t>(c->lat_e_waves) * (contra ? 3 : 1);
        // eight chains per wave on the diagonals with few enough waves
        const int form = (do_sums && c->lat_inside != 0 && (chains / 3 * kinds_e + 7) / 8 <= e_waves) ? 2 : 0;
        const bool wave_form = form != 0;
        // CONTRAfold: a cell's sums_rightmost_basepairs folds (d steps) precede its other
        // folds (d steps more); all but their last step needs nothing of diagonal d, so the
        // eight-chains launch of diagonal d-1 runs them ahead (flag 4: do so for d+1, flag 8:
        // this diagonal's were parked)
        const bool zr_ahead = form == 2 && kinds_e == 5;
        const int form_arg = form | (zr_ahead ? 4 : 0) | (form == 2 && zr_parked ? 8 : 0);
        // the closing-pair blocks of diagonal d+1 ride in the same launch as the wave-form
        // chains of diagonal d (one launch per diagonal, no cross-stream events: ~12 us per
        // diagonal less than the two-stream schedule, profiles/r02_latency_forms.txt)
        const bool merged = wave_form && c->lat_pairs != 0 && c->lat_merge != 0;
        if (pair_next && do_pair && !merged) {
          if (!have_a) {  // everything so far is on `st`
            HIPCHK(hipEventRecord(c->ev_a[pv], st));
            have_a = true;
          }
          HIPCHK(hipStreamWaitEvent(c->aux_stream, c->ev_a[pv], 0));
          if (c->lat_pairs != 0) {
            launch_pair_lat(b, contra, false, d + 1, gmax, active(d + 1), c->aux_stream);
          } else {
            launch_inside(b, contra, d, gmax, active(d + 1), block, false, true, c->aux_stream);
          }
          c->stats.launches_inside++;
        }
        if (have_b) HIPCHK(hipStreamWaitEvent(st, c->ev_b[pv], 0));
        if (do_sums) {
          const uint32_t pair_d = (merged && pair_next && do_pair) ? d + 1 : 0;
          if (wave_form || combine_due) {
            launch_inside_lat(b, contra, d, gmax, active(d >= 1 ? d - 1 : 0), form_arg, combine_due, pair_d, st);
            c->stats.launches_inside++;
          }
          if (!wave_form) {
            launch_inside(b, contra, d, gmax, active(d), block, true, false, st);
            c->stats.launches_inside++;
          }
          combine_due = wave_form;
        }
        zr_parked = zr_ahead;
        if (merged) {
          have_a = false;  // (recorded when a later diagonal needs it)
        } else {
          HIPCHK(hipEventRecord(c->ev_a[cu], st));
          have_a = true;
        }
        if (pair_next) {
          if (!merged) HIPCHK(hipEventRecord(c->ev_b[cu], c->aux_stream));
          have_b = !merged;
          pairs_done = heads_done = d + 1;
        } else {
          have_b = false;
        }
      }
      if (have_b) HIPCHK(hipStreamWaitEvent(st, c->ev_b[(gmax - 1) % ring], 0));
      if (combine_due)  // combine of the last diagonal
        launch_inside_lat(b, contra, gmax, gmax, active(gmax - 1), 0, true, 0, st);
    }
    for (uint32_t d = dmin_in; d < gmax && !lat_in;) {
      const bool fuse = c->fuse_inside != 0 && d >= 2 && d + 1 < gmax &&
                        !inside_is_split(d, gmax, active(d));
      if (fuse) {
        need_pairs(static_cast<int64_t>(d) + 1);
        const bool head = heads_done < static_cast<int64_t>(d) + 2 && d + 2 < gmax;
        if (contra && do_sums) {
          launch_inside_zr2(b, d, gmax, active(d), block, st);
          c->stats.launches_inside++;
        }
        launch_inside2(b, contra, d, gmax, active(d), block, do_sums, head && do_pair, st);
        c->stats.launches_inside++;
        if (head) heads_done = std::min<int64_t>(static_cast<int64_t>(d) + 3, gmax - 1);
        d += 2;
      } else {
        need_pairs(d);
        // the whole pair block of d+1 rides along unless its early part is parked already
        const bool pair_next = heads_done < static_cast<int64_t>(d) + 1 && d + 1 < gmax;
        launch_inside(b, contra, d, gmax, active(d), block, do_sums, pair_next && do_pair, st);
        c->stats.launches_inside++;
        if (pair_next) pairs_done = heads_done = d + 1;
        d += 1;
      }
    }
    if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 1], st));
    // (rnamc_fold_scores needs the sums_close key set only: no outside sweep; the output
    // triangle then holds -1 / expf of stale log-probabilities and is not looked at)
    if (!c->inside_only)
    // Outside sweep.  Launch d carries probs_multibranch and the pair tail of diagonal d and
    // the 2-loop half (pair head) of diagonal d-1: start one diagonal early.  Large launches
    // run the pair tail as its own kernel on a second stream beside the other two roles
    // (its register footprint would otherwise set their occupancy); both kernels of
    // diagonal d need both kernels of diagonal d+1.
    {
      const bool r_mb = (c->debug_roles & 4) != 0, r_pp = (c->debug_roles & 8) != 0;
      // (debug builds: bit 4 drops the 2-loop half, bit 5 the multibranch half)
      const bool r_head = r_pp && (c->debug_roles & 16) == 0, r_tail = r_pp && (c->debug_roles & 32) == 0;
      bool dual = false;  // the previous diagonal ran as two kernels
      // profiling: a pair of events around every kernel, on the stream it is launched on
      auto timed = [&](uint8_t cls, hipStream_t s, auto&& launch) {
        if (c->profile < 2) {
          launch();
          return;
        }
        const size_t x = c->kev_class.size();
        while (c->kev.size() < 2 * (x + 1)) {
          hipEvent_t e = nullptr;
          if (hipEventCreate(&e) != hipSuccess) {  // out of events: stop timing, keep running
            launch();
            return;
          }
          c->kev.push_back(e);
        }
        (void)hipEventRecord(c->kev[2 * x], s);
        launch();
        (void)hipEventRecord(c->kev[2 * x + 1], s);
        c->kev_class.push_back(cls);
      };
      if (lat) {
        // latency-form group: {probs_multibranch, multibranch half of the pair probabilities}
        // of diagonal d on `st` (k_outside_lat) beside the 2-loop half of diagonal d-1 on
        // aux_stream; both need both of diagonal d+1
        // (lat_merge: both in ONE launch on `st`, no events)
        const bool merged = c->lat_pairs != 0 && c->lat_merge != 0 && c->lat_split == 0;
        bool first = true;
        for (uint32_t d = gmax + 1; d-- > dmin_out && merged;) {
          const bool head = d >= 1 && d - 1 >= dmin_out && r_head;
          if (d < gmax || head) {
            timed(0, st, [&]() {
              launch_outside_lat(b, contra, d, gmax, active(d >= 1 ? d - 1 : 0), r_mb, r_tail, head, st);
            });
            c->stats.launches_outside++;
          }
        }
        for (uint32_t d = gmax + 1; d-- > dmin_out && !merged;) {
          const bool head = d >= 1 && d - 1 >= dmin_out;
          const uint32_t na = active(d >= 1 ? d - 1 : 0);
          const uint32_t pv = (d + 1) % ring, cu = d % ring;
          if (first) {
            HIPCHK(hipEventRecord(c->ev_a[pv], st));
          } else {
            HIPCHK(hipStreamWaitEvent(st, c->ev_b[pv], 0));
          }
          HIPCHK(hipStreamWaitEvent(c->aux_stream, c->ev_a[pv], 0));
          if (d < gmax) {
            if (c->lat_split != 0) {
              timed(1, st, [&]() { launch_outside_lat(b, contra, d, gmax, active(d), r_mb, false, false, st); });
              timed(0, st, [&]() { launch_outside_lat(b, contra, d, gmax, active(d), false, r_tail, false, st); });
            } else {
              timed(0, st, [&]() { launch_outside_lat(b, contra, d, gmax, active(d), r_mb, r_tail, false, st); });
            }
            c->stats.launches_outside++;
          }
          HIPCHK(hipEventRecord(c->ev_a[cu], st));
          if (head && r_head) {
            timed(3, c->aux_stream, [&]() {
              if (c->lat_pairs != 0) {
                launch_pair_lat(b, contra, true, d - 1, gmax, na, c->aux_stream);
              } else {
                launch_outside(b, contra, d, gmax, na, block, false, false, true, 4, c->aux_stream);
              }
            });
            c->stats.launches_outside++;
          }
          HIPCHK(hipEventRecord(c->ev_b[cu], c->aux_stream));
          first = false;
        }
        if (!first) HIPCHK(hipStreamWaitEvent(st, c->ev_b[dmin_out % ring], 0));
      }
      for (uint32_t d = gmax + 1; d-- > dmin_out && !lat;) {
        const bool head = d >= 1 && d - 1 >= dmin_out;
        const uint32_t na = active(d >= 1 ? d - 1 : 0);
        // worth it where the 2-loop blocks dominate a launch: the folds of a cell grow with
        // the length of the diagonal, its 496 probes do not
        const bool want_dual = c->dual_outside != 0 && d < gmax &&
                               gmax - d <= static_cast<uint64_t>(c->dual_max_diag) &&
                               static_cast<uint64_t>(gmax - d) * na >= c->dual_min_cells;
        if (!want_dual) {
          if (dual) {  // back to one stream: wait for the other kernel of d+1
            HIPCHK(hipStreamWaitEvent(st, c->ev_b[(d + 1) % ring], 0));
            dual = false;
          }
          timed(2, st, [&]() {
            launch_outside(b, contra, d, gmax, na, block, r_mb, r_tail, head && r_head, 7, st);
          });
          c->stats.launches_outside++;
        } else {
          // both kernels of diagonal d need both kernels of diagonal d+1
          hipEvent_t ea = c->ev_a[d % ring], eb = c->ev_b[d % ring];
          const uint32_t pv = (d + 1) % ring;
          if (!dual) {
            // first two-kernel diagonal: everything so far is on `st`
            HIPCHK(hipEventRecord(c->ev_a[pv], st));
          } else {
            HIPCHK(hipStreamWaitEvent(st, c->ev_b[pv], 0));
          }
          HIPCHK(hipStreamWaitEvent(c->aux_stream, c->ev_a[pv], 0));
          timed(0, st, [&]() {
            launch_outside(b, contra, d, gmax, na, block, r_mb, false, head && r_head, 5, st);
          });
          HIPCHK(hipEventRecord(ea, st));
          timed(1, c->aux_stream, [&]() {
            launch_outside(b, contra, d, gmax, na, block, false, r_tail, false, 2, c->aux_stream);
          });
          HIPCHK(hipEventRecord(eb, c->aux_stream));
          c->stats.launches_outside += 2;
          dual = true;
        }
      }
      if (dual) HIPCHK(hipStreamWaitEvent(st, c->ev_b[dmin_out % ring], 0));
    }
    if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 2], st));
    launch_finalize(b, nseq, gmax, dmin_out, st);
    c->stats.launches_other++;
    if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 3], st));
    HIPCHK(hipGetLastError());
    if (hooks) {
      rc = hooks->after(g, gb, nseq);
      if (rc) return rc;
    }
  }
  c->stats.n_groups = n_groups;
  c->stats.workspace_bytes = c->ws_floats * sizeof(float);
  if (prof) {
    HIPCHK(hipStreamSynchronize(st));
    for (size_t g = 0; g < n_groups; g++) {
      float a = 0, bms = 0, cc = 0;
      HIPCHK(hipEventElapsedTime(&a, c->events[4 * g + 0], c->events[4 * g + 1]));
      HIPCHK(hipEventElapsedTime(&bms, c->events[4 * g + 1], c->events[4 * g + 2]));
      HIPCHK(hipEventElapsedTime(&cc, c->events[4 * g + 2], c->events[4 * g + 3]));
      c->stats.ms_inside += a;
      c->stats.ms_outside += bms;
      c->stats.ms_other += cc;
    }
    for (size_t x = 0; x < c->kev_class.size(); x++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, c->kev[2 * x], c->kev[2 * x + 1]) != hipSuccess) continue;
      if (c->kev_class[x] == 0) {
        c->stats.ms_outside_main += ms;
        c->stats.launches_outside_main++;
      } else if (c->kev_class[x] == 1) {
        c->stats.ms_outside_tail += ms;
        c->stats.launches_outside_tail++;
      } else {
        c->stats.ms_outside_small += ms;
        c->stats.launches_outside_small++;
      }
    }
  }
  return RNAMC_OK;
}

// Length-dependent part of the generic 2-loop scores of the tree-order mode (TreeTabs), from
// the parameter block.  Turner: bulge_scores_init[len] | interior_scores_init[len] +
// max(ninio_coeff * |a-b|, ninio_max) (src/utils.rs:234-321); CONTRAfold: the cumulative
// bulge / interior length, symmetric / asymmetric and explicit terms (456-520).
void build_tree_tabs(const rnamc_params& P, TreeTabs& T) {
  std::memset(&T, 0, sizeof(T));
  const rnamc_turner_scores& t = P.turner;
  const rnamc_fold_score_sets& f = P.contra;
  for (uint32_t p = 0; p < 512; p++) {
    const uint32_t r = p >> 5, c = p & 31u;
    const bool first = c < 31u - r;
    const uint32_t a = first ? r : 30u - r, b = first ? c : c - (31u - r);
    if (!(r < 15u || c < 16u)) continue;  // not a probe slot
    const uint32_t len = a + b, diff = a > b ? a - b : b - a;
    const bool bulge = (a == 0u) != (b == 0u);
    if (len < 2u) continue;  // stack / 0x1: scored by the flat scorers
    if (bulge) {
      T.len[0][p] = t.bulge_scores_init[len];
      T.len[1][p] = f.bulge_scores_len_cumulative[len - 1u];
    } else if (a >= 1u && b >= 1u) {
      const float nin = t.ninio_coeff * static_cast<float>(diff);
      T.len[0][p] = t.interior_scores_init[len] + (nin > t.ninio_max ? nin : t.ninio_max);
      const float s0 = (a == b) ? f.interior_scores_symmetric_cumulative[a - 1u]
                                : f.interior_scores_asymmetric_cumulative[diff - 1u];
      const float se = (a <= RNAMC_MAX_INTERIOR_EXPLICIT && b <= RNAMC_MAX_INTERIOR_EXPLICIT)
                           ? f.interior_scores_explicit[a - 1u][b - 1u]
                           : 0.f;
      T.len[1][p] = (s0 + se) + f.interior_scores_len_cumulative[len - 2u];
    }
  }
  // the generic slots by class, then a + b (lane-per-cell sweeps); classes as slot_class of rnamc_tree.hip
  for (int m = 0; m < 2; m++) {
    uint32_t cnt = 0;
    for (uint32_t c = 0; c < 4u; c++) {
      T.gstart[m][c] = cnt;
      uint32_t in_class = 0;
      for (uint32_t s = 0; s <= 31u; s++) {
        for (uint32_t a = 0; a <= s && s <= 30u; a++) {
          const uint32_t b = s - a;
          const bool special = m == 0 ? ((a + b <= 1u) || (a >= 1u && a <= 2u && b >= 1u && b <= 2u)) : (a <= 1u && b <= 1u);
          if (special) continue;
          const uint32_t cls = ((a == 0u) != (b == 0u)) ? 0u
                               : (a == 1u || b == 1u) ? 1u
                               : ((a == 2u && b == 3u) || (a == 3u && b == 2u)) ? 2u : 3u;
          if (cls != c) continue;
          const uint32_t p = a <= 15u ? a * 32u + b : (30u - a) * 32u + b + a + 1u;  // (probe_slot's inverse)
          T.gslot[m][cnt] = a | (s << 8);
          T.glen[m][cnt] = T.len[m][p];
          cnt++;
          in_class++;
        }
        T.gcount[m][c][s] = in_class;
      }
      while (cnt % 8u != 0u) {  // (never counted: keeps an eight-wide read inside the list)
        T.gslot[m][cnt] = in_class ? T.gslot[m][cnt - 1] : 0u;
        T.glen[m][cnt] = 0.f;
        cnt++;
      }
    }
    // class 3 in runs of four consecutive a per level
    uint32_t ng = 0;
    for (uint32_t s = 0; s <= 31u; s++) {
      if (s <= 30u) {
        uint32_t amin = ~0u, amax = 0u;
        for (uint32_t a = 0; a <= s; a++) {
          const uint32_t b = s - a;
          const bool special = m == 0 ? ((a + b <= 1u) || (a >= 1u && a <= 2u && b >= 1u && b <= 2u)) : (a <= 1u && b <= 1u);
          const bool c3 = !special && a >= 2u && b >= 2u && !((a == 2u && b == 3u) || (a == 3u && b == 2u));
          if (!c3) continue;
          amin = std::min(amin, a);
          amax = std::max(amax, a);
        }
        for (uint32_t a0 = amin; amin != ~0u && a0 <= amax; a0 += 4u) {
          T.g4slot[m][ng] = a0 | (s << 8);
          for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t a = a0 + u, b = s - a;  // (a <= amax <= s - 2: b >= 2)
            const uint32_t p = a <= 15u ? a * 32u + b : (30u - a) * 32u + b + a + 1u;
            T.g4len[m][ng][u] = a <= amax ? T.len[m][p] : -INFINITY;
          }
          ng++;
        }
      }
      T.g4count[m][s] = ng;
    }
    // classes 0 and 1 by level: (0, s), (s, 0), (1, s - 1), (s - 1, 1)
    for (uint32_t s = 0; s < 32u; s++)
      for (uint32_t u = 0; u < 4u; u++) {
        T.elen[m][s][u] = -INFINITY;
        if (s < 2u || s > 30u) continue;
        const uint32_t a = u == 0u ? 0u : (u == 1u ? s : (u == 2u ? 1u : s - 1u)), b = s - a;
        const bool special = m == 0 ? ((a + b <= 1u) || (a >= 1u && a <= 2u && b >= 1u && b <= 2u)) : (a <= 1u && b <= 1u);
        const uint32_t cls = ((a == 0u) != (b == 0u)) ? 0u : (a == 1u || b == 1u) ? 1u : 3u;
        if (special || cls != (u < 2u ? 0u : 1u)) continue;
        if (u == 3u && s - 1u == 1u) continue;  // ((1, 1) once)
        const uint32_t p = a <= 15u ? a * 32u + b : (30u - a) * 32u + b + a + 1u;
        T.elen[m][s][u] = T.len[m][p];
      }
    for (uint32_t x = 0; x < 8u && ng < 128u; x++, ng++) {  // (never counted: a step's reads stay inside the list)
      T.g4slot[m][ng] = T.g4slot[m][ng - 1];
      for (uint32_t u = 0; u < 4u; u++) T.g4len[m][ng][u] = -INFINITY;
    }
  }
}

int ensure_tree_tabs(rnamc_ctx* c, hipStream_t st) {
  if (c->tree_tabs_valid && c->d_tree_tabs) return RNAMC_OK;
  if (!c->d_tree_tabs) HIPCHK(hipMalloc(&c->d_tree_tabs, sizeof(TreeTabs)));
  static thread_local TreeTabs tabs;  // (16 KB: not on the stack; the copy below is synchronous)
  build_tree_tabs(c->host_params, tabs);
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpy(c->d_tree_tabs, &tabs, sizeof(TreeTabs), hipMemcpyHostToDevice));
  c->tree_tabs_valid = true;
  return RNAMC_OK;
}

// Tree-order summation mode (rnamc_tree.hip): same grouping and per-diagonal sweep, dense
// n x n matrices, one workgroup per cell.  Fills c->descs / group_* like run_batch so that the
// host-buffer entry's drain thread works unchanged.
int run_batch_tree(rnamc_ctx* c, uint32_t n_seqs, const uint8_t* d_bases, const uint64_t* offsets,
                   bool contra, bool allows_short, float* d_out, const uint64_t* out_offsets,
                   float* d_logz, hipStream_t st, const GroupHooks* hooks = nullptr) {
  c->stats = rnamc_batch_stats{};
  c->kev_class.clear();
  c->descs.clear();
  c->group_begin.clear();
  c->group_out_floats.clear();
  if (n_seqs == 0) return RNAMC_OK;
  uint32_t max_n = 0;
  for (uint32_t s = 0; s < n_seqs; s++) {
    if (offsets[s + 1] < offsets[s]) return RNAMC_ERR_INVALID_ARG;
    const uint64_t n = offsets[s + 1] - offsets[s];
    if (n == 0) return RNAMC_ERR_EMPTY_SEQ;
    if (n > RNAMC_MAX_SEQ_LEN) return RNAMC_ERR_SEQ_TOO_LONG;
    max_n = std::max<uint32_t>(max_n, static_cast<uint32_t>(n));
  }
  int rc = ensure_hp_init(c, max_n);
  if (rc) return rc;
  rc = ensure_tree_tabs(c, st);
  if (rc) return rc;
  std::vector<uint32_t> order(n_seqs);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    return (offsets[a + 1] - offsets[a]) > (offsets[b + 1] - offsets[b]);
  });
  uint64_t ws_cap_floats = static_cast<uint64_t>(std::max<int64_t>(c->group_ws_bytes, 1)) / 4;
  // banding needs two diagonals per launch aligned to even diagonals, 32-bit float offsets INSIDE
  // one matrix (true for every n <= RNAMC_MAX_SEQ_LEN: ld * n < 2^32), and sequences long enough
  // to have a banded diagonal at all
  uint32_t band = (c->tree_two != 0 && c->tree_band >= 32) ? static_cast<uint32_t>(c->tree_band) & ~31u : 0u;
  if (band > 128u) band = 128u;
  {
    const uint64_t ld = ((static_cast<uint64_t>(max_n) + 31u) & ~31ull) + 32u;
    if (ld * max_n + 128ull >= (1ull << 32) || max_n < 3u * band + 2u) band = 0u;
  }
  if (band && !c->bulk_stream) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HIPCHK(hipStreamCreateWithPriority(&c->bulk_stream, hipStreamNonBlocking, lo));
  }
  if (band) {
    // The banded sweep lives on the mid-field kernels running BESIDE it.  Whether the side stream
    // owns a hardware queue depends on what else the process created (round 3: created late it
    // shared the sweep's queue, 157 ms instead of 49.5): detected, not assumed — once per caller
    // stream; a serialised side stream means the unbanded sweep (slower, never wrong).
    if (c->tree_side_force == 1 || c->tree_side_force == 2) {  // (knob "tree_side_stream": the verdict is given)
      c->side_verdict = static_cast<int>(c->tree_side_force);
      c->side_probed = true;
      c->side_probed_for = nullptr;
    } else if (!c->side_probed || c->side_probed_for != st) {
      c->side_verdict = tree_side_stream_probe(st, c->bulk_stream);
      c->side_probed = true;
      c->side_probed_for = st;
    }
  }
  // lane-per-cell sweeps: what a batch's fat launches want (a lone sequence keeps the wave-per-cell chain)
  uint32_t lane_mode = 0u;
  {
    const int64_t mode = c->tree_lane & 3;
    // (a plane of the sweep is one raw buffer there: msz * 4 bytes in a 32-bit record count)
    const uint64_t ld_max = ((static_cast<uint64_t>(max_n) + 31u) & ~31ull) + 32u;
    if (band && ld_max * max_n * 4ull + 1024ull < (1ull << 31) && (mode == 2 || (mode == 1 && offsets[n_seqs] - offsets[0] >= static_cast<uint64_t>(c->tree_lane_min_nt) && n_seqs > 1)))
      lane_mode = 3u;
  }
  // (a serialised side stream: the unbanded sweep — unless the mid-field kernels run in front of their
  // band on the sweep's own stream anyway; sums_external's walks then simply queue behind)
  if (band && c->side_verdict == 2 && !(lane_mode && c->tree_mid_sync != 0)) {
    band = 0u;
    lane_mode = 0u;
  }
  if (lane_mode && c->tree_mid_sync != 0 && static_cast<uint32_t>(c->tree_lane_band) < band)
    band = static_cast<uint32_t>(c->tree_lane_band);
  if (lane_mode && !c->group_ws_user) {
    // The batch form's launches cost ~8 us each whatever they hold, and a group's 36 n^2 floats per
    // sequence are what limits its size: twice the default workspace where the device has the room
    // (measured on a 1 000-sequence slice of the bench batch: 16 / 32 / 64 / 128 GB -> 1167 / 909 / 787 /
    // 753 ms per pass).
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const uint64_t have = static_cast<uint64_t>(free_b) + c->ws_floats * 4ull;
      const uint64_t margin = 24ull << 30;
      if (have > (128ull << 30) + margin) ws_cap_floats = (128ull << 30) / 4;
    } else {
      (void)hipGetLastError();
    }
  }
  // two groups side by side (see tree_dual): each gets half of the cap
  bool dual = lane_mode != 0u && c->tree_mid_sync != 0 && c->tree_dual != 0 && hooks == nullptr &&
              st != c->aux_stream && st != c->own_stream;
  if (dual) ws_cap_floats /= 2;
  std::vector<TreeSeq>& tseqs = c->h_tseqs;
  tseqs.clear();
  tseqs.reserve(n_seqs);
  uint64_t max_group_floats = 0;
  {
    uint64_t cur = 0, cur_nt = 0, cur_out = 0;
    uint32_t cnt = 0;
    for (uint32_t x = 0; x < n_seqs; x++) {
      const uint32_t s = order[x];
      const uint32_t n = static_cast<uint32_t>(offsets[s + 1] - offsets[s]);
      TreeSeq ts{};
      ts.n = n;
      ts.ld = ((n + 31u) & ~31u) + 32u;
      ts.msz = ((static_cast<uint64_t>(ts.ld) * n + 63ull) & ~63ull) + 64ull;
      const uint64_t vec = (static_cast<uint64_t>(n) + 64ull + 63ull) & ~63ull;
      ts.pk_words = static_cast<uint32_t>(((static_cast<uint64_t>(n) + 160) / 16 + 4 + 63) & ~63ull);
      // (mid-field ring: three products x 2 bands of diagonals x vec cells x {max, sum})
      // + the far ring (four diagonals x vec cells x {max, sum})
      const uint64_t mid_floats = band ? (3ull * (2ull * band) + 4ull) * vec * 2ull : 0ull;
      const uint64_t need = ts.msz * T_COUNT + 2ull * vec + ts.pk_words + mid_floats;
      if (cnt > 0 && (cnt >= static_cast<uint32_t>(c->group_max_seqs) ||
                      cur_nt + n > static_cast<uint64_t>(c->group_max_nt) ||
                      cur + need > ws_cap_floats)) {
        max_group_floats = std::max(max_group_floats, cur);
        c->group_out_floats.push_back(cur_out);
        cur = cur_nt = cur_out = 0;
        cnt = 0;
      }
      if (cnt == 0) c->group_begin.push_back(static_cast<uint32_t>(c->descs.size()));
      ts.seq_off = offsets[s];
      ts.ws_off = cur;
      ts.pk_off = cur + ts.msz * T_COUNT + 2ull * vec;
      ts.mid_off = ts.pk_off + ts.pk_words;
      ts.out_off = hooks ? cur_out : out_offsets[s];
      ts.batch_idx = s;
      tseqs.push_back(ts);
      SeqDesc sd{};  // host bookkeeping shared with the reference-order path
      sd.n = n;
      sd.seq_off = ts.seq_off;
      sd.ws_off = ts.ws_off;
      sd.out_off = ts.out_off;
      sd.batch_idx = s;
      c->descs.push_back(sd);
      cur += need;
      cur_nt += n;
      cur_out += rnamc_bpp_len(n);
      cnt++;
    }
    max_group_floats = std::max(max_group_floats, cur);
    c->group_out_floats.push_back(cur_out);
    c->group_begin.push_back(static_cast<uint32_t>(c->descs.size()));
  }
  if (c->group_begin.size() < 3) dual = false;  // (a single group)
  rc = ensure_ws(c, dual ? 2 * max_group_floats : max_group_floats);
  if (rc) return rc;
  if (dual && !c->ev_dual) {
    // (no new streams: a process's fifth and later streams share hardware queues on this runtime — the second
    // group's sweep would sit in the first one's queue, measured: 698 instead of 556 ms — so the second group
    // takes the two streams the context created first for the reference-order path, idle in this mode)
    c->dual_stream = c->aux_stream;
    c->bulk_stream2 = c->own_stream;
    HIPCHK(hipEventCreateWithFlags(&c->ev_dual, hipEventDisableTiming));
    for (size_t x = 0; x < c->ev_a.size(); x++) {
      hipEvent_t ea, eb;
      HIPCHK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
      c->ev_a2.push_back(ea);
      c->ev_b2.push_back(eb);
    }
  }
  if (c->tseqs_cap < tseqs.size()) {
    if (c->d_tseqs) {
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipFree(c->d_tseqs));
      c->d_tseqs = nullptr;
      c->tseqs_cap = 0;
    }
    const uint64_t cap = std::max<uint64_t>(tseqs.size(), 1024);
    HIPCHK(hipMalloc(&c->d_tseqs, cap * sizeof(TreeSeq)));
    c->tseqs_cap = cap;
  }
  HIPCHK(hipMemcpyAsync(c->d_tseqs, tseqs.data(), tseqs.size() * sizeof(TreeSeq),
                        hipMemcpyHostToDevice, st));
  const size_t n_groups = c->group_begin.size() - 1;
  const bool prof = c->profile != 0;
  if (prof) {
    while (c->events.size() < n_groups * 4) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      c->events.push_back(e);
    }
  }
  const uint32_t dmin_in = contra ? 0u : (RNAMC_MIN_SPAN_HAIRPIN_CLOSE - 1);
  const uint32_t dmin_out = (contra && allows_short) ? 1u : (RNAMC_MIN_SPAN_HAIRPIN_CLOSE - 1);
  if (dual) {  // (the second stream starts behind whatever the caller's stream holds: the descriptors' copy)
    HIPCHK(hipEventRecord(c->ev_dual, st));
    HIPCHK(hipStreamWaitEvent(c->dual_stream, c->ev_dual, 0));
  }
  for (size_t g = 0; g < n_groups; g++) {
    const bool odd = dual && (g & 1u) != 0u;
    hipStream_t gst = odd ? c->dual_stream : st;
    hipStream_t gbulk = odd ? c->bulk_stream2 : c->bulk_stream;
    std::vector<hipEvent_t>& gev_a = odd ? c->ev_a2 : c->ev_a;
    std::vector<hipEvent_t>& gev_b = odd ? c->ev_b2 : c->ev_b;
    const uint32_t gb = c->group_begin[g], ge = c->group_begin[g + 1];
    const uint32_t nseq = ge - gb;
    const uint32_t gmax = c->descs[gb].n;
    TreeBatch b{};
    b.seqs = c->d_tseqs + gb;
    b.one = tseqs[gb];
    b.use_one = nseq == 1 ? 1u : 0u;
    b.bases = d_bases;
    b.workspace = c->d_ws + (odd ? max_group_floats : 0);
    b.out = d_out;
    if (hooks) {
      rc = hooks->before(g, &b.out);
      if (rc) return rc;
    }
    b.log_partition = d_logz;
    b.params = c->d_params;
    b.tabs = c->d_tree_tabs;
    b.hp_init = c->d_hp_init;
    b.allows_short_hairpins = allows_short ? 1 : 0;
#ifdef RNAMC_DEBUG_KNOBS
    b.debug = static_cast<int>(c->tree_debug);
#endif
    b.ring = 2u * band;
    b.lane = lane_mode;
    auto active = [&](uint32_t d) {  // sequences with n > d form a prefix of the group
      uint32_t lo = 0, hi = nseq;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (c->descs[gb + mid].n > d) lo = mid + 1; else hi = mid;
      }
      return lo;
    };
    if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 0], gst));
    launch_tree_init(b, nseq, gmax, contra, 0, gst);
    launch_tree_static(b, contra, nseq, gmax, gst);
    c->stats.launches_other += 2;
    if (lane_mode) {
      launch_tlane_list(b, gmax, nseq, gst);
      c->stats.launches_other++;
    }
    const bool two = c->tree_two != 0;
    const uint32_t ering = static_cast<uint32_t>(c->ev_a.size());
    if (band) {
      // Banded sweep: launches are pairs (2m, 2m+1) (a lone first / last diagonal where the
      // range starts odd / ends even), so no launch straddles a band [x*band, (x+1)*band).
      // Inside: band x >= 3 takes the terms with both operand spans below thr = (x-1)*band from
      // k_tree_mid, which needs the diagonals below thr: enqueued on bulk_stream when band
      // x-1 starts, awaited when band x starts.  Rings: ev_a "the sweep reached a band boundary",
      // ev_b "mid-field of the band written".
      const uint32_t nb = (gmax + band - 1) / band;  // bands 0 .. nb-1
      // sums_external's first row and last column (k_tree_ext) trail the sweep by one band on
      // bulk_stream as well (the outside sweep is their only reader).
      auto boundary = [&](uint32_t x) -> int {  // "the sweep reached band x": bulk_stream may pass
        HIPCHK(hipEventRecord(gev_a[x % ering], gst));
        HIPCHK(hipStreamWaitEvent(gbulk, gev_a[x % ering], 0));
        return RNAMC_OK;
      };
      auto enqueue_mid = [&](bool outside, uint32_t x, uint32_t thr) -> int {
        const uint32_t dlo = x * band, dhi = std::min(gmax - 1, dlo + band - 1);
#ifdef RNAMC_DEBUG_KNOBS
        if (!(c->tree_debug & 32))  // (timing: the sweep without its mid-field kernels; results wrong)
#endif
        launch_tree_mid(b, outside, dlo, dhi, thr, gmax, active(dlo), c->tree_pol, gbulk);
        HIPCHK(hipEventRecord(gev_b[x % ering], gbulk));
        c->stats.launches_other++;
        return RNAMC_OK;
      };
      auto enqueue_ext = [&](uint32_t x) {  // band x of the inside sweep is enqueued whole
        const uint32_t dlo = std::max(dmin_in, x * band), dhi = std::min(gmax - 1, x * band + band - 1);
        if (dlo > dhi) return;
#ifdef RNAMC_DEBUG_KNOBS
        if (!(c->tree_debug & 64))
#endif
        launch_tree_ext(b, contra, dlo, dhi, gmax, active(dlo), gbulk);
        c->stats.launches_other++;
      };
      const bool sync_in = (lane_mode & 1u) != 0u && c->tree_mid_sync != 0;
      const bool sync_out = (lane_mode & 2u) != 0u && c->tree_mid_sync != 0;
      const bool ahead = c->tree_ahead != 0 && (c->tree_tpc == 0 || c->tree_tpc == 64);
      bool use_far = false;  // (the first launch of a sweep forms its blocks whole)
      uint32_t g_next = std::max(5u, dmin_in);  // lane-per-cell sweeps: the next diagonal without its generic 2-loop sums
      uint32_t d = dmin_in;
      uint32_t cur_band = ~0u;
      while (d < gmax) {
        const uint32_t x = d / band;
        if (x != cur_band) {
          // band x starts: everything below x*band is enqueued; sums_external of band x-1 and
          // the mid-field of band x+1 can go
          if (lane_mode && cur_band != ~0u) {  // (their readers want the band row- / column-major)
            launch_tlane_spread(b, false, cur_band * band, std::min(gmax - 1, cur_band * band + band - 1), gmax,
                                active(cur_band * band), gst);
            c->stats.launches_other++;
          }
          rc = boundary(x);
          if (rc) return rc;
          if (cur_band != ~0u) enqueue_ext(cur_band);
          cur_band = x;
          if (sync_in) {
            // (a batch: the band's mid-field in front of the band, on the sweep's own stream — every
            // diagonal below x * band is final, so the launches keep the terms of the band alone)
            if (x >= 1) {
              launch_tree_mid(b, false, x * band, std::min(gmax - 1, x * band + band - 1), x * band, gmax,
                              active(x * band), c->tree_pol, gst);
              c->stats.launches_other++;
            }
          } else {
            if (x + 1 >= 3 && x + 1 < nb) {
              rc = enqueue_mid(false, x + 1, x * band);
              if (rc) return rc;
            }
            if (x >= 3) HIPCHK(hipStreamWaitEvent(gst, gev_b[x % ering], 0));
          }
        }
        const uint32_t thr = sync_in ? x * band : (x >= 3 ? (x - 1) * band : 0u);
        if (lane_mode & 1u) {
          if (d == dmin_in) {  // (the first diagonal's closing-pair blocks: in FRONT of the generic sums below —
            // a batch of four enqueued ahead of it read X4 of this diagonal before it was written, which is
            // what made four diagonals a launch differ under Turner tables, whose first diagonal is 4)
            launch_tlane_inside(b, contra, ~0u, d, gmax, active(d), 0u, gst);
            c->stats.launches_inside++;
          }
          // (the generic 2-loop sums of diagonal d + 1, wanted by this launch's second role: up to three
          // diagonals at once — their slots read X4 up to their own diagonal minus four, i.e. up to d - 1;
          // X4 is complete up to d)
          while (g_next <= d + 1 && g_next < gmax) {
            const uint32_t gc = std::min<uint32_t>(static_cast<uint32_t>(c->tree_gen_batch), gmax - g_next);
            launch_tlane_gen(b, contra, false, g_next, gc, gmax, active(g_next), gst);
            c->stats.launches_inside++;
            g_next += gc;
          }
          launch_tlane_inside(b, contra, d, d + 1 < gmax ? d + 1 : ~0u, gmax, active(d), thr, gst);
          c->stats.launches_inside++;
          d++;
          continue;
        }
        const bool pair = (d % 2u == 0u) && d + 1 < gmax;
        // the next launch's diagonals: their 2-loop blocks' far parts ride in this launch
        const uint32_t nd0 = d + (pair ? 2u : 1u);
        const uint32_t ndc = (!ahead || nd0 >= gmax) ? 0u : ((nd0 % 2u == 0u && nd0 + 1 < gmax) ? 2u : 1u);
        launch_tree_inside(b, contra, d, gmax, active(d), c->tree_tpc, pair, thr, use_far, nd0, ndc, c->tree_pol, gst);
        use_far = ndc != 0u;
        c->stats.launches_inside++;
        d += pair ? 2 : 1;
      }
      if (cur_band != ~0u) {  // the last band's sums_external; the outside sweep reads them
        if (lane_mode) {
          launch_tlane_spread(b, false, cur_band * band, std::min(gmax - 1, cur_band * band + band - 1), gmax,
                              active(cur_band * band), gst);
          c->stats.launches_other++;
        }
        rc = boundary(cur_band + 1);
        if (rc) return rc;
        enqueue_ext(cur_band);
        HIPCHK(hipEventRecord(gev_b[(cur_band + 1) % ering], gbulk));
        HIPCHK(hipStreamWaitEvent(gst, gev_b[(cur_band + 1) % ering], 0));
      }
      if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 1], gst));
#ifdef RNAMC_DEBUG_KNOBS
      if (const char* dump = getenv("RNAMC_DUMP_MID")) {  // "<first slot>,<slots>,<path>": after the inside sweep
        int s0 = 0, ns = 1;
        char path[512] = {0};
        if (sscanf(dump, "%d,%d,%500s", &s0, &ns, path) == 3) {
          HIPCHK(hipDeviceSynchronize());
          const TreeSeq& t0 = tseqs[gb];
          std::vector<float> hbuf(static_cast<size_t>(ns) * t0.msz);
          HIPCHK(hipMemcpy(hbuf.data(), b.workspace + t0.ws_off + static_cast<uint64_t>(s0) * t0.msz,
                           hbuf.size() * sizeof(float), hipMemcpyDeviceToHost));
          if (FILE* fh = fopen(path, "wb")) {
            const uint64_t hdr[3] = {t0.n, t0.ld, t0.msz};
            fwrite(hdr, sizeof(hdr), 1, fh);
            fwrite(hbuf.data(), sizeof(float), hbuf.size(), fh);
            fclose(fh);
          }
        }
      }
#endif
      launch_tree_init(b, nseq, gmax, contra, 1, gst);
      c->stats.launches_other++;
      // Outside, from the top: band x takes the terms whose outside operand spans at least
      // thr = (x+2)*band (final once band x+2 is through) from k_tree_mid, enqueued when band x+1
      // starts.  (The first enqueue also orders bulk_stream after the inside sweep's last reads of
      // the ring and after launch_tree_init.)
      int64_t dd = static_cast<int64_t>(gmax) - 1;
      int64_t go_next = static_cast<int64_t>(gmax) - 5;  // (a generic enclosing 2-loop needs n >= d + 5)
      cur_band = ~0u;
      use_far = false;
      while (dd >= static_cast<int64_t>(dmin_out)) {
        const uint32_t du = static_cast<uint32_t>(dd);
        const uint32_t x = du / band;
        if (x != cur_band) {
          if (lane_mode && cur_band != ~0u) {  // (the band above is through: W and R for the mid-field kernels)
            launch_tlane_spread(b, true, cur_band * band, std::min(gmax - 1, cur_band * band + band - 1), gmax,
                                active(cur_band * band), gst);
            c->stats.launches_other++;
          }
          cur_band = x;
          if (sync_out) {
            if ((x + 1) * band < gmax) {  // (operands of span >= (x+1)*band: everything above this band)
              launch_tree_mid(b, true, x * band, std::min(gmax - 1, x * band + band - 1), (x + 1) * band, gmax,
                              active(x * band), c->tree_pol, gst);
              c->stats.launches_other++;
            }
          } else {
            if (x >= 1 && (x + 1) * band < gmax) {  // band x-1 has a mid-field: thr = (x+1)*band <= gmax-1
              rc = boundary(x);
              if (rc) return rc;
              rc = enqueue_mid(true, x - 1, (x + 1) * band);
              if (rc) return rc;
            }
            if ((x + 2) * band < gmax) HIPCHK(hipStreamWaitEvent(gst, gev_b[x % ering], 0));
          }
        }
        const uint32_t thr = sync_out ? ((x + 1) * band < gmax ? (x + 1) * band : 0u)
                                      : ((x + 2) * band < gmax ? (x + 2) * band : 0u);
        if (lane_mode & 2u) {
          // (the generic enclosing 2-loops of diagonal du, wanted by this launch: three diagonals downwards at
          // once — their slots read PX4 from their own diagonal plus four on, i.e. from du + 2)
          while (go_next >= static_cast<int64_t>(du) && go_next >= static_cast<int64_t>(dmin_out)) {
            const uint32_t gc = static_cast<uint32_t>(std::min<int64_t>(c->tree_gen_batch, go_next - static_cast<int64_t>(dmin_out) + 1));
            launch_tlane_gen(b, contra, true, static_cast<uint32_t>(go_next), gc, gmax,
                             active(static_cast<uint32_t>(go_next) - (gc - 1u)), gst);
            c->stats.launches_outside++;
            go_next -= gc;
          }
          if (du == gmax - 1) {  // (the top diagonal's enclosing 2-loops: none exist, the slots are written)
            launch_tlane_outside(b, contra, ~0u, du, gmax, active(du), 0u, gst);
            c->stats.launches_outside++;
          }
          // (sequences that enter the sweep with the next launch need their 2-loop sums too)
          const uint32_t dn = du > dmin_out ? du - 1 : ~0u;
          launch_tlane_outside(b, contra, du, dn, gmax, active(dn != ~0u ? dn : du), thr, gst);
          c->stats.launches_outside++;
          dd--;
          continue;
        }
        const bool pair = du % 2u == 1u && du - 1 >= dmin_out;
        const uint32_t lower = pair ? du - 1 : du;
        // the next launch (below): a pair when its top is odd and both diagonals are swept
        uint32_t nd0 = 0, ndc = 0;
        if (ahead && lower >= dmin_out + 1) {
          const uint32_t top = lower - 1;
          const bool npair = top % 2u == 1u && top - 1 >= dmin_out;
          nd0 = npair ? top - 1 : top;
          ndc = npair ? 2u : 1u;
        }
        // (sequences that enter the sweep with the next launch need their far parts too)
        launch_tree_outside(b, contra, lower, gmax, active(ndc ? nd0 : lower), c->tree_tpc, pair, thr, use_far,
                            nd0, ndc, c->tree_pol, gst);
        use_far = ndc != 0u;
        dd -= pair ? 2 : 1;
        c->stats.launches_outside++;
      }
    } else {
    for (uint32_t d = dmin_in; d < gmax; d += two ? 2 : 1) {
      launch_tree_inside(b, contra, d, gmax, active(d), c->tree_tpc, two, 0u, false, 0u, 0u, c->tree_pol, gst);
      c->stats.launches_inside++;
    }
    if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 1], gst));
    launch_tree_init(b, nseq, gmax, contra, 1, gst);
    c->stats.launches_other++;
    if (two) {
      // pairs (d+1, d) from the top; the lowest diagonal alone when their number is odd
      int64_t d = static_cast<int64_t>(gmax) - 1;
      for (; d - 1 >= static_cast<int64_t>(dmin_out); d -= 2) {
        launch_tree_outside(b, contra, static_cast<uint32_t>(d - 1), gmax, active(static_cast<uint32_t>(d - 1)),
                            c->tree_tpc, true, 0u, false, 0u, 0u, c->tree_pol, gst);
        c->stats.launches_outside++;
      }
      if (d >= static_cast<int64_t>(dmin_out)) {
        launch_tree_outside(b, contra, static_cast<uint32_t>(d), gmax, active(static_cast<uint32_t>(d)),
                            c->tree_tpc, false, 0u, false, 0u, 0u, c->tree_pol, gst);
        c->stats.launches_outside++;
      }
    } else
    for (uint32_t d = gmax; d-- > dmin_out;) {
      launch_tree_outside(b, contra, d, gmax, active(d), c->tree_tpc, false, 0u, false, 0u, 0u, c->tree_pol, gst);
      c->stats.launches_outside++;
    }
    }
    if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 2], gst));
    launch_tree_finalize(b, nseq, gmax, gst);
    c->stats.launches_other++;
    if (prof) HIPCHK(hipEventRecord(c->events[4 * g + 3], gst));
    HIPCHK(hipGetLastError());
    if (hooks) {
      rc = hooks->after(g, gb, nseq);
      if (rc) return rc;
    }
  }
  if (dual) {  // (the caller's stream ends behind the second one)
    HIPCHK(hipEventRecord(c->ev_dual, c->dual_stream));
    HIPCHK(hipStreamWaitEvent(st, c->ev_dual, 0));
  }
#ifdef RNAMC_DEBUG_KNOBS
  if (const char* dump = getenv("RNAMC_DUMP_SLOT")) {  // "<first slot>,<slots>,<path>": the first sequence's matrices, raw
    int s0 = 0, ns = 1;
    char path[512] = {0};
    if (sscanf(dump, "%d,%d,%500s", &s0, &ns, path) == 3 && !tseqs.empty()) {
      HIPCHK(hipDeviceSynchronize());
      const TreeSeq& t0 = tseqs[0];
      std::vector<float> hbuf(static_cast<size_t>(ns) * t0.msz);
      HIPCHK(hipMemcpy(hbuf.data(), c->d_ws + t0.ws_off + static_cast<uint64_t>(s0) * t0.msz, hbuf.size() * sizeof(float),
                       hipMemcpyDeviceToHost));
      if (FILE* fh = fopen(path, "wb")) {
        const uint64_t hdr[3] = {t0.n, t0.ld, t0.msz};
        fwrite(hdr, sizeof(hdr), 1, fh);
        fwrite(hbuf.data(), sizeof(float), hbuf.size(), fh);
        fclose(fh);
      }
    }
  }
#endif
  c->stats.tree_side_stream = static_cast<uint64_t>(c->side_probed ? c->side_verdict : 0);
  c->stats.n_groups = n_groups;
  c->stats.workspace_bytes = c->ws_floats * sizeof(float);
  if (prof) {
    HIPCHK(hipStreamSynchronize(st));
    for (size_t g = 0; g < n_groups; g++) {
      float a = 0, bms = 0, cc = 0;
      HIPCHK(hipEventElapsedTime(&a, c->events[4 * g + 0], c->events[4 * g + 1]));
      HIPCHK(hipEventElapsedTime(&bms, c->events[4 * g + 1], c->events[4 * g + 2]));
      HIPCHK(hipEventElapsedTime(&cc, c->events[4 * g + 2], c->events[4 * g + 3]));
      c->stats.ms_inside += a;
      c->stats.ms_outside += bms;
      c->stats.ms_other += cc;
    }
  }
  return RNAMC_OK;
}

// the sweep of the context's summation mode (rnamc_fold_scores needs the reference-order
// workspace layout and always takes that path)
int run_batch_mode(rnamc_ctx* c, uint32_t n_seqs, const uint8_t* d_bases, const uint64_t* offsets,
                   bool contra, bool allows_short, float* d_out, const uint64_t* out_offsets,
                   float* d_logz, hipStream_t st, const GroupHooks* hooks = nullptr) {
  if (c->summation_mode == 1 && !c->inside_only)
    return run_batch_tree(c, n_seqs, d_bases, offsets, contra, allows_short, d_out, out_offsets,
                          d_logz, st, hooks);
  return run_batch(c, n_seqs, d_bases, offsets, contra, allows_short, d_out, out_offsets, d_logz, st,
                   hooks);
}

// FoldScores of one sequence on the host, given the sums_close key set (packed
// diagonal-major, finite = key present).  Work is split by closing diagonal.
template <class Model>
void fold_scores_host(const Model& M, bool contra, bool allows_short, const uint8_t* s, uint32_t n,
                      const float* qb, float* hp, float* mb, float* ac, rnamc_twoloop_score* tl,
                      const std::vector<uint64_t>* tl_begin, std::vector<uint64_t>* tl_count) {
  const float nan = std::numeric_limits<float>::quiet_NaN();
  const float ninf = -std::numeric_limits<float>::infinity();
  auto tri = [n](uint32_t i, uint32_t j) {
    const uint64_t d = j - i;
    return d * n - d * (d - 1ull) / 2ull + i;
  };
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> pool;
  std::atomic<uint32_t> next{0};
  auto work = [&]() {
    for (;;) {
      const uint32_t d = next.fetch_add(1);
      if (d >= n) return;
      uint64_t cnt = 0;
      uint64_t w = tl_begin ? (*tl_begin)[d] : 0;
      for (uint32_t i = 0; i + d < n; i++) {
        const uint32_t j = i + d;
        const uint64_t x = tri(i, j);
        const bool act = canonical(s[i], s[j]) &&
                         ((contra && allows_short) || d + 1 >= RNAMC_MIN_SPAN_HAIRPIN_CLOSE);
        if (!tl_begin) {
          if (hp) hp[x] = (act && (!contra || d - 1 <= RNAMC_MAX_LOOP_LEN)) ? M.hairpin(s, n, i, j) : nan;
          const bool member = act && qb[x] > ninf;
          if (mb) mb[x] = member ? M.mbclose(s, n, i, j) : nan;
          if (ac) ac[x] = member ? M.accessible(s, n, i, j) : nan;
        }
        if (!act || d < 3) continue;
        // k in i+1 .. j-2, a = k-i-1 <= 30; l from j-1 down to k+1, a + b <= 30
        for (uint32_t k = i + 1; k + 1 < j && k - i - 1 <= RNAMC_MAX_2LOOP_LEN; k++) {
          for (uint32_t l = j - 1; l > k && (j - l - 1) + (k - i - 1) <= RNAMC_MAX_2LOOP_LEN; l--) {
            if (!(qb[tri(k, l)] > ninf)) continue;
            if (tl_begin) tl[w++] = rnamc_twoloop_score{i, j, k, l, M.twoloop(s, i, j, k, l)};
            cnt++;
          }
        }
      }
      if (tl_count) (*tl_count)[d] = cnt;
    }
  };
  for (unsigned t = 1; t < hw; t++) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
}

}  // namespace

namespace {
int validate_params(const rnamc_params* params) {
  if (params->abi_version != RNAMC_ABI_VERSION || params->struct_bytes != sizeof(rnamc_params)) {
    set_last_error("rnamc_params header does not match this library's ABI");
    return RNAMC_ERR_INVALID_ARG;
  }
  const rnamc_turner_scores& t = params->turner;
  if (t.num_special_hairpins > RNAMC_MAX_SPECIAL_HAIRPINS ||
      t.max_hairpin_len_extrapolation > RNAMC_MAX_LOOP_LEN || t.min_hairpin_len_extrapolation < 2 ||
      t.min_hairpin_len_extrapolation - 1 > RNAMC_MAX_LOOP_LEN ||
      t.min_hairpin_len > t.max_hairpin_len_extrapolation) {
    set_last_error("Turner hairpin limits out of range");
    return RNAMC_ERR_INVALID_ARG;
  }
  return RNAMC_OK;
}
}  // namespace

extern "C" {

int rnamc_ctx_create(const rnamc_params* params, int device, uint64_t workspace_bytes,
                     rnamc_ctx** out) {
  if (!params || !out) return RNAMC_ERR_INVALID_ARG;
  *out = nullptr;
  if (int rc = validate_params(params)) return rc;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_last_error("no HIP device visible: librnamc has no CPU fallback");
    return RNAMC_ERR_NO_DEVICE;
  }
  if (device < 0) {
    if (hipGetDevice(&device) != hipSuccess) return RNAMC_ERR_NO_DEVICE;
  }
  if (device >= count) return RNAMC_ERR_INVALID_ARG;
  DeviceGuard guard(device);
  if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
  rnamc_ctx* c = new (std::nothrow) rnamc_ctx();
  if (!c) return RNAMC_ERR_OOM;
  c->device = device;
  c->host_params = *params;
  auto fail = [&](int rc) {
    rnamc_ctx_destroy(c);
    return rc;
  };
  if (hipMalloc(&c->d_params, sizeof(rnamc_params)) != hipSuccess) return fail(RNAMC_ERR_OOM);
  if (hipMemcpy(c->d_params, params, sizeof(rnamc_params), hipMemcpyHostToDevice) != hipSuccess)
    return fail(RNAMC_ERR_HIP);
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess)
    return fail(RNAMC_ERR_HIP);
  {
    // the pair-tail kernel carries the longest dependent chains of a diagonal: its few
    // workgroups should be placed first, the other kernel fills the rest of the chip
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, hi) != hipSuccess)
      return fail(RNAMC_ERR_HIP);
#ifndef RNAMC_DBG_LAZY_BULK
    // the tree-order mode's side stream (mid-field products, lowest priority) is created HERE,
    // with the context, not at the first tree-order call: created as the process's fifth or later
    // stream (after the host entry's copy stream) it no longer gets a hardware queue of its own on
    // this runtime and its 100-600 us kernels sit in the queue of the sweep's 11 us launches
    // (measured: the n = 4096 tree-order sweep took 157 ms instead of 49.5 ms at the end of
    // bench.py's batch run)
    if (hipStreamCreateWithPriority(&c->bulk_stream, hipStreamNonBlocking, lo) != hipSuccess)
      return fail(RNAMC_ERR_HIP);
#endif
  }
  for (int x = 0; x < 16; x++) {
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess)
      return fail(RNAMC_ERR_HIP);
    c->ev_a.push_back(ea);
    c->ev_b.push_back(eb);
  }
  if (workspace_bytes) {
    int rc = ensure_ws(c, workspace_bytes / 4);
    if (rc) return fail(rc);
  }
  *out = c;
  return RNAMC_OK;
}

void rnamc_ctx_destroy(rnamc_ctx* c) {
  if (!c) return;
  {
    DeviceGuard guard(c->device);
    (void)hipDeviceSynchronize();
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->kev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_a) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_b) (void)hipEventDestroy(e);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->bulk_stream) (void)hipStreamDestroy(c->bulk_stream);
    for (hipEvent_t e : c->ev_a2) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_b2) (void)hipEventDestroy(e);
    if (c->ev_dual) (void)hipEventDestroy(c->ev_dual);

    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->st_bases) (void)hipFree(c->st_bases);
    for (int k = 0; k < 2; k++) {
      if (c->st_out[k]) (void)hipFree(c->st_out[k]);
      if (c->pinned[k]) (void)hipHostFree(c->pinned[k]);
      if (c->pinned_ev[k]) (void)hipEventDestroy(c->pinned_ev[k]);
      if (c->group_done[k]) (void)hipEventDestroy(c->group_done[k]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->st_logz) (void)hipFree(c->st_logz);
    if (c->d_params) (void)hipFree(c->d_params);
    if (c->d_hp_init) (void)hipFree(c->d_hp_init);
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_seqs) (void)hipFree(c->d_seqs);
    if (c->d_tseqs) (void)hipFree(c->d_tseqs);
    if (c->d_tree_tabs) (void)hipFree(c->d_tree_tabs);
  }
  delete c;
}

int rnamc_ctx_set_params(rnamc_ctx* c, const rnamc_params* params) {
  if (!c || !params) return RNAMC_ERR_INVALID_ARG;
  if (int rc = validate_params(params)) return rc;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  DeviceGuard guard(c->device);
  if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
  // work of earlier calls may still read the old tables
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(c->d_params, params, sizeof(rnamc_params), hipMemcpyHostToDevice));
  c->host_params = *params;
  c->tree_tabs_valid = false;
  c->hp_init_len = 0;  // the hairpin extrapolation table is derived from the Turner block
  c->fs_contra = c->fs_short = -1;
  c->fs_bases.clear();
  return RNAMC_OK;
}

int rnamc_ctx_set(rnamc_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return RNAMC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  const std::string k(name);
  if (k == "summation_mode" && (value == 0 || value == 1)) {
    c->summation_mode = value;
  } else if (k == "tree_side_stream" && value >= 0 && value <= 2) {
    c->tree_side_force = value;
    c->side_probed = false;
  } else if (k == "tree_waves" && value >= 64) {
    c->tree_pol.waves = static_cast<uint64_t>(value);
  } else if (k == "tree_short" && value >= 1) {
    c->tree_pol.short_terms = static_cast<uint32_t>(std::min<int64_t>(value, 1 << 30));
  } else if (k == "tree_mid_wgs" && value >= 1) {
    c->tree_pol.mid_wgs = static_cast<uint32_t>(std::min<int64_t>(value, 1 << 20));
  } else if (k == "tree_ahead_waves" && value >= 0) {
    c->tree_pol.ahead_waves = static_cast<uint64_t>(value);
  } else if (k == "tree_dual" && (value == 0 || value == 1)) {
    c->tree_dual = value;
#ifdef RNAMC_GEN_DIAGS
  } else if (k == "tree_gen_batch" && value >= 1 && value <= RNAMC_GEN_DIAGS) {  // (experiments: see k_tlane_gen)
#else
  } else if (k == "tree_gen_batch" && value >= 1 && value <= 3) {
#endif
    c->tree_gen_batch = value;
  } else if (k == "tree_lane_band" && value >= 32 && value <= 128 && value % 32 == 0) {
    c->tree_lane_band = value;
  } else if (k == "tree_mid_mx" && (value == 0 || value == 1)) {
    c->tree_pol.mid_mx = static_cast<uint32_t>(value);
  } else if (k == "tree_lane" && value >= 0 && value <= 2) {
    c->tree_lane = value;
  } else if (k == "tree_mid_sync" && (value == 0 || value == 1)) {
    c->tree_mid_sync = value;
  } else if (k == "tree_lane_min_nt" && value >= 0) {
    c->tree_lane_min_nt = value;
  } else if (k == "tree_xcd_rows" && (value == 0 || value == 1)) {
    c->tree_pol.xcd_rows = static_cast<uint32_t>(value);
  } else if (k == "tree_ahead") {
    c->tree_ahead = value;
  } else if (k == "tree_two") {
    c->tree_two = value;
  } else if (k == "tree_band" && value >= 0 && value <= 128 && value % 32 == 0) {
    c->tree_band = value;
  } else if (k == "tree_tpc" && (value == 0 || value == 64 || value == 128 || value == 256 || value == 1024)) {
    c->tree_tpc = value;
  } else if (k == "group_max_seqs" && value >= 1) {
    c->group_max_seqs = std::min<int64_t>(value, 65535);
  } else if (k == "group_max_nt" && value >= 1) {
    c->group_max_nt = value;
  } else if (k == "group_ws_bytes" && value >= 4) {
    c->group_ws_bytes = value;
    c->group_ws_user = true;
  } else if (k == "block_threads" && value >= 64 && value <= 256 && value % 64 == 0) {
    // (the sweep kernels are compiled with __launch_bounds__(256))
    c->block_threads = value;
  } else if (k == "profile") {
    c->profile = value;
  } else if (k == "order_inside" && value >= 0 && value <= 2) {
    c->order_inside = value;
  } else if (k == "order_outside" && value >= 0 && value <= 4) {
    c->order_outside = value;
  } else if (k == "dual_outside") {
    c->dual_outside = value;
  } else if (k == "dual_max_diag" && value >= 0) {
    c->dual_max_diag = value;
  } else if (k == "dual_min_cells" && value >= 0) {
    c->dual_min_cells = static_cast<uint64_t>(value);
  } else if (k == "fuse_inside") {
    c->fuse_inside = value;
  } else if (k == "latency_mode" && value >= 0 && value <= 2) {
    c->latency_mode = value;
  } else if (k == "lat_max_cells" && value >= 0) {
    c->lat_max_cells = value;
  } else if (k == "lat_inside" && (value == 0 || value == 2)) {
    c->lat_inside = value;
  } else if (k == "lat_split") {
    c->lat_split = value;
  } else if (k == "lat_merge") {
    c->lat_merge = value;
  } else if (k == "lat_zr_ahead") {
    c->lat_zr_ahead = value;
  } else if (k == "lat_e_waves" && value >= 0) {
    c->lat_e_waves = value;
  } else if (k == "lat_pairs") {
    c->lat_pairs = value;
#ifdef RNAMC_DEBUG_KNOBS  // result-changing: timing experiments only, never in a release build
  } else if (k == "debug_roles") {
    c->debug_roles = value;
  } else if (k == "tree_debug") {
    c->tree_debug = value;
#endif
  } else {
    return RNAMC_ERR_INVALID_ARG;
  }
  return RNAMC_OK;
}

int rnamc_bpp_batch_device(rnamc_ctx* c, uint32_t n_seqs, const uint8_t* d_bases,
                           const uint64_t* offsets, int uses_contra_model,
                           int allows_short_hairpins, float* d_bpp, const uint64_t* out_offsets,
                           float* d_log_partition, void* hip_stream) {
  if (!c || !offsets || !out_offsets || (n_seqs && (!d_bases || !d_bpp)))
    return RNAMC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  DeviceGuard guard(c->device);
  if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
  return run_batch_mode(c, n_seqs, d_bases, offsets, uses_contra_model != 0,
                        allows_short_hairpins != 0, d_bpp, out_offsets, d_log_partition,
                        static_cast<hipStream_t>(hip_stream));
}

int rnamc_bpp_batch(rnamc_ctx* c, uint32_t n_seqs, const uint8_t* bases, const uint64_t* offsets,
                    int uses_contra_model, int allows_short_hairpins, float* bpp,
                    const uint64_t* out_offsets, float* log_partition) {
  if (!c || !offsets || !out_offsets || (n_seqs && (!bases || !bpp))) return RNAMC_ERR_INVALID_ARG;
  if (n_seqs == 0) return RNAMC_OK;
  for (uint32_t s = 0; s < n_seqs; s++) {
    if (offsets[s + 1] < offsets[s]) return RNAMC_ERR_INVALID_ARG;
    const uint64_t n = offsets[s + 1] - offsets[s];
    if (n == 0) return RNAMC_ERR_EMPTY_SEQ;
    if (n > RNAMC_MAX_SEQ_LEN) return RNAMC_ERR_SEQ_TOO_LONG;
    for (uint64_t x = offsets[s]; x < offsets[s + 1]; x++)
      if (bases[x] > 3) return RNAMC_ERR_INVALID_BASE;
  }
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  DeviceGuard guard(c->device);
  if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
  const uint64_t base_lo = offsets[0], base_hi = offsets[n_seqs];
  std::vector<uint64_t> doff(n_seqs + 1);
  for (uint32_t s = 0; s <= n_seqs; s++) doff[s] = offsets[s] - base_lo;
  // Staging buffers live in the context and only grow (a caller that folds one record after
  // another, like the reference's binaries, would otherwise pay hipMalloc/hipFree per call).
  // The result is never staged whole: group g's triangles sit in st_out[g & 1] while a host
  // thread drains them (copy stream -> pinned bounce chunks -> the caller's buffers) and
  // group g+1 sweeps into the other buffer.
  auto grow = [&](void** p, uint64_t* cap, uint64_t need) -> hipError_t {
    if (*cap >= need && *p) return hipSuccess;
    if (*p) {
      (void)hipFree(*p);
      *p = nullptr;
      *cap = 0;
    }
    const uint64_t want = std::max<uint64_t>(need + need / 8, 4096);
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {  // the headroom is optional
      e = hipMalloc(p, std::max<uint64_t>(need, 1));
      if (e == hipSuccess) *cap = std::max<uint64_t>(need, 1);
      return e;
    }
    *cap = want;
    return hipSuccess;
  };
  constexpr uint64_t kChunkFloats = 16ull << 20;  // 64 MB bounce chunks
  if (!c->copy_stream) HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (int k = 0; k < 2; k++) {
    if (!c->pinned[k]) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->pinned[k]), kChunkFloats * sizeof(float), hipHostMallocDefault));
    if (!c->pinned_ev[k]) HIPCHK(hipEventCreateWithFlags(&c->pinned_ev[k], hipEventDisableTiming));
    if (!c->group_done[k]) HIPCHK(hipEventCreateWithFlags(&c->group_done[k], hipEventDisableTiming));
  }
  HIPCHK(hipStreamSynchronize(c->own_stream));
  HIPCHK(grow(reinterpret_cast<void**>(&c->st_bases), &c->st_bases_cap, base_hi - base_lo));
  HIPCHK(grow(reinterpret_cast<void**>(&c->st_logz), &c->st_logz_cap,
              static_cast<uint64_t>(n_seqs) * sizeof(float)));
  HIPCHK(hipMemcpyAsync(c->st_bases, bases + base_lo, base_hi - base_lo, hipMemcpyHostToDevice,
                        c->own_stream));

  // drain thread: one job per group, in order
  struct Job {
    size_t g;
    uint32_t first, count;
  };
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Job> jobs;
  size_t jobs_taken = 0, drained = 0;  // groups handed over / fully copied out
  bool stop = false;
  int drain_err = RNAMC_OK;
  std::string drain_msg;
  const int device = c->device;
  auto drain = [&]() {
    (void)hipSetDevice(device);
    for (;;) {
      Job job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || jobs_taken < jobs.size(); });
        if (jobs_taken >= jobs.size()) return;  // stop and nothing left
        job = jobs[jobs_taken++];
      }
      const int k = static_cast<int>(job.g & 1);
      hipError_t e = hipEventSynchronize(c->group_done[k]);
      const uint64_t total = c->group_out_floats[job.g];
      const float* src = c->st_out[k];
      const uint64_t nchunks = (total + kChunkFloats - 1) / kChunkFloats;
      auto issue = [&](uint64_t ch) {
        const uint64_t lo = ch * kChunkFloats, len = std::min(kChunkFloats, total - lo);
        hipError_t e2 = hipMemcpyAsync(c->pinned[ch & 1], src + lo, len * sizeof(float),
                                       hipMemcpyDeviceToHost, c->copy_stream);
        if (e2 == hipSuccess) e2 = hipEventRecord(c->pinned_ev[ch & 1], c->copy_stream);
        return e2;
      };
      uint32_t cur = job.first;  // descriptor whose triangle holds the next float to place
      if (e == hipSuccess && nchunks) e = issue(0);
      for (uint64_t ch = 0; ch < nchunks && e == hipSuccess; ch++) {
        if (ch + 1 < nchunks) e = issue(ch + 1);
        if (e == hipSuccess) e = hipEventSynchronize(c->pinned_ev[ch & 1]);
        if (e != hipSuccess) break;
        // scatter [lo, hi) of the group's staging buffer to the caller's triangles
        const uint64_t lo = ch * kChunkFloats, hi = std::min(total, lo + kChunkFloats);
        uint64_t pos = lo;
        while (pos < hi) {
          const SeqDesc& sd = c->descs[cur];
          const uint64_t s_lo = sd.out_off, s_hi = sd.out_off + rnamc_bpp_len(sd.n);
          if (pos >= s_hi) {
            cur++;
            continue;
          }
          const uint64_t upto = std::min(hi, s_hi);
          std::memcpy(bpp + out_offsets[sd.batch_idx] + (pos - s_lo), c->pinned[ch & 1] + (pos - lo),
                      (upto - pos) * sizeof(float));
          pos = upto;
        }
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess && drain_err == RNAMC_OK) {
          drain_err = RNAMC_ERR_HIP;
          drain_msg = std::string("result drain: ") + hipGetErrorString(e);
        }
        drained++;
      }
      cv.notify_all();
    }
  };
  std::thread drainer;
  try {
    drainer = std::thread(drain);
  } catch (...) {  // no thread: nothing has been enqueued yet
    set_last_error("rnamc_bpp_batch: could not start the result-drain thread");
    return RNAMC_ERR_OOM;
  }
  GroupHooks hooks;
  hooks.before = [&](size_t g, float** out_base) -> int {
    const int k = static_cast<int>(g & 1);
    {
      // buffer k was last used by group g-2: wait until it is copied out
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return g < 2 || drained + 2 > g; });
      if (drain_err) return drain_err;
    }
    const uint64_t need = std::max<uint64_t>(c->group_out_floats[g], 1) * sizeof(float);
    if (c->st_out_cap[k] < need) {
      // nothing on the device still writes to or reads from this buffer, but a free
      // synchronises the device: rare (buffers only grow)
      hipError_t e = grow(reinterpret_cast<void**>(&c->st_out[k]), &c->st_out_cap[k], need);
      if (e != hipSuccess) {
        set_last_error(std::string("result staging buffer: ") + hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? RNAMC_ERR_OOM : RNAMC_ERR_HIP;
      }
    }
    *out_base = c->st_out[k];
    return RNAMC_OK;
  };
  hooks.after = [&](size_t g, uint32_t first, uint32_t count) -> int {
    HIPCHK(hipEventRecord(c->group_done[g & 1], c->own_stream));
    {
      std::lock_guard<std::mutex> lk(mu);
      jobs.push_back(Job{g, first, count});
    }
    cv.notify_all();
    return RNAMC_OK;
  };
  int rc = run_batch_mode(c, n_seqs, c->st_bases, doff.data(), uses_contra_model != 0,
                          allows_short_hairpins != 0, nullptr, out_offsets, c->st_logz,
                          c->own_stream, &hooks);
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  cv.notify_all();
  drainer.join();
  if (rc == RNAMC_OK && drain_err) {
    set_last_error(drain_msg);
    rc = drain_err;
  }
  if (rc) {
    (void)hipStreamSynchronize(c->own_stream);
    return rc;
  }
  if (log_partition)
    HIPCHK(hipMemcpyAsync(log_partition, c->st_logz, n_seqs * sizeof(float), hipMemcpyDeviceToHost,
                          c->own_stream));
  HIPCHK(hipStreamSynchronize(c->own_stream));
  for (int k = 0; k < 2; k++)
    if (c->st_out_cap[k] > (24ull << 30)) {  // (two group buffers stay for the next call unless huge)
      (void)hipFree(c->st_out[k]);
      c->st_out[k] = nullptr;
      c->st_out_cap[k] = 0;
    }
  return RNAMC_OK;
}

int rnamc_ctx_last_stats(rnamc_ctx* c, rnamc_batch_stats* out) {
  if (!c || !out) return RNAMC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  *out = c->stats;
  return RNAMC_OK;
}

int rnamc_ctx_stats(rnamc_ctx* c, void* out, uint64_t out_bytes, uint64_t* lib_bytes) {
  if (!c || (!out && out_bytes)) return RNAMC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (lib_bytes) *lib_bytes = sizeof(rnamc_batch_stats);
  if (out && out_bytes)
    std::memcpy(out, &c->stats, static_cast<size_t>(std::min<uint64_t>(out_bytes, sizeof(rnamc_batch_stats))));
  return RNAMC_OK;
}

int rnamc_debug_fetch(rnamc_ctx* c, uint32_t seq_idx, int which, float* out_nxn) {
  if (!c || !out_nxn) return RNAMC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  if (c->group_begin.size() < 2) return RNAMC_ERR_INVALID_ARG;
  const size_t g = c->group_begin.size() - 2;
  const SeqDesc* sd = nullptr;
  for (uint32_t x = c->group_begin[g]; x < c->group_begin[g + 1]; x++)
    if (c->descs[x].batch_idx == seq_idx) sd = &c->descs[x];
  if (!sd) return RNAMC_ERR_INVALID_ARG;
  static const int kMat[7] = {M_QB, M_QA, M_Z, M_Q1D, M_MBC, M_PM, M_PM2};
  if (which < 0 || which > 6) return RNAMC_ERR_INVALID_ARG;
  const bool row_major = which >= 5;
  DeviceGuard guard(c->device);
  if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
  HIPCHK(hipDeviceSynchronize());
  const uint32_t n = sd->n;
  std::vector<float> packed(sd->tri_pad);  // column-major slots are a little larger than tri
  // probs_multibranch{,2} live interleaved ({pm, pm2} per cell) in the two adjacent slots
  const uint64_t slot = row_major ? static_cast<uint64_t>(M_PM) : static_cast<uint64_t>(kMat[which]);
  const size_t count = static_cast<size_t>(sd->tri_pad) * (row_major ? 2 : 1);
  packed.resize(count);
  HIPCHK(hipMemcpy(packed.data(), c->d_ws + sd->ws_off + slot * sd->tri_pad, count * sizeof(float),
                   hipMemcpyDeviceToHost));
  const float nan = std::numeric_limits<float>::quiet_NaN();
  for (uint64_t x = 0; x < static_cast<uint64_t>(n) * n; x++) out_nxn[x] = nan;
  for (uint32_t i = 0; i < n; i++)
    for (uint32_t j = i; j < n; j++) {
      const uint64_t d = j - i;
      const uint64_t cm = j >> 4, cr = j & 15u;
      const uint64_t idx = row_major ? 2ull * (16ull * (cm + 1ull) * (8ull * cm + cr) + i) + (which == 6 ? 1 : 0)
                                     : (d * n - d * (d - 1ull) / 2ull + i);
      out_nxn[static_cast<uint64_t>(i) * n + j] = packed[idx];
    }
  return RNAMC_OK;
}

int rnamc_fold_scores(rnamc_ctx* c, const uint8_t* bases, uint32_t n, int uses_contra_model,
                      int allows_short_hairpins, float* hairpin_scores,
                      float* multibranch_close_scores, float* accessible_scores,
                      rnamc_twoloop_score* twoloop_scores, uint64_t twoloop_cap,
                      uint64_t* twoloop_count) {
  if (!c || !bases) return RNAMC_ERR_INVALID_ARG;
  if (n == 0) return RNAMC_ERR_EMPTY_SEQ;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  const uint64_t tri_len = rnamc_bpp_len(n);
  const bool contra = uses_contra_model != 0, shorthp = allows_short_hairpins != 0;
  const bool cached = c->fs_contra == (contra ? 1 : 0) && c->fs_short == (shorthp ? 1 : 0) &&
                      c->fs_bases.size() == n && std::memcmp(c->fs_bases.data(), bases, n) == 0 &&
                      c->fs_qb.size() == tri_len;
  if (!cached) {
    // the device sweep of this one sequence leaves sums_close in the workspace
    const uint64_t offsets[2] = {0, n}, out_offsets[2] = {0, 0};
    c->fs_contra = c->fs_short = -1;
    c->fs_qb.assign(tri_len, 0.f);
    c->inside_only = true;
    int rc = rnamc_bpp_batch(c, 1, bases, offsets, uses_contra_model, allows_short_hairpins,
                             c->fs_qb.data(), out_offsets, nullptr);
    c->inside_only = false;
    if (rc) return rc;
    DeviceGuard guard(c->device);
    if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
    const SeqDesc& sd = c->descs.back();
    // sums_close, packed diagonal-major like the result
    HIPCHK(hipMemcpy(c->fs_qb.data(),
                     c->d_ws + sd.ws_off + static_cast<uint64_t>(M_QB) * sd.tri_pad,
                     tri_len * sizeof(float), hipMemcpyDeviceToHost));
    c->fs_bases.assign(bases, bases + n);
    c->fs_contra = contra ? 1 : 0;
    c->fs_short = shorthp ? 1 : 0;
  }
  const std::vector<float>& qb = c->fs_qb;
  std::vector<uint64_t> count(n, 0), begin(n + 1, 0);
  const Turner MT{c->host_params.turner, c->h_hp_init.data()};
  const Contra MC{c->host_params.contra};
  auto pass = [&](rnamc_twoloop_score* tl, const std::vector<uint64_t>* b, std::vector<uint64_t>* k) {
    if (contra)
      fold_scores_host(MC, true, shorthp, bases, n, qb.data(), hairpin_scores,
                       multibranch_close_scores, accessible_scores, tl, b, k);
    else
      fold_scores_host(MT, false, shorthp, bases, n, qb.data(), hairpin_scores,
                       multibranch_close_scores, accessible_scores, tl, b, k);
  };
  pass(nullptr, nullptr, &count);
  for (uint32_t d = 0; d < n; d++) begin[d + 1] = begin[d] + count[d];
  if (twoloop_count) *twoloop_count = begin[n];
  if (twoloop_scores) {
    if (twoloop_cap < begin[n]) {
      set_last_error("rnamc_fold_scores: twoloop_cap is smaller than the entry count");
      return RNAMC_ERR_INVALID_ARG;
    }
    pass(twoloop_scores, &begin, nullptr);
  }
  return RNAMC_OK;
}

int rnamc_fold_sums(rnamc_ctx* c, const uint8_t* bases, uint32_t n, int uses_contra_model,
                    int allows_short_hairpins, float* sums_external,
                    float* sums_rightmost_basepairs_external,
                    float* sums_rightmost_basepairs_multibranch, float* sums_close,
                    float* sums_accessible, float* sums_multibranch,
                    float* sums_1ormore_basepairs) {
  if (!c || !bases) return RNAMC_ERR_INVALID_ARG;
  if (n == 0) return RNAMC_ERR_EMPTY_SEQ;
  if (n > RNAMC_MAX_SEQ_LEN) return RNAMC_ERR_SEQ_TOO_LONG;  // (before anything is sized by n)
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  const uint64_t tri_len = rnamc_bpp_len(n);
  // the inside sweep alone (reference order whatever the context's mode is), results left in
  // the workspace; the fold-scores cache of the context is for another sequence afterwards
  const uint64_t offsets[2] = {0, n}, out_offsets[2] = {0, 0};
  c->fs_contra = c->fs_short = -1;
  std::vector<float> packed;
  try {  // nothing may throw across the C boundary
    packed.resize(tri_len);
  } catch (const std::exception&) {
    rnamc::set_last_error("rnamc_fold_sums: no host memory for a packed triangle");
    return RNAMC_ERR_OOM;
  }
  c->inside_only = true;
  int rc = rnamc_bpp_batch(c, 1, bases, offsets, uses_contra_model, allows_short_hairpins,
                           packed.data(), out_offsets, nullptr);
  c->inside_only = false;
  if (rc) return rc;
  DeviceGuard guard(c->device);
  if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
  const SeqDesc& sd = c->descs.back();
  const float ninf = -std::numeric_limits<float>::infinity();
  struct Want {
    float* out;
    int slot;       // workspace slot, or -1: the reference never writes this member in this model
    float initial;  // the reference's initial value (FoldSums::new, 213-226)
  };
  const Want wants[7] = {
      {sums_external, M_Z, 0.f},
      {sums_rightmost_basepairs_external, M_ZRE, ninf},
      {sums_rightmost_basepairs_multibranch, uses_contra_model ? M_ZRM : -1, ninf},
      {sums_close, M_QB, ninf},
      {sums_accessible, M_QA, ninf},
      {sums_multibranch, M_QM, ninf},
      {sums_1ormore_basepairs, M_Q1D, ninf},
  };
  for (const Want& w : wants) {
    if (!w.out) continue;
    for (uint64_t x = 0; x < static_cast<uint64_t>(n) * n; x++) w.out[x] = w.initial;
    if (w.slot < 0) continue;
    HIPCHK(hipMemcpy(packed.data(), c->d_ws + sd.ws_off + static_cast<uint64_t>(w.slot) * sd.tri_pad,
                     tri_len * sizeof(float), hipMemcpyDeviceToHost));
    uint64_t x = 0;  // diagonal-major: cell (i, i+d) at d*n - d(d-1)/2 + i
    for (uint32_t d = 0; d < n; d++)
      for (uint32_t i = 0; i + d < n; i++) w.out[static_cast<uint64_t>(i) * n + i + d] = packed[x++];
  }
  return RNAMC_OK;
}

int rnamc_centroid_fold_multi(rnamc_ctx* c, const float* bpp_packed, uint32_t n,
                              const float* centroid_thresholds, uint32_t n_thresholds,
                              uint32_t* pairs_out, uint32_t max_pairs, uint32_t* n_pairs,
                              float* expect_accuracy) {
  if (!c || !bpp_packed || !centroid_thresholds || !n_pairs || n == 0 || n_thresholds == 0)
    return RNAMC_ERR_INVALID_ARG;
  if (n > RNAMC_MAX_SEQ_LEN) return RNAMC_ERR_SEQ_TOO_LONG;
  if (n_thresholds > 65535u) return RNAMC_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  DeviceGuard guard(c->device);
  if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
  hipStream_t st = c->own_stream;
  const uint32_t ld = ((n + 31u) & ~31u) + 32u;
  const uint64_t msz = ((static_cast<uint64_t>(ld) * n + 63ull) & ~63ull) + 64ull;
  const uint64_t tri = rnamc_bpp_len(n);
  // workspace: per threshold two matrices, then the bpp triangle and the thresholds
  const uint64_t mats = 2ull * msz * n_thresholds;
  const uint64_t need = mats + ((tri + 63ull) & ~63ull) + ((n_thresholds + 63ull) & ~63ull);
  HIPCHK(hipStreamSynchronize(st));
  int rc = ensure_ws(c, need);
  if (rc) return rc;
  float* d_m = c->d_ws;
  float* d_bpp = d_m + mats;
  float* d_g = d_bpp + ((tri + 63ull) & ~63ull);
  HIPCHK(hipMemsetAsync(d_m, 0, mats * sizeof(float), st));  // M = 0 below and on the diagonal
  HIPCHK(hipMemcpyAsync(d_bpp, bpp_packed, tri * sizeof(float), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_g, centroid_thresholds, n_thresholds * sizeof(float), hipMemcpyHostToDevice, st));
  CentroidBatch a{};
  a.bpp = d_bpp;
  a.m = d_m;
  a.gammas = d_g;
  a.n = n;
  a.ld = ld;
  a.msz = msz;
  for (uint32_t d = 1; d < n; d++) launch_centroid(a, d, n_thresholds, st);
  HIPCHK(hipGetLastError());
  // the row-major matrices back to the host; traceback per threshold on host threads
  std::vector<float> host;
  try {
    host.resize(static_cast<size_t>(msz) * n_thresholds);
  } catch (...) {
    (void)hipStreamSynchronize(st);
    return RNAMC_ERR_OOM;
  }
  for (uint32_t g = 0; g < n_thresholds; g++)
    HIPCHK(hipMemcpyAsync(host.data() + static_cast<size_t>(g) * msz, d_m + 2ull * msz * g,
                          msz * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  auto prob = [&](uint32_t i, uint32_t j) { return bpp_packed[rnamc_bpp_index(n, i, j)]; };
  std::atomic<uint32_t> next{0};
  auto work = [&]() {
    for (;;) {
      const uint32_t g = next.fetch_add(1);
      if (g >= n_thresholds) return;
      const float* m = host.data() + static_cast<size_t>(g) * msz;
      auto M = [&](size_t r, size_t col) { return m[r * ld + col]; };
      n_pairs[g] = centroid_traceback(n, centroid_thresholds[g], M, prob,
                                      pairs_out ? pairs_out + static_cast<size_t>(g) * 2u * max_pairs : nullptr,
                                      max_pairs);
      if (expect_accuracy) expect_accuracy[g] = M(0, n - 1);
    }
  };
  const unsigned hw = std::max(1u, std::min<unsigned>({16u, std::thread::hardware_concurrency(), n_thresholds}));
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < hw; t++) {
    try {
      pool.emplace_back(work);
    } catch (...) {
      break;
    }
  }
  work();
  for (auto& t : pool) t.join();
  return RNAMC_OK;
}

int rnamc_durbin_batch(rnamc_ctx* c, const rnamc_align_scores* scores, uint32_t n_seqs,
                       const uint8_t* bases, const uint64_t* offsets, uint32_t n_pairs,
                       const uint32_t* pair_a, const uint32_t* pair_b, float* match_probs,
                       const uint64_t* out_offsets) {
  if (!c || !scores || !offsets || (n_seqs && !bases) ||
      (n_pairs && (!pair_a || !pair_b || !match_probs || !out_offsets)))
    return RNAMC_ERR_INVALID_ARG;
  if (n_pairs == 0) return RNAMC_OK;
  for (uint32_t s = 0; s < n_seqs; s++) {
    if (offsets[s + 1] < offsets[s]) return RNAMC_ERR_INVALID_ARG;
    const uint64_t n = offsets[s + 1] - offsets[s];
    // (the reference indexes [seq_len - 2]: a sequence is at least its two pseudo bases)
    if (n < 2) return RNAMC_ERR_EMPTY_SEQ;
    if (n > RNAMC_MAX_SEQ_LEN + 2ull) return RNAMC_ERR_SEQ_TOO_LONG;
    // real bases inside, anything (PSEUDO_BASE) at the two ends, which are never scored
    for (uint64_t x = offsets[s] + 1; x + 1 < offsets[s + 1]; x++)
      if (bases[x] > 3) return RNAMC_ERR_INVALID_BASE;
  }
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  DeviceGuard guard(c->device);
  if (!guard.ok) return RNAMC_ERR_NO_DEVICE;
  hipStream_t st = c->own_stream;
  const uint64_t base_lo = offsets[0], base_hi = offsets[n_seqs];
  uint8_t* d_bases = nullptr;
  DurbinPair* d_pairs = nullptr;
  float* d_out = nullptr;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(st);
    if (d_bases) (void)hipFree(d_bases);
    if (d_pairs) (void)hipFree(d_pairs);
    if (d_out) (void)hipFree(d_out);
  };
#define HIPCHK_D(expr)                                                     \
  do {                                                                     \
    hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) {                                                \
      set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));   \
      cleanup();                                                           \
      return (_e == hipErrorOutOfMemory) ? RNAMC_ERR_OOM : RNAMC_ERR_HIP;  \
    }                                                                      \
  } while (0)
  HIPCHK_D(hipMalloc(&d_bases, std::max<uint64_t>(base_hi - base_lo, 1)));
  HIPCHK_D(hipMemcpyAsync(d_bases, bases + base_lo, base_hi - base_lo, hipMemcpyHostToDevice, st));
  // pairs in chunks whose six matrices per pair fit the workspace budget
  const uint64_t ws_cap = static_cast<uint64_t>(std::max<int64_t>(c->group_ws_bytes, 1)) / 4;
  std::vector<DurbinPair> chunk;
  std::vector<float> h_out;
  for (uint32_t p0 = 0; p0 < n_pairs;) {
    chunk.clear();
    uint64_t ws = 0, out = 0;
    uint32_t max_cells = 0, p = p0;
    for (; p < n_pairs; p++) {
      if (pair_a[p] >= n_seqs || pair_b[p] >= n_seqs) {
        cleanup();
        return RNAMC_ERR_INVALID_ARG;
      }
      DurbinPair dp{};
      dp.n1 = static_cast<uint32_t>(offsets[pair_a[p] + 1] - offsets[pair_a[p]]);
      dp.n2 = static_cast<uint32_t>(offsets[pair_b[p] + 1] - offsets[pair_b[p]]);
      const uint64_t cells = static_cast<uint64_t>(dp.n1) * dp.n2;
      // (k_durbin_probs indexes pairs through blockIdx.y: at most 65535 per launch)
      if (!chunk.empty() && (ws + 6 * cells > ws_cap || chunk.size() >= 65535u)) break;
      dp.a_off = offsets[pair_a[p]] - base_lo;
      dp.b_off = offsets[pair_b[p]] - base_lo;
      dp.ws_off = ws;
      dp.out_off = out;
      chunk.push_back(dp);
      ws += 6 * cells;
      out += cells;
      max_cells = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint64_t>(max_cells, cells), 0xFFFFFFFFull));
    }
    int rc = ensure_ws(c, ws);
    if (rc) {
      cleanup();
      return rc;
    }
    if (d_pairs) HIPCHK_D(hipFree(d_pairs));
    d_pairs = nullptr;
    if (d_out) HIPCHK_D(hipFree(d_out));
    d_out = nullptr;
    HIPCHK_D(hipMalloc(&d_pairs, chunk.size() * sizeof(DurbinPair)));
    HIPCHK_D(hipMalloc(&d_out, out * sizeof(float)));
    HIPCHK_D(hipMemcpyAsync(d_pairs, chunk.data(), chunk.size() * sizeof(DurbinPair),
                            hipMemcpyHostToDevice, st));
    launch_durbin(d_pairs, static_cast<uint32_t>(chunk.size()), max_cells, d_bases, c->d_ws, d_out,
                  *scores, st);
    HIPCHK_D(hipGetLastError());
    // one D2H per chunk, scattered on the host (a FASTA of many short records makes tens of
    // thousands of pairs: one copy each would cost more than the sweeps)
    try {
      h_out.resize(out);
    } catch (...) {
      cleanup();
      return RNAMC_ERR_OOM;
    }
    HIPCHK_D(hipMemcpyAsync(h_out.data(), d_out, out * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK_D(hipStreamSynchronize(st));
    for (size_t x = 0; x < chunk.size(); x++)
      std::memcpy(match_probs + out_offsets[p0 + x], h_out.data() + chunk[x].out_off,
                  static_cast<uint64_t>(chunk[x].n1) * chunk[x].n2 * sizeof(float));
    p0 = p;
  }
#undef HIPCHK_D
  cleanup();
  return RNAMC_OK;
}

}  // extern "C"
