/*
 * rnamc.h — C ABI of librnamc.so: the MI355X-native McCaskill partition-function /
 * base-pairing-probability (bpp) hot path of heartsh/rna-algos.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own; the entry
 * points below are what a Rust `extern "C"` block inside the crate's
 * `src/mccaskill_algo.rs` would bind so that
 *
 *     pub fn mccaskill_algo<T>(seq, uses_contra_model, allows_short_hairpins,
 *                              fold_score_sets) -> (SparseProbMat<T>, FoldScores<T>)
 *     (reference: src/mccaskill_algo.rs:247-280)
 *
 * keeps its signature while its body becomes "pack -> rnamc_bpp_batch -> unpack"
 * (binding shown in INTEGRATION.md).  Plain pointers and sizes only; no C++ or
 * torch types cross this line.  Every function returns an `int` status
 * (RNAMC_OK == 0) and never aborts or throws; the Rust shim turns a non-zero
 * status into `panic!()` to preserve the reference's error behaviour
 * (src/utils.rs:570-572, src/mccaskill_algo.rs:526).
 *
 * Scoring tables are INPUTS.  The reference takes its Turner-2004 and
 * CONTRAfold v2.02 numbers from the third-party crate `rna-ss-params = "0.1"`
 * (Cargo.toml:12, glob-imported at src/utils.rs:8-10), which is not part of the
 * reference tree; `rnamc_params` below holds every table the path reads
 *  with fixed shapes, and is filled from a table file
 * (rnamc_params_load) or from the seeded synthetic generator
 * (rnamc_params_synthetic).
 */
#ifndef RNAMC_H
#define RNAMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: summation_mode knob (tree-order sums), rnamc_ctx_stats (sized copy), multi-device batch
 *    entry rnamc_bpp_batch_multi; rnamc_params itself is unchanged since 2 */
#define RNAMC_ABI_VERSION 3u

/* Compile-time limits.  In the reference these are constants of rna-ss-params
 * (recalled values): NUM_BASES, MAX_2LOOP_LEN (src/utils.rs:308),
 * MAX_LOOP_LEN (src/mccaskill_algo.rs:32-34,405), MIN_SPAN_HAIRPIN_CLOSE
 * (src/mccaskill_algo.rs:290), MAX_INTERIOR_* (src/mccaskill_algo.rs:35-36,43). */
#define RNAMC_NUM_BASES 4
#define RNAMC_MAX_2LOOP_LEN 30
#define RNAMC_MAX_LOOP_LEN 30
#define RNAMC_MIN_SPAN_HAIRPIN_CLOSE 5
#define RNAMC_MAX_INTERIOR_EXPLICIT 4
#define RNAMC_MAX_INTERIOR_SYMMETRIC 15
#define RNAMC_MAX_INTERIOR_ASYMMETRIC 28
#define RNAMC_MAX_SPECIAL_HAIRPINS 64
#define RNAMC_MAX_SPECIAL_HAIRPIN_LEN 16
/* Longest sequence the reference's callers can pass (T = u16,
 * src/bin/mccaskill_algo.rs:70-90). */
#define RNAMC_MAX_SEQ_LEN 65535u

/* Base codes (bytes2seq, src/utils.rs:562-577). */
#define RNAMC_A 0
#define RNAMC_C 1
#define RNAMC_G 2
#define RNAMC_U 3

/* Status codes. */
#define RNAMC_OK 0
#define RNAMC_ERR_INVALID_ARG 1   /* null pointer, bad size, bad ABI version      */
#define RNAMC_ERR_INVALID_BASE 2  /* byte outside ACGUacgu / code outside 0..3    */
#define RNAMC_ERR_EMPTY_SEQ 3     /* n == 0 (reference panics: mccaskill_algo.rs:526) */
#define RNAMC_ERR_SEQ_TOO_LONG 4  /* n > 65535                                     */
#define RNAMC_ERR_NO_DEVICE 5     /* no HIP device / HIP runtime failure at init   */
#define RNAMC_ERR_OOM 6           /* host or device allocation failed              */
#define RNAMC_ERR_HIP 7           /* a HIP call failed (see rnamc_last_error)      */
#define RNAMC_ERR_IO 8            /* table file could not be read / written        */
#define RNAMC_ERR_FORMAT 9        /* table file has wrong magic / version / size   */

/* ------------------------------------------------------------------------- */
/* The CONTRAfold parameter set: field-for-field mirror of `FoldScoreSets`
 * (reference: src/utils.rs:91-119; constructor/accumulate/transfer at
 * src/mccaskill_algo.rs:24-211).  All f32 (`Score`). */
typedef struct rnamc_fold_score_sets {
  float hairpin_scores_len[RNAMC_MAX_LOOP_LEN + 1];
  float bulge_scores_len[RNAMC_MAX_LOOP_LEN];
  float interior_scores_len[RNAMC_MAX_LOOP_LEN - 1];
  float interior_scores_symmetric[RNAMC_MAX_INTERIOR_SYMMETRIC];
  float interior_scores_asymmetric[RNAMC_MAX_INTERIOR_ASYMMETRIC];
  float stack_scores[4][4][4][4];
  float terminal_mismatch_scores[4][4][4][4];
  float dangling_scores_left[4][4][4];
  float dangling_scores_right[4][4][4];
  float helix_close_scores[4][4];
  float basepair_scores[4][4];
  float interior_scores_explicit[RNAMC_MAX_INTERIOR_EXPLICIT][RNAMC_MAX_INTERIOR_EXPLICIT];
  float bulge_scores_0x1[4];
  float interior_scores_1x1[4][4];
  float multibranch_score_base;
  float multibranch_score_basepair;
  float multibranch_score_unpair;
  float external_score_basepair;
  float external_score_unpair;
  /* cumulative ("at least") forms, filled by rnamc_fold_score_sets_accumulate */
  float hairpin_scores_len_cumulative[RNAMC_MAX_LOOP_LEN + 1];
  float bulge_scores_len_cumulative[RNAMC_MAX_LOOP_LEN];
  float interior_scores_len_cumulative[RNAMC_MAX_LOOP_LEN - 1];
  float interior_scores_symmetric_cumulative[RNAMC_MAX_INTERIOR_SYMMETRIC];
  float interior_scores_asymmetric_cumulative[RNAMC_MAX_INTERIOR_ASYMMETRIC];
} rnamc_fold_score_sets;

/* The Turner-2004 constants the path reads straight from rna-ss-params
 * (call sites: src/utils.rs:166-411, src/mccaskill_algo.rs:364,367,594).
 * Values are scores, i.e. free energies already multiplied by -1/(kT). */
typedef struct rnamc_turner_scores {
  float hairpin_scores_init[RNAMC_MAX_LOOP_LEN + 1];           /* HAIRPIN_SCORES_INIT            */
  float terminal_mismatch_scores_hairpin[4][4][4][4];          /* TERMINAL_MISMATCH_SCORES_HAIRPIN */
  float stack_scores[4][4][4][4];                              /* STACK_SCORES                   */
  float bulge_scores_init[RNAMC_MAX_2LOOP_LEN + 1];            /* BULGE_SCORES_INIT              */
  float interior_scores_init[RNAMC_MAX_2LOOP_LEN + 1];         /* INTERIOR_SCORES_INIT           */
  float interior_scores_1x1[4][4][4][4][4][4];                 /* INTERIOR_SCORES_1X1            */
  float interior_scores_1x2[4][4][4][4][4][4][4];              /* INTERIOR_SCORES_1X2            */
  float interior_scores_2x2[4][4][4][4][4][4][4][4];           /* INTERIOR_SCORES_2X2            */
  float terminal_mismatch_scores_1xmany[4][4][4][4];
  float terminal_mismatch_scores_2x3[4][4][4][4];
  float terminal_mismatch_scores_interior[4][4][4][4];
  float terminal_mismatch_scores_multibranch[4][4][4][4];
  float dangling_scores_5prime[4][4][4];
  float dangling_scores_3prime[4][4][4];
  float helix_augu_end_penalty;
  float coeff_hairpin_len_extrapolation;
  float ninio_coeff;
  float ninio_max;
  float init_multibranch_base;
  float coeff_num_branches;
  /* HAIRPIN_SCORES_SPECIAL: (whole hairpin incl. closing pair, score), compared by
   * slice equality (src/utils.rs:198-205).  bases are codes 0..3. */
  float special_hairpin_scores[RNAMC_MAX_SPECIAL_HAIRPINS];
  uint8_t special_hairpin_seqs[RNAMC_MAX_SPECIAL_HAIRPINS][RNAMC_MAX_SPECIAL_HAIRPIN_LEN];
  uint8_t special_hairpin_lens[RNAMC_MAX_SPECIAL_HAIRPINS];
  uint32_t num_special_hairpins;
  uint32_t min_hairpin_len;                /* MIN_HAIRPIN_LEN                 (recalled: 3)  */
  uint32_t max_hairpin_len_extrapolation;  /* MAX_HAIRPIN_LEN_EXTRAPOLATION   (recalled: 9)  */
  uint32_t min_hairpin_len_extrapolation;  /* MIN_HAIRPIN_LEN_EXTRAPOLATION   (recalled: 10) */
} rnamc_turner_scores;

typedef struct rnamc_params {
  uint32_t abi_version;   /* must be RNAMC_ABI_VERSION                     */
  uint32_t struct_bytes;  /* must be sizeof(rnamc_params)                   */
  uint64_t table_id;      /* provenance tag: synthetic seed, or file hash   */
  rnamc_turner_scores turner;
  rnamc_fold_score_sets contra;
} rnamc_params;

/* ------------------------------------------------------------------------- */
/* Misc. */
uint32_t rnamc_abi_version(void);
size_t rnamc_params_sizeof(void);
const char* rnamc_strerror(int status);
/* Thread-local detail of the last failing call on this thread (HIP error text). */
const char* rnamc_last_error(void);

/* bytes2seq (src/utils.rs:562-577): ASCII ACGUacgu -> codes 0..3; any other byte
 * -> RNAMC_ERR_INVALID_BASE (the reference panics). */
int rnamc_bytes2seq(const uint8_t* ascii, uint64_t n, uint8_t* codes);

/* Number of f32 slots of one sequence's packed bpp triangle: n(n+1)/2. */
uint64_t rnamc_bpp_len(uint32_t n);
/* Index of pair (i,j), i <= j < n, in the packed triangle (diagonal-major:
 * all pairs with j-i = d are contiguous, ordered by i). */
uint64_t rnamc_bpp_index(uint32_t n, uint32_t i, uint32_t j);

/* ------------------------------------------------------------------------- */
/* Parameter plumbing. */

/* FoldScoreSets::new(init_val)  (src/mccaskill_algo.rs:25-58). */
int rnamc_fold_score_sets_new(float init_val, rnamc_fold_score_sets* out);
/* FoldScoreSets::accumulate     (src/mccaskill_algo.rs:60-86). */
int rnamc_fold_score_sets_accumulate(rnamc_fold_score_sets* fss);
/* FoldScoreSets::transfer       (src/mccaskill_algo.rs:88-210): copy the
 * "compiled" CONTRAfold tables of `src` into `dst`, skipping (leaving as they
 * are) the entries whose closing pair -- and for stack_scores also the enclosed
 * pair -- is non-canonical, then accumulate. */
int rnamc_fold_score_sets_transfer(rnamc_fold_score_sets* dst, const rnamc_fold_score_sets* src);

/* Whole parameter block with every f32 set to init_val (header filled in). */
int rnamc_params_new(float init_val, rnamc_params* out);
/* Deterministic synthetic tables for BOTH models (SplitMix64 stream seeded with
 * `seed`, Turner-like magnitudes; contra tables pass through `transfer`). */
int rnamc_params_synthetic(uint64_t seed, rnamc_params* out);
/* Binary table file: 16-byte header {"RNAMCTBL", abi u32, bytes u32} + struct. */
int rnamc_params_save(const rnamc_params* p, const char* path);
int rnamc_params_load(const char* path, rnamc_params* out);
/* Enumerate the f32 array fields of rnamc_params (for host-language mirrors):
 * returns RNAMC_ERR_INVALID_ARG when idx is past the end. */
int rnamc_params_field(uint32_t idx, const char** name, uint64_t* byte_offset, uint64_t* count);
/* The non-float members, for host languages that do not mirror the struct:
 * HAIRPIN_SCORES_SPECIAL (src/utils.rs:198-205): n entries, sequence x as len[x] base codes
 * at seqs[x * RNAMC_MAX_SPECIAL_HAIRPIN_LEN ...]; and the three hairpin length constants
 * (src/utils.rs:174-183). */
int rnamc_params_set_special_hairpins(rnamc_params* p, uint32_t n, const uint8_t* seqs,
                                      const uint8_t* lens, const float* scores);
int rnamc_params_set_hairpin_limits(rnamc_params* p, uint32_t min_hairpin_len,
                                    uint32_t max_hairpin_len_extrapolation,
                                    uint32_t min_hairpin_len_extrapolation);

/* ------------------------------------------------------------------------- */
/* Device context: owns the uploaded tables, a workspace and streams on ONE
 * GPU.  Thread-safe: calls on one ctx are serialised by an internal mutex; use
 * one ctx per host thread (or per GPU) for concurrency. */
typedef struct rnamc_ctx rnamc_ctx;

/* device < 0 selects the current HIP device.  workspace_bytes == 0 sizes the DP
 * workspace lazily from the first batch (grown on demand). */
int rnamc_ctx_create(const rnamc_params* params, int device, uint64_t workspace_bytes,
                     rnamc_ctx** out);
void rnamc_ctx_destroy(rnamc_ctx* ctx);
/* Replace the tables of a live context (waits for the context's earlier work).  The
 * reference reads `&FoldScoreSets` on every call (src/mccaskill_algo.rs:247-280) and the
 * set is trainable (src/utils.rs:91-119): a host mirror that caches a context calls this
 * whenever the contents of the caller's set differ from the ones last uploaded. */
int rnamc_ctx_set_params(rnamc_ctx* ctx, const rnamc_params* params);
/* Knobs; returns RNAMC_ERR_INVALID_ARG for unknown names or values.
 *
 * "summation_mode" selects how every logsumexp sum of the sweep is evaluated:
 *   0 (default) reference order: each fold runs in the reference's k order with its cubic
 *     ln_exp_1p / expf pieces (src/utils.rs:579-655) — bit-identical to the reference CPU path
 *     (the parity gate; north_star's <= 1e-6);
 *   1 tree order: order-free sums (every lane keeps a running max and a sum of exp, merged by
 *     wave and LDS reductions; hardware exp2 / log2), the reference's recurrences with the
 *     cell-independent Theta(n^3) loops turned into prefix recurrences
 *     (rna_algos_amd/csrc/rnamc_tree.hip).  NOT bit-comparable with the reference: the
 *     reference's fold is approximate and order-dependent, so this mode differs from it by the
 *     reference's own error (measured max |dp| 1e-3 at n = 76, 7e-3 ... 2e-2 at n = 4096; ln Z by
 *     3e-2 ... 1e-1 at n = 4096) while agreeing with the exact (f64) value of the same
 *     recurrences to f32 rounding (|dp| < 1e-4 at n = 1024).  Key sets are identical.  About
 *     ten times faster than mode 0 on a lone long sequence.  rnamc_fold_scores and
 *     rnamc_debug_fetch always use mode 0.
 * Tuning (all optional): "group_max_seqs","group_max_nt","group_ws_bytes","block_threads",
 * "fuse_inside","dual_outside","dual_min_cells","dual_max_diag","order_inside","order_outside",
 * "latency_mode","lat_max_cells","lat_inside","lat_e_waves","lat_pairs",
 * "lat_merge","lat_zr_ahead","profile"; tree-order mode: "tree_two" (two anti-diagonals per
 * launch, default 1), "tree_tpc" (threads per cell: 64 / 128 / 256 / 1024, 0 = by diagonal size),
 * "tree_band" (width of a band of anti-diagonals whose cubic products take their mid-field from
 * the tiled kernel k_tree_mid one band ahead: 0 / 32 / 64 / 96 / 128, default 64; 0 = every launch
 * walks its sums whole), "tree_mid_wgs" (workgroups of a mid-field launch; default by length: 256 / 512 / 1024),
 * "tree_ahead" (banded sweeps: the far part of a launch's 2-loop blocks is summed by extra
 * workgroups of the previous launch, default 1), "tree_waves" (waves a tree-order launch may hold
 * at once when it picks threads per cell, default 5120), "tree_short" (sums of at most this many
 * terms take one wave per cell, default 256), "tree_ahead_waves" (waves up to which the ahead role
 * takes one wave per cell instead of one per row, default 16384), "tree_side_stream" (0: probe
 * whether the side stream runs beside the caller's stream and sweep unbanded if not — the default;
 * 1 / 2: take it as concurrent / serialised without probing), "tree_lane" (lane-per-cell sweeps,
 * rnamc_tree_lane.h: 0 never, 1 for calls of several sequences and at least "tree_lane_min_nt"
 * nucleotides — the default —, 2 always), "tree_mid_sync" (lane-per-cell sweeps: a band's mid-field
 * kernel runs in front of the band on the sweep's own stream, default 1), "tree_lane_band" (the band
 * width of those sweeps: 32 / 64 / 96 / 128, default 32, never wider than "tree_band"), "tree_gen_batch"
 * (diagonals whose generic 2-loop sums share a launch there: 1 .. 3, default 3), "tree_dual" (every other
 * group of such a call on a second stream with its own half of the workspace, so that two groups fill each
 * other's launch gaps: default 1; device-resident entry only), "tree_mid_mx" (the
 * mid-field products on the matrix cores, k_tree_mid_mx: exponentials per operand element, one
 * v_mfma_f32_32x32x2_f32 multiply-add per term; default 1, 0 = the VALU form k_tree_mid).
 * Every knob is per context. */
int rnamc_ctx_set(rnamc_ctx* ctx, const char* name, int64_t value);

/* mccaskill_algo over a batch (src/mccaskill_algo.rs:247-280 for each record, as
 * src/bin/mccaskill_algo.rs:64-93 does on its thread pool).
 *   bases        concatenated base codes 0..3, host memory
 *   offsets      n_seqs+1 prefix offsets into `bases`
 *   bpp          per sequence s the packed triangle of n_s(n_s+1)/2 f32 at
 *                bpp + out_offsets[s]; entry (i,j) at rnamc_bpp_index; a pair
 *                absent from the reference's SparseProbMat holds -1.0f
 *   log_partition  n_seqs f32: sums_external[0][n-1] (may be NULL)
 * The summation order of every logsumexp fold is the reference's unless the context's
 * "summation_mode" knob says otherwise (rnamc_ctx_set). */
int rnamc_bpp_batch(rnamc_ctx* ctx, uint32_t n_seqs, const uint8_t* bases,
                    const uint64_t* offsets, int uses_contra_model, int allows_short_hairpins,
                    float* bpp, const uint64_t* out_offsets, float* log_partition);

/* Same, with `bases`, `bpp`, `log_partition` already resident in device memory
 * of ctx's GPU; `offsets`/`out_offsets` stay host arrays.  Work is enqueued on
 * `hip_stream` (a hipStream_t, may be NULL for the default stream) and the call
 * returns without synchronising. */
int rnamc_bpp_batch_device(rnamc_ctx* ctx, uint32_t n_seqs, const uint8_t* d_bases,
                           const uint64_t* offsets, int uses_contra_model,
                           int allows_short_hairpins, float* d_bpp, const uint64_t* out_offsets,
                           float* d_log_partition, void* hip_stream);

/* ------------------------------------------------------------------------- */
/* The batch over SEVERAL devices: what the reference's binaries do with one pool task per
 * record on all cores (src/bin/mccaskill_algo.rs:58-93, src/bin/centroid_fold.rs:119-132).
 * A pool owns one device context (tables, workspace, streams) per listed device and one host
 * thread per context during a call.  The batch is cut into as many shards as there are
 * contexts — contiguous bands of the cost-sorted batch with equal total cost under the
 * measured sweep model (rnamc_shard_plan) — every shard runs through rnamc_bpp_batch of its
 * own context and writes straight into the caller's host triangles.  Sequences are never
 * split; there is no collective and no device-to-device traffic.
 *   devices / n_devices  HIP device ordinals, one context each (an ordinal may repeat: several
 *                        contexts on one GPU); n_devices == 0 = every visible device
 *   workspace_bytes      per context, as in rnamc_ctx_create */
typedef struct rnamc_pool rnamc_pool;
int rnamc_pool_create(const rnamc_params* params, const int* devices, uint32_t n_devices,
                      uint64_t workspace_bytes, rnamc_pool** out);
void rnamc_pool_destroy(rnamc_pool* pool);
uint32_t rnamc_pool_size(const rnamc_pool* pool);
/* context idx of the pool (owned by the pool), e.g. for rnamc_ctx_stats; NULL past the end */
rnamc_ctx* rnamc_pool_ctx(rnamc_pool* pool, uint32_t idx);
/* rnamc_ctx_set_params / rnamc_ctx_set on every context of the pool */
int rnamc_pool_set_params(rnamc_pool* pool, const rnamc_params* params);
int rnamc_pool_set(rnamc_pool* pool, const char* name, int64_t value);
/* Arguments as rnamc_bpp_batch.  The whole batch is validated first: a bad record fails the
 * call with its status before any device is touched.  A device-side failure of one shard is
 * returned once the other shards have finished (no buffer is left in use). */
int rnamc_bpp_batch_multi(rnamc_pool* pool, uint32_t n_seqs, const uint8_t* bases,
                          const uint64_t* offsets, int uses_contra_model,
                          int allows_short_hairpins, float* bpp, const uint64_t* out_offsets,
                          float* log_partition);
/* The partition rnamc_bpp_batch_multi uses (host only, no device needed): shard_of_seq[s] in
 * 0 .. n_shards-1.  Shard k holds a band of the batch sorted by length (longest first,
 * stable); cuts lie where the running cost a*n(n^2-1)/6 + b*n^2 reaches k/n_shards of the total. */
int rnamc_shard_plan(uint32_t n_seqs, const uint64_t* offsets, uint32_t n_shards,
                     uint32_t* shard_of_seq);
/* The cost model behind the partition, seconds of one reference-order sweep of an n-nt sequence
 * on an MI355X: RNAMC_COST_S_PER_CELL_K * n(n^2-1)/6 (the Theta(n^3) folds) +
 * RNAMC_COST_S_PER_N2 * n^2 (the 496-probe 2-loop blocks); fitted on one box (DESIGN.md section
 * 6).  THE one source of the two constants: rnamc_shard_plan, bench.py's shards and
 * rna_algos_amd.workloads.sweep_cost all evaluate rnamc_sweep_cost. */
#define RNAMC_COST_S_PER_CELL_K 3.25e-12
#define RNAMC_COST_S_PER_N2 6.6e-10
int rnamc_sweep_cost(uint32_t count, const uint64_t* lengths, double* cost_s);

/* Per-kernel accounting of the last batch call on this ctx (launch counts and
 * device time by HIP events on the launch stream; the latter only when
 * profiling was switched on with rnamc_ctx_set(ctx,"profile",1)). */
typedef struct rnamc_batch_stats {
  uint64_t n_groups;
  uint64_t launches_inside;
  uint64_t launches_outside;  /* kernels of the outside sweeps (two per diagonal on large launches) */
  uint64_t launches_other;
  double ms_inside;   /* sum of event-timed inside sweeps  */
  double ms_outside;  /* sum of event-timed outside sweeps */
  double ms_other;
  uint64_t workspace_bytes;
  /* per-kernel accounting of the outside sweep (rnamc_ctx_set(ctx,"profile",2) only; the
   * extra event records cost about 2 % of the sweep): launches and summed
   * durations (HIP events around every launch, on the stream it is launched on) of
   *   main : k_outside<.,5>  probs_multibranch + 2-loop half of the pair probabilities
   *   tail : k_outside<.,2>  multibranch half of the pair probabilities (second stream)
   *   small: k_outside<.,7>  all three roles in one kernel (launches too small to split) */
  uint64_t launches_outside_main, launches_outside_tail, launches_outside_small;
  double ms_outside_main, ms_outside_tail, ms_outside_small;
  /* tree-order mode, banded sweep: does the context's side stream (mid-field products) run BESIDE
   * the caller's stream?  0 not probed (no banded call yet), 1 yes, 2 no — the two share a
   * hardware queue (the process holds too many streams): the sweep then runs unbanded (slower on
   * long sequences, never wrong).  Probed once per context and stream, ~0.2 ms
   * (rnamc_tree.hip, tree_side_stream_probe). */
  uint64_t tree_side_stream;
} rnamc_batch_stats;
int rnamc_ctx_last_stats(rnamc_ctx* ctx, rnamc_batch_stats* out);
/* The same with the caller's idea of the struct size: copies min(out_bytes, sizeof) bytes, so a
 * binding compiled against an older (shorter) rnamc_batch_stats is never overrun.  Returns the
 * library's sizeof(rnamc_batch_stats) through *lib_bytes (may be NULL). */
int rnamc_ctx_stats(rnamc_ctx* ctx, void* out, uint64_t out_bytes, uint64_t* lib_bytes);

/* Debug / test hook: copy one DP matrix of sequence `seq_idx` of the LAST group
 * of the last batch call back to the host as a dense n*n row-major f32 matrix
 * (cells outside the stored triangle = NaN).  which: 0 sums_close, 1
 * sums_accessible, 2 sums_external, 3 sums_1ormore_basepairs, 4
 * multibranch_close_scores, 5 probs_multibranch, 6 probs_multibranch2. */
int rnamc_debug_fetch(rnamc_ctx* ctx, uint32_t seq_idx, int which, float* out_nxn);

/* ------------------------------------------------------------------------- */
/* FoldScores<T>, the second element of mccaskill_algo's result
 * (src/mccaskill_algo.rs:14-19): the loop scores the inside pass looked up, keyed like
 * its hash maps.  The key sets depend on the DP (a pair is a key of
 * multibranch_close_scores / accessible_scores iff its sums_close is finite, :333-338 /
 * :457-462; (i,j,k,l) is a key of twoloop_scores iff (k,l) has a sums_close entry, :318-320
 * / :429-431), so the call runs the device sweep for the sequence, reads the sums_close
 * key set back and scores on the host with the functions the kernels use.
 *   hairpin_scores, multibranch_close_scores, accessible_scores: packed triangles of
 *     rnamc_bpp_len(n) floats (rnamc_bpp_index), NaN = key absent; each may be NULL.
 *   twoloop_scores: up to twoloop_cap entries, closing pair (i,j) diagonal-major, then
 *     k ascending, l descending (the reference's visiting order); *twoloop_count gets the
 *     full count.  With twoloop_scores == NULL only the count is produced; a non-NULL
 *     buffer that is too small gives RNAMC_ERR_INVALID_ARG (count still set).
 * The context keeps the key set of the last sequence it swept: the usual pair of calls
 * (count, allocate, fill) on the same sequence and flags runs ONE device sweep. */
typedef struct rnamc_twoloop_score {
  uint32_t i, j, k, l; /* (i,j) closes, (k,l) is enclosed */
  float score;
} rnamc_twoloop_score;
int rnamc_fold_scores(rnamc_ctx* ctx, const uint8_t* bases, uint32_t n, int uses_contra_model,
                      int allows_short_hairpins, float* hairpin_scores,
                      float* multibranch_close_scores, float* accessible_scores,
                      rnamc_twoloop_score* twoloop_scores, uint64_t twoloop_cap,
                      uint64_t* twoloop_count);

/* FoldSums<T> (src/mccaskill_algo.rs:3-11, 213-226): what the reference's first stage,
 * `pub fn get_fold_sums` / `get_fold_sums_contra` (src/mccaskill_algo.rs:282-378 / 380-516),
 * returns.  The inside sweep of ONE sequence on the device, always in reference order, then the
 * seven members as n x n row-major matrices like the reference's Vec<Vec<f32>> (any pointer may
 * be NULL).  Cells the reference never writes hold its initial values: sums_external 0 (lower
 * triangle and spans it skips included), the other dense matrices -inf; the two sparse maps
 * (sums_close, sums_accessible) come dense with -inf for an absent key (the reference inserts
 * finite values only, src/mccaskill_algo.rs:332-338 / 456-462).  Under Turner the reference never
 * writes sums_rightmost_basepairs_multibranch: all -inf.  The second stage (`get_basepair_probs*`)
 * is not offered alone: it reads the caller's FoldScores maps instead of the sequence; the whole
 * path is rnamc_bpp_batch. */
int rnamc_fold_sums(rnamc_ctx* ctx, const uint8_t* bases, uint32_t n, int uses_contra_model,
                    int allows_short_hairpins, float* sums_external,
                    float* sums_rightmost_basepairs_external,
                    float* sums_rightmost_basepairs_multibranch, float* sums_close,
                    float* sums_accessible, float* sums_multibranch,
                    float* sums_1ormore_basepairs);

/* ------------------------------------------------------------------------- */
/* Consumers of the path's output, host side. */

/* centroid_fold (src/centroid_fold.rs:25-105) driven off a packed bpp triangle
 * as produced above.  pairs_out receives up to max_pairs (i,j) pairs in the
 * reference's push order; *n_pairs the count; *expect_accuracy the score. */
int rnamc_centroid_fold(const float* bpp_packed, uint32_t n, float centroid_threshold,
                        uint32_t* pairs_out, uint32_t max_pairs, uint32_t* n_pairs,
                        float* expect_accuracy);

/* The same fold for SEVERAL thresholds at once, the Theta(n^3) fill on ctx's GPU (the reference
 * runs it for 18 gammas per record, src/bin/centroid_fold.rs:147-161): one anti-diagonal sweep of
 * (max,+) reductions for all thresholds off one copy of the bpp triangle, then the traceback of
 * every threshold on host threads.  (max,+) is order-free in f32, so the matrices, and with them
 * pairs, push order and expect_accuracy, are bit-identical to rnamc_centroid_fold's.
 *   pairs_out        n_thresholds blocks of max_pairs (i,j) pairs (may be NULL)
 *   n_pairs          n_thresholds counts;  expect_accuracy  n_thresholds scores (may be NULL) */
int rnamc_centroid_fold_multi(rnamc_ctx* ctx, const float* bpp_packed, uint32_t n,
                              const float* centroid_thresholds, uint32_t n_thresholds,
                              uint32_t* pairs_out, uint32_t max_pairs, uint32_t* n_pairs,
                              float* expect_accuracy);

/* ------------------------------------------------------------------------- */
/* Durbin pair-HMM nucleotide match probabilities (SURVEY.md 8f-4;
 * reference: src/durbin_algo.rs:73-242, constants src/compiled_align_scores.rs). */

/* AlignScores (src/durbin_algo.rs:4-14). */
typedef struct rnamc_align_scores {
  float match2match_score;
  float match2insert_score;
  float insert_extend_score;
  float insert_switch_score;
  float init_match_score;
  float init_insert_score;
  float insert_scores[RNAMC_NUM_BASES];
  float match_scores[RNAMC_NUM_BASES][RNAMC_NUM_BASES];
} rnamc_align_scores;

/* AlignScores::new(init_val) (src/durbin_algo.rs:26-40). */
int rnamc_align_scores_new(float init_val, rnamc_align_scores* out);
/* AlignScores::transfer (src/durbin_algo.rs:42-57): the compiled CONTRAlign constants of
 * src/compiled_align_scores.rs:2-19 (they are part of the reference tree, unlike the
 * McCaskill tables). */
int rnamc_align_scores_transfer(rnamc_align_scores* scores);

/* durbin_algo (src/durbin_algo.rs:73-77) over a batch of sequence pairs, as
 * src/bin/durbin_algo.rs:55-75 runs one pool task per pair.
 *   bases     concatenated codes of ALL sequences, each carrying PSEUDO_BASE (= 4,
 *             src/utils.rs:122) at both ends as the reference's callers build them
 *             (src/bin/durbin_algo.rs:49-51); real bases are codes 0..3
 *   offsets   n_seqs+1 prefix offsets into `bases` (every sequence has length >= 2)
 *   pair_a/b  n_pairs indices into the sequences: pair p aligns pair_a[p] with pair_b[p]
 *   match_probs  per pair p a dense row-major ProbMat of len(a) x len(b) f32 at
 *             match_probs + out_offsets[p] (border rows / columns are 0, as in the reference)
 * Each pair's forward and backward sweeps run by anti-diagonal on the GPU; every logsumexp
 * fold is the reference's, in its order. */
int rnamc_durbin_batch(rnamc_ctx* ctx, const rnamc_align_scores* scores, uint32_t n_seqs,
                       const uint8_t* bases, const uint64_t* offsets, uint32_t n_pairs,
                       const uint32_t* pair_a, const uint32_t* pair_b, float* match_probs,
                       const uint64_t* out_offsets);

#ifdef __cplusplus
}
#endif
#endif /* RNAMC_H */
