// Drop-in body for the reference crate's `src/mccaskill_algo.rs` public entry point.
//
// NOT COMPILED IN THIS REPOSITORY (the build image has no Rust toolchain); kept short so
// that a maintainer can check it by eye, and `tests/test_rust_shim_cpu.py` parses the
// `extern "C"` block below against include/rnamc.h (every symbol, arity, integer width and
// pointer-ness).  It keeps the signature of
//     pub fn mccaskill_algo<T>(seq, uses_contra_model, allows_short_hairpins,
//                              fold_score_sets) -> (SparseProbMat<T>, FoldScores<T>)
// (reference: src/mccaskill_algo.rs:247-255) and replaces lines 256-279 by
// pack -> FFI -> unpack.  `FoldSums`, `FoldScores`, `FoldScoreSets` and the four stage
// functions (`get_fold_sums{,_contra}`, `get_basepair_probs{,_contra}`) stay as they are in
// the crate: they remain CPU Rust.
//
// Semantics kept from the reference (src/mccaskill_algo.rs:247-280):
//  * `fold_score_sets` is READ ON EVERY CALL: the device tables are re-uploaded whenever the
//    contents of the argument differ from what the context holds (content hash, not object
//    identity; nothing is cached per set);
//  * `FoldScores<T>` is always filled (cargo feature "no-fold-scores" opts OUT for callers
//    that only take `.0`, like the crate's own binaries);
//  * errors panic, with librnamc's message.
//
// Link: build.rs (`cargo:rustc-link-lib=dylib=rnamc` + search path of librnamc.so).

use std::os::raw::{c_char, c_int, c_void};
use std::sync::Mutex;
use utils::*;

#[repr(C)]
pub struct RnamcCtx {
  _private: [u8; 0],
}

#[repr(C)]
pub struct RnamcPool {
  _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct TwoloopScore {
  i: u32,
  j: u32,
  k: u32,
  l: u32,
  score: f32,
}

extern "C" {
  // include/rnamc.h
  fn rnamc_params_sizeof() -> usize;
  fn rnamc_strerror(status: c_int) -> *const c_char;
  fn rnamc_last_error() -> *const c_char;
  fn rnamc_params_new(init_val: f32, out: *mut c_void) -> c_int;
  fn rnamc_params_save(p: *const c_void, path: *const c_char) -> c_int;
  fn rnamc_params_field(idx: u32, name: *mut *const c_char, byte_offset: *mut u64, count: *mut u64) -> c_int;
  fn rnamc_params_set_special_hairpins(p: *mut c_void, n: u32, seqs: *const u8, lens: *const u8, scores: *const f32) -> c_int;
  fn rnamc_params_set_hairpin_limits(p: *mut c_void, min_hairpin_len: u32, max_hairpin_len_extrapolation: u32, min_hairpin_len_extrapolation: u32) -> c_int;
  fn rnamc_pool_create(params: *const c_void, devices: *const c_int, n_devices: u32, workspace_bytes: u64, out: *mut *mut RnamcPool) -> c_int;
  fn rnamc_pool_ctx(pool: *mut RnamcPool, idx: u32) -> *mut RnamcCtx;
  fn rnamc_pool_set_params(pool: *mut RnamcPool, params: *const c_void) -> c_int;
  fn rnamc_pool_set(pool: *mut RnamcPool, name: *const c_char, value: i64) -> c_int;
  fn rnamc_bpp_len(n: u32) -> u64;
  fn rnamc_bpp_batch_multi(pool: *mut RnamcPool, n_seqs: u32, bases: *const u8, offsets: *const u64, uses_contra_model: c_int, allows_short_hairpins: c_int, bpp: *mut f32, out_offsets: *const u64, log_partition: *mut f32) -> c_int;
  fn rnamc_fold_scores(ctx: *mut RnamcCtx, bases: *const u8, n: u32, uses_contra_model: c_int, allows_short_hairpins: c_int, hairpin_scores: *mut f32, multibranch_close_scores: *mut f32, accessible_scores: *mut f32, twoloop_scores: *mut TwoloopScore, twoloop_cap: u64, twoloop_count: *mut u64) -> c_int;
  fn rnamc_fold_sums(ctx: *mut RnamcCtx, bases: *const u8, n: u32, uses_contra_model: c_int, allows_short_hairpins: c_int, sums_external: *mut f32, sums_rightmost_basepairs_external: *mut f32, sums_rightmost_basepairs_multibranch: *mut f32, sums_close: *mut f32, sums_accessible: *mut f32, sums_multibranch: *mut f32, sums_1ormore_basepairs: *mut f32) -> c_int;
}

// panic with librnamc's own words (the reference panics on the same inputs:
// src/utils.rs:570-572, src/mccaskill_algo.rs:526)
fn check(status: c_int, what: &str) {
  if status != 0 {
    let (a, b) = unsafe {
      (
        std::ffi::CStr::from_ptr(rnamc_strerror(status)).to_string_lossy().into_owned(),
        std::ffi::CStr::from_ptr(rnamc_last_error()).to_string_lossy().into_owned(),
      )
    };
    panic!("{}: rnamc status {} ({}) {}", what, status, a, b);
  }
}

// One pool per process: a device context (tables, workspace, streams) on EVERY visible GPU,
// reused across calls; `key` is the hash of the FoldScoreSets contents its tables were built
// from.  A batch is sharded over the pool's devices inside librnamc (rnamc_bpp_batch_multi).
struct State {
  pool: *mut RnamcPool,
  key: u64,
}
unsafe impl Send for State {}
static STATE: Mutex<Option<State>> = Mutex::new(None);

fn content_key(f: &FoldScoreSets) -> u64 {
  // FNV-1a over the bytes of the set (f32 arrays and scalars only: no padding)
  let bytes = unsafe {
    std::slice::from_raw_parts(f as *const FoldScoreSets as *const u8, std::mem::size_of::<FoldScoreSets>())
  };
  let mut h = 0xcbf29ce484222325u64;
  for &b in bytes {
    h = (h ^ b as u64).wrapping_mul(0x100000001b3);
  }
  h
}

// Runs `f` with a context whose tables are `fold_score_sets`.  The lock is held across the
// FFI call, so a caller on another thread with a different set cannot swap the tables under
// it (calls on one context are serialised inside librnamc anyway).
fn with_pool<R>(fold_score_sets: &FoldScoreSets, f: impl FnOnce(*mut RnamcPool) -> R) -> R {
  let key = content_key(fold_score_sets);
  let mut guard = STATE.lock().unwrap_or_else(|e| e.into_inner());
  let stale = match guard.as_ref() {
    Some(st) => st.key != key,
    None => true,
  };
  if stale {
    let params = build_params(fold_score_sets);
    let p = params.as_ptr() as *const c_void;
    match guard.as_mut() {
      Some(st) => {
        check(unsafe { rnamc_pool_set_params(st.pool, p) }, "rnamc_pool_set_params");
        st.key = key;
      }
      None => {
        let mut pool = std::ptr::null_mut();
        // Which devices: RNAMC_DEVICES="0,2" lists them; under a one-process-per-GPU launch
        // (LOCAL_RANK set) the rank's own device only — a rank that saw every device would open
        // streams, tables and a workspace on all of them and shard its batch across its peers'
        // GPUs; otherwise n_devices = 0: one context per visible GPU (the reference's binary
        // takes every core: src/bin/mccaskill_algo.rs:44-48)
        let devices: Vec<c_int> = match (std::env::var("RNAMC_DEVICES"), std::env::var("LOCAL_RANK")) {
          (Ok(v), _) => v.split(',').filter_map(|x| x.trim().parse::<c_int>().ok()).collect(),
          (_, Ok(r)) => r.trim().parse::<c_int>().map(|d| vec![d]).unwrap_or_default(),
          _ => Vec::new(),
        };
        let dev_ptr = if devices.is_empty() { std::ptr::null() } else { devices.as_ptr() };
        check(unsafe { rnamc_pool_create(p, dev_ptr, devices.len() as u32, 0, &mut pool) }, "rnamc_pool_create");
        // RNA_ALGOS_SUMMATION_MODE=1 opts in to the tree-order sums (about 10x faster on a lone
        // long sequence; NOT bit-comparable with the reference CPU path: include/rnamc.h,
        // rnamc_ctx_set).  The default, 0, is the reference's own summation order.
        if let Ok(v) = std::env::var("RNA_ALGOS_SUMMATION_MODE") {
          let name = std::ffi::CString::new("summation_mode").unwrap();
          check(unsafe { rnamc_pool_set(pool, name.as_ptr(), v.parse::<i64>().unwrap_or(0)) }, "rnamc_pool_set");
        }
        *guard = Some(State { pool, key });
      }
    }
  }
  f(guard.as_ref().unwrap().pool)
}

// the first context of the pool, for the per-sequence entries (rnamc_fold_scores)
fn with_context<R>(fold_score_sets: &FoldScoreSets, f: impl FnOnce(*mut RnamcCtx) -> R) -> R {
  with_pool(fold_score_sets, |pool| f(unsafe { rnamc_pool_ctx(pool, 0) }))
}

// Fills a rnamc_params block: every f32 table by name (rnamc_params_field gives name, byte
// offset and element count; multi-dimensional arrays are row-major like the Rust arrays),
// then the special hairpins and the three hairpin length constants.  "turner.*" come from
// rna_ss_params::compiled_scores_turner (what the reference's Turner branch reads directly,
// src/utils.rs:166-411), "contra.*" from the FoldScoreSets the caller built.
pub fn build_params(f: &FoldScoreSets) -> Vec<u8> {
  let mut params = vec![0u8; unsafe { rnamc_params_sizeof() }];
  check(unsafe { rnamc_params_new(0., params.as_mut_ptr() as *mut c_void) }, "rnamc_params_new");
  unsafe fn flat<A>(a: &A) -> &[f32] {
    std::slice::from_raw_parts(a as *const A as *const f32, std::mem::size_of::<A>() / 4)
  }
  let scalar = |x: &Score| -> Vec<f32> { vec![*x] };
  let mut idx = 0u32;
  loop {
    let (mut name, mut off, mut cnt) = (std::ptr::null(), 0u64, 0u64);
    if unsafe { rnamc_params_field(idx, &mut name, &mut off, &mut cnt) } != 0 {
      break;
    }
    idx += 1;
    let name = unsafe { std::ffi::CStr::from_ptr(name) }.to_str().unwrap();
    let src: Vec<f32> = unsafe {
      match name {
        "turner.hairpin_scores_init" => flat(&HAIRPIN_SCORES_INIT).to_vec(),
        "turner.terminal_mismatch_scores_hairpin" => flat(&TERMINAL_MISMATCH_SCORES_HAIRPIN).to_vec(),
        "turner.stack_scores" => flat(&STACK_SCORES).to_vec(),
        "turner.bulge_scores_init" => flat(&BULGE_SCORES_INIT).to_vec(),
        "turner.interior_scores_init" => flat(&INTERIOR_SCORES_INIT).to_vec(),
        "turner.interior_scores_1x1" => flat(&INTERIOR_SCORES_1X1).to_vec(),
        "turner.interior_scores_1x2" => flat(&INTERIOR_SCORES_1X2).to_vec(),
        "turner.interior_scores_2x2" => flat(&INTERIOR_SCORES_2X2).to_vec(),
        "turner.terminal_mismatch_scores_1xmany" => flat(&TERMINAL_MISMATCH_SCORES_1XMANY).to_vec(),
        "turner.terminal_mismatch_scores_2x3" => flat(&TERMINAL_MISMATCH_SCORES_2X3).to_vec(),
        "turner.terminal_mismatch_scores_interior" => flat(&TERMINAL_MISMATCH_SCORES_INTERIOR).to_vec(),
        "turner.terminal_mismatch_scores_multibranch" => flat(&TERMINAL_MISMATCH_SCORES_MULTIBRANCH).to_vec(),
        "turner.dangling_scores_5prime" => flat(&DANGLING_SCORES_5PRIME).to_vec(),
        "turner.dangling_scores_3prime" => flat(&DANGLING_SCORES_3PRIME).to_vec(),
        "turner.helix_augu_end_penalty" => scalar(&HELIX_AUGU_END_PENALTY),
        "turner.coeff_hairpin_len_extrapolation" => scalar(&COEFF_HAIRPIN_LEN_EXTRAPOLATION),
        "turner.ninio_coeff" => scalar(&NINIO_COEFF),
        "turner.ninio_max" => scalar(&NINIO_MAX),
        "turner.init_multibranch_base" => scalar(&INIT_MULTIBRANCH_BASE),
        "turner.coeff_num_branches" => scalar(&COEFF_NUM_BRANCHES),
        "turner.special_hairpin_scores" => continue, // set below with the sequences
        "contra.hairpin_scores_len" => flat(&f.hairpin_scores_len).to_vec(),
        "contra.bulge_scores_len" => flat(&f.bulge_scores_len).to_vec(),
        "contra.interior_scores_len" => flat(&f.interior_scores_len).to_vec(),
        "contra.interior_scores_symmetric" => flat(&f.interior_scores_symmetric).to_vec(),
        "contra.interior_scores_asymmetric" => flat(&f.interior_scores_asymmetric).to_vec(),
        "contra.stack_scores" => flat(&f.stack_scores).to_vec(),
        "contra.terminal_mismatch_scores" => flat(&f.terminal_mismatch_scores).to_vec(),
        "contra.dangling_scores_left" => flat(&f.dangling_scores_left).to_vec(),
        "contra.dangling_scores_right" => flat(&f.dangling_scores_right).to_vec(),
        "contra.helix_close_scores" => flat(&f.helix_close_scores).to_vec(),
        "contra.basepair_scores" => flat(&f.basepair_scores).to_vec(),
        "contra.interior_scores_explicit" => flat(&f.interior_scores_explicit).to_vec(),
        "contra.bulge_scores_0x1" => flat(&f.bulge_scores_0x1).to_vec(),
        "contra.interior_scores_1x1" => flat(&f.interior_scores_1x1).to_vec(),
        "contra.multibranch_score_base" => scalar(&f.multibranch_score_base),
        "contra.multibranch_score_basepair" => scalar(&f.multibranch_score_basepair),
        "contra.multibranch_score_unpair" => scalar(&f.multibranch_score_unpair),
        "contra.external_score_basepair" => scalar(&f.external_score_basepair),
        "contra.external_score_unpair" => scalar(&f.external_score_unpair),
        "contra.hairpin_scores_len_cumulative" => flat(&f.hairpin_scores_len_cumulative).to_vec(),
        "contra.bulge_scores_len_cumulative" => flat(&f.bulge_scores_len_cumulative).to_vec(),
        "contra.interior_scores_len_cumulative" => flat(&f.interior_scores_len_cumulative).to_vec(),
        "contra.interior_scores_symmetric_cumulative" => flat(&f.interior_scores_symmetric_cumulative).to_vec(),
        "contra.interior_scores_asymmetric_cumulative" => flat(&f.interior_scores_asymmetric_cumulative).to_vec(),
        other => panic!("unknown rnamc_params field {}", other),
      }
    };
    // a table whose shape differs from librnamc's compile-time limits (RNAMC_MAX_LOOP_LEN
    // ...: recalled values of rna-ss-params constants) stops here, loudly
    assert_eq!(src.len() as u64, cnt, "shape of {}", name);
    let dst = &mut params[off as usize..off as usize + 4 * cnt as usize];
    dst.copy_from_slice(unsafe { std::slice::from_raw_parts(src.as_ptr() as *const u8, dst.len()) });
  }
  // the limits librnamc was compiled with must be the crate's
  assert_eq!(MAX_2LOOP_LEN, 30, "RNAMC_MAX_2LOOP_LEN");
  assert_eq!(MAX_LOOP_LEN, 30, "RNAMC_MAX_LOOP_LEN");
  assert_eq!(MIN_SPAN_HAIRPIN_CLOSE, 5, "RNAMC_MIN_SPAN_HAIRPIN_CLOSE");
  assert_eq!(MAX_INTERIOR_EXPLICIT, 4, "RNAMC_MAX_INTERIOR_EXPLICIT");
  // HAIRPIN_SCORES_SPECIAL: (sequence, score) pairs (src/utils.rs:198-205)
  const W: usize = 16; // RNAMC_MAX_SPECIAL_HAIRPIN_LEN
  let n = HAIRPIN_SCORES_SPECIAL.len();
  let (mut seqs, mut lens, mut scores) = (vec![0u8; n * W], vec![0u8; n], vec![0f32; n]);
  for (x, (hairpin, score)) in HAIRPIN_SCORES_SPECIAL.iter().enumerate() {
    for (y, &b) in hairpin.iter().enumerate() {
      seqs[x * W + y] = b as u8;
    }
    lens[x] = hairpin.len() as u8;
    scores[x] = *score;
  }
  let p = params.as_mut_ptr() as *mut c_void;
  check(
    unsafe { rnamc_params_set_special_hairpins(p, n as u32, seqs.as_ptr(), lens.as_ptr(), scores.as_ptr()) },
    "rnamc_params_set_special_hairpins",
  );
  check(
    unsafe {
      rnamc_params_set_hairpin_limits(
        p,
        MIN_HAIRPIN_LEN as u32,
        MAX_HAIRPIN_LEN_EXTRAPOLATION as u32,
        MIN_HAIRPIN_LEN_EXTRAPOLATION as u32,
      )
    },
    "rnamc_params_set_hairpin_limits",
  );
  params
}

// table file for hosts without the crate (Python mirror, tests): bindings/rust/dump_tables.rs
pub fn save_params(f: &FoldScoreSets, path: &str) {
  let params = build_params(f);
  let c_path = std::ffi::CString::new(path).unwrap();
  check(unsafe { rnamc_params_save(params.as_ptr() as *const c_void, c_path.as_ptr()) }, "rnamc_params_save");
}

// diagonal-major packed triangle -> SparseProbMat<T>; absent pairs hold -1.0
fn unpack_probs<T: HashIndex>(packed: &[f32], n: usize) -> SparseProbMat<T> {
  let mut basepair_probs = SparseProbMat::<T>::default();
  let mut x = 0;
  for d in 0..n {
    for i in 0..n - d {
      let p = packed[x];
      x += 1;
      if p >= -0.5 {
        basepair_probs.insert((T::from_usize(i).unwrap(), T::from_usize(i + d).unwrap()), p);
      }
    }
  }
  basepair_probs
}

// The whole FASTA in one call, sharded over the process's GPUs (every visible one unless
// RNAMC_DEVICES / LOCAL_RANK say otherwise: with_pool): what
// src/bin/mccaskill_algo.rs:58-93 and src/bin/centroid_fold.rs:119-132 do with one pool task
// per record on all cores.  Called from that pool instead, every per-sequence call would
// serialise on the pool and run as a latency-bound group of one on one device.  Returns the
// `.0` of mccaskill_algo per record (the binaries discard `.1`).
pub fn mccaskill_algo_batch<T>(
  seqs: &[SeqSlice],
  uses_contra_model: bool,
  allows_short_hairpins: bool,
  fold_score_sets: &FoldScoreSets,
) -> Vec<SparseProbMat<T>>
where
  T: HashIndex,
{
  let mut bases = Vec::<u8>::new();
  let (mut offsets, mut out_offsets) = (vec![0u64], vec![0u64]);
  for seq in seqs {
    bases.extend(seq.iter().map(|&x| x as u8));
    offsets.push(bases.len() as u64);
    out_offsets.push(out_offsets.last().unwrap() + unsafe { rnamc_bpp_len(seq.len() as u32) });
  }
  let mut packed = vec![0f32; (*out_offsets.last().unwrap() as usize).max(1)];
  with_pool(fold_score_sets, |pool| {
    check(
      unsafe {
        rnamc_bpp_batch_multi(
          pool,
          seqs.len() as u32,
          bases.as_ptr(),
          offsets.as_ptr(),
          uses_contra_model as c_int,
          allows_short_hairpins as c_int,
          packed.as_mut_ptr(),
          out_offsets.as_ptr(),
          std::ptr::null_mut(),
        )
      },
      "rnamc_bpp_batch_multi",
    )
  });
  seqs
    .iter()
    .enumerate()
    .map(|(s, seq)| unpack_probs::<T>(&packed[out_offsets[s] as usize..out_offsets[s + 1] as usize], seq.len()))
    .collect()
}

pub fn mccaskill_algo<T>(
  seq: SeqSlice,
  uses_contra_model: bool,
  allows_short_hairpins: bool,
  fold_score_sets: &FoldScoreSets,
) -> (SparseProbMat<T>, FoldScores<T>)
where
  T: HashIndex,
{
  let basepair_probs = mccaskill_algo_batch::<T>(&[seq], uses_contra_model, allows_short_hairpins, fold_score_sets)
    .pop()
    .unwrap();
  // FoldScores<T> is part of the reference's result and filled by default; callers that
  // only take `.0` (src/bin/*.rs, tests/tests.rs, benches/benches.rs) may build the crate
  // with feature "no-fold-scores" and get empty maps instead of ~n^2 * 186 hash inserts.
  let fold_scores = if cfg!(feature = "no-fold-scores") {
    FoldScores::<T>::new()
  } else {
    get_fold_scores::<T>(seq, uses_contra_model, allows_short_hairpins, fold_score_sets)
  };
  (basepair_probs, fold_scores)
}

// The four maps of src/mccaskill_algo.rs:14-19 through rnamc_fold_scores: three packed
// triangles (NaN = key absent) and the list of twoloop_scores inserts.  The first call
// counts, the second fills; librnamc keeps the key set of the sequence between the two, so
// the device sweep runs once.
fn get_fold_scores<T: HashIndex>(
  seq: SeqSlice,
  uses_contra_model: bool,
  allows_short_hairpins: bool,
  fold_score_sets: &FoldScoreSets,
) -> FoldScores<T> {
  let n = seq.len();
  let bases: Vec<u8> = seq.iter().map(|&x| x as u8).collect();
  let len = unsafe { rnamc_bpp_len(n as u32) } as usize;
  let (mut hp, mut mb, mut ac) = (vec![0f32; len], vec![0f32; len], vec![0f32; len]);
  let (c, s) = (uses_contra_model as c_int, allows_short_hairpins as c_int);
  let p = bases.as_ptr();
  let tl = with_context(fold_score_sets, |ctx| {
    let mut count = 0u64;
    check(
      unsafe {
        rnamc_fold_scores(ctx, p, n as u32, c, s, hp.as_mut_ptr(), mb.as_mut_ptr(), ac.as_mut_ptr(),
          std::ptr::null_mut(), 0, &mut count)
      },
      "rnamc_fold_scores (count)",
    );
    let mut tl = vec![TwoloopScore::default(); count as usize];
    check(
      unsafe {
        rnamc_fold_scores(ctx, p, n as u32, c, s, std::ptr::null_mut(), std::ptr::null_mut(),
          std::ptr::null_mut(), tl.as_mut_ptr(), count, &mut count)
      },
      "rnamc_fold_scores (fill)",
    );
    tl
  });
  let mut out = FoldScores::<T>::new();
  let t = |x: usize| T::from_usize(x).unwrap();
  let mut x = 0;
  for d in 0..n {
    for i in 0..n - d {
      let key = (t(i), t(i + d));
      if !hp[x].is_nan() { out.hairpin_scores.insert(key, hp[x]); }
      if !mb[x].is_nan() { out.multibranch_close_scores.insert(key, mb[x]); }
      if !ac[x].is_nan() { out.accessible_scores.insert(key, ac[x]); }
      x += 1;
    }
  }
  for e in tl.iter() {
    let key = (t(e.i as usize), t(e.j as usize), t(e.k as usize), t(e.l as usize));
    out.twoloop_scores.insert(key, e.score);
  }
  out
}

// `FoldSums<T>` of the reference's first stage (src/mccaskill_algo.rs:282-378 / 380-516) from the
// device: the inside sweep alone through rnamc_fold_sums.  A maintainer who wants the GPU behind
// `get_fold_sums{,_contra}` as well calls this from their bodies (and `get_fold_scores` above for
// the `&mut FoldScores<T>` they fill on the way); the crate's own callers never use the stages
// separately (src/mccaskill_algo.rs:256-279 is the only call site).
pub fn fold_sums_device<T: HashIndex>(
  seq: SeqSlice,
  uses_contra_model: bool,
  allows_short_hairpins: bool,
  fold_score_sets: &FoldScoreSets,
) -> FoldSums<T> {
  let n = seq.len();
  let bases: Vec<u8> = seq.iter().map(|&x| x as u8).collect();
  let mut m: Vec<Vec<f32>> = (0..7).map(|_| vec![0f32; n * n]).collect();
  with_context(fold_score_sets, |ctx| {
    let p: Vec<*mut f32> = m.iter_mut().map(|v| v.as_mut_ptr()).collect();
    check(
      unsafe {
        rnamc_fold_sums(ctx, bases.as_ptr(), n as u32, uses_contra_model as c_int,
          allows_short_hairpins as c_int, p[0], p[1], p[2], p[3], p[4], p[5], p[6])
      },
      "rnamc_fold_sums",
    );
  });
  let rows = |v: &Vec<f32>| -> SumMat { v.chunks(n).map(|r| r.to_vec()).collect() };
  let sparse = |v: &Vec<f32>| -> SparseSumMat<T> {
    let mut out = SparseSumMat::<T>::default();
    for i in 0..n {
      for j in i..n {
        let x = v[i * n + j];
        // the reference inserts finite sums only (src/mccaskill_algo.rs:332-338 / 456-462)
        if x.is_finite() { out.insert((T::from_usize(i).unwrap(), T::from_usize(j).unwrap()), x); }
      }
    }
    out
  };
  FoldSums {
    sums_external: rows(&m[0]),
    sums_rightmost_basepairs_external: rows(&m[1]),
    sums_rightmost_basepairs_multibranch: rows(&m[2]),
    sums_close: sparse(&m[3]),
    sums_accessible: sparse(&m[4]),
    sums_multibranch: rows(&m[5]),
    sums_1ormore_basepairs: rows(&m[6]),
  }
}
