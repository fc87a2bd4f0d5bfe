// Drop-in body for the reference crate's `src/mccaskill_algo.rs` public entry point.
//
// NOT COMPILED IN THIS REPOSITORY (the build image has no Rust toolchain); kept
// short so that a maintainer can check it by eye.  It keeps the signature of
//     pub fn mccaskill_algo<T>(seq, uses_contra_model, allows_short_hairpins,
//                              fold_score_sets) -> (SparseProbMat<T>, FoldScores<T>)
// (reference: src/mccaskill_algo.rs:247-255) and replaces lines 256-279 by
// pack -> FFI -> unpack.  `FoldSums`, `FoldScores`, `FoldScoreSets` and the four
// stage functions stay as they are in the crate (their structs are not touched).
//
// Link: `cargo:rustc-link-lib=dylib=rnamc` + search path of librnamc.so (build.rs).

use std::os::raw::{c_char, c_int, c_void};
use std::sync::OnceLock;
use utils::*;

#[repr(C)]
pub struct RnamcCtx {
  _private: [u8; 0],
}

extern "C" {
  // include/rnamc.h
  fn rnamc_params_sizeof() -> usize;
  fn rnamc_params_load(path: *const c_char, out: *mut c_void) -> c_int;
  fn rnamc_params_new(init_val: f32, out: *mut c_void) -> c_int;
  fn rnamc_params_field(idx: u32, name: *mut *const c_char, off: *mut u64, cnt: *mut u64) -> c_int;
  fn rnamc_params_set_special_hairpins(p: *mut c_void, n: u32, seqs: *const u8, lens: *const u8,
    scores: *const f32) -> c_int;
  fn rnamc_params_set_hairpin_limits(p: *mut c_void, min_len: u32, max_extrap: u32,
    min_extrap: u32) -> c_int;
  fn rnamc_ctx_create(params: *const c_void, device: c_int, ws: u64, out: *mut *mut RnamcCtx) -> c_int;
  fn rnamc_bpp_len(n: u32) -> u64;
  fn rnamc_bpp_batch(
    ctx: *mut RnamcCtx,
    n_seqs: u32,
    bases: *const u8,
    offsets: *const u64,
    uses_contra_model: c_int,
    allows_short_hairpins: c_int,
    bpp: *mut f32,
    out_offsets: *const u64,
    log_partition: *mut f32,
  ) -> c_int;
  fn rnamc_fold_scores(
    ctx: *mut RnamcCtx,
    bases: *const u8,
    n: u32,
    uses_contra_model: c_int,
    allows_short_hairpins: c_int,
    hairpin_scores: *mut f32,
    multibranch_close_scores: *mut f32,
    accessible_scores: *mut f32,
    twoloop_scores: *mut TwoloopScore,
    twoloop_cap: u64,
    twoloop_count: *mut u64,
  ) -> c_int;
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

struct Ctx(*mut RnamcCtx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {} // calls on one ctx are serialised inside librnamc

// One context per process: the tables are copied into a rnamc_params block by copy_tables.
fn context(fold_score_sets: &FoldScoreSets) -> &'static Ctx {
  static CTX: OnceLock<Ctx> = OnceLock::new();
  CTX.get_or_init(|| unsafe {
    let mut params = vec![0u8; rnamc_params_sizeof()];
    assert_eq!(rnamc_params_new(0., params.as_mut_ptr() as *mut c_void), 0);
    copy_tables(&mut params, fold_score_sets);
    let mut ctx = std::ptr::null_mut();
    assert_eq!(rnamc_ctx_create(params.as_ptr() as *const c_void, -1, 0, &mut ctx), 0);
    Ctx(ctx)
  })
}

// Fills the rnamc_params block: every f32 table by name (rnamc_params_field gives name, byte
// offset and element count; multi-dimensional arrays are row-major like the Rust arrays),
// then the special hairpins and the three hairpin length constants.  "turner.*" come from
// rna_ss_params::compiled_scores_turner, "contra.*" from the FoldScoreSets the caller built.
fn copy_tables(params: &mut [u8], f: &FoldScoreSets) {
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
    assert_eq!(src.len() as u64, cnt, "shape of {}", name);
    let dst = &mut params[off as usize..off as usize + 4 * cnt as usize];
    dst.copy_from_slice(unsafe { std::slice::from_raw_parts(src.as_ptr() as *const u8, dst.len()) });
  }
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
  unsafe {
    assert_eq!(rnamc_params_set_special_hairpins(p, n as u32, seqs.as_ptr(), lens.as_ptr(), scores.as_ptr()), 0);
    assert_eq!(rnamc_params_set_hairpin_limits(p, MIN_HAIRPIN_LEN as u32,
      MAX_HAIRPIN_LEN_EXTRAPOLATION as u32, MIN_HAIRPIN_LEN_EXTRAPOLATION as u32), 0);
  }
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
  let n = seq.len();
  let bases: Vec<u8> = seq.iter().map(|&x| x as u8).collect();
  let offsets = [0u64, n as u64];
  let len = unsafe { rnamc_bpp_len(n as u32) } as usize;
  let out_offsets = [0u64, len as u64];
  let mut packed = vec![0f32; len.max(1)];
  let mut log_partition = 0f32;
  let status = unsafe {
    rnamc_bpp_batch(
      context(fold_score_sets).0,
      1,
      bases.as_ptr(),
      offsets.as_ptr(),
      uses_contra_model as c_int,
      allows_short_hairpins as c_int,
      packed.as_mut_ptr(),
      out_offsets.as_ptr(),
      &mut log_partition,
    )
  };
  if status != 0 {
    panic!(); // the reference panics on empty / invalid input (src/mccaskill_algo.rs:526)
  }
  // diagonal-major packed triangle -> SparseProbMat<T>; absent pairs hold -1.0
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
  // FoldScores<T>: no in-crate caller reads it (src/bin/*.rs take `.0`), so it is filled
  // only when the crate is built with feature "fold-scores" (downstream crates that read it).
  let fold_scores = if cfg!(feature = "fold-scores") {
    get_fold_scores::<T>(&bases, uses_contra_model, allows_short_hairpins, fold_score_sets)
  } else {
    FoldScores::<T>::new()
  };
  (basepair_probs, fold_scores)
}

// The four maps of src/mccaskill_algo.rs:14-19 through rnamc_fold_scores: three packed
// triangles (NaN = key absent) and the list of twoloop_scores inserts.
fn get_fold_scores<T: HashIndex>(
  bases: &[u8],
  uses_contra_model: bool,
  allows_short_hairpins: bool,
  fold_score_sets: &FoldScoreSets,
) -> FoldScores<T> {
  let n = bases.len();
  let len = unsafe { rnamc_bpp_len(n as u32) } as usize;
  let (mut hp, mut mb, mut ac) = (vec![0f32; len], vec![0f32; len], vec![0f32; len]);
  let ctx = context(fold_score_sets).0;
  let (c, s) = (uses_contra_model as c_int, allows_short_hairpins as c_int);
  let mut count = 0u64;
  let p = bases.as_ptr();
  unsafe {
    let st = rnamc_fold_scores(ctx, p, n as u32, c, s, hp.as_mut_ptr(), mb.as_mut_ptr(),
      ac.as_mut_ptr(), std::ptr::null_mut(), 0, &mut count);
    assert_eq!(st, 0);
  }
  let mut tl = vec![TwoloopScore::default(); count as usize];
  unsafe {
    let st = rnamc_fold_scores(ctx, p, n as u32, c, s, std::ptr::null_mut(), std::ptr::null_mut(),
      std::ptr::null_mut(), tl.as_mut_ptr(), count, &mut count);
    assert_eq!(st, 0);
  }
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
