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

// One context per process: the tables of `fold_score_sets` are copied into a
// rnamc_params block field by field (names from rnamc_params_field: "contra.<field>"
// for every FoldScoreSets field, "turner.<CONST>" filled from rna_ss_params::*).
fn context(fold_score_sets: &FoldScoreSets) -> &'static Ctx {
  static CTX: OnceLock<Ctx> = OnceLock::new();
  CTX.get_or_init(|| unsafe {
    let mut params = vec![0u8; rnamc_params_sizeof()];
    assert_eq!(rnamc_params_new(0., params.as_mut_ptr() as *mut c_void), 0);
    copy_tables(&mut params, fold_score_sets); // see INTEGRATION.md §3 (mechanical)
    let mut ctx = std::ptr::null_mut();
    assert_eq!(rnamc_ctx_create(params.as_ptr() as *const c_void, -1, 0, &mut ctx), 0);
    Ctx(ctx)
  })
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
