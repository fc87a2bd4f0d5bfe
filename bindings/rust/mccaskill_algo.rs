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
  // FoldScores<T>: no in-crate caller reads it (src/bin/*.rs take `.0`); callers that do
  // (downstream crates) keep calling the crate's own get_fold_sums*, which still fills it.
  (basepair_probs, FoldScores::<T>::new())
}
