// Drop-in body for the reference crate's `src/durbin_algo.rs` entry point
//     pub fn durbin_algo(seq_pair: &SeqPair, align_scores: &AlignScores) -> ProbMat
// (reference: src/durbin_algo.rs:73-77).  NOT COMPILED IN THIS REPOSITORY (no
// Rust toolchain in the build image); tests/test_rust_shim_cpu.py checks the `extern "C"` block
// against include/rnamc.h.  `AlignScores`, `AlignSums`, `get_align_sums` stay as they are in the
// crate.  Sequences carry PSEUDO_BASE at both ends, as the reference's callers build them.

use std::os::raw::{c_char, c_int, c_void};
use utils::*;

#[repr(C)]
pub struct RnamcCtx {
  _private: [u8; 0],
}

// rnamc_align_scores (include/rnamc.h): field for field the reference's AlignScores
#[repr(C)]
pub struct AlignScoresC {
  match2match_score: f32,
  match2insert_score: f32,
  insert_extend_score: f32,
  insert_switch_score: f32,
  init_match_score: f32,
  init_insert_score: f32,
  insert_scores: [f32; 4],
  match_scores: [[f32; 4]; 4],
}

extern "C" {
  // include/rnamc.h
  fn rnamc_strerror(status: c_int) -> *const c_char;
  fn rnamc_last_error() -> *const c_char;
  fn rnamc_params_sizeof() -> usize;
  fn rnamc_params_new(init_val: f32, out: *mut c_void) -> c_int;
  fn rnamc_ctx_create(params: *const c_void, device: c_int, workspace_bytes: u64, out: *mut *mut RnamcCtx) -> c_int;
  fn rnamc_durbin_batch(ctx: *mut RnamcCtx, scores: *const AlignScoresC, n_seqs: u32, bases: *const u8, offsets: *const u64, n_pairs: u32, pair_a: *const u32, pair_b: *const u32, match_probs: *mut f32, out_offsets: *const u64) -> c_int;
}

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

struct Ctx(*mut RnamcCtx);
unsafe impl Send for Ctx {}
unsafe impl Sync for Ctx {} // calls on one ctx are serialised inside librnamc

// The pair-HMM reads none of the folding tables (AlignScores travel with every call), so one
// context per process built on an all-zero parameter block is exact, not a cache of a set.
fn context() -> &'static Ctx {
  static CTX: std::sync::OnceLock<Ctx> = std::sync::OnceLock::new();
  CTX.get_or_init(|| unsafe {
    let mut params = vec![0u8; rnamc_params_sizeof()];
    check(rnamc_params_new(0., params.as_mut_ptr() as *mut c_void), "rnamc_params_new");
    let mut ctx = std::ptr::null_mut();
    check(rnamc_ctx_create(params.as_ptr() as *const c_void, -1, 0, &mut ctx), "rnamc_ctx_create");
    Ctx(ctx)
  })
}

fn pack_scores(s: &AlignScores) -> AlignScoresC {
  AlignScoresC {
    match2match_score: s.match2match_score,
    match2insert_score: s.match2insert_score,
    insert_extend_score: s.insert_extend_score,
    insert_switch_score: s.insert_switch_score,
    init_match_score: s.init_match_score,
    init_insert_score: s.init_insert_score,
    insert_scores: s.insert_scores,
    match_scores: s.match_scores,
  }
}

// All pairs of a FASTA in one device batch: what src/bin/durbin_algo.rs:55-75 does with one pool
// task per pair.  `pairs` index into `seqs`.
pub fn durbin_algo_batch(seqs: &[SeqSlice], pairs: &[(usize, usize)], align_scores: &AlignScores) -> Vec<ProbMat> {
  let mut bases = Vec::<u8>::new();
  let mut offsets = vec![0u64];
  for seq in seqs {
    bases.extend(seq.iter().map(|&x| x as u8));
    offsets.push(bases.len() as u64);
  }
  let pair_a: Vec<u32> = pairs.iter().map(|p| p.0 as u32).collect();
  let pair_b: Vec<u32> = pairs.iter().map(|p| p.1 as u32).collect();
  let mut out_offsets = vec![0u64];
  for p in pairs {
    out_offsets.push(out_offsets.last().unwrap() + (seqs[p.0].len() * seqs[p.1].len()) as u64);
  }
  let mut flat = vec![0f32; (*out_offsets.last().unwrap() as usize).max(1)];
  let scores = pack_scores(align_scores);
  check(
    unsafe {
      rnamc_durbin_batch(context().0, &scores, seqs.len() as u32, bases.as_ptr(), offsets.as_ptr(),
        pairs.len() as u32, pair_a.as_ptr(), pair_b.as_ptr(), flat.as_mut_ptr(), out_offsets.as_ptr())
    },
    "rnamc_durbin_batch",
  );
  pairs
    .iter()
    .enumerate()
    .map(|(x, p)| {
      let (n1, n2) = (seqs[p.0].len(), seqs[p.1].len());
      let base = out_offsets[x] as usize;
      (0..n1).map(|i| flat[base + i * n2..base + (i + 1) * n2].to_vec()).collect()
    })
    .collect()
}

pub fn durbin_algo(seq_pair: &SeqPair, align_scores: &AlignScores) -> ProbMat {
  durbin_algo_batch(&[seq_pair.0, seq_pair.1], &[(0, 1)], align_scores).pop().unwrap()
}
