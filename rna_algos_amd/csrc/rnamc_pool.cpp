// rnamc_pool.cpp — the batch entry over SEVERAL devices behind the C ABI.
//
// The reference fans a FASTA out over all cores inside the binary, one pool task per record
// (src/bin/mccaskill_algo.rs:58-93, src/bin/centroid_fold.rs:119-132).  The drop-in's
// equivalent is one device context + one host thread per GPU: the batch is cut into as many
// shards as there are contexts, every shard runs through rnamc_bpp_batch of its own context
// (H2D, sweeps, D2H on that device's streams) and writes straight into the caller's host
// triangles.  Sequences are independent and never split: there is no
// collective and no device-to-device traffic.
//
// Shards are contiguous bands of the cost-sorted batch with equal total cost under the
// measured model of one sweep (a * n(n^2-1)/6 + b * n^2: the Theta(n^3) folds plus the
// Theta(496 n^2) 2-loop blocks), so every device's lock-step groups hold
// sequences of similar length and stay large on every diagonal.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "rnamc_internal.h"

struct rnamc_pool {
  std::vector<rnamc_ctx*> ctxs;
  std::vector<int> devices;
  std::mutex mu;  // one batch at a time per pool
};

namespace {

double sweep_cost(uint64_t n) {
  const double x = static_cast<double>(n);
  return RNAMC_COST_S_PER_CELL_K * (x * (x * x - 1.0) / 6.0) + RNAMC_COST_S_PER_N2 * x * x;
}

// shard of every sequence: bands of the cost-sorted order (longest first, stable) with equal
// total cost; cut k lies at the first position whose running cost reaches k / n_shards of the
// total (the position itself goes to the next band, as numpy.searchsorted(side="left") cuts)
void plan(uint32_t n_seqs, const uint64_t* offsets, uint32_t n_shards, uint32_t* shard_of_seq,
          std::vector<uint32_t>* order_out) {
  n_shards = std::min(n_shards, std::max(n_seqs, 1u));  // fewer sequences than shards: one each
  std::vector<uint32_t> order(n_seqs);
  std::iota(order.begin(), order.end(), 0u);
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    return (offsets[a + 1] - offsets[a]) > (offsets[b + 1] - offsets[b]);
  });
  std::vector<double> csum(n_seqs);
  double run = 0.0;
  for (uint32_t x = 0; x < n_seqs; x++) {
    run += sweep_cost(offsets[order[x] + 1] - offsets[order[x]]);
    csum[x] = run;
  }
  std::vector<uint32_t> cuts;
  for (uint32_t k = 1; k < n_shards; k++) {
    uint32_t cut = static_cast<uint32_t>(
        std::lower_bound(csum.begin(), csum.end(), run * static_cast<double>(k) / n_shards) - csum.begin());
    // no empty band while there are sequences to give (a few sequences of very different cost:
    // one per device beats two idle devices)
    if (n_seqs >= n_shards) {
      const uint32_t lo = (cuts.empty() ? 0u : cuts.back()) + 1u, hi = n_seqs - (n_shards - k);
      cut = std::min(std::max(cut, lo), hi);
    }
    cuts.push_back(cut);
  }
  uint32_t shard = 0;
  for (uint32_t x = 0; x < n_seqs; x++) {
    while (shard + 1 < n_shards && cuts[shard] <= x) shard++;  // past every cut at or before x
    shard_of_seq[order[x]] = shard;
  }
  if (order_out) *order_out = std::move(order);
}

}  // namespace

extern "C" {

int rnamc_sweep_cost(uint32_t count, const uint64_t* lengths, double* cost_s) {
  if (count && (!lengths || !cost_s)) return RNAMC_ERR_INVALID_ARG;
  for (uint32_t x = 0; x < count; x++) cost_s[x] = sweep_cost(lengths[x]);
  return RNAMC_OK;
}

int rnamc_shard_plan(uint32_t n_seqs, const uint64_t* offsets, uint32_t n_shards,
                     uint32_t* shard_of_seq) {
  if (!offsets || !shard_of_seq || n_shards == 0) return RNAMC_ERR_INVALID_ARG;
  for (uint32_t s = 0; s < n_seqs; s++)
    if (offsets[s + 1] < offsets[s]) return RNAMC_ERR_INVALID_ARG;
  if (n_seqs) plan(n_seqs, offsets, n_shards, shard_of_seq, nullptr);
  return RNAMC_OK;
}

int rnamc_pool_create(const rnamc_params* params, const int* devices, uint32_t n_devices,
                      uint64_t workspace_bytes, rnamc_pool** out) {
  if (!params || !out || (n_devices && !devices)) return RNAMC_ERR_INVALID_ARG;
  *out = nullptr;
  std::vector<int> devs;
  if (n_devices == 0) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
      rnamc::set_last_error("no HIP device visible: librnamc has no CPU fallback");
      return RNAMC_ERR_NO_DEVICE;
    }
    for (int d = 0; d < count; d++) devs.push_back(d);
  } else {
    devs.assign(devices, devices + n_devices);
  }
  rnamc_pool* p = new (std::nothrow) rnamc_pool();
  if (!p) return RNAMC_ERR_OOM;
  for (int d : devs) {
    rnamc_ctx* c = nullptr;
    const int rc = rnamc_ctx_create(params, d, workspace_bytes, &c);
    if (rc) {
      rnamc_pool_destroy(p);
      return rc;
    }
    p->ctxs.push_back(c);
    p->devices.push_back(d);
  }
  *out = p;
  return RNAMC_OK;
}

void rnamc_pool_destroy(rnamc_pool* p) {
  if (!p) return;
  for (rnamc_ctx* c : p->ctxs) rnamc_ctx_destroy(c);
  delete p;
}

uint32_t rnamc_pool_size(const rnamc_pool* p) { return p ? static_cast<uint32_t>(p->ctxs.size()) : 0u; }

rnamc_ctx* rnamc_pool_ctx(rnamc_pool* p, uint32_t idx) {
  return (p && idx < p->ctxs.size()) ? p->ctxs[idx] : nullptr;
}

int rnamc_pool_set_params(rnamc_pool* p, const rnamc_params* params) {
  if (!p || !params) return RNAMC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(p->mu);
  for (rnamc_ctx* c : p->ctxs)
    if (int rc = rnamc_ctx_set_params(c, params)) return rc;
  return RNAMC_OK;
}

int rnamc_pool_set(rnamc_pool* p, const char* name, int64_t value) {
  if (!p || !name) return RNAMC_ERR_INVALID_ARG;
  std::lock_guard<std::mutex> lock(p->mu);
  for (rnamc_ctx* c : p->ctxs)
    if (int rc = rnamc_ctx_set(c, name, value)) return rc;
  return RNAMC_OK;
}

int rnamc_bpp_batch_multi(rnamc_pool* p, uint32_t n_seqs, const uint8_t* bases,
                          const uint64_t* offsets, int uses_contra_model,
                          int allows_short_hairpins, float* bpp, const uint64_t* out_offsets,
                          float* log_partition) {
  if (!p || p->ctxs.empty() || !offsets || !out_offsets || (n_seqs && (!bases || !bpp)))
    return RNAMC_ERR_INVALID_ARG;
  if (n_seqs == 0) return RNAMC_OK;
  // the whole batch is validated before any device sees a part of it: one bad record fails the
  // call with its status and nothing is computed (the reference panics before its pool starts
  // for bad bytes, src/bin/mccaskill_algo.rs:50-56)
  for (uint32_t s = 0; s < n_seqs; s++) {
    if (offsets[s + 1] < offsets[s]) return RNAMC_ERR_INVALID_ARG;
    const uint64_t n = offsets[s + 1] - offsets[s];
    if (n == 0) return RNAMC_ERR_EMPTY_SEQ;
    if (n > RNAMC_MAX_SEQ_LEN) return RNAMC_ERR_SEQ_TOO_LONG;
    for (uint64_t x = offsets[s]; x < offsets[s + 1]; x++)
      if (bases[x] > 3) return RNAMC_ERR_INVALID_BASE;
  }
  std::lock_guard<std::mutex> lock(p->mu);
  const uint32_t n_shards = static_cast<uint32_t>(std::min<size_t>(p->ctxs.size(), n_seqs));
  std::vector<uint32_t> shard_of(n_seqs), order;
  plan(n_seqs, offsets, n_shards, shard_of.data(), &order);
  struct Shard {
    std::vector<uint32_t> members;  // batch indices, longest first
    std::vector<uint8_t> bases;
    std::vector<uint64_t> offsets, out_offsets;
    std::vector<float> logz;
    int status = RNAMC_OK;
    std::string error;
  };
  std::vector<Shard> shards(n_shards);
  for (uint32_t x = 0; x < n_seqs; x++) shards[shard_of[order[x]]].members.push_back(order[x]);
  for (Shard& sh : shards) {
    uint64_t total = 0;
    for (uint32_t s : sh.members) total += offsets[s + 1] - offsets[s];
    sh.bases.resize(total);
    sh.offsets.assign(1, 0);
    for (uint32_t s : sh.members) {
      const uint64_t n = offsets[s + 1] - offsets[s];
      std::memcpy(sh.bases.data() + sh.offsets.back(), bases + offsets[s], n);
      sh.offsets.push_back(sh.offsets.back() + n);
      sh.out_offsets.push_back(out_offsets[s]);  // results go straight into the caller's triangles
    }
    sh.out_offsets.push_back(0);  // (entry n of the prefix form: unused by rnamc_bpp_batch)
    sh.logz.resize(sh.members.size());
  }
  auto work = [&](uint32_t k) {
    Shard& sh = shards[k];
    if (sh.members.empty()) return;
    sh.status = rnamc_bpp_batch(p->ctxs[k], static_cast<uint32_t>(sh.members.size()), sh.bases.data(),
                                sh.offsets.data(), uses_contra_model, allows_short_hairpins, bpp,
                                sh.out_offsets.data(), log_partition ? sh.logz.data() : nullptr);
    if (sh.status) sh.error = rnamc_last_error();  // (thread-local: carry it to the caller's thread)
  };
  std::vector<std::thread> pool;
  for (uint32_t k = 1; k < n_shards; k++) {
    try {
      pool.emplace_back(work, k);
    } catch (...) {  // no thread: this shard runs on the caller's thread below
      work(k);
    }
  }
  work(0);
  for (std::thread& t : pool) t.join();
  for (const Shard& sh : shards)
    if (sh.status) {
      rnamc::set_last_error(sh.error);
      return sh.status;
    }
  if (log_partition)
    for (const Shard& sh : shards)
      for (size_t x = 0; x < sh.members.size(); x++) log_partition[sh.members[x]] = sh.logz[x];
  return RNAMC_OK;
}

}  // extern "C"
