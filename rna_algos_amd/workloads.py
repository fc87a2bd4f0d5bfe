"""Synthetic inputs of BASELINE.json's configs: SplitMix64 streams,
base = top 2 bits of each output (uniform iid over A,C,G,U)."""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def splitmix_u64(seed, count):
    """First `count` outputs of SplitMix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        state = np.uint64(seed) + _GAMMA * np.arange(1, count + 1, dtype=np.uint64)
        return _mix(state)


def synthetic_seq(n, seed):
    return (splitmix_u64(seed, n) >> np.uint64(62)).astype(np.uint8)


def batch_lengths(count=10000, master_seed=10000):
    """C4: length_s = 256 + (u64 mod 1793) in [256, 2048]."""
    return (256 + (splitmix_u64(master_seed, count) % np.uint64(1793))).astype(np.int64)


def batch_seq(s, master_seed=10000, lengths=None):
    """Sequence s of the C4 batch: seeded with master*2^32 + s."""
    n = int(lengths[s]) if lengths is not None else int(batch_lengths(s + 1, master_seed)[s])
    return synthetic_seq(n, (master_seed << 32) + s)


def batch(count=10000, master_seed=10000):
    lens = batch_lengths(count, master_seed)
    return [synthetic_seq(int(lens[s]), (master_seed << 32) + s) for s in range(count)]


def pair_cost(n):
    """T(n) = n(n^2-1)/6: (cell,k) iterations of each Theta(n^3) loop."""
    n = np.asarray(n, dtype=np.float64)
    return n * (n * n - 1.0) / 6.0


def sweep_cost(n):
    """Measured cost model of one sequence on an MI355X, seconds: the Theta(n^3) folds plus the
    Theta(n^2 * 496) 2-loop blocks (fit to a 512-sequence group of n ~ 2000 and a 4000-sequence
    group of n = 256..964).  Used to balance shards.  Evaluated by the
    library (rnamc_sweep_cost; constants RNAMC_COST_* of include/rnamc.h): ONE source for
    rnamc_shard_plan, bench.py's shards and this function."""
    from . import _lib
    a = np.ascontiguousarray(np.atleast_1d(np.asarray(n)), dtype=np.uint64)
    out = np.empty(a.shape[0], dtype=np.float64)
    _lib.check(_lib.lib().rnamc_sweep_cost(a.shape[0], a.ctypes.data, out.ctypes.data))
    return out if np.ndim(n) else float(out[0])


def paired_fraction(seq):
    """f = (#canonical pairs with span >= 5) / (n^2/2), measured on the input."""
    s = np.asarray(seq, dtype=np.int64)
    n = s.shape[0]
    cnt = 0
    for d in range(4, n):
        t = s[:n - d] + s[d:]
        cnt += int(np.count_nonzero((t == 3) | (t == 5)))
    return cnt / (n * n / 2.0)


def algorithmic_bytes(lengths, contra, f=0.375):
    """streamed-operand bytes of the whole job."""
    lengths = np.asarray(lengths, dtype=np.float64)
    T = pair_cost(lengths).sum()
    n2 = (lengths * lengths).sum()
    per_t = 16.0 + (4.0 if contra else 0.0) + 8.0 + 12.0 * f
    return per_t * T + 12.0 * 496.0 * f * n2 / 2.0 + 36.0 * n2
