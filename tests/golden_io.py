"""Reader / writer of the golden-vector container `RNAMCGLD` that
bindings/rust/dump_tables.rs writes from the reference CPU path (and that
tests/make_golden.py writes from the oracle under synthetic tables, so that the reader is
exercised in this image too).  Layout: magic "RNAMCGLD", u32 version = 1, u32 n_records; per
record {n, uses_contra, allows_short, 0} as u32, n base codes padded to 4 bytes, then the
packed diagonal-major triangle of n(n+1)/2 f32 (-1.0 = pair absent)."""
import numpy as np

MAGIC = b"RNAMCGLD"


def write(path, records):
    """records: iterable of (seq codes u8[n], contra, short, packed f32[n(n+1)/2])"""
    records = list(records)
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(np.array([1, len(records)], dtype="<u4").tobytes())
        for seq, contra, short, packed in records:
            n = len(seq)
            assert packed.shape[0] == n * (n + 1) // 2
            fh.write(np.array([n, int(contra), int(short), 0], dtype="<u4").tobytes())
            fh.write(np.asarray(seq, dtype=np.uint8).tobytes())
            fh.write(b"\0" * (-n % 4))
            fh.write(np.asarray(packed, dtype="<f4").tobytes())


def read(path):
    raw = open(path, "rb").read()
    if raw[:8] != MAGIC:
        raise ValueError(f"{path}: not an RNAMCGLD file")
    version, count = np.frombuffer(raw, dtype="<u4", count=2, offset=8)
    if version != 1:
        raise ValueError(f"{path}: unknown RNAMCGLD version {version}")
    off = 16
    out = []
    for _ in range(int(count)):
        n, contra, short, _ = (int(x) for x in np.frombuffer(raw, dtype="<u4", count=4, offset=off))
        off += 16
        seq = np.frombuffer(raw, dtype=np.uint8, count=n, offset=off).copy()
        off += n + (-n % 4)
        m = n * (n + 1) // 2
        packed = np.frombuffer(raw, dtype="<f4", count=m, offset=off).copy()
        off += 4 * m
        out.append((seq, bool(contra), bool(short), packed))
    if off != len(raw):
        raise ValueError(f"{path}: {len(raw) - off} trailing bytes")
    return out
