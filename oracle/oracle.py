"""ctypes wrapper around oracle/liboracle.so (the CPU restatement in deacon_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  See the header of deacon_oracle.c for the
parity status ("parity unpinned" at value level; XXH3 pinned against the C xxHash).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "deacon_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [
        ("k", C.c_uint32),
        ("w", C.c_uint32),
        ("abs_threshold", C.c_uint64),
        ("rel_threshold", C.c_double),
        ("prefix_length", C.c_uint64),
        ("deplete", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.dor_xxh3_64_u64.restype = C.c_uint64
    L.dor_xxh3_64_u64.argtypes = [C.c_uint64]
    L.dor_xxh3_64_u128.restype = C.c_uint64
    L.dor_xxh3_64_u128.argtypes = [C.c_uint64, C.c_uint64]
    for name in ("dor_canonical_minimizer_positions", "dor_canonical_minimizer_positions_naive"):
        f = getattr(L, name)
        f.restype = C.c_int64
        f.argtypes = [u8p, C.c_uint64, C.c_uint32, C.c_uint32, u32p, C.c_uint64]
    L.dor_kmer_hash.restype = C.c_uint64
    L.dor_kmer_hash.argtypes = [u8p, C.c_uint32]
    L.dor_minimizer_hashes_and_positions.restype = C.c_int64
    L.dor_minimizer_hashes_and_positions.argtypes = [
        u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, u32p, C.c_uint64]
    L.dor_canonicalise_nucleotide.restype = C.c_uint8
    L.dor_canonicalise_nucleotide.argtypes = [C.c_uint8]
    L.dor_scaled_entropy.restype = C.c_float
    L.dor_scaled_entropy.argtypes = [u8p, C.c_uint32]
    L.dor_index_minimizer_hashes.restype = C.c_int64
    L.dor_index_minimizer_hashes.argtypes = [
        u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, u64p, C.c_uint64]
    L.dor_required_hits.restype = C.c_uint64
    L.dor_required_hits.argtypes = [C.c_uint64, C.c_double, C.c_uint64]
    L.dor_meets_filtering_criteria.restype = C.c_int
    L.dor_meets_filtering_criteria.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_int]
    L.dor_set_new.restype = C.c_void_p
    L.dor_set_new.argtypes = [C.c_uint64]
    L.dor_set_free.restype = None
    L.dor_set_free.argtypes = [C.c_void_p]
    L.dor_set_insert.restype = C.c_int
    L.dor_set_insert.argtypes = [C.c_void_p, C.c_uint64]
    L.dor_set_contains.restype = C.c_int
    L.dor_set_contains.argtypes = [C.c_void_p, C.c_uint64]
    L.dor_set_len.restype = C.c_uint64
    L.dor_set_len.argtypes = [C.c_void_p]
    L.dor_set_insert_many.restype = C.c_int
    L.dor_set_insert_many.argtypes = [C.c_void_p, u64p, C.c_uint64]
    L.dor_set_insert_many_mt.restype = C.c_int
    L.dor_set_insert_many_mt.argtypes = [C.c_void_p, u64p, C.c_uint64, C.c_int]
    L.dor_set_dump.restype = C.c_uint64
    L.dor_set_dump.argtypes = [C.c_void_p, u64p, C.c_uint64]
    L.dor_count_distinct_hits.restype = C.c_uint64
    L.dor_count_distinct_hits.argtypes = [C.c_void_p, u64p, C.c_uint64]
    L.dor_filter_batch.restype = C.c_int
    L.dor_filter_batch.argtypes = [
        C.c_void_p, u8p, u64p, u32p, C.c_uint64, C.POINTER(Params), u8p, u32p, u32p]
    L.dor_filter_batch_mt.restype = C.c_int
    L.dor_filter_batch_mt.argtypes = [
        C.c_void_p, u8p, u64p, u32p, C.c_uint64, C.POINTER(Params), u8p, u32p, u32p, C.c_int]
    L.dor_filter_batch_tuned_mt.restype = C.c_int
    L.dor_filter_batch_tuned_mt.argtypes = [
        C.c_void_p, u8p, u64p, u32p, C.c_uint64, C.POINTER(Params), u8p, u32p, u32p, C.c_int]
    L.dor_set_variant.restype = C.c_int
    L.dor_set_variant.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.dor_should_keep_hashes.restype = C.c_int
    L.dor_should_keep_hashes.argtypes = [
        C.c_void_p, u64p, u64p, C.c_uint64, C.c_uint64, C.c_double, C.c_int, u8p, u32p, u32p]
    L.dor_index_read_header.restype = C.c_int
    L.dor_index_read_header.argtypes = [C.c_char_p, u8p, u8p, u64p]
    L.dor_index_read_keys.restype = C.c_int64
    L.dor_index_read_keys.argtypes = [C.c_char_p, u64p, C.c_uint64]
    L.dor_index_write.restype = C.c_int
    L.dor_index_write.argtypes = [C.c_char_p, C.c_uint8, C.c_uint8, u64p, C.c_uint64]
    L.dor_index_add_sequence.restype = C.c_int
    L.dor_index_add_sequence.argtypes = [C.c_void_p, u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float]
    _lib = L
    return L


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _bytes_arr(seq):
    if isinstance(seq, np.ndarray):
        return np.ascontiguousarray(seq, dtype=np.uint8)
    return np.frombuffer(bytes(seq), dtype=np.uint8) if len(seq) else np.zeros(0, np.uint8)


VARIANTS = [(rot, bits, comb) for rot in (1, 7) for bits in (16, 32) for comb in ("add", "xor")]
DEFAULT_VARIANT = (1, 16, "add")


def set_variant(nt_rot=1, cmp_bits=16, combine="add"):
    """The three details of A4 the reference's tests cannot separate (deacon_oracle.c, dor_set_variant)."""
    if lib().dor_set_variant(nt_rot, cmp_bits, {"add": 0, "xor": 1}[combine]) != 0:
        raise ValueError("bad minimizer variant")


def xxh3_64_u64(v):
    return int(lib().dor_xxh3_64_u64(C.c_uint64(v & (2**64 - 1))))


def xxh3_64_u128(v):
    return int(lib().dor_xxh3_64_u128(C.c_uint64(v & (2**64 - 1)), C.c_uint64(v >> 64)))


def codes_of(seq):
    return (_bytes_arr(seq) >> 1) & 3


def canonical_minimizer_positions(seq, k, w, naive=False):
    """simd-minimizers `canonical_minimizer_positions` restated (A2+A4) on an ASCII sequence."""
    codes = np.ascontiguousarray(codes_of(seq), dtype=np.uint8)
    n = len(codes)
    out = np.zeros(max(n, 1), np.uint32)
    f = lib().dor_canonical_minimizer_positions_naive if naive else lib().dor_canonical_minimizer_positions
    c = f(_p(codes, C.c_uint8), n, k, w, _p(out, C.c_uint32), len(out))
    if c < 0:
        raise ValueError(f"oracle error {c}")
    return out[:c].copy()


def minimizer_hashes_and_positions(seq, k, w, prefix_length=0):
    """src/filter_common.rs:211 get_minimizer_hashes_and_positions -> (hashes u64[], positions u32[])."""
    s = _bytes_arr(seq)
    cap = max(len(s), 1)
    h = np.zeros(cap, np.uint64)
    p = np.zeros(cap, np.uint32)
    c = lib().dor_minimizer_hashes_and_positions(
        _p(s, C.c_uint8), len(s), prefix_length, k, w, _p(h, C.c_uint64), _p(p, C.c_uint32), cap)
    if c < 0:
        raise ValueError(f"oracle error {c}")
    return h[:c].copy(), p[:c].copy()


def index_minimizer_hashes(seq, k, w, entropy_threshold=0.0):
    """src/minimizers.rs:53 compute_minimizer_hashes (index-side variant)."""
    s = _bytes_arr(seq)
    cap = max(len(s), 1)
    h = np.zeros(cap, np.uint64)
    c = lib().dor_index_minimizer_hashes(
        _p(s, C.c_uint8), len(s), k, w, C.c_float(entropy_threshold), _p(h, C.c_uint64), cap)
    if c < 0:
        raise ValueError(f"oracle error {c}")
    return h[:c].copy()


def scaled_entropy(kmer, k):
    s = _bytes_arr(kmer)
    return float(lib().dor_scaled_entropy(_p(s, C.c_uint8), k))


def canonicalise_nucleotide(c):
    return int(lib().dor_canonicalise_nucleotide(c))


def required_hits(abs_threshold, rel_threshold, total):
    return int(lib().dor_required_hits(abs_threshold, C.c_double(rel_threshold), total))


def meets_filtering_criteria(hits, total, abs_threshold, rel_threshold, deplete):
    return bool(lib().dor_meets_filtering_criteria(hits, total, abs_threshold,
                                                   C.c_double(rel_threshold), int(deplete)))


class Index:
    """Stand-in for the reference's FxHashSet<u64> index (membership only)."""

    def __init__(self, keys=(), k=31, w=15, threads=1):
        self.k, self.w = k, w
        keys = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64))
        self._h = lib().dor_set_new(max(len(keys), 8))
        if not self._h:
            raise MemoryError
        if len(keys):
            if threads > 1:
                rc = lib().dor_set_insert_many_mt(self._h, _p(keys, C.c_uint64), len(keys), threads)
            else:
                rc = lib().dor_set_insert_many(self._h, _p(keys, C.c_uint64), len(keys))
            if rc != 0:
                raise MemoryError(f"oracle set insert error {rc}")

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().dor_set_free(self._h)
            except TypeError:  # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    def __len__(self):
        return int(lib().dor_set_len(self._h))

    def __contains__(self, key):
        return bool(lib().dor_set_contains(self._h, C.c_uint64(int(key))))

    def add_sequence(self, seq, entropy_threshold=0.0):
        s = _bytes_arr(seq)
        rc = lib().dor_index_add_sequence(self._h, _p(s, C.c_uint8), len(s), self.k, self.w,
                                          C.c_float(entropy_threshold))
        if rc != 0:
            raise ValueError(f"oracle error {rc}")

    def keys(self):
        n = len(self)
        out = np.zeros(max(n, 1), np.uint64)
        c = lib().dor_set_dump(self._h, _p(out, C.c_uint64), len(out))
        return out[:c].copy()

    @classmethod
    def build(cls, seqs, k=31, w=15, entropy_threshold=0.0):
        """src/index.rs:167 build (minus FASTX parsing)."""
        if (k + w - 1) % 2 == 0:
            raise ValueError("Constraint violated: k + w - 1 must be odd")
        idx = cls((), k, w)
        for s in seqs:
            idx.add_sequence(s, entropy_threshold)
        return idx

    def write(self, path):
        keys = self.keys()
        rc = lib().dor_index_write(os.fsencode(path), self.k, self.w, _p(keys, C.c_uint64), len(keys))
        if rc != 0:
            raise OSError(f"oracle index write error {rc}")

    @classmethod
    def read(cls, path):
        k, w, n = C.c_uint8(), C.c_uint8(), C.c_uint64()
        rc = lib().dor_index_read_header(os.fsencode(path), C.byref(k), C.byref(w), C.byref(n))
        if rc != 0:
            raise OSError(f"oracle index header error {rc}")
        keys = np.zeros(max(n.value, 1), np.uint64)
        c = lib().dor_index_read_keys(os.fsencode(path), _p(keys, C.c_uint64), len(keys))
        if c < 0:
            raise OSError(f"oracle index read error {c}")
        return cls(keys[:c], k.value, w.value)


def concat_reads(reads):
    """list of bytes -> (bases u8[], offsets u64[n+1])."""
    lens = np.fromiter((len(r) for r in reads), dtype=np.uint64, count=len(reads))
    offsets = np.zeros(len(reads) + 1, np.uint64)
    np.cumsum(lens, out=offsets[1:])
    bases = np.frombuffer(b"".join(bytes(r) for r in reads), dtype=np.uint8).copy() if len(reads) else np.zeros(0, np.uint8)
    return bases, offsets


def filter_batch(index, bases, offsets, unit_id=None, abs_threshold=2, rel_threshold=0.01,
                 prefix_length=0, deplete=False, threads=1, tuned=False):
    """Per-unit (keep, hits, total): A1-A8 end to end on the CPU.  tuned: the "port-tuned" form of the same
    arithmetic (dor_filter_batch_tuned_mt), default A4 rules only."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n_reads = len(offsets) - 1
    if unit_id is not None:
        unit_id = np.ascontiguousarray(unit_id, dtype=np.uint32)
        n_units = int(unit_id[-1]) + 1 if n_reads else 0
    else:
        n_units = n_reads
    keep = np.zeros(max(n_units, 1), np.uint8)
    hits = np.zeros(max(n_units, 1), np.uint32)
    total = np.zeros(max(n_units, 1), np.uint32)
    p = Params(index.k, index.w, abs_threshold, rel_threshold, prefix_length, int(deplete))
    if len(bases) == 0:
        bases = np.zeros(1, np.uint8)
    uid = _p(unit_id, C.c_uint32) if unit_id is not None else None
    if tuned:
        rc = lib().dor_filter_batch_tuned_mt(index._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), uid,
                                             n_reads, C.byref(p), _p(keep, C.c_uint8), _p(hits, C.c_uint32),
                                             _p(total, C.c_uint32), threads)
    elif threads > 1:
        rc = lib().dor_filter_batch_mt(index._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), uid,
                                       n_reads, C.byref(p), _p(keep, C.c_uint8), _p(hits, C.c_uint32),
                                       _p(total, C.c_uint32), threads)
    else:
        rc = lib().dor_filter_batch(index._h, _p(bases, C.c_uint8), _p(offsets, C.c_uint64), uid,
                                    n_reads, C.byref(p), _p(keep, C.c_uint8), _p(hits, C.c_uint32),
                                    _p(total, C.c_uint32))
    if rc != 0:
        raise ValueError(f"oracle error {rc}")
    return keep[:n_units].astype(bool), hits[:n_units].copy(), total[:n_units].copy()


def should_keep_hashes(index, hashes, hash_offsets, abs_threshold=2, rel_threshold=0.01, deplete=False):
    """src/remote_filter.rs:230 unpaired_should_keep / :266 paired_should_keep (hashes precomputed)."""
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    hash_offsets = np.ascontiguousarray(hash_offsets, dtype=np.uint64)
    n = len(hash_offsets) - 1
    keep = np.zeros(max(n, 1), np.uint8)
    hits = np.zeros(max(n, 1), np.uint32)
    total = np.zeros(max(n, 1), np.uint32)
    if len(hashes) == 0:
        hashes = np.zeros(1, np.uint64)
    lib().dor_should_keep_hashes(index._h, _p(hashes, C.c_uint64), _p(hash_offsets, C.c_uint64), n,
                                 abs_threshold, C.c_double(rel_threshold), int(deplete),
                                 _p(keep, C.c_uint8), _p(hits, C.c_uint32), _p(total, C.c_uint32))
    return keep[:n].astype(bool), hits[:n].copy(), total[:n].copy()
