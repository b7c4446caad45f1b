"""Host-side mirror of the reference's filter interface over the HIP C ABI.

Names, argument meaning and results follow the Rust items they stand for (paths under the reference's src/):

  Index                              <- index::load_minimizer_hashes            (index.rs:80-107)
  FilterProcessor                    <- local_filter::FilterProcessor          (local_filter.rs:153-285)
    .should_keep_sequence(seq)          (local_filter.rs:221-252)  -> (keep, hit_count, num_minimizers)
    .should_keep_pair(seq1, seq2)       (local_filter.rs:254-285)
    .filter_batch(...)                  the whole paraseq per-record loop   (local_filter.rs:346-528)
    .stats()                            ProcessingStats                        (local_filter.rs:179-187)
  get_minimizer_hashes_and_positions <- filter_common.rs:211-310
  unpaired_should_keep / paired_should_keep <- remote_filter.rs:230-301

Everything here is plumbing: all arithmetic happens in lib/libdeacon_hip.so on the GPU.
"""
import ctypes as C
import os

import numpy as np

from . import _native as N
from ._native import DeaconHipError, Params

DEFAULT_KMER_LENGTH = 31  # minimizers.rs:4
DEFAULT_WINDOW_SIZE = 15  # minimizers.rs:5


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _as_u8(seq):
    if isinstance(seq, np.ndarray):
        return np.ascontiguousarray(seq, dtype=np.uint8)
    if isinstance(seq, str):
        seq = seq.encode()
    return np.frombuffer(bytes(seq), dtype=np.uint8)


def concat_reads(reads):
    """list of bytes-like -> (bases u8[], offsets u64[n+1]) in the layout dcn_filter_batch takes."""
    n = len(reads)
    offsets = np.zeros(n + 1, np.uint64)
    if n:
        np.cumsum(np.fromiter((len(r) for r in reads), dtype=np.uint64, count=n), out=offsets[1:])
    joined = b"".join(bytes(r) if not isinstance(r, str) else r.encode() for r in reads)
    bases = np.frombuffer(joined, dtype=np.uint8).copy() if joined else np.zeros(0, np.uint8)
    return bases, offsets


def pack_ascii(bases, allow_newline=False):
    """Concatenated ASCII -> (packed u32[2*ceil(n/32)], invmask u32[ceil(n/32)]) in the layout
    dcn_filter_batch_packed takes (PackedSeqVec::from_ascii + the mask loop, filter_common.rs:238-258).
    A newline byte in the input raises: the ASCII entry points strip one from the end of a read
    (filter_common.rs:229), the packed ones cannot, so such a batch must go through filter_batch / submit."""
    bases = _as_u8(bases)
    g = (len(bases) + 31) // 32
    packed = np.zeros(max(2 * g, 1), np.uint32)
    mask = np.zeros(max(g, 1), np.uint32)
    nl = C.c_uint32()
    N.check(N.lib().dcn_pack_ascii(_ptr(bases) if len(bases) else None, len(bases), _ptr(packed), _ptr(mask), C.byref(nl)))
    if nl.value and not allow_newline:
        raise ValueError("pack_ascii: the batch holds a newline byte; a read ending in one is shortened by the ASCII entry "
                         "points (filter_common.rs:229) but not by the packed ones -- strip line ends or use filter_batch")
    return packed[:2 * g], mask[:g]


def set_minimizer_variant(nt_rot=1, cmp_bits=16, combine="add"):
    """Process-wide parity-pinning switch (include/deacon_hip.h): ntHash rotation per base, hash bits the window
    minimum compares, and how the strands' hashes are combined ("add" | "xor").  (1, 16, "add") are the rules of
    SURVEY.md 8a row A4; set it before any index is built or loaded."""
    N.check(N.lib().dcn_set_minimizer_variant(int(nt_rot), int(cmp_bits), {"add": 0, "xor": 1}[combine]))


def get_minimizer_variant():
    r, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    N.check(N.lib().dcn_get_minimizer_variant(C.byref(r), C.byref(b), C.byref(c)))
    return r.value, b.value, ("add", "xor")[c.value]


class PendingBatch:
    """A batch in flight (dcn_filter_batch_submit): holds the arrays the library reads and writes until wait()."""

    def __init__(self, proc, ticket, n_units, keep, hits, total, inputs):
        self.proc, self.ticket, self.n_units = proc, ticket, n_units
        self.keep, self.hits, self.total, self._inputs = keep, hits, total, inputs

    def wait(self):
        N.check(N.lib().dcn_filter_batch_wait(self.proc._h, self.ticket))
        self._inputs = None
        keep = self.keep[:self.n_units].astype(bool)
        if self.hits is None:
            return keep
        return keep, self.hits[:self.n_units], self.total[:self.n_units]


class PinnedBuffer:
    """Page-locked host memory from dcn_host_alloc, viewed as a numpy array: batches built in one go over PCIe
    without the staging copy.  Keep the object alive while `.array` is in use."""

    def __init__(self, count, dtype=np.uint8):
        dtype = np.dtype(dtype)
        self._p = C.c_void_p()
        self.nbytes = int(count) * dtype.itemsize
        N.check(N.lib().dcn_host_alloc(self.nbytes, C.byref(self._p)))
        raw = (C.c_uint8 * max(self.nbytes, 1)).from_address(self._p.value)
        self.array = np.frombuffer(raw, dtype=dtype, count=int(count))

    def close(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            self.array = None
            N.lib().dcn_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Index:
    """Device-resident minimizer set (stands for Arc<FxHashSet<u64>> + IndexHeader)."""

    def __init__(self, handle, device):
        self._h = handle
        self.device = device
        k, w, n = C.c_uint8(), C.c_uint8(), C.c_uint64()
        N.check(N.lib().dcn_index_header(self._h, C.byref(k), C.byref(w), C.byref(n)))
        self.kmer_length, self.window_size, self.n_keys = k.value, w.value, n.value

    @classmethod
    def from_keys(cls, keys, kmer_length=DEFAULT_KMER_LENGTH, window_size=DEFAULT_WINDOW_SIZE, device=0):
        keys = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64))
        h = C.c_void_p()
        N.check(N.lib().dcn_index_from_keys(_ptr(keys), len(keys), kmer_length, window_size, device, C.byref(h)))
        return cls(h, device)

    @classmethod
    def from_file(cls, path, device=0):
        h = C.c_void_p()
        N.check(N.lib().dcn_index_from_file(os.fsencode(path), device, C.byref(h)))
        return cls(h, device)

    @classmethod
    def build(cls, seqs, kmer_length=DEFAULT_KMER_LENGTH, window_size=DEFAULT_WINDOW_SIZE, entropy_threshold=0.0,
              capacity_keys=0, device=0):
        """index::build (index.rs:167-308) for sequences in memory: index-side minimizers merged on the device."""
        bases, offsets = concat_reads(seqs)
        h = C.c_void_p()
        N.check(N.lib().dcn_index_build(_ptr(bases) if len(bases) else None, _ptr(offsets), len(seqs), kmer_length,
                                        window_size, float(entropy_threshold), int(capacity_keys), device, C.byref(h)))
        return cls(h, device)

    @classmethod
    def union(cls, indexes):
        """index::union (index.rs:563-664): set union; all inputs must share k and w."""
        arr = (C.c_void_p * len(indexes))(*[i._h for i in indexes])
        h = C.c_void_p()
        N.check(N.lib().dcn_index_union(arr, len(indexes), C.byref(h)))
        return cls(h, indexes[0].device)

    def clone(self, device):
        """Replica on another (or the same) device, copied device to device (dcn_index_clone)."""
        h = C.c_void_p()
        N.check(N.lib().dcn_index_clone(self._h, int(device), C.byref(h)))
        return type(self)(h, int(device))

    @property
    def table_bytes(self):
        """device memory of the hash table (dcn_index_memory)"""
        b = C.c_uint64()
        N.check(N.lib().dcn_index_memory(self._h, C.byref(b)))
        return b.value

    def diff(self, other):
        """index::diff (index.rs:421-536): the minimizers of self that are not in other."""
        h = C.c_void_p()
        N.check(N.lib().dcn_index_diff(self._h, other._h, C.byref(h)))
        return type(self)(h, self.device)

    def keys(self):
        """The distinct minimizer hashes (arbitrary order, like iterating the reference's set)."""
        out = np.zeros(max(self.n_keys, 1), np.uint64)
        n = C.c_uint64()
        N.check(N.lib().dcn_index_keys(self._h, _ptr(out), len(out), C.byref(n)))
        return out[:n.value]

    def write(self, path):
        """index::write_minimizers (index.rs:130-164): the reference's index file format."""
        N.check(N.lib().dcn_index_write_file(self._h, os.fsencode(path)))

    def header(self):
        return self.kmer_length, self.window_size, self.n_keys

    def __len__(self):
        return self.n_keys

    def contains(self, keys):
        keys = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64))
        out = np.zeros(len(keys), np.uint8)
        N.check(N.lib().dcn_index_contains(self._h, _ptr(keys), len(keys), _ptr(out)))
        return out.astype(bool)

    def contains_device(self, d_keys, n, d_out, stream=None):
        """Device-resident probe: d_keys / d_out are raw device pointers; asynchronous on `stream`."""
        N.check(N.lib().dcn_index_contains_device(self._h, d_keys, n, d_out, stream))

    def probe_ceiling(self, d_keys=None, n=64_000_000, reps=5):
        """Home-group reads per second this table serves for a key stream (device pointer; None: uniformly random
        groups) with nothing else running -- dcn_index_probe_ceiling."""
        r = C.c_double()
        N.check(N.lib().dcn_index_probe_ceiling(self._h, d_keys, n, reps, C.byref(r)))
        return r.value

    def close(self):
        if getattr(self, "_h", None):
            N.lib().dcn_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FilterProcessor:
    """One pipeline context bound to an index: decides keep/drop for units (reads or pairs)."""

    def __init__(self, index, abs_threshold=2, rel_threshold=0.01, prefix_length=0, deplete=False,
                 max_batch_bases=1 << 26, max_batch_reads=1 << 20):
        self.index = index
        self.abs_threshold = int(abs_threshold)
        self.rel_threshold = float(rel_threshold)
        self.prefix_length = int(prefix_length)
        self.deplete = bool(deplete)
        self.max_batch_bases = int(max_batch_bases)
        self.max_batch_reads = int(max_batch_reads)
        self._h = C.c_void_p()
        N.check(N.lib().dcn_ctx_create(index._h, self.max_batch_bases, self.max_batch_reads, C.byref(self._h)))

    # -- parameters -----------------------------------------------------------------------------------
    def _params(self):
        return Params(self.abs_threshold, self.rel_threshold, self.prefix_length, 1 if self.deplete else 0, 0)

    # -- the batch seam ---------------------------------------------------------------------------------
    def filter_batch(self, bases, offsets, unit_id=None, counts=True):
        """bases: concatenated ASCII; offsets[n_reads+1]; unit_id groups mates -> (keep bool[], hits, total).
        counts=False asks for the decisions only (returns just keep): the kernels may then stop probing a read once
        its decision is fixed, which is what `deacon filter` needs outside --debug."""
        bases = _as_u8(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = len(offsets) - 1
        if unit_id is not None:
            unit_id = np.ascontiguousarray(unit_id, dtype=np.uint32)
            n_units = int(unit_id[-1]) + 1 if n_reads else 0
        else:
            n_units = n_reads
        keep = np.zeros(max(n_units, 1), np.uint8)
        p = self._params()
        if not counts:
            N.check(N.lib().dcn_filter_batch(self._h, _ptr(bases) if len(bases) else None, _ptr(offsets),
                                             _ptr(unit_id), n_reads, C.byref(p), _ptr(keep), None, None))
            return keep[:n_units].astype(bool)
        hits = np.zeros(max(n_units, 1), np.uint32)
        total = np.zeros(max(n_units, 1), np.uint32)
        N.check(N.lib().dcn_filter_batch(self._h, _ptr(bases) if len(bases) else None, _ptr(offsets),
                                         _ptr(unit_id), n_reads, C.byref(p), _ptr(keep), _ptr(hits), _ptr(total)))
        return keep[:n_units].astype(bool), hits[:n_units], total[:n_units]

    def _outputs(self, offsets, unit_id, counts, out):
        n_reads = len(offsets) - 1
        n_units = (int(unit_id[-1]) + 1 if n_reads else 0) if unit_id is not None else n_reads
        if out is not None:
            keep, hits, total = out
        else:
            keep = np.zeros(max(n_units, 1), np.uint8)
            hits = np.zeros(max(n_units, 1), np.uint32) if counts else None
            total = np.zeros(max(n_units, 1), np.uint32) if counts else None
        return n_reads, n_units, keep, hits, total

    def submit(self, bases, offsets, unit_id=None, counts=True, out=None):
        """dcn_filter_batch_submit: returns a PendingBatch; up to two may be in flight per processor.
        out = (keep u8[], hits u32[] | None, total u32[] | None) lets the caller supply (page-locked) result arrays."""
        bases = _as_u8(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if unit_id is not None:
            unit_id = np.ascontiguousarray(unit_id, dtype=np.uint32)
        n_reads, n_units, keep, hits, total = self._outputs(offsets, unit_id, counts, out)
        p = self._params()
        t = C.c_uint64()
        N.check(N.lib().dcn_filter_batch_submit(self._h, _ptr(bases) if len(bases) else None, _ptr(offsets),
                                                _ptr(unit_id), n_reads, C.byref(p), _ptr(keep), _ptr(hits),
                                                _ptr(total), C.byref(t)))
        return PendingBatch(self, t.value, n_units, keep, hits, total, (bases, offsets, unit_id, p))

    def submit_packed(self, packed, invmask, offsets, unit_id=None, counts=True, out=None):
        """dcn_filter_batch_packed_submit: the batch as a 2-bit stream + invalid mask (see pack_ascii)."""
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        invmask = np.ascontiguousarray(invmask, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if unit_id is not None:
            unit_id = np.ascontiguousarray(unit_id, dtype=np.uint32)
        n_reads, n_units, keep, hits, total = self._outputs(offsets, unit_id, counts, out)
        g = (int(offsets[-1]) + 31) // 32 if n_reads else 0
        if len(packed) < 2 * g or len(invmask) < g:
            raise ValueError("packed / invmask must hold whole 32-base groups of the batch")
        p = self._params()
        t = C.c_uint64()
        N.check(N.lib().dcn_filter_batch_packed_submit(self._h, _ptr(packed), _ptr(invmask), _ptr(offsets),
                                                       _ptr(unit_id), n_reads, C.byref(p), _ptr(keep), _ptr(hits),
                                                       _ptr(total), C.byref(t)))
        return PendingBatch(self, t.value, n_units, keep, hits, total, (packed, invmask, offsets, unit_id, p))

    def filter_batch_packed(self, packed, invmask, offsets, unit_id=None, counts=True):
        """dcn_filter_batch_packed (blocking)."""
        return self.submit_packed(packed, invmask, offsets, unit_id, counts).wait()

    def filter_reads(self, reads, paired=False):
        """reads: list of sequences; paired=True: reads 2i and 2i+1 are the mates of pair i."""
        bases, offsets = concat_reads(reads)
        unit_id = (np.arange(len(reads), dtype=np.uint32) // 2) if paired else None
        return self.filter_batch(bases, offsets, unit_id)

    def filter_batch_device(self, d_bases, d_offsets, n_reads, n_bases, d_keep, d_hits=None, d_total=None,
                            d_unit_id=None, n_units=None):
        """Same computation on device-resident inputs; arguments are raw device pointers (ints).  Asynchronous:
        call synchronize() before reading the outputs."""
        p = self._params()
        if n_units is None:
            n_units = n_reads
        N.check(N.lib().dcn_filter_batch_device(self._h, d_bases, d_offsets, d_unit_id, n_reads, n_bases, n_units,
                                                C.byref(p), d_keep, d_hits, d_total))

    def synchronize(self):
        N.check(N.lib().dcn_ctx_synchronize(self._h))

    def reserve_records(self, n_records):
        N.check(N.lib().dcn_ctx_reserve_records(self._h, int(n_records)))

    @property
    def stream(self):
        return N.lib().dcn_ctx_stream(self._h)

    # -- the per-read seam of the reference ----------------------------------------------------------------
    def should_keep_sequence(self, seq):
        keep, hits, total = self.filter_reads([seq])
        return bool(keep[0]), int(hits[0]), int(total[0])

    def should_keep_pair(self, seq1, seq2):
        keep, hits, total = self.filter_reads([seq1, seq2], paired=True)
        return bool(keep[0]), int(hits[0]), int(total[0])

    # -- counters -------------------------------------------------------------------------------------------
    def stats(self):
        c = (C.c_uint64 * N.N_STATS)()
        N.check(N.lib().dcn_ctx_stats(self._h, c))
        return dict(zip(N.STAT_NAMES, (int(x) for x in c)))

    def reset_stats(self):
        N.check(N.lib().dcn_ctx_reset_stats(self._h))

    def set_profiling(self, enable=True):
        """True / 1: every stage; 2: the scan stage only (cheaper: two events per batch); False / 0: off"""
        N.check(N.lib().dcn_ctx_set_profiling(self._h, 2 if enable == 2 else (1 if enable else 0)))

    def profile(self):
        """-> ({stage: accumulated device ms}, batches measured); HIP events on the context's stream."""
        ms = (C.c_double * N.N_STAGES)()
        n = C.c_uint64()
        N.check(N.lib().dcn_ctx_profile(self._h, ms, C.byref(n)))
        return dict(zip(N.STAGE_NAMES, (float(x) for x in ms))), int(n.value)

    def summary(self, elapsed_seconds):
        """The numeric fields of FilterSummary (filter_common.rs:11-38; filled at local_filter.rs:780-821)."""
        s = self.stats()
        seqs_in, bp_in = s["total_seqs"], s["total_bp"]
        seqs_out, bp_out = seqs_in - s["filtered_seqs"], s["output_bp"]

        def prop(a, b):
            return a / b if b else 0.0

        return {
            "k": self.index.kmer_length, "w": self.index.window_size,
            "abs_threshold": self.abs_threshold, "rel_threshold": self.rel_threshold,
            "prefix_length": self.prefix_length, "deplete": self.deplete,
            "seqs_in": seqs_in, "seqs_out": seqs_out, "seqs_out_proportion": prop(seqs_out, seqs_in),
            "seqs_removed": s["filtered_seqs"], "seqs_removed_proportion": prop(s["filtered_seqs"], seqs_in),
            "bp_in": bp_in, "bp_out": bp_out, "bp_out_proportion": prop(bp_out, bp_in),
            "bp_removed": s["filtered_bp"], "bp_removed_proportion": prop(s["filtered_bp"], bp_in),
            "time": elapsed_seconds,
            "seqs_per_second": int(seqs_in / elapsed_seconds) if elapsed_seconds > 0 else 0,
            "bp_per_second": int(bp_in / elapsed_seconds) if elapsed_seconds > 0 else 0,
        }

    # -- minimizers (parity seam) ------------------------------------------------------------------------------
    def minimizer_hashes_batch(self, bases, offsets, prefix_length=None):
        """-> (out_offsets u64[n+1], hashes u64[], positions u32[]) for every read of the batch."""
        bases = _as_u8(bases)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n_reads = len(offsets) - 1
        pl = self.prefix_length if prefix_length is None else int(prefix_length)
        out_off = np.zeros(n_reads + 1, np.uint64)
        cap = max(int(len(bases)), 1)
        hashes = np.zeros(cap, np.uint64)
        pos = np.zeros(cap, np.uint32)
        N.check(N.lib().dcn_minimizer_hashes_batch(self._h, _ptr(bases) if len(bases) else None, _ptr(offsets),
                                                   n_reads, pl, _ptr(out_off), _ptr(hashes), _ptr(pos), cap))
        n = int(out_off[-1])
        return out_off, hashes[:n], pos[:n]

    def should_keep_hashes(self, hashes, hash_offsets):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        hash_offsets = np.ascontiguousarray(hash_offsets, dtype=np.uint64)
        n_units = len(hash_offsets) - 1
        keep = np.zeros(max(n_units, 1), np.uint8)
        hits = np.zeros(max(n_units, 1), np.uint32)
        total = np.zeros(max(n_units, 1), np.uint32)
        p = self._params()
        N.check(N.lib().dcn_should_keep_hashes(self._h, _ptr(hashes) if len(hashes) else None, _ptr(hash_offsets),
                                               n_units, C.byref(p), _ptr(keep), _ptr(hits), _ptr(total)))
        return keep[:n_units].astype(bool), hits[:n_units], total[:n_units]

    def close(self):
        if getattr(self, "_h", None):
            N.lib().dcn_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stats_allreduce(processors):
    """dcn_stats_allreduce: the six counters summed over several processors of this process."""
    arr = (C.c_void_p * len(processors))(*[p._h for p in processors])
    c = (C.c_uint64 * N.N_STATS)()
    N.check(N.lib().dcn_stats_allreduce(arr, len(processors), c))
    return dict(zip(N.STAT_NAMES, (int(x) for x in c)))


def get_minimizer_hashes_and_positions(processor, seq, prefix_length=0):
    """filter_common.rs:211-310 for one read -> (hashes u64[], positions u32[])."""
    bases, offsets = concat_reads([seq])
    _, h, p = processor.minimizer_hashes_batch(bases, offsets, prefix_length)
    return h, p


def _should_keep(processor, per_unit_hashes, abs_threshold, rel_threshold, deplete):
    lens = np.fromiter((len(h) for h in per_unit_hashes), dtype=np.uint64, count=len(per_unit_hashes))
    off = np.zeros(len(per_unit_hashes) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    flat = (np.concatenate([np.asarray(h, dtype=np.uint64) for h in per_unit_hashes])
            if len(per_unit_hashes) and off[-1] else np.zeros(0, np.uint64))
    saved = (processor.abs_threshold, processor.rel_threshold, processor.deplete)
    processor.abs_threshold, processor.rel_threshold, processor.deplete = int(abs_threshold), float(rel_threshold), bool(deplete)
    try:
        keep, hits, total = processor.should_keep_hashes(flat, off)
    finally:
        processor.abs_threshold, processor.rel_threshold, processor.deplete = saved
    return [(bool(k), int(h), int(t)) for k, h, t in zip(keep, hits, total)]


def unpaired_should_keep(processor, input_minimizers, abs_threshold, rel_threshold, deplete):
    """remote_filter.rs:230-264: one Vec<u64> of minimizer hashes per read -> [(keep, hits, total)]."""
    return _should_keep(processor, input_minimizers, abs_threshold, rel_threshold, deplete)


def paired_should_keep(processor, input_minimizers, abs_threshold, rel_threshold, deplete):
    """remote_filter.rs:266-301: one Vec<u64> per pair (both mates' hashes concatenated)."""
    return _should_keep(processor, input_minimizers, abs_threshold, rel_threshold, deplete)
