"""Client side of the reference's server engine (`deacon client`, src/remote_filter.rs): minimizer hashes
are computed locally -- here on the GPU by dcn_minimizer_hashes_batch -- and sent to a server that holds
the index (src/remote_filter.rs:336-357 unpaired, :403-421 paired; wire structs src/server_common.rs:9-58).

k and w come from the server's /index_header (server_common.rs:63-81), exactly as the reference's client
asks for them before it starts reading records.
"""
import json
import urllib.error
import urllib.request

import numpy as np

from .filter import FilterProcessor, Index, concat_reads


class ServerError(RuntimeError):
    pass


def _get(url):
    try:
        with urllib.request.urlopen(url) as r:
            return r.read()
    except urllib.error.HTTPError as e:
        raise ServerError(f"Server returned an error: {e.code}") from e


def get_server_index_header(server_address):
    """server_common.rs:63-81 -> {"format_version", "kmer_length", "window_size"}."""
    return json.loads(_get(server_address.rstrip("/") + "/index_header"))


def get_server_index_version(server_address):
    return _get(server_address.rstrip("/") + "/index_version").decode()


def post_filter_request(server_address, paired, units, abs_threshold, rel_threshold, deplete, kmer_length, debug):
    """POST one UnpairedFilterRequest / PairedFilterRequest; returns FilterResponse.should_output as
    [(keep, hit_count, total_minimizers, hit_kmers)]."""
    body = json.dumps({"input": units, "abs_threshold": int(abs_threshold), "rel_threshold": float(rel_threshold),
                       "deplete": bool(deplete), "kmer_length": int(kmer_length), "debug": bool(debug)},
                      separators=(",", ":")).encode()
    route = "/should_output_paired" if paired else "/should_output_unpaired"
    req = urllib.request.Request(server_address.rstrip("/") + route, data=body,
                                 headers={"Content-Type": "application/json"}, method="POST")
    try:
        with urllib.request.urlopen(req) as r:
            out = json.loads(r.read())["should_output"]
    except urllib.error.HTTPError as e:
        raise ServerError(f"Server returned an error: {e.code}") from e
    return [(bool(k), int(h), int(t), list(kmers)) for k, h, t, kmers in out]


def _effective(seq, prefix_length, kmer_length):
    """effective sequence of get_minimizer_hashes_and_positions (filter_common.rs:217-229)"""
    if len(seq) < kmer_length:
        return b""
    if prefix_length > 0 and len(seq) > prefix_length:
        seq = seq[:prefix_length]
    return seq[:-1] if seq.endswith(b"\n") else seq


class RemoteFilter:
    """Stands for the reference's filter loop built with `--features server`: minimizers here, probing there."""

    def __init__(self, server_address, abs_threshold=2, rel_threshold=0.01, prefix_length=0, deplete=False,
                 debug=False, device=0):
        self.server_address = server_address.rstrip("/")
        header = get_server_index_header(self.server_address)
        self.kmer_length, self.window_size = int(header["kmer_length"]), int(header["window_size"])
        self.abs_threshold, self.rel_threshold = int(abs_threshold), float(rel_threshold)
        self.prefix_length, self.deplete, self.debug = int(prefix_length), bool(deplete), bool(debug)
        # the minimizer kernels need only k and w: an empty device set carries them
        self._index = Index.from_keys(np.zeros(0, np.uint64), self.kmer_length, self.window_size, device=device)
        self._processor = FilterProcessor(self._index, prefix_length=self.prefix_length)

    def minimizers(self, reads):
        bases, offsets = concat_reads(reads)
        off, hashes, positions = self._processor.minimizer_hashes_batch(bases, offsets, self.prefix_length)
        return off, hashes, positions

    def filter_reads(self, reads, paired=False):
        """reads: list of bytes; paired=True takes mates interleaved (r1, r2, r1, r2, ...).
        Returns [(keep, hit_count, total_minimizers, hit_kmers)] per read or per pair."""
        reads = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
        if paired and len(reads) % 2:
            raise ValueError("paired input needs an even number of reads")
        if not reads:
            return []
        off, hashes, positions = self.minimizers(reads)
        off = [int(x) for x in off]
        hashes, positions = hashes.tolist(), positions.tolist()
        units = []
        if paired:
            for i in range(0, len(reads), 2):
                lo, hi = off[i], off[i + 2]
                # get_paired_minimizer_hashes_and_positions leaves the sequence list empty
                # (filter_common.rs:326-345: it extends by hashes.len() - positions.len() == 0 copies)
                units.append([hashes[lo:hi], positions[lo:hi], []])
        else:
            for i, r in enumerate(reads):
                lo, hi = off[i], off[i + 1]
                units.append([hashes[lo:hi], positions[lo:hi],
                              list(_effective(r, self.prefix_length, self.kmer_length))])
        return post_filter_request(self.server_address, paired, units, self.abs_threshold, self.rel_threshold,
                                   self.deplete, self.kmer_length, self.debug)

    def close(self):
        self._processor.close()
        self._index.close()
