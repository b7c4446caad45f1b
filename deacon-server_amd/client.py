"""Client side of the reference's server engine (`deacon client`, src/remote_filter.rs): minimizer hashes
are computed locally -- here on the GPU by dcn_minimizer_hashes_batch -- and sent to a server that holds
the index (src/remote_filter.rs:336-357 unpaired, :403-421 paired; wire structs src/server_common.rs:9-58).

k and w come from the server's /index_header (server_common.rs:63-81), exactly as the reference's client
asks for them before it starts reading records.
"""
import argparse
import bz2
import gzip
import io
import json
import lzma
import os
import subprocess
import sys
import time
import urllib.error
import urllib.request

import numpy as np

from .filter import FilterProcessor, Index, concat_reads

VERSION = "0.4.0"          # the tool's version (cli/deacon_hip_cli.cpp)
BATCH_RECORDS = 10000      # records (or pairs) per request, as the reference's client (remote_filter.rs:727, :919, :1150)


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


# ---- `deacon client` (src/main.rs:97-157, src/remote_filter.rs:431-695): the file-to-file command ------------------------
#
#   python -m deacon_server_amd.client <SERVER_ADDRESS> [INPUT] [INPUT2] [-o OUT] [-O OUT2] [-a N] [-r F] [-p N] [-d] [-R]
#                                      [-s SUMMARY] [-t N] [--compression-level N] [--debug] [-q]
#
# Records are read and written on the host in plain Python (a request is JSON over HTTP: the wire, not the parser, sets the
# pace -- the reference says as much of its own client); the minimizers of each batch come from the GPU.

class ClientError(RuntimeError):
    pass


_MAGIC = ((b"\x1f\x8b", "gz"), (b"BZh", "bz2"), (b"\xfd7zXZ\x00", "xz"), (b"\x28\xb5\x2f\xfd", "zst"))


def _open_input(path):
    """Compression is found by content, as the reference's reader does.  zstd is not in this Python's standard library: such
    an input is decoded by the tool's own reader (`deacon-hip cat`, no GPU involved) in a child process."""
    raw = sys.stdin.buffer if path == "-" else open(path, "rb")
    raw = raw if isinstance(raw, io.BufferedReader) else io.BufferedReader(raw)
    head = raw.peek(6)[:6]
    kind = next((k for magic, k in _MAGIC if head.startswith(magic)), None)
    if kind == "gz":
        return io.BufferedReader(gzip.GzipFile(fileobj=raw))
    if kind == "bz2":
        return io.BufferedReader(bz2.BZ2File(raw))
    if kind == "xz":
        return io.BufferedReader(lzma.LZMAFile(raw))
    if kind == "zst":
        tool = _tool()
        if path == "-" or not os.path.exists(tool):
            raise ClientError("zstd input needs a file path and the deacon-hip tool beside this package")
        raw.close()
        return subprocess.Popen([tool, "cat", path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    return raw


def read_fastx(stream):
    """(id, seq, qual | None) per record; the id is the whole header line, sequence lines are joined, CR is dropped --
    what the reference's parser hands to RecordData (remote_filter.rs:31-36)."""
    line = stream.readline()
    while line and not line.strip():
        line = stream.readline()
    n = 0
    while line:
        n += 1
        line = line.rstrip(b"\r\n")
        if line[:1] == b">":
            rid, parts = line[1:], []
            line = stream.readline()
            while line and line[:1] != b">":
                parts.append(line.rstrip(b"\r\n"))
                line = stream.readline()
            yield rid, b"".join(parts), None
        elif line[:1] == b"@":
            rid = line[1:]
            seq = stream.readline().rstrip(b"\r\n")
            plus = stream.readline()
            while plus and plus[:1] != b"+":  # a sequence over several lines
                seq += plus.rstrip(b"\r\n")
                plus = stream.readline()
            if not plus:
                raise ClientError(f"Truncated FASTQ record {n}")
            qual = b""
            while len(qual) < len(seq):
                q = stream.readline()
                if not q:
                    break
                qual += q.rstrip(b"\r\n")
            if len(qual) != len(seq):
                raise ClientError(f"FASTQ sequence and quality lengths differ in record {n}")
            yield rid, seq, qual
            line = stream.readline()
            while line and not line.strip():
                line = stream.readline()
        else:
            raise ClientError(f"Record {n} starts with neither '>' nor '@'")


def _tool():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "deacon-hip")


class _ToolCompressor:
    """A .zst output: this Python has no zstd, so the records go through the tool's own writer (`deacon-hip compress zst LEVEL`,
    a child process that never touches the GPU) into the file."""

    def __init__(self, path, kind, level):
        self._file = open(path, "wb")
        self._proc = subprocess.Popen([_tool(), "compress", kind, str(level)], stdin=subprocess.PIPE, stdout=self._file)

    def write(self, data):
        self._proc.stdin.write(data)

    def flush(self):
        self._proc.stdin.flush()

    def close(self):
        self._proc.stdin.close()
        rc = self._proc.wait()
        self._file.close()
        if rc != 0:
            raise ClientError(f"the compressor of {self._file.name} failed (exit code {rc})")


def _open_output(path, level):
    """get_writer (remote_filter.rs:189-228): the extension chooses the codec, the level is checked against it
    (validate_compression_level, :66-100)."""
    if path == "-":
        return sys.stdout.buffer
    if path.endswith(".gz"):
        if not 1 <= level <= 9:
            raise ClientError(f"Invalid gzip compression level {level}. Must be between 1 and 9.")
        return gzip.open(path, "wb", compresslevel=level)
    if path.endswith(".xz"):
        if not 0 <= level <= 9:
            raise ClientError(f"Invalid xz compression level {level}. Must be between 0 and 9.")
        return lzma.open(path, "wb", preset=level)
    if path.endswith(".zst"):
        if not 1 <= level <= 22:
            raise ClientError(f"Invalid zstd compression level {level}. Must be between 1 and 22.")
        if not os.path.exists(_tool()):
            raise ClientError("zstd output needs the deacon-hip tool beside this package (build it: __graft_entry__.build())")
        return _ToolCompressor(path, "zst", level)
    return open(path, "wb")


def _record_bytes(rid, seq, qual, rename, number):
    """output_fastx_record_from_parts (remote_filter.rs:1402-1443)"""
    name = str(number).encode() if rename else rid
    if qual is None:
        return b">" + name + b"\n" + seq + b"\n"
    return b"@" + name + b"\n" + seq + b"\n+\n" + qual + b"\n"


def _duration(sec):  # Rust's {:.2?} of a Duration
    return f"{sec:.2f}s" if sec >= 1 else f"{sec * 1e3:.2f}ms" if sec >= 1e-3 else f"{sec * 1e6:.2f}µs"


def _batches(it, n):
    batch = []
    for x in it:
        batch.append(x)
        if len(batch) == n:
            yield batch
            batch = []
    if batch:
        yield batch


def _pairs_of_two(r1, r2):
    """both files in step; the shorter one ends the run (remote_filter.rs:929-955: `if let (Some, Some)`)"""
    yield from zip(r1, r2)


def _pairs_interleaved(r):
    n = 0
    for first in r:
        n += 1
        second = next(r, None)
        if second is None:
            raise ClientError(f"Uneven number of interleaved sequence pairs. Found {n} records.")
        n += 1
        yield first, second


def run_client(server_address, input_path="-", input2_path=None, output_path="-", output2_path=None, abs_threshold=2,
               rel_threshold=0.01, prefix_length=0, deplete=False, rename=False, summary_path=None, threads=8,
               compression_level=2, debug=False, quiet=False, device=0, remote_filter=None, log=None):
    """remote_filter::run.  Returns the summary as a dict (FilterSummary, filter_common.rs:11-38).
    `remote_filter`: a ready RemoteFilter (tests put one with a CPU minimizer backend here); by default the GPU one."""
    log = log or sys.stderr
    start = time.perf_counter()
    paired_stdin = input_path == "-" and input2_path == "-"
    input_type = "interleaved" if paired_stdin else "paired" if input2_path is not None else "single"
    options = [f"abs_threshold={abs_threshold}, rel_threshold={rel_threshold}"]
    if prefix_length > 0:
        options.append(f"prefix_length={prefix_length}")
    if rename:
        options.append("rename")
    if threads > 0:
        options.append(f"threads={threads}")
    log.write(f"Deacon v{VERSION}; mode: {'deplete' if deplete else 'search'}; input: {input_type}; options: {', '.join(options)}\n")

    # the reference's client narrows the prefix length to a u8 (`config.prefix_length as u8`, remote_filter.rs:526, :551, :574):
    # -p 300 searches the first 44 bases there.  Same here, said aloud; the summary keeps the number that was given.
    effective_prefix = prefix_length & 0xFF
    if effective_prefix != prefix_length:
        log.write(f"Warning: prefix length {prefix_length} is taken modulo 256 = {effective_prefix}, as the reference's client does\n")
    rf = remote_filter or RemoteFilter(server_address, abs_threshold, rel_threshold, effective_prefix, deplete, debug, device)
    own_rf = remote_filter is None
    try:
        log.write(f"Loaded index (k={rf.kmer_length}, w={rf.window_size}) in {_duration(time.perf_counter() - start)}\n")
        out1 = _open_output(output_path, compression_level)
        out2 = _open_output(output2_path, compression_level) if output2_path is not None and input2_path is not None else None
        total_seqs = filtered_seqs = total_bp = output_bp = filtered_bp = counter = 0
        filtering_start = time.perf_counter()

        if input2_path is None:
            for batch in _batches(read_fastx(_open_input(input_path)), BATCH_RECORDS):
                answers = rf.filter_reads([seq for _, seq, _ in batch])
                for (rid, seq, qual), (keep, hits, total, kmers) in zip(batch, answers):
                    total_seqs += 1
                    total_bp += len(seq)
                    if debug:
                        log.write(f"DEBUG: {rid.decode(errors='replace')} hits={hits}/{total} keep={str(keep).lower()} kmers=[{','.join(kmers)}]\n")
                    if keep:
                        output_bp += len(seq)
                        counter += 1
                        out1.write(_record_bytes(rid, seq, qual, rename, counter))
                    else:
                        filtered_seqs += 1
                        filtered_bp += len(seq)
                out1.flush()
        else:
            if paired_stdin:
                pairs = _pairs_interleaved(read_fastx(_open_input("-")))
            else:
                pairs = _pairs_of_two(read_fastx(_open_input(input_path)), read_fastx(_open_input(input2_path)))
            for batch in _batches(pairs, BATCH_RECORDS):
                reads = [seq for pair in batch for _, seq, _ in pair]
                answers = rf.filter_reads(reads, paired=True)
                for ((id1, s1, q1), (id2, s2, q2)), (keep, hits, total, kmers) in zip(batch, answers):
                    total_seqs += 2
                    total_bp += len(s1) + len(s2)
                    if debug and hits > 0:
                        log.write(f"DEBUG: {id1.decode(errors='replace')}/{id2.decode(errors='replace')} hits={hits}/{total} "
                                  f"keep={str(keep).lower()} kmers=[{','.join(kmers)}]\n")
                    if keep:
                        output_bp += len(s1) + len(s2)
                        counter += 2
                        out1.write(_record_bytes(id1, s1, q1, rename, counter - 1))
                        (out2 or out1).write(_record_bytes(id2, s2, q2, rename, counter))
                    else:
                        filtered_seqs += 2
                        filtered_bp += len(s1) + len(s2)
                out1.flush()
                if out2 is not None:
                    out2.flush()
        for o in (out1, out2):
            if o is not None and o is not sys.stdout.buffer:
                o.close()
            elif o is not None:
                o.flush()

        total_time = time.perf_counter() - start
        seqs_per_sec = total_seqs / total_time
        bp_per_sec = total_bp / total_time
        output_seqs = total_seqs - filtered_seqs
        prop = lambda a, b: a / b if b > 0 else 0.0
        if not quiet:
            log.write(f"Retained {output_seqs}/{total_seqs} sequences ({prop(output_seqs, total_seqs) * 100:.3f}%), "
                      f"{output_bp}/{total_bp} bp ({prop(output_bp, total_bp) * 100:.3f}%)\n")
            log.write(f"Completed in {_duration(total_time)}. Speed: {seqs_per_sec:.0f} seqs/s ({bp_per_sec / 1e6:.1f} Mbp/s)\n")
        summary = {
            "version": f"deacon {VERSION}",
            "index": get_server_index_version(rf.server_address),  # get_summary_index (filter_common.rs:46-75): "address:file@hash"
            "input": input_path, "input2": input2_path, "output": output_path, "output2": output2_path,
            "k": rf.kmer_length, "w": rf.window_size, "abs_threshold": abs_threshold, "rel_threshold": rel_threshold,
            "prefix_length": prefix_length, "deplete": deplete, "rename": rename,
            "seqs_in": total_seqs, "seqs_out": output_seqs, "seqs_out_proportion": prop(output_seqs, total_seqs),
            "seqs_removed": filtered_seqs, "seqs_removed_proportion": prop(filtered_seqs, total_seqs),
            "bp_in": total_bp, "bp_out": output_bp, "bp_out_proportion": prop(output_bp, total_bp),
            "bp_removed": filtered_bp, "bp_removed_proportion": prop(filtered_bp, total_bp),
            "time": total_time, "seqs_per_second": int(seqs_per_sec), "bp_per_second": int(bp_per_sec),
        }
        if summary_path is not None:
            with open(summary_path, "w") as f:
                json.dump(summary, f, indent=2)
            log.write(f'Summary saved to "{summary_path}"\n')
        return summary
    finally:
        if own_rf:
            rf.close()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="deacon-hip client",
                                 description="Alternate version of filter: minimizers are computed here (on the GPU), the index "
                                             "is held by a server (python -m deacon_server_amd.server IDX -p PORT)")
    ap.add_argument("server_address", help="Server address to connect to (including port), e.g. http://127.0.0.1:8888")
    ap.add_argument("input", nargs="?", default="-", help="Optional path to fastx file (or - for stdin)")
    ap.add_argument("input2", nargs="?", default=None, help="Optional path to second paired fastx file (or - for interleaved stdin)")
    ap.add_argument("-o", "--output", default="-", help="Path to output fastx file (or - for stdout; detects .gz, .zst and .xz)")
    ap.add_argument("-O", "--output2", default=None, help="Optional path to second paired output fastx file")
    ap.add_argument("-a", "--abs-threshold", type=int, default=2, help="Minimum absolute number of minimizer hits for a match")
    ap.add_argument("-r", "--rel-threshold", type=float, default=0.01, help="Minimum relative proportion (0.0-1.0) of minimizer hits for a match")
    ap.add_argument("-p", "--prefix-length", type=int, default=0, help="Search only the first N nucleotides per sequence (0 = entire sequence)")
    ap.add_argument("-d", "--deplete", action="store_true", help="Discard matching sequences (invert filtering behaviour)")
    ap.add_argument("-R", "--rename", action="store_true", help="Replace sequence headers with incrementing numbers")
    ap.add_argument("-s", "--summary", default=None, help="Path to JSON summary output file")
    ap.add_argument("-t", "--threads", type=int, default=8, help="Accepted for compatibility (the minimizers are the GPU's)")
    ap.add_argument("--compression-level", type=int, default=2, help="Output compression level (1-9 for gz & xz; 1-22 for zstd)")
    ap.add_argument("--debug", action="store_true", help="Output sequences with minimizer hits to stderr")
    ap.add_argument("-q", "--quiet", action="store_true", help="Suppress progress reporting")
    ap.add_argument("--device", type=int, default=0, help="GPU that computes the minimizers")
    a = ap.parse_args(argv)
    if not 1 <= a.abs_threshold <= 65535:
        ap.error("invalid value for --abs-threshold: must be 1..65535")
    if a.prefix_length < 0:
        ap.error("invalid value for --prefix-length")
    if a.output2 is not None and a.input2 is None:  # src/main.rs:385-389
        sys.stderr.write("Warning: --output2 specified but no second input file provided. --output2 will be ignored.\n")
    try:
        run_client(a.server_address, a.input, a.input2, a.output, a.output2, a.abs_threshold, a.rel_threshold, a.prefix_length,
                   a.deplete, a.rename, a.summary, a.threads, a.compression_level, a.debug, a.quiet, a.device)
    except (ClientError, ServerError, OSError, urllib.error.URLError) as e:
        sys.stderr.write(f"Error: {e}\n")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
