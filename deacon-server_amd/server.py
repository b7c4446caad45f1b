"""Server surface of the reference over the GPU-resident index.

Routes and JSON shapes are the reference's (src/server.rs:48-58, src/server_common.rs:9-58):

  GET  /                        "Index loaded with N minimizers and header: IndexHeader { .. }"   (server.rs:90-105)
  GET  /index_header            {"format_version":2,"kmer_length":K,"window_size":W}              (server.rs:109-112)
  GET  /index_version           "<index path>@<sha256 of the index file>"                         (server.rs:68-75, 116-119)
  POST /should_output_unpaired  UnpairedFilterRequest -> FilterResponse                           (server.rs:144-164)
  POST /should_output_paired    PairedFilterRequest   -> FilterResponse                           (server.rs:120-142)

  request  {"input": [[hashes u64[], positions u32[], seq], ...], "abs_threshold", "rel_threshold",
            "deplete", "kmer_length", "debug"}        seq = u8[] (unpaired) or [u8[], u8[]] (paired)
  response {"should_output": [[keep, hit_count, total_minimizers, hit_kmers[]], ...]}

The decisions come from dcn_should_keep_hashes (probe + distinct count + threshold on the device table); the
reference takes a global mutex around every request (server.rs:121,145) and so does this: one filter context,
one request on the device at a time.  Only the `debug` k-mer strings are assembled on the host, from the
per-hash membership flags of dcn_index_contains, in the order sequence_matches / pair_matches emit them
(src/filter_common.rs:129-198).
"""
import argparse
import hashlib
import json
import sys
import threading
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

import numpy as np

BODY_LIMIT = 2147483648  # DefaultBodyLimit::max, server.rs:58
INDEX_FORMAT_VERSION = 2  # index.rs:17-31

_REQUEST_FIELDS = ("input", "abs_threshold", "rel_threshold", "deplete", "kmer_length", "debug")


class RequestError(Exception):
    """A request the reference's Json extractor would reject; carries the HTTP status."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def index_version_string(index_path):
    """server.rs:68-75: index path + '@' + sha256 hex digest of the index file's bytes."""
    h = hashlib.sha256()
    with open(index_path, "rb") as f:
        for block in iter(lambda: f.read(1 << 24), b""):
            h.update(block)
    return str(index_path) + "@" + h.hexdigest()


class GpuBackend:
    """The resident device table plus one filter context (what server.rs keeps in its INDEX global)."""

    def __init__(self, index_path, device=0):
        from .filter import FilterProcessor, Index
        self.index = Index.from_file(str(index_path), device=device)
        self.processor = FilterProcessor(self.index)

    def header(self):
        return self.index.kmer_length, self.index.window_size, len(self.index)

    def should_keep(self, flat_hashes, hash_offsets, abs_threshold, rel_threshold, deplete):
        p = self.processor
        p.abs_threshold, p.rel_threshold, p.deplete = int(abs_threshold), float(rel_threshold), bool(deplete)
        return p.should_keep_hashes(flat_hashes, hash_offsets)

    def contains(self, flat_hashes):
        return self.index.contains(flat_hashes)


def _parse_request(body, paired):
    try:
        req = json.loads(body)
    except (ValueError, UnicodeDecodeError) as e:
        raise RequestError(400, f"Failed to parse the request body as JSON: {e}")
    if not isinstance(req, dict):
        raise RequestError(422, "Failed to deserialize the JSON body into the target type: expected a map")
    for field in _REQUEST_FIELDS:
        if field not in req:
            raise RequestError(422, f"Failed to deserialize the JSON body into the target type: missing field `{field}`")
    try:
        abs_threshold = req["abs_threshold"]
        kmer_length = req["kmer_length"]
        rel_threshold = req["rel_threshold"]
        if isinstance(abs_threshold, bool) or not isinstance(abs_threshold, int) or abs_threshold < 0:
            raise ValueError("abs_threshold: expected usize")
        if isinstance(kmer_length, bool) or not isinstance(kmer_length, int) or not 0 <= kmer_length <= 255:
            raise ValueError("kmer_length: expected u8")
        if isinstance(rel_threshold, bool) or not isinstance(rel_threshold, (int, float)):
            raise ValueError("rel_threshold: expected f64")
        if not isinstance(req["deplete"], bool) or not isinstance(req["debug"], bool):
            raise ValueError("deplete/debug: expected bool")
        units = req["input"]
        if not isinstance(units, list):
            raise ValueError("input: expected a sequence")
        hashes, positions, seqs = [], [], []
        for unit in units:
            if not isinstance(unit, list) or len(unit) != 3:
                raise ValueError("input: expected a tuple of size 3")
            hashes.append(np.asarray(unit[0], dtype=np.uint64).reshape(-1))
            positions.append(np.asarray(unit[1], dtype=np.uint32).reshape(-1))
            if paired:
                if not isinstance(unit[2], list) or any(not isinstance(s, list) for s in unit[2]):
                    raise ValueError("input: expected Vec<Vec<u8>>")
                seqs.append([bytes(bytearray(s)) for s in unit[2]])
            else:
                if not isinstance(unit[2], list):
                    raise ValueError("input: expected Vec<u8>")
                seqs.append(bytes(bytearray(unit[2])))
    except (ValueError, TypeError, OverflowError) as e:
        raise RequestError(422, f"Failed to deserialize the JSON body into the target type: {e}")
    return hashes, positions, seqs, abs_threshold, float(rel_threshold), req["deplete"], kmer_length, req["debug"]


def _debug_kmers(hashes, positions, seq, member, kmer_length, paired):
    """hit_kmers of sequence_matches (filter_common.rs:143-153) / pair_matches (:184-196): one string per
    first occurrence of a hash that is in the index, in input order."""
    out = []
    seen = set()
    for i, h in enumerate(hashes.tolist()):
        if not member[i] or h in seen:
            continue
        seen.add(h)
        if i >= len(positions):
            continue
        pos = int(positions[i])
        if paired:
            # pair_matches indexes the sequence list with the minimizer index i (filter_common.rs:187-193)
            if i >= len(seq):
                continue
            s = seq[i]
            if pos + kmer_length > len(s):
                continue
        else:
            s = seq
            if pos + kmer_length > len(s):
                # the reference slices out of bounds and panics here; report it as a rejected request
                raise RequestError(422, "minimizer position outside the sequence")
        out.append(s[pos:pos + kmer_length].decode("utf-8", errors="replace"))
    return out


def should_output(backend, body, paired):
    """unpaired_should_keep / paired_should_keep (remote_filter.rs:230-301) for one request body."""
    hashes, positions, seqs, abs_threshold, rel_threshold, deplete, kmer_length, debug = _parse_request(body, paired)
    n = len(hashes)
    offsets = np.zeros(n + 1, np.uint64)
    if n:
        np.cumsum(np.fromiter((len(h) for h in hashes), dtype=np.uint64, count=n), out=offsets[1:])
    flat = np.concatenate(hashes) if n and offsets[-1] else np.zeros(0, np.uint64)
    if n:
        keep, hits, total = backend.should_keep(flat, offsets, abs_threshold, rel_threshold, deplete)
    else:
        keep, hits, total = [], [], []
    kmers = [[] for _ in range(n)]
    if debug and len(flat):
        member = np.asarray(backend.contains(flat), dtype=bool)
        for u in range(n):
            lo, hi = int(offsets[u]), int(offsets[u + 1])
            kmers[u] = _debug_kmers(hashes[u], positions[u], seqs[u], member[lo:hi], kmer_length, paired)
    return {"should_output": [[bool(keep[u]), int(hits[u]), int(total[u]), kmers[u]] for u in range(n)]}


class _Handler(BaseHTTPRequestHandler):
    protocol_version = "HTTP/1.1"

    def log_message(self, fmt, *args):  # the reference logs connections only under RUST_LOG=trace
        if self.server.app.verbose:
            sys.stderr.write("%s - %s\n" % (self.address_string(), fmt % args))

    def _send(self, status, payload, content_type):
        data = payload if isinstance(payload, bytes) else payload.encode()
        self.send_response(status)
        self.send_header("Content-Type", content_type)
        self.send_header("Content-Length", str(len(data)))
        self.end_headers()
        self.wfile.write(data)

    def _text(self, status, text):
        self._send(status, text, "text/plain; charset=utf-8")

    def do_GET(self):
        app = self.server.app
        path = self.path.split("?", 1)[0]
        if path == "/":
            k, w, n = app.backend.header()
            self._text(200, f"Index loaded with {n} minimizers and header: IndexHeader {{ format_version: "
                            f"{INDEX_FORMAT_VERSION}, kmer_length: {k}, window_size: {w} }}")
        elif path == "/index_header":
            k, w, _ = app.backend.header()
            self._send(200, json.dumps({"format_version": INDEX_FORMAT_VERSION, "kmer_length": k, "window_size": w},
                                       separators=(",", ":")), "application/json")
        elif path == "/index_version":
            self._text(200, app.index_version)
        elif path in ("/should_output_paired", "/should_output_unpaired"):
            self._text(405, "")
        else:
            self._text(404, "")

    def do_POST(self):
        app = self.server.app
        path = self.path.split("?", 1)[0]
        if path in ("/", "/index_header", "/index_version"):
            return self._text(405, "")
        if path not in ("/should_output_paired", "/should_output_unpaired"):
            return self._text(404, "")
        ctype = (self.headers.get("Content-Type") or "").split(";")[0].strip().lower()
        if ctype != "application/json" and not ctype.endswith("+json"):
            return self._text(415, "Expected request with `Content-Type: application/json`")
        try:
            length = int(self.headers.get("Content-Length") or "")
        except ValueError:
            return self._text(411, "")
        if length > BODY_LIMIT:
            return self._text(413, "length limit exceeded")
        body = self.rfile.read(length)
        try:
            with app.lock:  # server.rs:121,145 -- requests are serialised on the index
                result = should_output(app.backend, body, paired=path.endswith("_paired"))
        except RequestError as e:
            return self._text(e.status, str(e))
        except Exception as e:  # device/library failure: report it, keep serving
            return self._text(500, f"{type(e).__name__}: {e}")
        self._send(200, json.dumps(result, separators=(",", ":")), "application/json")


class DeaconServer:
    """run_server (server.rs:36-65): load the index once, then answer requests until shutdown()."""

    def __init__(self, index_path, port, host="0.0.0.0", device=0, backend=None, verbose=False):
        self.index_version = index_version_string(index_path)
        self.backend = backend if backend is not None else GpuBackend(index_path, device)
        self.lock = threading.Lock()
        self.verbose = verbose
        self.httpd = ThreadingHTTPServer((host, port), _Handler)
        self.httpd.daemon_threads = True
        self.httpd.app = self
        self.port = self.httpd.server_address[1]
        self._thread = None

    def serve_forever(self):
        self.httpd.serve_forever()

    def start(self):
        """Serve on a background thread (what the reference's tests do with run_with_server!)."""
        self._thread = threading.Thread(target=self.httpd.serve_forever, daemon=True)
        self._thread.start()
        return self

    def shutdown(self):
        self.httpd.shutdown()
        self.httpd.server_close()
        if self._thread is not None:
            self._thread.join()
            self._thread = None


def main(argv=None):
    ap = argparse.ArgumentParser(prog="deacon-hip server", description="Serve filter decisions from a GPU-resident index")
    ap.add_argument("index", help="path to a minimizer index file")
    ap.add_argument("-p", "--port", type=int, default=8888)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args(argv)
    sys.stderr.write(f"Loading index from: {a.index}\n")
    server = DeaconServer(a.index, a.port, host=a.host, device=a.device, verbose=a.verbose)
    sys.stderr.write("Loaded index!\n")
    try:
        server.serve_forever()
    except KeyboardInterrupt:
        pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
