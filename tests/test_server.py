"""Server surface: routes, wire shapes and decisions of src/server.rs,
src/server_common.rs and the client half of src/remote_filter.rs.

CPU tests drive the HTTP layer with the oracle standing in for the device table (the oracle is only the
checker's backend here, never the product's); GPU tests run the real server on the device-resident index
with the GPU client and compare every answer with the oracle."""
import hashlib
import json
import urllib.error
import urllib.request

import numpy as np
import pytest

from conftest import mutate, random_reads

K, W = 31, 15


class OracleBackend:
    def __init__(self, O, index):
        self.O, self.index = O, index

    def header(self):
        return self.index.k, self.index.w, len(self.index)

    def should_keep(self, flat, offsets, abs_threshold, rel_threshold, deplete):
        return self.O.should_keep_hashes(self.index, flat, offsets, abs_threshold, rel_threshold, deplete)

    def contains(self, flat):
        return np.array([int(h) in self.index for h in flat], dtype=bool)


def _workload(rng, O, n_reads=40):
    genome = random_reads(rng, 1, 6000, 6000)[0]
    index = O.Index.build([genome], K, W)
    reads = []
    for i in range(n_reads):
        if i % 3 == 0:
            reads.append(random_reads(rng, 1, 80, 200)[0])
        else:
            s = int(rng.integers(0, len(genome) - 200))
            reads.append(mutate(rng, genome[s:s + int(rng.integers(60, 200))], 0.02))
    reads += [b"", b"ACGT", b"A" * 31, b"ACGTN" * 30]
    return genome, index, reads


def _expected_unit(keys, hashes, positions, seq, k, abs_threshold, rel_threshold, deplete, debug, O, paired=False):
    """sequence_matches / pair_matches + meets_filtering_criteria (filter_common.rs:99-198), in plain Python"""
    seen, kmers = set(), []
    for i, h in enumerate(hashes):
        if h in keys and h not in seen:
            seen.add(h)
            if debug and i < len(positions):
                if paired:
                    if i < len(seq) and positions[i] + k <= len(seq[i]):
                        kmers.append(bytes(seq[i][positions[i]:positions[i] + k]).decode())
                else:
                    kmers.append(bytes(seq[positions[i]:positions[i] + k]).decode())
    keep = O.meets_filtering_criteria(len(seen), len(hashes), abs_threshold, rel_threshold, deplete)
    return (bool(keep), len(seen), len(hashes), kmers)


def _units_unpaired(O, reads, prefix_length=0):
    units = []
    for r in reads:
        h, p = O.minimizer_hashes_and_positions(r, K, W, prefix_length)
        eff = b"" if len(r) < K else (r[:prefix_length] if prefix_length and len(r) > prefix_length else r)
        eff = eff[:-1] if eff.endswith(b"\n") else eff
        units.append([[int(x) for x in h], [int(x) for x in p], list(eff)])
    return units


def _http(method, url, body=None, headers=None):
    req = urllib.request.Request(url, data=body, headers=headers or {}, method=method)
    try:
        with urllib.request.urlopen(req) as r:
            return r.status, r.read(), r.headers.get("Content-Type")
    except urllib.error.HTTPError as e:
        return e.code, e.read(), e.headers.get("Content-Type")


@pytest.fixture()
def oracle_server(oracle, tmp_path, dcn):
    from deacon_server_amd import server as S
    rng = np.random.default_rng(71)
    genome, index, reads = _workload(rng, oracle)
    path = tmp_path / "ref.idx"
    index.write(str(path))
    srv = S.DeaconServer(path, 0, host="127.0.0.1", backend=OracleBackend(oracle, index)).start()
    yield srv, f"http://127.0.0.1:{srv.port}", index, reads, path
    srv.shutdown()


def test_get_routes(oracle_server):
    srv, url, index, reads, path = oracle_server
    st, body, ctype = _http("GET", url + "/")
    assert st == 200
    assert body.decode() == (f"Index loaded with {len(index)} minimizers and header: IndexHeader "
                             f"{{ format_version: 2, kmer_length: {K}, window_size: {W} }}")
    st, body, ctype = _http("GET", url + "/index_header")
    assert st == 200 and ctype == "application/json"
    assert json.loads(body) == {"format_version": 2, "kmer_length": K, "window_size": W}
    st, body, _ = _http("GET", url + "/index_version")
    assert st == 200
    assert body.decode() == str(path) + "@" + hashlib.sha256(path.read_bytes()).hexdigest()
    assert _http("GET", url + "/nope")[0] == 404
    assert _http("GET", url + "/should_output_unpaired")[0] == 405
    assert _http("POST", url + "/index_header", b"{}", {"Content-Type": "application/json"})[0] == 405


def test_rejected_requests(oracle_server):
    srv, url, index, reads, path = oracle_server
    route = url + "/should_output_unpaired"
    js = {"Content-Type": "application/json"}
    good = {"input": [], "abs_threshold": 2, "rel_threshold": 0.01, "deplete": False, "kmer_length": K, "debug": False}
    assert _http("POST", route, json.dumps(good).encode(), js)[0] == 200
    assert _http("POST", route, json.dumps(good).encode(), {"Content-Type": "text/plain"})[0] == 415
    assert _http("POST", route, b"{not json", js)[0] == 400
    for field in good:
        bad = {k: v for k, v in good.items() if k != field}
        assert _http("POST", route, json.dumps(bad).encode(), js)[0] == 422, field
    for field, value in (("abs_threshold", -1), ("abs_threshold", 1.5), ("kmer_length", 256), ("deplete", 1),
                         ("input", [[[1], [0]]]), ("input", [[[-1], [0], []]]), ("input", [[["x"], [0], []]])):
        bad = dict(good)
        bad[field] = value
        assert _http("POST", route, json.dumps(bad).encode(), js)[0] == 422, (field, value)


@pytest.mark.parametrize("abs_threshold,rel_threshold,deplete,debug", [
    (2, 0.01, False, False), (1, 0.0, True, False), (2, 0.01, True, True), (1, 0.5, False, True), (3, 1.0, False, False)])
def test_unpaired_decisions_over_http(oracle_server, oracle, abs_threshold, rel_threshold, deplete, debug):
    from deacon_server_amd import client as CL
    srv, url, index, reads, path = oracle_server
    keys = set(int(x) for x in index.keys())
    units = _units_unpaired(oracle, reads)
    got = CL.post_filter_request(url, False, units, abs_threshold, rel_threshold, deplete, K, debug)
    want = [_expected_unit(keys, u[0], u[1], bytes(u[2]), K, abs_threshold, rel_threshold, deplete, debug, oracle)
            for u in units]
    assert got == want
    assert any(g[0] for g in got) and not all(g[0] for g in got)
    if debug:
        assert any(g[3] for g in got)


def test_paired_decisions_over_http(oracle_server, oracle):
    from deacon_server_amd import client as CL
    srv, url, index, reads, path = oracle_server
    keys = set(int(x) for x in index.keys())
    per_read = _units_unpaired(oracle, reads[:40])
    units = []
    for i in range(0, 40, 2):
        a, b = per_read[i], per_read[i + 1]
        # one case keeps per-minimizer sequences so pair_matches' `all_sequences[i]` indexing is exercised
        seqs = [a[2], b[2]] if i == 2 else []
        units.append([a[0] + b[0], a[1] + b[1], seqs])
    for debug in (False, True):
        got = CL.post_filter_request(url, True, units, 2, 0.01, False, K, debug)
        want = [_expected_unit(keys, u[0], u[1], [bytes(s) for s in u[2]], K, 2, 0.01, False, debug, oracle, paired=True)
                for u in units]
        assert got == want


# ---- GPU: the real server on the device-resident table, with the GPU client -------------------------------

@pytest.fixture()
def gpu_server(oracle, tmp_path, dcn):
    from deacon_server_amd import server as S
    rng = np.random.default_rng(72)
    genome, index, reads = _workload(rng, oracle, n_reads=300)
    path = tmp_path / "ref.idx"
    index.write(str(path))
    srv = S.DeaconServer(path, 0, host="127.0.0.1").start()
    yield srv, f"http://127.0.0.1:{srv.port}", index, reads
    srv.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize("abs_threshold,rel_threshold,deplete,prefix_length", [
    (2, 0.01, False, 0), (1, 0.0, True, 0), (2, 0.2, False, 100), (1, 1.0, True, 0)])
def test_gpu_client_server_unpaired(gpu_server, oracle, abs_threshold, rel_threshold, deplete, prefix_length):
    from deacon_server_amd import client as CL
    srv, url, index, reads = gpu_server
    assert CL.get_server_index_header(url) == {"format_version": 2, "kmer_length": K, "window_size": W}
    rf = CL.RemoteFilter(url, abs_threshold, rel_threshold, prefix_length, deplete)
    got = rf.filter_reads(reads)
    bases, offsets = oracle.concat_reads(reads)
    keep, hits, total = oracle.filter_batch(index, bases, offsets, None, abs_threshold, rel_threshold,
                                            prefix_length, deplete)
    assert [(g[0], g[1], g[2]) for g in got] == [(bool(k), int(h), int(t)) for k, h, t in zip(keep, hits, total)]
    assert all(g[3] == [] for g in got)
    rf.close()


@pytest.mark.gpu
def test_gpu_client_server_paired(gpu_server, oracle):
    from deacon_server_amd import client as CL
    srv, url, index, reads = gpu_server
    reads = reads[:len(reads) // 2 * 2]
    rf = CL.RemoteFilter(url, 2, 0.01, 0, False)
    got = rf.filter_reads(reads, paired=True)
    bases, offsets = oracle.concat_reads(reads)
    unit_id = (np.arange(len(reads)) // 2).astype(np.uint32)
    keep, hits, total = oracle.filter_batch(index, bases, offsets, unit_id, 2, 0.01, 0, False)
    assert [(g[0], g[1], g[2]) for g in got] == [(bool(k), int(h), int(t)) for k, h, t in zip(keep, hits, total)]
    rf.close()


@pytest.mark.gpu
def test_gpu_server_debug_kmers(gpu_server, oracle):
    from deacon_server_amd import client as CL
    srv, url, index, reads = gpu_server
    keys = set(int(x) for x in index.keys())
    rf = CL.RemoteFilter(url, 1, 0.0, 0, False, debug=True)
    got = rf.filter_reads(reads)
    units = _units_unpaired(oracle, reads)
    want = [_expected_unit(keys, u[0], u[1], bytes(u[2]), K, 1, 0.0, False, True, oracle) for u in units]
    assert got == want
    assert sum(len(g[3]) for g in got) > 100
    rf.close()


# ---- the client as a command (src/main.rs:97-157, remote_filter::run): file to file through the server -------------------

def _oracle_remote_filter(CL, O, url, abs_threshold=2, rel_threshold=0.01, prefix_length=0, deplete=False, debug=False):
    """A RemoteFilter whose minimizers come from the oracle instead of the GPU: the CPU tests' stand-in for the device
    (the product's RemoteFilter has no such path)."""
    class OracleRemoteFilter(CL.RemoteFilter):
        def __init__(self):
            self.server_address = url
            header = CL.get_server_index_header(url)
            self.kmer_length, self.window_size = int(header["kmer_length"]), int(header["window_size"])
            self.abs_threshold, self.rel_threshold = abs_threshold, rel_threshold
            self.prefix_length, self.deplete, self.debug = prefix_length, deplete, debug

        def minimizers(self, reads):
            off, hs, ps = [0], [], []
            for r in reads:
                h, p = O.minimizer_hashes_and_positions(r, self.kmer_length, self.window_size, self.prefix_length)
                hs.append(np.asarray(h, np.uint64))
                ps.append(np.asarray(p, np.uint32))
                off.append(off[-1] + len(h))
            return (np.asarray(off, np.uint64), np.concatenate(hs) if hs else np.zeros(0, np.uint64),
                    np.concatenate(ps) if ps else np.zeros(0, np.uint32))

        def close(self):
            pass
    return OracleRemoteFilter()


def _fastq_bytes(recs):
    return b"".join(b"@" + i + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in recs)


def _named(reads):
    return [(f"read{i} desc {i}".encode(), r) for i, r in enumerate(reads) if r]


def test_read_fastx_shapes():
    import io
    from deacon_server_amd import client as CL
    fa = b"\n>a one\nACGT\nAC\n\n>b\n>c\nGG\r\n"
    assert list(CL.read_fastx(io.BytesIO(fa))) == [(b"a one", b"ACGTAC", None), (b"b", b"", None), (b"c", b"GG", None)]
    fq = b"@q1 x\nACGT\n+\nIIII\n\n@q2\nAC\nGT\n+q2\nII\nII\n@q3\n\n+\n\n"
    assert list(CL.read_fastx(io.BytesIO(fq))) == [(b"q1 x", b"ACGT", b"IIII"), (b"q2", b"ACGT", b"IIII"), (b"q3", b"", b"")]
    with pytest.raises(CL.ClientError):
        list(CL.read_fastx(io.BytesIO(b"@q\nACGT\n+\nII\n")))
    with pytest.raises(CL.ClientError):
        list(CL.read_fastx(io.BytesIO(b"@q\nACGT\n")))
    with pytest.raises(CL.ClientError):
        list(CL.read_fastx(io.BytesIO(b"ACGT\n")))
    assert list(CL.read_fastx(io.BytesIO(b""))) == []


def _decoded(path):
    """any compressed file -> bytes, through the tool's own reader (no GPU involved)"""
    import os
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deacon-server_amd", "bin", "deacon-hip")
    return subprocess.run([tool, "cat", str(path)], check=True, capture_output=True).stdout


# server_tests.rs:148-410 (the client command to a file, .gz / .zst / .xz by extension, --deplete, --rename) with the outputs
# checked byte for byte instead of for existence
@pytest.mark.parametrize("deplete,rename,ext", [(False, False, ""), (True, True, ""), (False, True, ".gz"), (False, False, ".zst"),
                                                (True, False, ".xz")])
def test_client_command_single_over_http(oracle_server, oracle, tmp_path, deplete, rename, ext):
    import gzip
    import io
    import subprocess
    from deacon_server_amd import client as CL
    srv, url, index, reads, path = oracle_server
    recs = _named(reads)
    gz = ext != ""
    inp = tmp_path / ("in.fastq" + ext)
    if ext == ".gz":
        inp.write_bytes(gzip.compress(_fastq_bytes(recs)))
    elif ext:  # written by the tool's own writer, read back by the client (zstd through `deacon-hip cat`, xz through lzma)
        inp.write_bytes(subprocess.run([CL._tool(), "compress", ext[1:], "3"], input=_fastq_bytes(recs), check=True,
                                       capture_output=True).stdout)
    else:
        inp.write_bytes(_fastq_bytes(recs))
    out = tmp_path / ("out.fastq" + ext)
    log = io.StringIO()
    summary = CL.run_client(url, str(inp), None, str(out), None, 2, 0.01, 0, deplete, rename, str(tmp_path / "s.json"),
                            remote_filter=_oracle_remote_filter(CL, oracle, url, deplete=deplete), log=log)
    bases, offsets = oracle.concat_reads([s for _, s in recs])
    keep, hits, total = oracle.filter_batch(index, bases, offsets, None, 2, 0.01, 0, deplete)
    kept = [(i, s) for (i, s), k in zip(recs, keep) if k]
    assert 0 < len(kept) < len(recs)
    want = _fastq_bytes([(str(n + 1).encode(), s) if rename else (i, s) for n, (i, s) in enumerate(kept)])
    assert (_decoded(out) if gz else out.read_bytes()) == want
    if ext == ".gz":
        assert gzip.decompress(out.read_bytes()) == want
    assert summary == {**json.loads((tmp_path / "s.json").read_text())}
    assert (summary["seqs_in"], summary["seqs_out"], summary["seqs_removed"]) == (len(recs), len(kept), len(recs) - len(kept))
    assert summary["bp_in"] == sum(len(s) for _, s in recs) and summary["bp_out"] == sum(len(s) for _, s in kept)
    assert summary["bp_removed"] == summary["bp_in"] - summary["bp_out"]
    assert (summary["k"], summary["w"], summary["deplete"], summary["rename"]) == (K, W, deplete, rename)
    assert summary["index"] == CL.get_server_index_version(url) and summary["input2"] is None
    text = log.getvalue()
    assert text.startswith(f"Deacon v{CL.VERSION}; mode: {'deplete' if deplete else 'search'}; input: single; options: "
                           f"abs_threshold=2, rel_threshold=0.01{', rename' if rename else ''}, threads=8\n")
    assert f"Loaded index (k={K}, w={W}) in " in text and f"Retained {len(kept)}/{len(recs)} sequences (" in text
    assert "Completed in " in text and "Summary saved to " in text


def test_client_command_pairs_over_http(oracle_server, oracle, tmp_path, monkeypatch):
    import io
    import types
    from deacon_server_amd import client as CL
    srv, url, index, reads, path = oracle_server
    recs = _named(reads)
    recs = recs[:len(recs) // 2 * 2]
    r1, r2 = recs[0::2], recs[1::2]
    (tmp_path / "r1.fq").write_bytes(_fastq_bytes(r1))
    (tmp_path / "r2.fa").write_bytes(b"".join(b">" + i + b"\n" + s[:50] + b"\n" + s[50:] + b"\n" for i, s in r2))  # FASTA mates, two lines
    bases, offsets = oracle.concat_reads([s for _, s in recs])
    unit_id = (np.arange(len(recs)) // 2).astype(np.uint32)
    keep, hits, total = oracle.filter_batch(index, bases, offsets, unit_id, 2, 0.01, 0, False)
    assert 0 < int(keep.sum()) < len(keep)
    want1 = _fastq_bytes([p for p, k in zip(r1, keep) if k])
    want2 = b"".join(b">" + i + b"\n" + s + b"\n" for (i, s), k in zip(r2, keep) if k)
    log = io.StringIO()
    # two files in, two files out
    s = CL.run_client(url, str(tmp_path / "r1.fq"), str(tmp_path / "r2.fa"), str(tmp_path / "o1.fq"), str(tmp_path / "o2.fa"),
                      debug=True, remote_filter=_oracle_remote_filter(CL, oracle, url, debug=True), log=log)
    assert (tmp_path / "o1.fq").read_bytes() == want1 and (tmp_path / "o2.fa").read_bytes() == want2
    assert s["seqs_in"] == len(recs) and s["seqs_out"] == 2 * int(keep.sum()) and s["input2"] == str(tmp_path / "r2.fa")
    debug_lines = [ln for ln in log.getvalue().splitlines() if ln.startswith("DEBUG: ")]
    assert len(debug_lines) == int((hits > 0).sum())  # pairs are only reported when something hit (remote_filter.rs:1000)
    first = int(np.flatnonzero(hits > 0)[0])
    assert debug_lines[0] == (f"DEBUG: {r1[first][0].decode()}/{r2[first][0].decode()} hits={hits[first]}/{total[first]} "
                              f"keep={'true' if keep[first] else 'false'} kmers=[]")
    # two files in, one file out: mates interleaved, renamed in output order
    CL.run_client(url, str(tmp_path / "r1.fq"), str(tmp_path / "r2.fa"), str(tmp_path / "both.fx"), None, rename=True,
                  remote_filter=_oracle_remote_filter(CL, oracle, url), log=io.StringIO())
    n, want = 0, b""
    for a, b, k in zip(r1, r2, keep):
        if k:
            want += _fastq_bytes([(str(n + 1).encode(), a[1])]) + b">" + str(n + 2).encode() + b"\n" + b[1] + b"\n"
            n += 2
    assert (tmp_path / "both.fx").read_bytes() == want
    # interleaved stdin; an odd record count is an error that names the count
    inter = _fastq_bytes([x for pair in zip(r1, r2) for x in pair])
    monkeypatch.setattr(CL.sys, "stdin", types.SimpleNamespace(buffer=io.BytesIO(inter)))
    CL.run_client(url, "-", "-", str(tmp_path / "i.fq"), None, remote_filter=_oracle_remote_filter(CL, oracle, url), log=io.StringIO())
    assert (tmp_path / "i.fq").read_bytes() == _fastq_bytes([x for a, b, k in zip(r1, r2, keep) if k for x in (a, b)])
    monkeypatch.setattr(CL.sys, "stdin", types.SimpleNamespace(buffer=io.BytesIO(_fastq_bytes(recs[:3]))))
    with pytest.raises(CL.ClientError, match="Uneven number of interleaved sequence pairs. Found 3 records."):
        CL.run_client(url, "-", "-", str(tmp_path / "j.fq"), None, remote_filter=_oracle_remote_filter(CL, oracle, url), log=io.StringIO())


def test_client_command_refuses_levels_like_the_reference(oracle_server, oracle, tmp_path):  # remote_filter.rs:66-100
    import io
    from deacon_server_amd import client as CL
    srv, url, index, reads, path = oracle_server
    (tmp_path / "in.fq").write_bytes(_fastq_bytes(_named(reads)))
    for name, level, text in (("o.fq.gz", 10, "Invalid gzip compression level 10. Must be between 1 and 9."),
                              ("o.fq.zst", 23, "Invalid zstd compression level 23. Must be between 1 and 22."),
                              ("o.fq.xz", 10, "Invalid xz compression level 10. Must be between 0 and 9.")):
        with pytest.raises(CL.ClientError, match=text):
            CL.run_client(url, str(tmp_path / "in.fq"), None, str(tmp_path / name), None, compression_level=level,
                          remote_filter=_oracle_remote_filter(CL, oracle, url), log=io.StringIO())


def test_client_command_prefix_is_a_u8_like_the_reference(oracle_server, oracle, tmp_path):
    import io
    from deacon_server_amd import client as CL
    srv, url, index, reads, path = oracle_server
    recs = _named(reads)
    (tmp_path / "in.fq").write_bytes(_fastq_bytes(recs))
    log = io.StringIO()
    s = CL.run_client(url, str(tmp_path / "in.fq"), None, str(tmp_path / "o.fq"), None, 1, 0.0, 256 + 60,
                      remote_filter=_oracle_remote_filter(CL, oracle, url, 1, 0.0, prefix_length=60), log=log)
    bases, offsets = oracle.concat_reads([x for _, x in recs])
    keep, _, _ = oracle.filter_batch(index, bases, offsets, None, 1, 0.0, 60, False)
    assert s["seqs_out"] == int(keep.sum()) and s["prefix_length"] == 316
    assert "modulo 256 = 60" in log.getvalue()


@pytest.mark.gpu
def test_gpu_client_command_file_to_file(gpu_server, oracle, tmp_path, capsys):
    from deacon_server_amd import client as CL
    srv, url, index, reads = gpu_server
    recs = _named(reads)
    (tmp_path / "in.fq").write_bytes(_fastq_bytes(recs))
    for deplete in (False, True):
        argv = [url, str(tmp_path / "in.fq"), "-o", str(tmp_path / "out.fq"), "-s", str(tmp_path / "s.json"), "-p", "120"]
        assert CL.main(argv + (["-d"] if deplete else [])) == 0
        bases, offsets = oracle.concat_reads([s for _, s in recs])
        keep, hits, total = oracle.filter_batch(index, bases, offsets, None, 2, 0.01, 120, deplete)
        assert (tmp_path / "out.fq").read_bytes() == _fastq_bytes([r for r, k in zip(recs, keep) if k])
        summary = json.loads((tmp_path / "s.json").read_text())
        assert summary["seqs_out"] == int(keep.sum()) and summary["prefix_length"] == 120 and summary["deplete"] is deplete
    assert "Retained " in capsys.readouterr().err
    # -O without a second input: a warning, no second file (server_tests.rs:992-1026)
    assert CL.main([url, str(tmp_path / "in.fq"), "-o", str(tmp_path / "o1.fq"), "-O", str(tmp_path / "o2.fq")]) == 0
    assert "Warning: --output2 specified but no second input" in capsys.readouterr().err
    assert (tmp_path / "o1.fq").exists() and not (tmp_path / "o2.fq").exists()
    # two files of mates, two outputs, one of them .zst (server_tests.rs:413-450, :870-990)
    pairs = recs[:len(recs) // 2 * 2]
    (tmp_path / "r1.fq").write_bytes(_fastq_bytes(pairs[0::2]))
    (tmp_path / "r2.fq").write_bytes(_fastq_bytes(pairs[1::2]))
    assert CL.main([url, str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), "-o", str(tmp_path / "p1.fq"), "-O",
                    str(tmp_path / "p2.fq.zst"), "-d", "-R"]) == 0
    bases, offsets = oracle.concat_reads([s for _, s in pairs])
    unit_id = (np.arange(len(pairs)) // 2).astype(np.uint32)
    keep, _, _ = oracle.filter_batch(index, bases, offsets, unit_id, 2, 0.01, 0, True)
    kept = [j for j, k in enumerate(keep) if k]
    assert 0 < len(kept) < len(keep)
    assert (tmp_path / "p1.fq").read_bytes() == _fastq_bytes([(str(2 * n + 1).encode(), pairs[2 * j][1]) for n, j in enumerate(kept)])
    assert _decoded(tmp_path / "p2.fq.zst") == _fastq_bytes([(str(2 * n + 2).encode(), pairs[2 * j + 1][1]) for n, j in enumerate(kept)])
    assert CL.main(["http://127.0.0.1:9", str(tmp_path / "in.fq")]) == 1  # nobody listens there: an error, not a traceback
