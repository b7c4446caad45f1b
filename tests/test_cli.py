"""f2: the `deacon-hip` command-line driver, checked the way the reference checks its own binary
(tests/filter_tests.rs, tests/index_tests.rs, tests/cli_tests.rs: black-box runs on small FASTA/FASTQ files), plus
random inputs whose kept records are compared with the oracle's decisions.  Sequence literals come from
tests/golden/reference_cases.json (data of the reference's tests)."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, mutate, random_reads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "deacon-server_amd", "bin", "deacon-hip")
CASES = {c["id"]: c for c in json.load(open(os.path.join(GOLDEN, "reference_cases.json")))["cases"]}
SEQ1, SEQ2 = CASES["C-7"]["ref"]  # create_test_fasta / create_test_fastq of tests/filter_tests.rs:8-21
SC2 = CASES["C-1"]["units"][0][0]
SC2_REV = CASES["C-2"]["units"][0][0]
SC2B, SC2B_REV = CASES["C-3-fwd"]["units"][0][1], CASES["C-3-rev"]["units"][0][1]


def run(*args, stdin=None, check=True, env=None):
    p = subprocess.run([BIN, *map(str, args)], input=stdin, capture_output=True, env=env)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p


def fasta(path, recs):
    path.write_text("".join(f">{i}\n{s}\n" for i, s in recs))


def fastq(path, recs):
    path.write_text("".join(f"@{i}\n{s}\n+\n{'~' * len(s)}\n" for i, s in recs))


def build_index(tmp_path, recs, name="ref", k=None, w=None):
    fa = tmp_path / f"{name}.fasta"
    fasta(fa, recs)
    idx = tmp_path / f"{name}.idx"
    args = ["index", "build", fa, "-o", idx]
    if k:
        args += ["-k", k]
    if w:
        args += ["-w", w]
    run(*args)
    return idx


# ---- no GPU needed ---------------------------------------------------------------------------------------------
def test_binary_exists_version_and_usage():  # tests/cli_tests.rs
    assert os.path.exists(BIN), "build with __graft_entry__.build()"
    assert run("--version").stdout.startswith(b"deacon-hip ")
    p = run(check=False)
    assert p.returncode == 2 and b"Usage" in p.stderr


def test_every_subcommand_prints_its_options():  # src/main.rs:17-235 (clap: help on stdout, exit code 0)
    wanted = {
        ("filter",): [b"<INDEX>", b"--output2", b"--abs-threshold", b"--rel-threshold", b"--prefix-length", b"--deplete",
                      b"--rename", b"--summary", b"--threads", b"--compression-level", b"--debug", b"--quiet", b"--gpus"],
        ("index",): [b"build", b"info", b"union", b"diff"],
        ("index", "build"): [b"-k <K>", b"-w <W>", b"--output", b"--capacity", b"--entropy-threshold", b"[default: 31]"],
        ("index", "info"): [b"<INDEX>"],
        ("index", "union"): [b"<INPUTS>...", b"--output"],
        ("index", "diff"): [b"<FIRST>", b"<SECOND>", b"--kmer-length", b"--window-size"],
    }
    for sub, words in wanted.items():
        for flag in ("--help", "-h"):
            p = run(*sub, flag)
            assert p.stdout.startswith(b"Usage") or b"\n\nUsage: deacon-hip " + " ".join(sub).encode() in p.stdout, sub
            for word in words:
                assert word in p.stdout, (sub, word)
    # server / client (src/main.rs:86-157) are the package's Python, started by the tool as a child process
    p = run("client", "--help")
    assert b"usage: deacon-hip client" in p.stdout and b"--output2" in p.stdout and b"server_address" in p.stdout
    p = run("server", check=False)
    assert p.returncode == 2 and b"deacon-hip server" in p.stderr
    # the flag is found behind other arguments too, before anything is opened
    assert b"Usage: deacon-hip filter" in run("filter", "no-such.idx", "-d", "--help").stdout


def test_filter_fails_loudly_without_gpu(tmp_path, dcn):
    import ctypes
    n = ctypes.c_int()
    if dcn._native.lib().dcn_device_count(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    (tmp_path / "x.idx").write_bytes(bytes([2, 31, 15, 0]))
    fastq(tmp_path / "r.fq", [("a", "ACGT" * 20)])
    p = run("filter", tmp_path / "x.idx", tmp_path / "r.fq", check=False)
    assert p.returncode == 1 and b"Error" in p.stderr


def bgzf_compress(data, block=65280, level=6, eof=True):
    """BGZF (SAM spec 4.1), written by hand: gzip members of <= 64 KB whose extra field 'BC' holds the member's size - 1"""
    import struct
    import zlib
    out = bytearray()

    def member(chunk):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = c.compress(chunk) + c.flush()
        out.extend(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(raw) + 25) + raw +
                   struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    for i in range(0, len(data), block):
        member(data[i:i + block])
    if eof:
        member(b"")
    return bytes(out)


def test_blocked_gzip_input_is_inflated_member_by_member(tmp_path):
    """`deacon-hip cat` = the tool's input side alone (no GPU).  BGZF members are found by their length field and inflated on
    several threads; the bytes must be those of the stream decoder (DCN_CLI_NO_BGZF) for: a BGZF file with and without its
    end-of-file member, BGZF followed by an ordinary member and the other way round (`cat a.bgz b.gz`), an ordinary gzip
    file, tiny and empty inputs, stdin; a truncated or corrupted member is an error, not silence."""
    rng = np.random.default_rng(3)
    data = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGTN"), int(rng.integers(20, 400))).astype(np.uint8)), b"I" * 7)
                    for i in range(40_000))
    cases = {"bgzf": (bgzf_compress(data), data), "bgzf-no-eof": (bgzf_compress(data, eof=False), data),
             "bgzf-then-gzip": (bgzf_compress(data[:3_000_000], eof=False) + gzip.compress(data[3_000_000:], 1), data),
             "gzip-then-bgzf": (gzip.compress(data[:1_000_000], 1) + bgzf_compress(data[1_000_000:]), data),
             "gzip": (gzip.compress(data, 1), data), "small-blocks": (bgzf_compress(data[:500_000], block=777), data[:500_000]),
             "tiny": (bgzf_compress(b"@a\nACGT\n+\nIIII\n"), b"@a\nACGT\n+\nIIII\n"), "empty": (bgzf_compress(b""), b"")}
    for name, (blob, want) in cases.items():
        f = tmp_path / f"{name}.gz"
        f.write_bytes(blob)
        for extra in ({}, {"DCN_CLI_NO_BGZF": "1"}, {"DCN_CLI_BGZF_THREADS": "1"}, {"DCN_CLI_BGZF_THREADS": "7"}):
            p = run("cat", f, env=dict(os.environ, **extra))
            assert p.stdout == want, (name, extra, len(p.stdout), len(want))
        assert run("cat", "-", stdin=blob).stdout == want, name
    whole = cases["bgzf"][0]
    for bad in (whole[:len(whole) // 2], whole[:70_000] + bytes([whole[70_000] ^ 0x55]) + whole[70_001:]):
        (tmp_path / "bad.gz").write_bytes(bad)
        p = run("cat", tmp_path / "bad.gz", check=False)
        assert p.returncode == 1 and b"gzip stream" in p.stderr


def test_bzip2_input_is_read_like_the_references_reader_reads_it(tmp_path):
    """the reference's reader sniffs bzip2 too (niffler's default formats; README documents gz / zst / xz): one stream, a
    concatenation of streams (pbzip2, `cat a.bz2 b.bz2`), stdin; a cut or damaged stream is an error (no GPU: `cat`)"""
    import bz2
    data = b"".join(b"@r%d\nACGTACGTACGGTTAACC\n+\nIIIIIIIIIIIIIIIIII\n" % i for i in range(60_000))
    for name, blob in (("one", bz2.compress(data)), ("two", bz2.compress(data[:1_000_000]) + bz2.compress(data[1_000_000:], 1))):
        f = tmp_path / f"{name}.fq.bz2"
        f.write_bytes(blob)
        assert run("cat", f).stdout == data, name
        assert run("cat", "-", stdin=blob).stdout == data, name
    whole = bz2.compress(data)
    p = run("cat", "-", stdin=whole[:-20], check=False)
    assert p.returncode == 1 and b"truncated bzip2 stream" in p.stderr
    hurt = bytearray(whole)
    hurt[len(hurt) // 2] ^= 0x10
    p = run("cat", "-", stdin=bytes(hurt), check=False)
    assert p.returncode == 1 and b"bzip2 stream" in p.stderr


# ---- the reference's filter tests ------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@gpu
def test_filter_to_file_and_summary(tmp_path):  # filter_tests.rs:92-128
    idx = build_index(tmp_path, [("seq1", "A" * 100)])
    fastq(tmp_path / "reads.fastq", [("seq1", SEQ1), ("seq2", SEQ2)])
    out, summ = tmp_path / "filtered.fastq", tmp_path / "summary.json"
    run("filter", idx, tmp_path / "reads.fastq", "--output", out, "--summary", summ)
    assert out.read_text() == ""
    s = json.loads(summ.read_text())
    assert s["seqs_in"] == 2 and s["seqs_out"] == 0 and s["bp_in"] == len(SEQ1) + len(SEQ2)
    assert s["k"] == 31 and s["w"] == 15 and s["abs_threshold"] == 2 and s["rel_threshold"] == 0.01
    assert set(s) == {"version", "index", "input", "input2", "output", "output2", "k", "w", "abs_threshold",
                      "rel_threshold", "prefix_length", "deplete", "rename", "seqs_in", "seqs_out",
                      "seqs_out_proportion", "seqs_removed", "seqs_removed_proportion", "bp_in", "bp_out",
                      "bp_out_proportion", "bp_removed", "bp_removed_proportion", "time", "seqs_per_second",
                      "bp_per_second"}  # FilterSummary, src/filter_common.rs:11-38


def _zstd():
    import ctypes
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    z.ZSTD_isError.restype = ctypes.c_uint
    z.ZSTD_isError.argtypes = [ctypes.c_size_t]
    return z


def zstd_compress(data, level=3):
    import ctypes
    z = _zstd()
    buf = ctypes.create_string_buffer(z.ZSTD_compressBound(len(data)))
    n = z.ZSTD_compress(buf, len(buf), data, len(data), level)
    assert not z.ZSTD_isError(n)
    return buf.raw[:n]


def zstd_decompress(data, max_out=1 << 26):
    import ctypes
    z = _zstd()
    buf = ctypes.create_string_buffer(max_out)
    n = z.ZSTD_decompress(buf, max_out, data, len(data))
    assert not z.ZSTD_isError(n)
    return buf.raw[:n]


@gpu
def test_filter_compressed_outputs_and_inputs(tmp_path):  # filter_tests.rs:131-215 (gzip, zstd, xz by extension)
    import lzma
    idx = build_index(tmp_path, [("seq1", SEQ1), ("seq2", SEQ2)])
    fastq(tmp_path / "reads.fastq", [("seq1", SEQ1), ("seq2", SEQ2)])
    plain = run("filter", idx, tmp_path / "reads.fastq").stdout
    assert plain.count(b"@seq") == 2
    decoders = {"gz": gzip.decompress, "zst": zstd_decompress, "xz": lzma.decompress}
    for ext, dec in decoders.items():  # written by this build, read by an independent decoder
        out = tmp_path / f"filtered.fastq.{ext}"
        run("filter", idx, tmp_path / "reads.fastq", "-o", out)
        assert out.stat().st_size > 0 and dec(out.read_bytes()) == plain, ext
        assert run("filter", idx, out).stdout == plain, ext  # and read back by this build (format found by content)
    # compressed outputs are one member / frame / stream per batch, compressed on the formatter threads: many small
    # batches must still decode to the plain output, and an output with nothing kept is a valid empty file of its format
    rng = np.random.default_rng(3)
    many = [(f"r{i}", SEQ1 if i % 3 else "".join("ACGT"[c] for c in rng.integers(0, 4, 120))) for i in range(600)]
    fastq(tmp_path / "many.fastq", many)
    env = dict(os.environ, DCN_CLI_CHUNK_MB="1", DCN_CLI_MAX_BATCH_READS="50")
    plain_many = run("filter", idx, tmp_path / "many.fastq").stdout
    assert plain_many.count(b"@r") == 400
    for ext, dec in decoders.items():
        out = tmp_path / f"many.fastq.{ext}"
        run("filter", idx, tmp_path / "many.fastq", "-o", out, "-t", "4", env=env)
        assert dec(out.read_bytes()) == plain_many, ext
        run("filter", "-d", idx, tmp_path / "reads.fastq", "-o", out)
        assert out.stat().st_size > 0 and dec(out.read_bytes()) == b"", ext
    # a .gz output is a file of BGZF members (round 4): every member carries its size ('BC'), none holds more than 64 KB, the
    # last one is bgzip's 28-byte end-of-file marker -- and the tool's own input side reads it member by member
    big = [(f"b{i}", "".join("ACGT"[c] for c in rng.integers(0, 4, 150))) for i in range(3000)] + [("hit", SEQ1)]
    fastq(tmp_path / "big.fastq", big)
    run("filter", "-d", idx, tmp_path / "big.fastq", "-o", tmp_path / "big.fastq.gz")
    blob = (tmp_path / "big.fastq.gz").read_bytes()
    pos, members, sizes = 0, 0, []
    while pos < len(blob):
        assert blob[pos:pos + 4] == b"\x1f\x8b\x08\x04" and blob[pos + 12:pos + 16] == b"BC\x02\x00"
        total = int.from_bytes(blob[pos + 16:pos + 18], "little") + 1
        sizes.append(int.from_bytes(blob[pos + total - 4:pos + total], "little"))
        pos += total
        members += 1
    assert pos == len(blob) and members >= 15 and max(sizes) <= 65536 and sizes[-1] == 0 and total == 28
    want = run("filter", "-d", idx, tmp_path / "big.fastq").stdout
    assert gzip.decompress(blob) == want and want.count(b"@b") == 3000
    assert run("cat", tmp_path / "big.fastq.gz").stdout == want
    assert run("cat", tmp_path / "big.fastq.gz", env=dict(os.environ, DCN_CLI_NO_BGZF="1")).stdout == want
    run("filter", "-d", idx, tmp_path / "big.fastq", "-o", tmp_path / "one.fastq.gz", env=dict(os.environ, DCN_CLI_GZIP_ONE_MEMBER="1"))
    assert gzip.decompress((tmp_path / "one.fastq.gz").read_bytes()) == want and (tmp_path / "one.fastq.gz").read_bytes()[3] != 4
    for ext, lo, hi in (("gz", 1, 9), ("zst", 1, 22), ("xz", 0, 9)):  # validate_compression_level, local_filter.rs:95-107
        p = run("filter", idx, tmp_path / "reads.fastq", "-o", tmp_path / f"x.{ext}", "--compression-level", hi + 1, check=False)
        assert p.returncode != 0 and b"compression level" in p.stderr, ext
        run("filter", idx, tmp_path / "reads.fastq", "-o", tmp_path / f"x.{ext}", "--compression-level", hi)
    # compressed inputs written by independent encoders, several members / frames / streams in one file, and stdin
    raw = (tmp_path / "reads.fastq").read_bytes()
    half = raw.index(b"@seq2")
    members = {"gz": gzip.compress(raw[:half]) + gzip.compress(raw[half:]),
               "zst": zstd_compress(raw[:half]) + zstd_compress(raw[half:]),
               "xz": lzma.compress(raw[:half]) + lzma.compress(raw[half:])}
    for ext, blob in members.items():
        f = tmp_path / f"in.fastq.{ext}"
        f.write_bytes(blob)
        assert run("filter", idx, f).stdout == plain, ext
        assert run("filter", idx, "-", stdin=blob).stdout == plain, ext
    assert run("filter", idx, "-", stdin=raw).stdout == plain
    p = run("filter", idx, "-", stdin=members["zst"][:-7], check=False)
    assert p.returncode != 0 and b"zstd" in p.stderr


@gpu
def test_filter_deplete_rename_min_matches_prefix(tmp_path):  # filter_tests.rs:218-341
    idx = build_index(tmp_path, [("seq1", SEQ1), ("seq2", SEQ2)])
    fastq(tmp_path / "reads.fastq", [("seq1", SEQ1), ("seq2", SEQ2)])
    fq = tmp_path / "reads.fastq"
    assert run("filter", idx, fq).stdout.count(b"@seq") == 2
    assert run("filter", "--deplete", idx, fq).stdout == b""
    renamed = run("filter", "--rename", idx, fq).stdout.decode().split("\n")
    assert renamed[0] == "@1" and renamed[4] == "@2" and renamed[1] == SEQ1
    assert run("filter", "--abs-threshold", "1", idx, fq).stdout.count(b"@seq") == 2
    assert run("filter", "--prefix-length", "6", idx, fq).stdout == b""  # shorter than k: no minimizers, no crash


@gpu
def test_filter_paired_modes(tmp_path):  # filter_tests.rs:344-583, 726-940
    idx = build_index(tmp_path, [("seq1", SEQ1), ("seq2", SEQ2)])
    r1, r2 = tmp_path / "r1.fastq", tmp_path / "r2.fastq"
    fastq(r1, [("read1", SEQ1), ("read2", SEQ2)])
    fastq(r2, [("read1", SEQ1), ("read2", SEQ2)])
    out = run("filter", idx, r1, r2).stdout.decode()
    assert out.count("@read") == 4 and out.split("\n")[0] == "@read1" and out.split("\n")[4] == "@read1"  # interleaved
    assert run("filter", "-d", idx, r1, r2).stdout == b""
    ren = run("filter", "-R", idx, r1, r2).stdout.decode().split("\n")
    assert [ren[i] for i in (0, 4, 8, 12)] == ["@1", "@2", "@3", "@4"]
    assert run("filter", "-a", "1", idx, r1, r2).stdout.count(b"@read") == 4
    # interleaved stdin: `- -`
    inter = "".join(f"@p{i}/{m}\n{s}\n+\n{'~' * len(s)}\n" for i, s in ((1, SEQ1), (2, SEQ2)) for m in (1, 2))
    got = run("filter", idx, "-", "-", stdin=inter.encode()).stdout.decode()
    assert got == inter
    # single stdin
    single = (tmp_path / "r1.fastq").read_bytes()
    assert run("filter", idx, "-", stdin=single).stdout == single
    # --output2: mates to separate files; gzip too; ignored with a warning for single input
    o1, o2 = tmp_path / "o1.fastq", tmp_path / "o2.fastq"
    run("filter", idx, r1, r2, "-o", o1, "-O", o2)
    assert o1.read_text() == r1.read_text() and o2.read_text() == r2.read_text()
    g1, g2 = tmp_path / "o1.fastq.gz", tmp_path / "o2.fastq.gz"
    run("filter", idx, r1, r2, "-o", g1, "-O", g2)
    assert gzip.open(g1).read() == r1.read_bytes() and gzip.open(g2).read() == r2.read_bytes()
    p = run("filter", idx, r1, "-o", o1, "-O", tmp_path / "ignored.fastq")
    assert b"--output2 will be ignored" in p.stderr and not (tmp_path / "ignored.fastq").exists()


@gpu
def test_filter_sc2_strands_and_pairs(tmp_path):  # filter_tests.rs:586-723
    idx = build_index(tmp_path, [("mn908947.3_0:60", SC2)])
    for name, seq in (("fwd", SC2), ("rev", SC2_REV)):
        fq = tmp_path / f"{name}.fastq"
        fastq(fq, [(f"mn908947.3_0:60_{name}", seq)])
        assert run("filter", "-d", "-a", "1", "-r", "0.01", idx, fq).stdout == b""
        assert run("filter", "-d", idx, fq).stdout == b""  # default -a 2: the read shares >= 2 minimizers
        assert run("filter", idx, fq).stdout.count(b"@mn9") == 1
    for m1, m2 in ((SC2, SC2B), (SC2_REV, SC2B_REV)):
        fastq(tmp_path / "m1.fastq", [("a/1", m1)])
        fastq(tmp_path / "m2.fastq", [("a/2", m2)])
        assert run("filter", "-d", idx, tmp_path / "m1.fastq", tmp_path / "m2.fastq").stdout == b""


@gpu
def test_shared_minimizer_counted_once(tmp_path):  # filter_tests.rs:943-1015
    c = CASES["C-4"]
    idx = build_index(tmp_path, [("reference", c["ref"][0])])
    fasta(tmp_path / "r1.fasta", [("read1/1", c["units"][0][0])])
    fasta(tmp_path / "r2.fasta", [("read1/2", c["units"][0][1])])
    summ = tmp_path / "s.json"
    out = run("filter", "--deplete", idx, tmp_path / "r1.fasta", tmp_path / "r2.fasta", "--summary", summ,
              "--abs-threshold", "2", "--rel-threshold", "0.01").stdout
    assert out.count(b">read1") == 2
    assert json.loads(summ.read_text())["seqs_out"] == 2


@gpu
def test_proportional_thresholds(tmp_path, oracle):  # filter_tests.rs:1018-1130 (the reference only checks success)
    idx = build_index(tmp_path, [("seq1", SEQ1), ("seq2", SEQ2)])
    fq = tmp_path / "reads.fastq"
    fastq(fq, [("seq1", SEQ1), ("seq2", SEQ2)])
    oidx = oracle.Index.build([SEQ1.encode(), SEQ2.encode()])
    b, o = oracle.concat_reads([SEQ1.encode(), SEQ2.encode()])
    for rel in ("0.0", "0.01", "0.1", "0.5", "1.0"):
        keep, hits, total = oracle.filter_batch(oidx, b, o, abs_threshold=1, rel_threshold=float(rel))
        assert run("filter", "-a", "1", "-r", rel, idx, fq).stdout.count(b"@seq") == int(keep.sum()), rel
    # periodic reads: 56 minimizers but few distinct ones, so a proportion of 1.0 cannot be met by distinct hits
    assert total.tolist() == [56, 56] and hits.max() < 56


@gpu
def test_multiline_fasta_and_newline_mapping(tmp_path):  # filter_tests.rs:1133-1251
    c5 = CASES["C-5"]
    idx = build_index(tmp_path, [("ref", c5["ref"][0])], k=31, w=1)
    q = tmp_path / "query.fasta"
    q.write_text(">query\nACGTTTAAGGCCAACC\nACACACACACACATT\n")
    out = run("filter", "-a", "1", idx, q).stdout.decode()
    assert ">query" in out and c5["ref"][0] in out
    assert run("filter", "-a", "1", idx, "-", stdin=q.read_bytes()).stdout.decode() == out  # the stream (chunk) reader
    many = "".join(f">q{i}\nACGTTTAAGGCCAACC\nACACACACACACATT\n" for i in range(60000)).encode()
    got = run("filter", "-a", "1", idx, "-", stdin=many, env=dict(os.environ, DCN_CLI_CHUNK_MB="1")).stdout
    assert got.count(b">q") == 60000 and got == run("filter", "-a", "1", idx, "-", stdin=many, env=dict(os.environ, DCN_CLI_NO_CHUNK_READER="1")).stdout
    ref = tmp_path / "nl.fa"
    ref.write_text(">reference\nAAAAA\nAAAAA\nAAAAA\nAAAAA\n")
    idx2 = tmp_path / "nl.idx"
    run("index", "build", "-k", "5", "-w", "5", ref, "-o", idx2)
    q2 = tmp_path / "q2.fa"
    q2.write_text(">query\nAAAAACAAAAACAAAAACAAAAA\n")
    assert b">query" not in run("filter", "-a", "1", "-r", "0.0", idx2, q2).stdout


@gpu
def test_large_kmer_filter(tmp_path):  # filter_tests.rs:1254-1296
    idx = build_index(tmp_path, [("seq1", SEQ1), ("seq2", SEQ2)], k=41, w=15)
    fq = tmp_path / "reads.fastq"
    fastq(fq, [("seq1", SEQ1), ("seq2", SEQ2)])
    assert run("filter", idx, fq, "-a", "1", "-r", "0.0").stdout.count(b"@seq") == 2


@gpu
def test_index_build_and_info(tmp_path):  # index_tests.rs:10-166
    idx = build_index(tmp_path, [("seq1", SEQ1), ("seq2", SEQ2)])
    assert idx.stat().st_size > 4
    p = run("index", "info", idx)
    assert b"K-mer length (k): 31" in p.stderr and b"Window size (w): 15" in p.stderr
    idx2 = build_index(tmp_path, [("seq1", SEQ1), ("seq2", SEQ2)], name="custom", k=15, w=11)
    p = run("index", "info", idx2)
    assert b"K-mer length (k): 15" in p.stderr and b"Window size (w): 11" in p.stderr
    p = run("index", "build", tmp_path / "ref.fasta", "-k", "31", "-w", "16", check=False)  # k+w-1 even
    assert p.returncode != 0 and b"must be odd" in p.stderr
    # stdout output: `deacon index build ref.fa > ref.idx` (how the reference's tests build indexes)
    raw = run("index", "build", tmp_path / "ref.fasta", "-q").stdout
    assert raw[:3] == bytes([2, 31, 15]) and len(raw) == idx.stat().st_size


@gpu
def test_index_union_and_diff(tmp_path, oracle):  # index_tests.rs:86-341
    rng = np.random.default_rng(3)
    g1 = random_reads(rng, 1, 20_000, 20_000)[0].decode()
    g2 = g1[10_000:] + random_reads(rng, 1, 10_000, 10_000)[0].decode()  # overlaps the second half of g1
    i1 = build_index(tmp_path, [("a", g1)], name="one")
    i2 = build_index(tmp_path, [("b", g2)], name="two")
    k1 = set(oracle.Index.read(i1).keys().tolist())
    k2 = set(oracle.Index.read(i2).keys().tolist())
    u = tmp_path / "union.idx"
    run("index", "union", i1, i2, "-o", u)
    assert set(oracle.Index.read(u).keys().tolist()) == k1 | k2  # index_tests.rs:86-127: size >= each input
    d = tmp_path / "diff.idx"
    p = run("index", "diff", i1, i2, "-o", d)
    assert set(oracle.Index.read(d).keys().tolist()) == k1 - k2 and 0 < len(k1 - k2) < len(k1)
    assert f"Removed {len(k1 & k2)} minimizers, {len(k1 - k2)} remaining".encode() in p.stderr
    # three ways to name the second operand agree in count and file size (index_tests.rs:168-341)
    d2, d3 = tmp_path / "diff_fastx.idx", tmp_path / "diff_auto.idx"
    run("index", "diff", i1, tmp_path / "two.fasta", "-k", "31", "-w", "15", "-o", d2)
    run("index", "diff", i1, tmp_path / "two.fasta", "-o", d3)  # FASTX detected, k and w from the first index
    assert d.stat().st_size == d2.stat().st_size == d3.stat().st_size
    assert set(oracle.Index.read(d2).keys().tolist()) == set(oracle.Index.read(d3).keys().tolist()) == k1 - k2
    # mismatching parameters are refused
    i3 = build_index(tmp_path, [("a", g1)], name="other", k=15, w=11)
    assert run("index", "union", i1, i3, "-o", tmp_path / "x.idx", check=False).returncode != 0
    assert run("index", "diff", i1, tmp_path / "two.fasta", "-k", "15", "-w", "11", check=False).returncode != 0


# ---- random data against the oracle ------------------------------------------------------------------------------
@gpu
@pytest.mark.parametrize("paired", [False, True])
def test_cli_matches_oracle_on_random_reads(tmp_path, oracle, paired):
    rng = np.random.default_rng(55)
    genome = random_reads(rng, 1, 60_000, 60_000)[0]
    (tmp_path / "g.fa").write_bytes(b">chr1 test genome\n" + b"\n".join(genome[i:i + 70] for i in range(0, len(genome), 70)) + b"\n")
    idx = tmp_path / "g.idx"
    run("index", "build", tmp_path / "g.fa", "-o", idx, "-q")
    oidx = oracle.Index.build([genome])
    assert np.array_equal(np.sort(oracle.Index.read(idx).keys()), np.sort(oidx.keys()))
    reads = []
    for i in range(6000):
        ln = int(rng.integers(25, 300))
        if rng.random() < 0.5:
            s = int(rng.integers(0, len(genome) - ln))
            reads.append(mutate(rng, genome[s:s + ln], 0.02))
        else:
            reads.append(random_reads(rng, 1, ln, ln, p_n=0.005)[0])
    b, o = oracle.concat_reads(reads)
    summ = tmp_path / "s.json"
    if paired:
        fastq(tmp_path / "r1.fq", [(f"r{i}/1 x", reads[2 * i].decode()) for i in range(3000)])
        fastq(tmp_path / "r2.fq", [(f"r{i}/2 x", reads[2 * i + 1].decode()) for i in range(3000)])
        out = run("filter", "-d", idx, tmp_path / "r1.fq", tmp_path / "r2.fq", "-s", summ).stdout.decode()
        keep, _, _ = oracle.filter_batch(oidx, b, o, (np.arange(6000) // 2).astype(np.uint32), deplete=True)
        want_ids = [f"r{i}/{m} x" for i in range(3000) if keep[i] for m in (1, 2)]
        lens = np.array([len(r) for r in reads]).reshape(-1, 2).sum(1)
    else:
        fastq(tmp_path / "r.fq", [(f"r{i} x", r.decode()) for i, r in enumerate(reads)])
        out = run("filter", "-d", idx, tmp_path / "r.fq", "-s", summ).stdout.decode()
        keep, _, _ = oracle.filter_batch(oidx, b, o, deplete=True)
        want_ids = [f"r{i} x" for i in range(6000) if keep[i]]
        lens = np.array([len(r) for r in reads])
    got_ids = [l[1:] for l in out.split("\n")[0::4] if l]
    assert got_ids == want_ids  # same records, input order
    s = json.loads(summ.read_text())
    assert s["seqs_in"] == 6000 and s["seqs_out"] == len(want_ids)
    assert s["bp_out"] == int(lens[keep].sum()) and s["bp_removed"] == int(lens[~keep].sum())
    assert s["deplete"] is True and abs(s["seqs_out_proportion"] - len(want_ids) / 6000) < 1e-12


@pytest.mark.gpu
def test_debug_lines(tmp_path, oracle):  # src/local_filter.rs:354-363, 424-434
    rng = np.random.default_rng(56)
    genome = random_reads(rng, 1, 20_000, 20_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    oidx = oracle.Index.build([genome])
    keys = set(int(x) for x in oidx.keys())
    reads = [mutate(rng, genome[s:s + 150], 0.03) for s in range(0, 6000, 300)] + random_reads(rng, 5, 100, 150) + [b"ACGT"]
    reads[3] = reads[3][:60] + reads[3][:60]  # repeated minimizers: listed once
    fastq(tmp_path / "r.fq", [(f"r{i} d", r.decode()) for i, r in enumerate(reads)])
    err = run("filter", idx, tmp_path / "r.fq", "--debug", "-p", 120).stderr.decode().splitlines()
    lines = [l for l in err if l.startswith("DEBUG: ")]
    assert len(lines) == len(reads)
    for i, r in enumerate(reads):
        h, p = oracle.minimizer_hashes_and_positions(r, 31, 15, 120)
        seen, kmers = set(), []
        for hv, pv in zip(h.tolist(), p.tolist()):
            if hv in keys and hv not in seen:
                seen.add(hv)
                kmers.append(r[pv:pv + 31].decode())
        keep = oracle.meets_filtering_criteria(len(seen), len(h), 2, 0.01, False)
        assert lines[i] == (f"DEBUG: r{i} d hits={len(seen)}/{len(h)} keep={'true' if keep else 'false'} "
                            f"kmers=[{','.join(kmers)}]"), i
    assert any("kmers=[A" in l or "kmers=[C" in l or "kmers=[G" in l or "kmers=[T" in l for l in lines)
    # pairs: "id1/id2", only pairs with hits, and the empty k-mer list the reference produces
    fastq(tmp_path / "p1.fq", [(f"p{i}/1", reads[2 * i].decode()) for i in range(12)])
    fastq(tmp_path / "p2.fq", [(f"p{i}/2", reads[2 * i + 1].decode()) for i in range(12)])
    err = run("filter", idx, tmp_path / "p1.fq", tmp_path / "p2.fq", "--debug").stderr.decode().splitlines()
    lines = [l for l in err if l.startswith("DEBUG: ")]
    b, o = oracle.concat_reads(reads[:24])
    keep, hits, total = oracle.filter_batch(oidx, b, o, (np.arange(24) // 2).astype(np.uint32))
    want = [f"DEBUG: p{i}/1/p{i}/2 hits={hits[i]}/{total[i]} keep={'true' if keep[i] else 'false'} kmers=[]"
            for i in range(12) if hits[i] > 0]
    assert lines == want and 0 < len(want) < 12


@gpu
@pytest.mark.parametrize("paired", [False, True])
def test_batches_cut_into_several_calls(tmp_path, oracle, paired, monkeypatch):
    """A parsed chunk that exceeds the context (very short records) is filtered in several calls with rebased
    offsets / unit ids; forced here with a tiny per-call read limit."""
    rng = np.random.default_rng(57)
    genome = random_reads(rng, 1, 30_000, 30_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    oidx = oracle.Index.build([genome])
    reads = []
    for i in range(5000):
        ln = int(rng.integers(20, 200))
        s = int(rng.integers(0, len(genome) - ln))
        reads.append(mutate(rng, genome[s:s + ln], 0.02) if i % 2 else random_reads(rng, 1, ln, ln)[0])
    b, o = oracle.concat_reads(reads)
    monkeypatch.setenv("DCN_CLI_MAX_BATCH_READS", "338")
    if paired:
        fastq(tmp_path / "r1.fq", [(f"r{i}/1", reads[2 * i].decode()) for i in range(2500)])
        fastq(tmp_path / "r2.fq", [(f"r{i}/2", reads[2 * i + 1].decode()) for i in range(2500)])
        out = run("filter", idx, tmp_path / "r1.fq", tmp_path / "r2.fq", "-R").stdout.decode()
        keep, _, _ = oracle.filter_batch(oidx, b, o, (np.arange(5000) // 2).astype(np.uint32))
        want = [reads[2 * i + m].decode() for i in range(2500) if keep[i] for m in (0, 1)]
    else:
        fastq(tmp_path / "r.fq", [(f"r{i}", r.decode()) for i, r in enumerate(reads)])
        out = run("filter", idx, tmp_path / "r.fq", "-R").stdout.decode()
        keep, _, _ = oracle.filter_batch(oidx, b, o)
        want = [r.decode() for r, k_ in zip(reads, keep) if k_]
    lines = out.split("\n")
    assert lines[1::4][:len(want)] == want and len([l for l in lines[1::4] if l]) == len(want)
    assert [l for l in lines[0::4] if l] == [f"@{i + 1}" for i in range(len(want))]  # --rename numbers across calls


def _long_and_short_records(rng, genome, n_short=400):
    """records of every kind a piecewise scan has to get right: host-derived and random, N runs and scattered N (k-mers that
    fail the ACGT test on and around piece seams), homopolymers and short-period repeats (ties: the same position chosen
    by consecutive windows across a seam, positions that re-appear), lengths around the piece size"""
    recs = []
    for ln in (70_000, 23_000, 3_001, 2_999, 6_044, 6_045, 6_046, 45_000):
        s = int(rng.integers(0, len(genome) - ln))
        recs.append(mutate(rng, genome[s:s + ln], 0.03))
    r = bytearray(random_reads(rng, 1, 40_000, 40_000, p_n=0.02)[0])        # an N every 50 bases: hardly a clean window
    recs.append(bytes(r))
    r = bytearray(mutate(rng, genome[1000:31_000], 0.01))
    r[2_900:3_100] = b"N" * 200                                              # an N run across the first seam
    r[8_990:9_010] = b"n" * 20
    recs.append(bytes(r))
    recs.append(b"A" * 9_000 + genome[500:4_000] + b"AC" * 4_000 + b"ACGT" * 2_500 + genome[7_000:9_000])
    recs.append((b"ACGTTGCA" * 3 + b"N") * 1_500)
    for _ in range(n_short):
        ln = int(rng.integers(25, 400))
        s = int(rng.integers(0, len(genome) - ln))
        recs.append(mutate(rng, genome[s:s + ln], 0.02) if rng.random() < 0.5 else random_reads(rng, 1, ln, ln, p_n=0.01)[0])
    order = rng.permutation(len(recs))
    # (in front, in this order: a long-long pair, a long-short pair and a short-long pair for the paired mode)
    return [recs[0], recs[1], recs[2], recs[-1], recs[-2], recs[4]] + [recs[i] for i in order if i not in (0, 1, 2, 4, len(recs) - 1, len(recs) - 2)]


@gpu
@pytest.mark.parametrize("mode", ["search", "deplete", "prefix", "paired", "gzip-stdin"])
def test_records_longer_than_the_largest_call(tmp_path, oracle, monkeypatch, mode):
    """the reference takes records of any length (src/local_filter.rs:346-374).  With the largest call
    shrunk to 20 kbp and pieces of 3 kbp, a dozen records go piece by piece -- dozens of seams each, on N runs, ties and
    repeats -- among ordinary reads; kept records == the oracle's decisions on the whole records."""
    rng = np.random.default_rng(61)
    genome = random_reads(rng, 1, 80_000, 80_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    oidx = oracle.Index.build([genome])
    recs = _long_and_short_records(rng, genome)
    if len(recs) % 2:
        recs.pop()
    monkeypatch.setenv("DCN_CLI_MAX_BATCH_BASES", "20000")
    monkeypatch.setenv("DCN_CLI_GIANT_PIECE", "3000")
    b, o = oracle.concat_reads(recs)
    flags = {"search": ["-a", 2, "-r", 0.01], "deplete": ["-d", "-a", 1, "-r", 0.02], "prefix": ["-p", 7000, "-r", 0.05], "paired": ["-d"],
             "gzip-stdin": ["-a", 2, "-r", 0.01]}[mode]
    kw = {"search": dict(abs_threshold=2, rel_threshold=0.01), "deplete": dict(abs_threshold=1, rel_threshold=0.02, deplete=True),
          "prefix": dict(rel_threshold=0.05, prefix_length=7000), "paired": dict(deplete=True),
          "gzip-stdin": dict(abs_threshold=2, rel_threshold=0.01)}[mode]
    if mode == "paired":
        fasta(tmp_path / "r1.fa", [(f"r{i}/1", recs[2 * i].decode()) for i in range(len(recs) // 2)])
        fasta(tmp_path / "r2.fa", [(f"r{i}/2", recs[2 * i + 1].decode()) for i in range(len(recs) // 2)])
        out = run("filter", *flags, idx, tmp_path / "r1.fa", tmp_path / "r2.fa").stdout.decode()
        keep, _, _ = oracle.filter_batch(oidx, b, o, (np.arange(len(recs)) // 2).astype(np.uint32), **kw)
        want = [f"r{i}/{m}" for i in range(len(recs) // 2) if keep[i] for m in (1, 2)]
    else:
        # 70-column FASTA: the long records are multi-line
        (tmp_path / "r.fa").write_bytes(b"".join(b">r%d\n" % i + b"\n".join(r[j:j + 70] for j in range(0, len(r), 70)) + b"\n"
                                                 for i, r in enumerate(recs)))
        if mode == "gzip-stdin":  # the chunk reader's path: a compressed stream on stdin, chunks of 1 MB (records cross them)
            monkeypatch.setenv("DCN_CLI_CHUNK_MB", "1")
            out = run("filter", *flags, idx, "-", stdin=gzip.compress((tmp_path / "r.fa").read_bytes(), 1), env=dict(os.environ)).stdout.decode()
        else:
            out = run("filter", *flags, idx, tmp_path / "r.fa").stdout.decode()
        keep, _, _ = oracle.filter_batch(oidx, b, o, **kw)
        want = [f"r{i}" for i in range(len(recs)) if keep[i]]
    got = [l[1:] for l in out.split("\n") if l.startswith(">")]
    assert got == want and 0 < len(want) < len(recs)


@gpu
def test_debug_lines_of_a_record_longer_than_the_largest_call(tmp_path, oracle, monkeypatch):
    """--debug lists the k-mers of the hits (src/local_filter.rs:354-363): for a record that goes piece by piece the
    positions come back read-relative across the seams"""
    rng = np.random.default_rng(62)
    genome = random_reads(rng, 1, 30_000, 30_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    oidx = oracle.Index.build([genome])
    keys = set(int(x) for x in oidx.keys())
    r = bytearray(random_reads(rng, 1, 26_000, 26_000)[0])
    r[5_000:5_400] = genome[100:500]
    r[11_950:12_100] = genome[9_000:9_150]        # a hit region across a seam of 6 kbp pieces
    r[20_000:20_200] = genome[100:300]            # the same k-mers again: listed once
    reads = [bytes(r), genome[2_000:2_150], random_reads(rng, 1, 120, 120)[0]]
    fastq(tmp_path / "r.fq", [(f"r{i}", x.decode()) for i, x in enumerate(reads)])
    monkeypatch.setenv("DCN_CLI_MAX_BATCH_BASES", "12000")
    monkeypatch.setenv("DCN_CLI_GIANT_PIECE", "6000")
    err = run("filter", idx, tmp_path / "r.fq", "--debug").stderr.decode().splitlines()
    lines = [l for l in err if l.startswith("DEBUG: ")]
    assert len(lines) == 3
    for i, x in enumerate(reads):
        h, p = oracle.minimizer_hashes_and_positions(x, 31, 15)
        seen, kmers = set(), []
        for hv, pv in zip(h.tolist(), p.tolist()):
            if hv in keys and hv not in seen:
                seen.add(hv)
                kmers.append(x[pv:pv + 31].decode())
        keep = oracle.meets_filtering_criteria(len(seen), len(h), 2, 0.01, False)
        assert lines[i] == f"DEBUG: r{i} hits={len(seen)}/{len(h)} keep={'true' if keep else 'false'} kmers=[{','.join(kmers)}]", i
    assert lines[0].count(",") > 20


@gpu
def test_a_300_mbp_record_among_short_reads(tmp_path, oracle):
    """the shape VERDICT r3 names: a chromosome-sized FASTA record (300 Mbp, far beyond the 84 Mbp of the largest call) in a
    file of short reads, -a 2 -r 0.01, -r 0.001 and -d, default batch sizes; keep == the oracle's decision on the whole record"""
    rng = np.random.default_rng(63)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    genome = alpha[rng.integers(0, 4, 2_000_000)].tobytes()
    (tmp_path / "g.fa").write_bytes(b">g\n" + genome + b"\n")
    idx = tmp_path / "g.idx"
    run("index", "build", tmp_path / "g.fa", "-o", idx, "-q")
    oidx = oracle.Index.build([genome])
    big = alpha[rng.integers(0, 4, 300_000_000)]
    big[150_000_000:150_400_000] = np.frombuffer(genome[:400_000], np.uint8)     # 0.13 % of it from the index: ~0.3 % of its minimizers
    big[33_554_000:33_555_000] = ord("N")                                          # an N run where the first seam falls
    big = big.tobytes()
    shorts = [genome[s:s + 150] for s in range(0, 3000, 150)] + random_reads(rng, 20, 150, 150)
    recs = shorts[:25] + [big] + shorts[25:]
    with open(tmp_path / "r.fa", "wb") as f:
        for i, r in enumerate(recs):
            f.write(b">r%d\n" % i + r + b"\n")
    b, o = oracle.concat_reads(recs)
    # (49,866 of its 37.5 M minimizers hit: no match at the default -r 0.01, which asks for 375 k; a match at -r 0.001)
    for flags, kw in ((["-a", 2, "-r", 0.01], dict(abs_threshold=2, rel_threshold=0.01)), (["-r", 0.001], dict(rel_threshold=0.001)),
                      (["-d", "-r", 0.001], dict(deplete=True, rel_threshold=0.001))):
        keep, hits, total = oracle.filter_batch(oidx, b, o, threads=8, **kw)
        out = tmp_path / "out.fa"
        summ = tmp_path / "s.json"
        run("filter", *flags, idx, tmp_path / "r.fa", "-o", out, "-s", summ)
        got = [l[1:].decode().strip() for l in open(out, "rb") if l.startswith(b">")]
        assert got == [f"r{i}" for i in range(len(recs)) if keep[i]]
        s = json.loads(summ.read_text())
        assert s["seqs_in"] == len(recs) and s["bp_in"] == len(b)
    assert int(total[25]) > 30_000_000 and int(hits[25]) > int(total[25]) // 1000 and not keep[25]   # (the last run was -d: the match is dropped)


@gpu
def test_filter_reads_blocked_gzip_like_plain_files(tmp_path, oracle):
    """the same reads as BGZF (single file through the chunk reader; two files of mates through the record reader) and as
    plain files give the same output"""
    rng = np.random.default_rng(64)
    genome = random_reads(rng, 1, 60_000, 60_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    reads = []
    for i in range(30_000):
        ln = int(rng.integers(40, 300))
        s_ = int(rng.integers(0, len(genome) - ln))
        reads.append(mutate(rng, genome[s_:s_ + ln], 0.02) if i % 2 else random_reads(rng, 1, ln, ln)[0])
    fq = "".join(f"@r{i}\n{r.decode()}\n+\n{'I' * len(r)}\n" for i, r in enumerate(reads)).encode()
    (tmp_path / "r.fq").write_bytes(fq)
    (tmp_path / "r.fq.gz").write_bytes(bgzf_compress(fq))
    assert run("filter", idx, tmp_path / "r.fq.gz").stdout == run("filter", idx, tmp_path / "r.fq").stdout
    m1 = "".join(f"@r{i}/1\n{reads[2 * i].decode()}\n+\n{'I' * len(reads[2 * i])}\n" for i in range(15_000)).encode()
    m2 = "".join(f"@r{i}/2\n{reads[2 * i + 1].decode()}\n+\n{'I' * len(reads[2 * i + 1])}\n" for i in range(15_000)).encode()
    for name, blob in (("m1.fq", m1), ("m2.fq", m2), ("m1.fq.gz", bgzf_compress(m1)), ("m2.fq.gz", bgzf_compress(m2, block=4000))):
        (tmp_path / name).write_bytes(blob)
    a = run("filter", "-d", idx, tmp_path / "m1.fq.gz", tmp_path / "m2.fq.gz").stdout
    assert a == run("filter", "-d", idx, tmp_path / "m1.fq", tmp_path / "m2.fq").stdout and len(a) > 1000


@gpu
@pytest.mark.parametrize("paired", [False, True])
def test_multi_gpu_driver_gives_the_single_context_output(tmp_path, oracle, paired, monkeypatch):
    """`--devices 0,0,0` = three pipeline contexts on the one GPU of the box (index replicated device to device,
    calls dealt round-robin, two in flight per context, merged by sequence number): byte-identical output to the
    single-context run, same summary counters.  Mixed long + short reads, gzip in (BASELINE configs[4]'s stream)."""
    rng = np.random.default_rng(58)
    genome = random_reads(rng, 1, 80_000, 80_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    oidx = oracle.Index.build([genome])
    reads = []
    for i in range(4000):
        ln = int(rng.integers(30, 250)) if i % 50 else int(rng.integers(3_000, 30_000))
        s = int(rng.integers(0, len(genome) - ln))
        reads.append(mutate(rng, genome[s:s + ln], 0.03) if i % 2 else random_reads(rng, 1, ln, ln, p_n=0.002)[0])
    b, o = oracle.concat_reads(reads)
    monkeypatch.setenv("DCN_CLI_MAX_BATCH_READS", "500")  # many calls, so that every context gets work
    outs = {}
    # "three-ascii": the batches stay ASCII (DCN_CLI_NO_PACKED_PARSE), so the library's host pool packs them -- three contexts'
    # packing jobs side by side on its shared workers (round 4)
    for name, extra in (("one", []), ("three", ["--devices", "0,0,0"]), ("gpus1", ["--gpus", "1"]), ("three-ascii", ["--devices", "0,0,0"])):
        if name == "three-ascii":
            monkeypatch.setenv("DCN_CLI_NO_PACKED_PARSE", "1")
        summ = tmp_path / f"{name}.json"
        if paired:
            for m in (1, 2):
                with gzip.open(tmp_path / f"r{m}.fq.gz", "wb") as f:
                    f.write("".join(f"@r{i}/{m}\n{reads[2 * i + m - 1].decode()}\n+\n{'I' * len(reads[2 * i + m - 1])}\n"
                                    for i in range(2000)).encode())
            p = run("filter", "-d", idx, tmp_path / "r1.fq.gz", tmp_path / "r2.fq.gz", "-s", summ, *extra)
        else:
            with gzip.open(tmp_path / "r.fq.gz", "wb") as f:
                f.write("".join(f"@r{i}\n{r.decode()}\n+\n{'I' * len(r)}\n" for i, r in enumerate(reads)).encode())
            p = run("filter", "-d", idx, tmp_path / "r.fq.gz", "-s", summ, *extra)
        sj = json.loads(summ.read_text())
        outs[name] = (p.stdout, {k_: sj[k_] for k_ in ("seqs_in", "seqs_out", "bp_in", "bp_out", "seqs_removed", "bp_removed")})
    assert outs["three"] == outs["one"] == outs["gpus1"] == outs["three-ascii"]
    uid = (np.arange(4000) // 2).astype(np.uint32) if paired else None
    keep, _, _ = oracle.filter_batch(oidx, b, o, uid, deplete=True)
    ids = [l[1:] for l in outs["three"][0].decode().split("\n")[0::4] if l]
    want = [f"r{i}/{m}" for i in range(2000) if keep[i] for m in (1, 2)] if paired else [f"r{i}" for i in range(4000) if keep[i]]
    assert ids == want
    assert b"no such HIP device" in run("filter", idx, tmp_path / ("r1.fq.gz" if paired else "r.fq.gz"), "--devices", "0,7",
                                        check=False).stderr


@gpu
def test_parallel_parser_on_hostile_fastq(tmp_path, oracle, monkeypatch):
    """The mmap parser cuts the file at record boundaries it has to find from the middle of nowhere: qualities that
    start with '@' or '+', ids containing '+', Windows line ends and a missing final newline must not move a cut."""
    rng = np.random.default_rng(58)
    genome = random_reads(rng, 1, 30_000, 30_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    oidx = oracle.Index.build([genome])
    reads, lines = [], []
    for i in range(40_000):
        ln = int(rng.integers(40, 120))
        s = int(rng.integers(0, len(genome) - ln))
        r = mutate(rng, genome[s:s + ln], 0.02) if i % 3 else random_reads(rng, 1, ln, ln)[0]
        reads.append(r)
        q = bytearray(rng.integers(33, 74, ln, dtype=np.uint8).tobytes())
        q[0] = ord("@") if i % 2 else ord("+")
        if ln > 1 and i % 5 == 0:
            q[1] = ord("@")
        lines.append(b"@r%d +x @y\n%s\n+\n%s\n" % (i, r, bytes(q)))
    data = b"".join(lines)
    monkeypatch.setenv("DCN_CLI_CHUNK_MB", "1")
    b, o = oracle.concat_reads(reads)
    keep, _, _ = oracle.filter_batch(oidx, b, o)
    want = [f"r{i} +x @y" for i in range(len(reads)) if keep[i]]
    repeated_id = b"".join(l.replace(b"\n+\n", b"\n+" + l[1:l.index(b"\n")] + b"\n", 1) if i % 7 == 0 else l for i, l in enumerate(lines))
    for name, payload in (("unix.fq", data), ("nofinalnl.fq", data[:-1]), ("dos.fq", data.replace(b"\n", b"\r\n")),
                          ("plusid.fq", repeated_id)):
        (tmp_path / name).write_bytes(payload)
        out = run("filter", idx, tmp_path / name, "-t", 6).stdout
        got = [l[1:].rstrip(b"\r").decode() for l in out.split(b"\n")[0::4] if l]
        assert got == want, name
        # the same bytes as a stream (stdin, gzip file): the chunk reader cuts it into 1 MB pieces of whole records for
        # the same parser pool, the tail of every piece travelling to the next; the one-thread reader agrees
        assert run("filter", idx, "-", "-t", 6, stdin=payload).stdout == out, name
        (tmp_path / (name + ".gz")).write_bytes(gzip.compress(payload, 1))
        assert run("filter", idx, tmp_path / (name + ".gz"), "-t", 6).stdout == out, name
        monkeypatch.setenv("DCN_CLI_NO_CHUNK_READER", "1")
        assert run("filter", idx, "-", "-t", 6, stdin=payload).stdout == out, name
        monkeypatch.delenv("DCN_CLI_NO_CHUNK_READER")
        # the chunk parsers hand the batch over 2-bit packed (dcn_filter_batch_packed); the ASCII hand-over agrees, also
        # when a chunk is cut into several calls (a later piece's bits are moved to base 0 of a stream of its own)
        monkeypatch.setenv("DCN_CLI_NO_PACKED_PARSE", "1")
        assert run("filter", idx, tmp_path / name, "-t", 6).stdout == out, name
        monkeypatch.delenv("DCN_CLI_NO_PACKED_PARSE")
        monkeypatch.setenv("DCN_CLI_MAX_BATCH_READS", "997")
        assert run("filter", idx, tmp_path / name, "-t", 6).stdout == out, name
        assert run("filter", idx, tmp_path / (name + ".gz"), "-t", 6).stdout == out, name
        monkeypatch.delenv("DCN_CLI_MAX_BATCH_READS")
        # three ways to the same bytes: the shared output mapping (default for a plain file from a plain file: kept
        # records copied to their final place by the formatter threads), the gather writer (pipes / stdout: ranges of
        # the mapped input where a record already has its output form, formatted pieces otherwise, one writev per
        # 1024 of them) and the copying formatter + write (compressed outputs; DCN_CLI_NO_GATHER)
        for extra in ([], ["-R"]):
            run("filter", idx, tmp_path / name, "-t", 6, "-o", tmp_path / "mapped.fq", *extra)
            monkeypatch.setenv("DCN_CLI_NO_MMAP_OUT", "1")
            run("filter", idx, tmp_path / name, "-t", 6, "-o", tmp_path / "gathered.fq", *extra)
            monkeypatch.setenv("DCN_CLI_NO_GATHER", "1")
            run("filter", idx, tmp_path / name, "-t", 6, "-o", tmp_path / "streamed.fq", *extra)
            monkeypatch.delenv("DCN_CLI_NO_GATHER")
            monkeypatch.delenv("DCN_CLI_NO_MMAP_OUT")
            assert (tmp_path / "gathered.fq").read_bytes() == (tmp_path / "streamed.fq").read_bytes(), (name, extra)
            assert (tmp_path / "mapped.fq").read_bytes() == (tmp_path / "streamed.fq").read_bytes(), (name, extra)
            if not extra:
                assert (tmp_path / "gathered.fq").read_bytes() == out, name
    # FASTA in (single- and multi-line records mixed) through the mapping
    fa = b"".join((b">s%d d\n%s\n" % (i, r)) if i % 2 else (b">s%d d\n%s\n%s\n" % (i, r[:30], r[30:])) for i, r in enumerate(reads[:5000]))
    (tmp_path / "in.fa").write_bytes(fa)
    out = run("filter", idx, tmp_path / "in.fa").stdout
    run("filter", idx, tmp_path / "in.fa", "-o", tmp_path / "mapped.fa")
    assert (tmp_path / "mapped.fa").read_bytes() == out
    assert [l[1:].decode() for l in out.split(b"\n")[0::2] if l] == [f"s{i} d" for i in range(5000) if keep[i]]
    # nothing kept: an empty file, not the reservation
    run("filter", idx, tmp_path / "in.fa", "-a", 60000, "-o", tmp_path / "none.fa")
    assert (tmp_path / "none.fa").stat().st_size == 0


@gpu
def test_paired_files_in_parallel_match_the_record_reader(tmp_path, monkeypatch):
    """two plain files of mates are cut at the same record NUMBERS (their bytes differ: ids of different lengths,
    mate 2 longer than mate 1, CRLF in one file, a blank line here and there) and parsed on the worker pool; the
    record-by-record reader (DCN_CLI_NO_PAIR_MMAP) must give the same bytes for every output form"""
    rng = np.random.default_rng(77)
    genome = random_reads(rng, 1, 40_000, 40_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    n = 30_000
    l1, l2 = [], []
    for i in range(n):
        a, b = int(rng.integers(40, 100)), int(rng.integers(60, 160))
        s = int(rng.integers(0, len(genome) - 200))
        m1 = genome[s:s + a] if i % 2 else random_reads(rng, 1, a, a)[0]
        m2 = genome[s + 20:s + 20 + b] if i % 2 else random_reads(rng, 1, b, b)[0]
        l1.append(b"@p%d/1\n%s\n+\n%s\n" % (i, m1, b"I" * a) + (b"\n" if i % 5000 == 7 else b""))
        l2.append(b"@pair-%d/2 extra\n%s\n+\n%s\n" % (i, m2, b"@" * b))
    r1, r2 = tmp_path / "r1.fq", tmp_path / "r2.fq"
    r1.write_bytes(b"".join(l1))
    r2.write_bytes(b"".join(l2).replace(b"\n", b"\r\n"))
    monkeypatch.setenv("DCN_CLI_CHUNK_MB", "1")
    for extra in ([], ["-d"], ["-R"], ["-a", "1", "-r", "0.0"]):
        outs = {}
        for mode in ("pool", "reader"):
            if mode == "reader":
                monkeypatch.setenv("DCN_CLI_NO_PAIR_MMAP", "1")
            outs[mode] = [run("filter", idx, r1, r2, "-t", 5, *extra).stdout]
            run("filter", idx, r1, r2, "-t", 5, "-o", tmp_path / f"{mode}.fq", *extra)
            outs[mode].append((tmp_path / f"{mode}.fq").read_bytes())
            run("filter", idx, r1, r2, "-t", 5, "-o", tmp_path / f"{mode}1.fq", "-O", tmp_path / f"{mode}2.fq", *extra)
            outs[mode] += [(tmp_path / f"{mode}1.fq").read_bytes(), (tmp_path / f"{mode}2.fq").read_bytes()]
            monkeypatch.delenv("DCN_CLI_NO_PAIR_MMAP", raising=False)
        assert outs["pool"] == outs["reader"], extra
        assert outs["pool"][0] == outs["pool"][1] and (extra == ["-d"] or len(outs["pool"][0]) > 100_000)
        # -o / -O to two plain files goes through two output mappings (round 3); the write(2) path must agree
        monkeypatch.setenv("DCN_CLI_NO_MMAP_OUT", "1")
        run("filter", idx, r1, r2, "-t", 5, "-o", tmp_path / "w1.fq", "-O", tmp_path / "w2.fq", *extra)
        monkeypatch.delenv("DCN_CLI_NO_MMAP_OUT")
        assert [(tmp_path / "w1.fq").read_bytes(), (tmp_path / "w2.fq").read_bytes()] == outs["pool"][2:4], extra
        assert outs["pool"][2].count(b"\n") == outs["pool"][3].count(b"\n") > 0 or extra == ["-d"]  # mate for mate
        # a second output that cannot be mapped (not a regular file) sends BOTH through write(2): same first file
        run("filter", idx, r1, r2, "-t", 5, "-o", tmp_path / "n1.fq", "-O", "/dev/null", *extra)
        assert (tmp_path / "n1.fq").read_bytes() == outs["pool"][2], extra
    # a file of mates that ends early, or runs on, is an error either way
    r2.write_bytes(b"".join(l2[:-3]))
    p = run("filter", idx, r1, r2, check=False)
    assert p.returncode != 0 and b"unpaired" in p.stderr
    r2.write_bytes(b"".join(l2 + l2[:2]))
    p = run("filter", idx, r1, r2, check=False)
    assert p.returncode != 0 and b"more records" in p.stderr


@gpu
def test_malformed_inputs_get_the_same_answer_from_every_reader(tmp_path, monkeypatch):
    """the mapped parser (plain file), the chunk reader (the same bytes gzip-compressed) and the record reader
    (DCN_CLI_NO_CHUNK_READER): same exit code, same output, same last line of stderr for records cut short, quality strings of
    the wrong length, a missing '+' line, garbage, blank lines, CRLF, an empty file, a file of blank lines"""
    idx = build_index(tmp_path, [("g", SEQ1)])
    good = "@r1\n" + SEQ1[:60] + "\n+\n" + "I" * 60 + "\n"
    cases = {
        "ok": good * 3, "no final newline": (good * 3)[:-1],
        "last quality line missing": good * 2 + "@r3\nACGTACGT\n+\n",
        "last record cut in the sequence": good * 2 + "@r3\nACGTAC",
        "quality shorter": good + "@r2\nACGTACGTAC\n+\nIIII\n" + good, "quality longer": good + "@r2\nACGT\n+\nIIIIIIII\n" + good,
        "no plus line": good + "@r2\nACGTACGT\nIIIIIIII\n" + good, "starts with garbage": "xyz\n" + good,
        "garbage between records": good + "garbage\n" + good, "empty": "", "only newlines": "\n\n\n", "leading blank line": "\n" + good,
        "fasta empty sequence": ">a\n>b\n" + SEQ1 + "\n", "crlf": (good * 2).replace("\n", "\r\n"), "blank lines between": good + "\n\n" + good,
    }
    for name, text in cases.items():
        (tmp_path / "x.fq").write_bytes(text.encode())
        (tmp_path / "x.fq.gz").write_bytes(gzip.compress(text.encode(), 1))
        seen = set()
        for path, env in (("x.fq", {}), ("x.fq.gz", {}), ("x.fq.gz", {"DCN_CLI_NO_CHUNK_READER": "1"})):
            p = run("filter", idx, tmp_path / path, "-q", "-a", 1, check=False, env=dict(os.environ, **env))
            err = p.stderr.decode().strip().splitlines()
            seen.add((p.returncode, p.stdout, err[-1] if err else ""))
        assert len(seen) == 1, (name, seen)
        rc = next(iter(seen))[0]
        assert (rc == 0) == (name in ("ok", "no final newline", "empty", "only newlines", "leading blank line", "fasta empty sequence", "crlf",
                                      "blank lines between")), (name, seen)


@gpu
def test_paired_compressed_files_go_through_two_chunk_readers(tmp_path, monkeypatch):
    """R1.fastq.gz + R2.fastq.gz -- the usual shape of a short-read run: each stream has a reader thread of its own that cuts
    it into chunks of whole records; the second delivers exactly as many records per batch as the first counted, whatever
    their bytes (ids of different lengths, mate 2 longer, CRLF in one file, blank lines).  The record-by-record reader
    (DCN_CLI_NO_CHUNK_READER) must give the same bytes, for gzip + BGZF, plain + gzip, gzip + plain, and chunks far smaller
    and far larger than the files."""
    rng = np.random.default_rng(78)
    genome = random_reads(rng, 1, 40_000, 40_000)[0]
    idx = build_index(tmp_path, [("g", genome.decode())])
    n = 24_000
    l1, l2 = [], []
    for i in range(n):
        a, b = int(rng.integers(40, 100)), int(rng.integers(60, 160))
        s = int(rng.integers(0, len(genome) - 200))
        m1 = genome[s:s + a] if i % 2 else random_reads(rng, 1, a, a)[0]
        m2 = genome[s + 20:s + 20 + b] if i % 2 else random_reads(rng, 1, b, b)[0]
        l1.append(b"@p%d/1\n%s\n+\n%s\n" % (i, m1, b"I" * a) + (b"\n" if i % 5000 == 7 else b""))
        l2.append(b"@pair-%d/2 extra\n%s\n+\n%s\n" % (i, m2, b"@" * b))
    t1, t2 = b"".join(l1), b"".join(l2).replace(b"\n", b"\r\n")
    (tmp_path / "r1.fq").write_bytes(t1)
    (tmp_path / "r2.fq").write_bytes(t2)
    (tmp_path / "r1.fq.gz").write_bytes(gzip.compress(t1, 4))
    (tmp_path / "r2.fq.gz").write_bytes(gzip.compress(t2, 1))
    (tmp_path / "r2.bgzf.fq.gz").write_bytes(bgzf_compress(t2))
    want = {}
    for extra in ([], ["-d"], ["-R"]):
        monkeypatch.setenv("DCN_CLI_NO_CHUNK_READER", "1")
        want[tuple(extra)] = run("filter", idx, tmp_path / "r1.fq.gz", tmp_path / "r2.fq.gz", "-t", 5, *extra).stdout
        monkeypatch.delenv("DCN_CLI_NO_CHUNK_READER")
        assert extra == ["-d"] or len(want[tuple(extra)]) > 100_000
    for chunk_mb, gz_chunk in (("1", "40000"), ("64", None)):
        monkeypatch.setenv("DCN_CLI_CHUNK_MB", chunk_mb)
        if gz_chunk:
            monkeypatch.setenv("DCN_CLI_GZ_CHUNK", gz_chunk)
            monkeypatch.setenv("DCN_CLI_GZ_THREADS", "3")
        for f1, f2 in (("r1.fq.gz", "r2.fq.gz"), ("r1.fq.gz", "r2.bgzf.fq.gz"), ("r1.fq", "r2.fq.gz"), ("r1.fq.gz", "r2.fq")):
            for extra in ([], ["-d"], ["-R"]):
                assert run("filter", idx, tmp_path / f1, tmp_path / f2, "-t", 5, *extra).stdout == want[tuple(extra)], (chunk_mb, f1, f2, extra)
        # two output files: mate for mate
        run("filter", idx, tmp_path / "r1.fq.gz", tmp_path / "r2.fq.gz", "-o", tmp_path / "o1.fq", "-O", tmp_path / "o2.fq.gz")
        o1, o2 = (tmp_path / "o1.fq").read_bytes(), gzip.decompress((tmp_path / "o2.fq.gz").read_bytes())
        assert o1.count(b"\n") == o2.count(b"\n") > 0
        inter = want[()].split(b"\n")
        assert o1.split(b"\n")[:4] == inter[:4] and o2.split(b"\n")[:4] == inter[4:8]
        monkeypatch.delenv("DCN_CLI_GZ_CHUNK", raising=False)
        monkeypatch.delenv("DCN_CLI_GZ_THREADS", raising=False)
    monkeypatch.setenv("DCN_CLI_CHUNK_MB", "1")
    # a file of mates that ends early, or runs on, is an error (the same words as the record reader's)
    (tmp_path / "short.fq.gz").write_bytes(gzip.compress(b"".join(l2[:-3]), 1))
    p = run("filter", idx, tmp_path / "r1.fq.gz", tmp_path / "short.fq.gz", check=False)
    assert p.returncode != 0 and b"unpaired" in p.stderr
    (tmp_path / "long.fq.gz").write_bytes(gzip.compress(b"".join(l2 + l2[:2]), 1))
    p = run("filter", idx, tmp_path / "r1.fq.gz", tmp_path / "long.fq.gz", check=False)
    assert p.returncode != 0 and b"more records" in p.stderr
    # interleaved mates on stdin (plain and gzip): the chunk reader cuts an even number of records per chunk
    inter = b"".join(a + b for a, b in zip(l1, l2))
    for extra in ([], ["-d"], ["-R"]):
        for blob in (inter, gzip.compress(inter, 1)):
            assert run("filter", idx, "-", "-", "-t", 5, *extra, stdin=blob).stdout == want[tuple(extra)], extra
    monkeypatch.setenv("DCN_CLI_NO_CHUNK_READER", "1")
    assert run("filter", idx, "-", "-", "-t", 5, stdin=inter).stdout == want[()]
    monkeypatch.delenv("DCN_CLI_NO_CHUNK_READER")
    p = run("filter", idx, "-", "-", stdin=inter + l1[0], check=False)
    assert p.returncode != 0 and b"unpaired" in p.stderr
    # FASTA mates (records of several lines)
    fa1 = b"".join(b">a%d\n%s\n%s\n" % (i, genome[i * 7:i * 7 + 60], genome[i * 7 + 60:i * 7 + 90]) for i in range(3000))
    fa2 = b"".join(b">b%d\n%s\n" % (i, genome[i * 7 + 100:i * 7 + 170]) for i in range(3000))
    (tmp_path / "a.fa.gz").write_bytes(gzip.compress(fa1, 1))
    (tmp_path / "b.fa.gz").write_bytes(gzip.compress(fa2, 1))
    got = run("filter", idx, tmp_path / "a.fa.gz", tmp_path / "b.fa.gz", "-a", 1).stdout
    monkeypatch.setenv("DCN_CLI_NO_CHUNK_READER", "1")
    assert got == run("filter", idx, tmp_path / "a.fa.gz", tmp_path / "b.fa.gz", "-a", 1).stdout and got.count(b">") == 6000


# ---- BASELINE.json configs[0] at its stated shape ------------------------------------------------
def test_fastq_record_helpers_round_trip(tmp_path):
    import bench_cli
    rng = np.random.default_rng(0)
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (1000, bench_cli.READ_LEN))]
    rec = bench_cli.fastq_records(seqs, first_id=123_456_000)
    lines = rec.tobytes().split(b"\n")
    assert lines[0] == b"@123456000" and lines[1] == seqs[0].tobytes() and lines[2] == b"+" and lines[3] == b"I" * 150
    p = tmp_path / "x.fq"
    rec[::3].tofile(p)
    assert bench_cli.ids_of_output(p).tolist() == list(range(123_456_000, 123_457_000, 3))
    rec[:10].tofile(p)
    assert bench_cli.ids_of_output(p, limit_id=123_456_004).tolist() == [123_456_000 + i for i in range(4)]


@pytest.mark.gpu
def test_config1_plumbing_at_its_stated_shape(tmp_path):
    """4,641,652 bp genome -> `deacon-hip index build` -> 10,000 x 150 bp FASTQ, -a 2 -r 0.01, search and -d: the index
    key set and the ids of the kept records of both modes equal the oracle's (the leg bench.py reports as cli.plumbing)."""
    import bench_cli
    r = bench_cli.plumbing(str(tmp_path), threads=4)
    assert r["index_build"]["key_set_equals_oracle"] and r["index_build"]["keys"] > 500_000
    for mode in ("search", "deplete"):
        assert r[mode]["kept_ids_equal_oracle"], r[mode]
        assert r[mode]["seqs_in"] == 10_000 and r[mode]["bp_in"] == 1_500_000
    # the two modes are complements of each other, and the reads drawn from the genome are the ones found
    assert r["search"]["seqs_out"] + r["deplete"]["seqs_out"] == 10_000
    assert 4_900 <= r["search"]["seqs_out"] <= 5_100
    assert r["decisions_match"]


# ---- the host-side parser under AddressSanitizer / UBSan (CPU build only: GPU sanitizers are not available on the pool) ------
def test_parser_pool_under_sanitizers(tmp_path):
    """`deacon-hip bench-parse` (the parser pool of the mapped-input path, no GPU) built with -fsanitize=address,undefined
    and run over hostile FASTA / FASTQ: CRLF, '+id' lines, qualities starting with '@', blank lines, empty sequences, a
    missing final newline, truncated records, multi-line and one-base-per-line FASTA, binary garbage.  Malformed input
    must end in the tool's own error; a sanitizer report fails the test."""
    exe = tmp_path / "deacon-hip-asan"
    pkg = os.path.join(ROOT, "deacon-server_amd")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
           "-I", os.path.join(ROOT, "include"), os.path.join(pkg, "cli", "deacon_hip_cli.cpp"), "-o", str(exe),
           "-L", os.path.join(pkg, "lib"), "-ldeacon_hip", "-lz", "-ldl", "-lpthread", f"-Wl,-rpath,{os.path.join(pkg, 'lib')}",
           "-Wl,--allow-shlib-undefined"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0 and ("asan" in p.stderr.lower() or "ubsan" in p.stderr.lower() or "sanitize" in p.stderr.lower()):
        pytest.skip("no sanitizer runtime for g++ here")
    assert p.returncode == 0, p.stderr[-2000:]
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)

    def seq(n):
        return alpha[rng.integers(0, len(alpha), n)].tobytes()

    def fq(n, crlf=False, plus_id=False, at_qual=False, blank=False, no_final_nl=False, trunc=0, empty=False):
        out = []
        for i in range(n):
            ln = int(rng.integers(0 if empty else 1, 400))
            q = (b"@" + bytes(rng.integers(33, 74, ln - 1, dtype=np.uint8))) if (at_qual and ln) else b"I" * ln
            rec = b"@r%d some comment\n" % i + seq(ln) + b"\n+" + (b"r%d" % i if plus_id and i % 3 == 0 else b"") + b"\n" + q + b"\n"
            out.append(rec + (b"\n" if blank and i % 500 == 7 else b""))
        data = b"".join(out)
        if crlf:
            data = data.replace(b"\n", b"\r\n")
        if no_final_nl:
            data = data.rstrip(b"\r\n")
        return data[:len(data) - trunc] if trunc else data

    def fa(n, width, crlf=False):
        out = []
        for i in range(n):
            s = seq(int(rng.integers(1, 3000)))
            out.append(b">c%d desc\n" % i + b"\n".join(s[j:j + width] for j in range(0, len(s), width)) + b"\n")
        data = b"".join(out)
        return data.replace(b"\n", b"\r\n") if crlf else data

    n = 50_000  # ~ 20 MB: several 4 MB chunks per file
    files = {"plain.fq": (fq(n), True), "crlf.fq": (fq(n, crlf=True), True), "plus.fq": (fq(n, plus_id=True), True),
             "atq.fq": (fq(n, at_qual=True), True), "blank.fq": (fq(n, blank=True), True), "empty.fq": (fq(n, empty=True), True),
             "nofinal.fq": (fq(n, no_final_nl=True), True), "trunc1.fq": (fq(2000, trunc=1), True),
             "trunc5.fq": (fq(2000, trunc=5), False), "trunc200.fq": (fq(2000, trunc=200), False),
             "multi.fa": (fa(8000, 60), True), "multicrlf.fa": (fa(8000, 60, crlf=True), True), "w1.fa": (fa(800, 1), True),
             "garbage.fq": (b"@" + bytes(rng.integers(0, 256, 3_000_000, dtype=np.uint8)), False),
             "onlyat.fq": (b"@\n" * 50_000, False), "tiny.fq": (b"@a\nA\n+\nI", True)}
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    flags = open("/proc/cpuinfo").read()
    packs = " avx2" in flags and " bmi2" in flags
    for name, (data, ok) in files.items():
        path = tmp_path / name
        path.write_bytes(data)
        for t in (1, 5):
            r = subprocess.run([str(exe), "bench-parse", str(path), "-t", str(t)], capture_output=True, text=True, env=env, timeout=300)
            report = r.stdout + r.stderr
            assert "AddressSanitizer" not in report and "runtime error" not in report, (name, t, report[-3000:])
            if ok:
                assert r.returncode == 0 and "parsed" in r.stdout, (name, t, report[-500:])
            else:
                assert r.returncode == 1 and "Error:" in r.stderr, (name, t, report[-500:])
        # bench-parse takes the packing parser where the host has AVX2 + BMI2 (the sequences go 2-bit packed into the batch
        # while the records are parsed); the ASCII form stays the fallback, and --verify compares the two chunk by chunk
        # with dcn_pack_ascii of the ASCII form as the judge, here with chunks small enough to end inside 32-base groups
        r = subprocess.run([str(exe), "bench-parse", str(path), "-t", "3", "--ascii"], capture_output=True, text=True, env=env, timeout=300)
        report = r.stdout + r.stderr
        assert "AddressSanitizer" not in report and "runtime error" not in report, (name, report[-3000:])
        assert (r.returncode == 0 and "(ascii)" in r.stdout) if ok else (r.returncode == 1 and "Error:" in r.stderr), (name, report[-500:])
        if ok and packs:
            r = subprocess.run([str(exe), "bench-parse", str(path), "--verify"], capture_output=True, text=True, timeout=300,
                               env=dict(env, DCN_CLI_CHUNK_KB="256"))
            report = r.stdout + r.stderr
            assert "AddressSanitizer" not in report and "runtime error" not in report, (name, report[-3000:])
            assert r.returncode == 0 and "verified" in r.stdout, (name, report[-500:])
            if name.endswith(".fq"):
                assert " (0 packed)" not in r.stdout, r.stdout
