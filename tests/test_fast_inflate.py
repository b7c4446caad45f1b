"""The command-line tool's own gzip decoder (deacon-server_amd/cli/fast_inflate.hpp) against zlib, through `deacon-hip cat`
(= the tool's input side alone, no GPU).  The reference reads compressed inputs through its readers' decoders
(src/filter_common.rs:40-76 of the reference: paraseq / niffler over flate2); what has to hold here is that the bytes that
come out are the bytes zlib makes of the same file, for every kind of deflate block and gzip header, and that a damaged
file is an error and never different bytes."""
import gzip
import os
import random
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "deacon-server_amd", "bin", "deacon-hip")
ZLIB = dict(os.environ, DCN_CLI_ZLIB_INFLATE="1")
# the decoder three ways: one stream on several threads (parallel_gzip.hpp) with chunks so small that a test file is dozens of
# them and single blocks are longer than a chunk's overlap; the same with the shipped chunk size; one thread (fast_inflate.hpp alone)
SMALL_CHUNKS = dict(os.environ, DCN_CLI_GZ_THREADS="3", DCN_CLI_GZ_CHUNK="30000")
ONE_THREAD = dict(os.environ, DCN_CLI_NO_PARALLEL_GZ="1")
WAYS = {"small-chunks": SMALL_CHUNKS, "default": dict(os.environ, DCN_CLI_GZ_THREADS="4"), "one-thread": ONE_THREAD}


def cat(blob, env=None, count=False):
    p = subprocess.run([BIN, "cat", "-"] + (["--count"] if count else []), input=blob, capture_output=True, env=env)
    return p.returncode, p.stdout, p.stderr.decode()


def fastq_text(rng, n, quals=b"I"):
    alpha = np.frombuffer(b"ACGT", np.uint8)
    out = []
    for i in range(n):
        m = int(rng.integers(30, 300))
        q = bytes(rng.choice(np.frombuffer(quals, np.uint8), m)) if len(quals) > 1 else quals * m
        out.append(b"@r%d\n%s\n+\n%s\n" % (i, alpha[rng.integers(0, 4, m)].tobytes(), q))
    return b"".join(out)


def member(data, flg=0, extra=b"", name=b"", comment=b"", level=6):
    """one gzip member with the optional header fields FLG names (RFC 1952 2.3)"""
    raw = zlib.compress(data, level)[2:-4]
    h = bytearray(b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\xff")
    if flg & 4:
        h += struct.pack("<H", len(extra)) + extra
    if flg & 8:
        h += name + b"\0"
    if flg & 16:
        h += comment + b"\0"
    if flg & 2:
        h += struct.pack("<H", zlib.crc32(bytes(h)) & 0xFFFF)
    return bytes(h) + raw + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


@pytest.fixture(scope="module")
def payloads():
    rng = np.random.default_rng(5)
    return {
        "empty": b"", "one": b"x",
        "fastq": fastq_text(rng, 20_000),
        "fastq-quals": fastq_text(rng, 8_000, quals=bytes(range(33, 74))),
        "zeros": bytes(3_000_000),
        "random": rng.integers(0, 256, 1_000_000, dtype=np.uint8).tobytes(),
        "text": b"the quick brown fox jumps over the lazy dog. " * 30_000,
        "runs": b"".join(bytes([i % 251]) * (i % 700 + 1) for i in range(4000)),
        "period3": b"abc" * 300_000, "period7": b"abcdefg" * 200_000,
        "skewed": bytes(rng.choice([65, 66, 67, 200, 201, 7], 1_000_000, p=[.9, .05, .02, .01, .01, .01]).astype(np.uint8)),
        # 286 literal/length and 30 distance codes all in use, code lengths up to 15: long codes past the tables' bits
        "geometric": bytes(np.minimum(rng.geometric(0.08, 1_500_000), 255).astype(np.uint8)),
    }


def test_every_kind_of_block_decodes_to_zlibs_bytes(payloads):
    """stored (level 0), fixed-Huffman, dynamic at several levels, RLE / Huffman-only / filtered strategies, and streams cut
    into many blocks by sync and full flushes (empty stored blocks between them)"""
    for name, data in payloads.items():
        blobs = {f"level{lv}": gzip.compress(data, lv) for lv in (0, 1, 6, 9)}
        for sname, strat in (("fixed", zlib.Z_FIXED), ("rle", zlib.Z_RLE), ("huffman", zlib.Z_HUFFMAN_ONLY), ("filtered", zlib.Z_FILTERED)):
            c = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strat)
            blobs[sname] = c.compress(data) + c.flush()
        c = zlib.compressobj(5, zlib.DEFLATED, 31)
        parts = []
        for i in range(0, len(data), 7777):
            parts += [c.compress(data[i:i + 7777]), c.flush(zlib.Z_SYNC_FLUSH if i % 3 else zlib.Z_FULL_FLUSH)]
        blobs["flushed"] = b"".join(parts) + c.flush()
        for kind, blob in blobs.items():
            for way in ("small-chunks", "one-thread"):
                rc, out, err = cat(blob, env=WAYS[way])
                assert rc == 0 and out == data, (name, kind, way, rc, len(out), len(data), err)


def test_members_header_fields_and_padding(payloads):
    fq = payloads["fastq"]
    blob = (member(fq[:100_000], 8, name=b"a.fq") + member(b"") +
            member(fq[100_000:300_000], 4 | 8 | 16 | 2, extra=b"XY\x03\x00abc", name=b"n", comment=b"c", level=1) + member(fq[300_000:], level=9))
    assert cat(blob, env=ZLIB)[:2] == (0, fq)
    # a header longer than one read of the input (an extra field of 60 KB and a long name)
    big = member(fq[:50_000], 4 | 8, extra=b"ZZ" + struct.pack("<H", 60_000) + bytes(60_000), name=b"n" * 70_000)
    # many small members that are not BGZF (no length field): each ends inside some chunk
    many = b"".join(member(fq[i:i + 40_000], level=1 + i % 9) for i in range(0, len(fq), 40_000))
    for way, env in WAYS.items():
        assert cat(blob, env=env)[:2] == (0, fq), way
        assert cat(big + member(fq[50_000:]), env=env)[:2] == (0, fq), way
        assert cat(many, env=env)[:2] == (0, fq), way
        # a header field that lies about the header: FHCRC
        lying = bytearray(member(fq[:1000], 2))
        lying[10] ^= 1
        assert cat(bytes(lying), env=env)[0] == 1, way
        # what follows the last member: zlib's reader refuses anything that is not a member; so does ours, with the same words
        for tail in (bytes(100), b"garbage"):
            a, b = cat(blob + tail, env=env), cat(blob + tail, env=ZLIB)
            assert a[0] == b[0] == 1 and "invalid gzip stream" in a[2] and "invalid gzip stream" in b[2], way


def test_large_input_crosses_the_decoders_buffers(payloads):
    """more than the decoder's 8 MB output buffer and 4 MB input buffer, blocks and matches across every refill; a slow pipe
    that delivers the file in small pieces"""
    data = payloads["fastq-quals"] * 12 + payloads["random"] * 3 + payloads["text"] * 4
    for lv in (1, 6):
        blob = gzip.compress(data, lv)
        for way, env in WAYS.items():
            rc, out, err = cat(blob, env=env)
            assert rc == 0 and out == data, (lv, way, err)
    # stored blocks only, larger than the input buffer in total (nothing for the block search to find: the driver decodes alone)
    for way, env in WAYS.items():
        assert cat(gzip.compress(data[:20_000_000], 0), env=env)[1] == data[:20_000_000], way
    blob = gzip.compress(data[:6_000_000], 6)
    p = subprocess.Popen([BIN, "cat", "-"], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    import threading
    got = []
    t = threading.Thread(target=lambda: got.append(p.stdout.read()))
    t.start()
    for i in range(0, len(blob), 1013):
        p.stdin.write(blob[i:i + 1013])
        if i % (1013 * 64) == 0:
            p.stdin.flush()
    p.stdin.close()
    t.join()
    assert p.wait() == 0 and got[0] == data[:6_000_000]


def test_damaged_files_are_errors_never_other_bytes(payloads):
    fq = payloads["fastq"]
    blob = gzip.compress(fq, 6)
    for cut in (1, 5, 12, 1000, len(blob) // 2, len(blob) - 9, len(blob) - 3):
        for way, env in WAYS.items():
            rc, out, err = cat(blob[:cut], count=True, env=env)
            if cut < 2:
                continue  # (a byte that is not a gzip magic is plain text to the tool)
            assert rc == 1 and "truncated gzip stream" in err, (cut, way, rc, err)
    random.seed(3)
    light = gzip.compress(fq[:300_000], 1)
    accepted = 0
    for t in range(120):
        b = bytearray(blob[:200_000]) if t % 2 else bytearray(light)
        for _ in range(random.randint(1, 4)):
            i = random.randrange(10, len(b))
            b[i] ^= 1 << random.randrange(8)
        rc, out, err = cat(bytes(b), env=SMALL_CHUNKS if t % 4 < 2 else ONE_THREAD)
        assert rc in (0, 1), (t, rc, err)  # never a signal
        if rc == 0:  # (rare: the flips cancelled out or hit a header field nobody checks) -- then zlib reads the same bytes
            accepted += 1
            assert out == gzip.decompress(bytes(b)), t
    assert accepted < 10


def test_blocked_gzip_members_take_the_same_decoder(payloads):
    """BGZF members are inflated side by side (tests/test_cli.py has the member logic); here: the fast decoder on members of
    every block kind, against zlib's on the same file"""
    from test_cli import bgzf_compress
    for name in ("fastq", "fastq-quals", "random", "zeros", "geometric", "empty"):
        data = payloads[name][:2_000_000]
        blob = bgzf_compress(data)
        for env in (None, ZLIB, dict(os.environ, DCN_CLI_BGZF_THREADS="3")):
            rc, out, err = cat(blob, env=env)
            assert rc == 0 and out == data, (name, err)
    # members of a few bytes: their last symbols are shorter than what one lookup may see (a decoder that asks for a lookup's
    # worth of bits AHEAD of each symbol refuses them -- this one did, before the tool's own compressor showed it)
    tiny = payloads["fastq"][:3000] + b"I" * 200
    for block in (1, 7, 16, 33):
        for env in (None, ZLIB):
            assert cat(bgzf_compress(tiny, block=block), env=env)[:2] == (0, tiny), block
    # a member whose CRC or length field lies
    blob = bytearray(bgzf_compress(payloads["fastq"][:200_000]))
    first = (blob[16] | blob[17] << 8) + 1
    blob[first - 8] ^= 1
    rc, out, err = cat(bytes(blob))
    assert rc == 1 and "gzip stream" in err


def test_decoder_stays_inside_its_buffers_under_the_sanitizers(tmp_path):
    """tests/cpp/fast_inflate_test.cpp with AddressSanitizer and UBSan (CPU build: the decoder is host code)"""
    exe = tmp_path / "fast_inflate_test"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "deacon-server_amd", "cli"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "cpp", "fast_inflate_test.cpp"), "-lz"], check=True)
    p = subprocess.run([str(exe)], capture_output=True)
    assert p.returncode == 0 and p.stdout.strip().endswith(b"bad 0"), (p.stdout[-500:], p.stderr[-3000:])


@pytest.mark.parametrize("sanitizer,rounds", [("address,undefined", 36), ("thread", 10)])
def test_parallel_reader_under_the_sanitizers(tmp_path, sanitizer, rounds):
    """tests/cpp/parallel_gzip_test.cpp: one stream on 2-4 threads with 20 KB and 150 KB chunks, whole / cut short / damaged,
    under AddressSanitizer + UBSan and under ThreadSanitizer (CPU builds: the reader is host code)"""
    exe = tmp_path / "parallel_gzip_test"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", f"-fsanitize={sanitizer}", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "deacon-server_amd", "cli"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "cpp", "parallel_gzip_test.cpp"), "-lz"], check=True)
    p = subprocess.run([str(exe), str(rounds)], capture_output=True)
    assert p.returncode == 0 and p.stdout.strip().endswith(b"bad 0"), (p.stdout[-500:], p.stderr[-3000:])


def gz_write(data, level=2, env=None):
    """`deacon-hip gz [level]`: the .gz writer alone -- stdin to BGZF members on stdout, as the formatter threads write them"""
    p = subprocess.run([BIN, "gz", str(level)], input=data, capture_output=True, env=env)
    assert p.returncode == 0, p.stderr
    return p.stdout


def test_gz_writer_own_compressor_and_zlib(payloads):
    """.gz outputs at levels 1-3 go through the tool's own compressor (cli/fast_deflate.hpp), 4+ and DCN_CLI_ZLIB_DEFLATE=1 through
    zlib: either way python's gzip, zlib's inflate and the tool's own readers give the input back; the file is BGZF (every
    member <= 64 KB with its size in the BC field, the empty member last); what does not compress is stored, not blown up; the
    own compressor is not larger than zlib at the same level on FASTQ"""
    from test_cli import bgzf_compress  # noqa: F401  (the same layout the reader tests write)
    for name, data in payloads.items():
        data = data[:3_000_000]
        for level, env in ((1, None), (2, None), (3, None), (2, dict(os.environ, DCN_CLI_ZLIB_DEFLATE="1")), (5, None)):
            blob = gz_write(data, level, env)
            assert gzip.decompress(blob) == data, (name, level)
            assert cat(blob)[:2] == (0, data), (name, level)
            # members: header with the BC subfield, sizes that add up, the end-of-file member
            at, n_members = 0, 0
            while at < len(blob):
                assert blob[at:at + 4] == b"\x1f\x8b\x08\x04" and blob[at + 12:at + 16] == b"BC\x02\x00", (name, at)
                size = (blob[at + 16] | blob[at + 17] << 8) + 1
                assert 26 <= size <= 65536
                at += size
                n_members += 1
            # (bgzip / htslib know an untruncated file by these exact 28 bytes)
            assert at == len(blob) and blob[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
            assert n_members == (len(data) + 65279) // 65280 + 1 or len(data) == 0
            assert len(blob) <= len(data) + (34 if env is None and level <= 3 else 64) * n_members + 64, (name, level, len(blob), len(data))
    fq = payloads["fastq-quals"]
    assert len(gz_write(fq, 2)) <= len(gz_write(fq, 2, dict(os.environ, DCN_CLI_ZLIB_DEFLATE="1"))) * 1.01


def test_compressor_under_the_sanitizers(tmp_path):
    """tests/cpp/fast_deflate_test.cpp: 4,000 inputs of 0 ... 65,535 bytes through the compressor into buffers of exactly
    bound(n) bytes, read back by zlib and by the tool's decoder, under AddressSanitizer + UBSan"""
    exe = tmp_path / "fast_deflate_test"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "deacon-server_amd", "cli"), "-o", str(exe),
                    os.path.join(ROOT, "tests", "cpp", "fast_deflate_test.cpp"), "-lz"], check=True)
    p = subprocess.run([str(exe)], capture_output=True)
    assert p.returncode == 0 and p.stdout.strip().endswith(b"bad 0"), (p.stdout[-500:], p.stderr[-3000:])
