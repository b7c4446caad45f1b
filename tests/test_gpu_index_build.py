"""f1: index build on the GPU -- index::build / minimizers::fill_minimizer_hashes / write_minimizers --
against the oracle's index-side restatement.  Key SETS must be identical."""
import numpy as np
import pytest

from conftest import random_reads

pytestmark = pytest.mark.gpu

IUPAC = b"ACGTNRYSWKMBDHVacgtnryswkmbdhv"


def messy_sequences(rng, n, min_len, max_len):
    seqs = []
    alpha = np.frombuffer(IUPAC, dtype=np.uint8)
    for _ in range(n):
        ln = int(rng.integers(min_len, max_len + 1))
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, ln)].copy()
        m = rng.random(ln) < 0.01
        s[m] = alpha[rng.integers(0, len(alpha), int(m.sum()))]
        if rng.random() < 0.3:
            s[rng.random(ln) < 0.5] |= 0x20
        seqs.append(s.tobytes())
    return seqs


@pytest.mark.parametrize("k,w", [(31, 15), (15, 11), (41, 15), (5, 5), (21, 9)])
def test_index_build_matches_oracle(oracle, dcn, k, w):
    rng = np.random.default_rng(k * 10 + w)
    seqs = messy_sequences(rng, 40, 0, 5000) + [b"", b"ACGT", b"N" * 200, b"ACGTNNNNNACGT" * 30, b"A" * 300]
    want = oracle.Index.build(seqs, k=k, w=w)
    got = dcn.Index.build(seqs, k, w)
    assert got.header() == (k, w, len(want))
    assert sorted(got.keys().tolist()) == sorted(want.keys().tolist())


def test_index_build_long_sequences_and_chunk_seams(oracle, dcn, monkeypatch):
    rng = np.random.default_rng(5)
    seqs = messy_sequences(rng, 3, 150_000, 400_000) + messy_sequences(rng, 20, 100, 3000)
    want = sorted(oracle.Index.build(seqs).keys().tolist())
    for chunk in ("4096", "50000", "1000003"):
        monkeypatch.setenv("DCN_BUILD_CHUNK_BASES", chunk)
        got = dcn.Index.build(seqs)
        assert sorted(got.keys().tolist()) == want, chunk
    monkeypatch.delenv("DCN_BUILD_CHUNK_BASES")
    got = dcn.Index.build(seqs, capacity_keys=10)  # tiny hint: the table must grow and rehash
    assert sorted(got.keys().tolist()) == want


@pytest.mark.parametrize("thr", [0.01, 0.3, 0.5, 0.75, 0.95])
def test_index_build_entropy_floor(oracle, dcn, thr):
    rng = np.random.default_rng(8)
    seqs = messy_sequences(rng, 20, 200, 4000)
    seqs += [b"A" * 100 + random_reads(rng, 1, 200, 200)[0] + b"AT" * 60 + b"ACG" * 40 + b"AAAAAAAAAAT" * 12,
             b"GCGCGCGCGCGCGCGCGCGCGCGCGCGCGCGCGCGC" * 5 + b"AACCGGTT" * 20]
    want = oracle.Index.build(seqs, entropy_threshold=thr)
    got = dcn.Index.build(seqs, entropy_threshold=thr)
    assert sorted(got.keys().tolist()) == sorted(want.keys().tolist())
    assert len(want) < len(oracle.Index.build(seqs)) or thr < 0.02


def test_index_file_roundtrip_both_ways(oracle, dcn, tmp_path):
    rng = np.random.default_rng(9)
    seqs = messy_sequences(rng, 10, 1000, 20_000)
    built = dcn.Index.build(seqs, 31, 15)
    p1 = tmp_path / "gpu.idx"
    built.write(str(p1))
    via_oracle = oracle.Index.read(p1)  # the oracle's bincode reader accepts the GPU-written file
    assert (via_oracle.k, via_oracle.w) == (31, 15)
    assert sorted(via_oracle.keys().tolist()) == sorted(built.keys().tolist())
    reloaded = dcn.Index.from_file(str(p1))
    assert reloaded.header() == built.header()
    assert reloaded.contains(built.keys()).all()
    # file size: 3 header bytes + varint count + 9 bytes per (large) hash, as README.md:52 implies
    n = built.n_keys
    assert abs(p1.stat().st_size - (3 + 3 + 9 * n)) <= 8 + n // 1000
    # and the other direction
    p2 = tmp_path / "cpu.idx"
    oracle.Index.build(seqs).write(p2)
    assert sorted(dcn.Index.from_file(str(p2)).keys().tolist()) == sorted(built.keys().tolist())


def test_built_index_filters_like_oracle_index(oracle, dcn):
    """End to end: build on GPU, filter on GPU == build on CPU, filter on CPU."""
    rng = np.random.default_rng(10)
    genome = messy_sequences(rng, 1, 100_000, 100_000)[0]
    oidx = oracle.Index.build([genome])
    gidx = dcn.Index.build([genome])
    reads = []
    for i in range(3000):
        ln = int(rng.integers(50, 250))
        if i % 2:
            s = int(rng.integers(0, len(genome) - ln))
            reads.append(genome[s:s + ln])
        else:
            reads.append(random_reads(rng, 1, ln, ln)[0])
    b, o = oracle.concat_reads(reads)
    want = oracle.filter_batch(oidx, b, o, deplete=True)
    proc = dcn.FilterProcessor(gidx, deplete=True, max_batch_bases=1 << 20, max_batch_reads=4096)
    got = proc.filter_batch(b, o)
    for g, w_ in zip(got, want):
        assert g.tolist() == w_.tolist()


def test_union_and_diff_set_algebra(oracle, dcn):
    rng = np.random.default_rng(12)
    a = rng.integers(0, 2**63, 200_000, dtype=np.uint64)
    b = np.concatenate([a[50_000:120_000], rng.integers(0, 2**63, 100_000, dtype=np.uint64), np.zeros(1, np.uint64)])
    c = np.concatenate([a[:10], np.zeros(1, np.uint64)])
    ia, ib, ic = (dcn.Index.from_keys(x, 31, 15) for x in (a, b, c))
    sa, sb, sc = set(a.tolist()), set(b.tolist()), set(c.tolist())
    u = dcn.Index.union([ia, ib, ic])
    assert u.n_keys == len(sa | sb | sc) and set(u.keys().tolist()) == sa | sb | sc
    assert set(ia.diff(ib).keys().tolist()) == sa - sb
    assert set(ib.diff(ia).keys().tolist()) == sb - sa          # keeps key 0 (only in b)
    assert set(ib.diff(ic).keys().tolist()) == sb - sc          # drops key 0 (in both)
    assert ia.diff(ia).n_keys == 0
    assert dcn.Index.union([ia]).n_keys == len(sa)
    other = dcn.Index.from_keys(a[:10], 15, 11)
    with pytest.raises(dcn.DeaconHipError):
        dcn.Index.union([ia, other])
    with pytest.raises(dcn.DeaconHipError):
        ia.diff(other)


def test_table_sizing_rule_changes_memory_not_results(dcn, monkeypatch):
    """the table is a power-of-two number of 16-byte groups with >= S slots per key (S = 8 while the table stays
    small against the device, 4 otherwise, DCN_TABLE_SLOTS_PER_KEY fixes it): dcn_index_memory reports the size;
    membership, the key export and a clone do not depend on it, also at a load where most probes walk"""
    rng = np.random.default_rng(77)
    keys = np.unique(np.concatenate([rng.integers(0, 2**64, 300_000, dtype=np.uint64), np.zeros(1, np.uint64)]))
    absent = np.setdiff1d(rng.integers(0, 2**64, 100_000, dtype=np.uint64), keys)
    sizes = {}
    for s in (None, 2, 4, 8, 16):
        if s is None:
            monkeypatch.delenv("DCN_TABLE_SLOTS_PER_KEY", raising=False)
        else:
            monkeypatch.setenv("DCN_TABLE_SLOTS_PER_KEY", str(s))
        idx = dcn.Index.from_keys(keys, 31, 15)
        sizes[s] = idx.table_bytes
        assert idx.n_keys == len(keys)
        assert idx.contains(keys).all() and not idx.contains(absent).any()
        assert np.array_equal(np.sort(idx.keys()), keys)
        rep = idx.clone(0)
        assert rep.table_bytes == idx.table_bytes and rep.contains(keys[::7]).all()
        rep.close()
        idx.close()
    groups = lambda s: max(64, 1 << int(np.ceil(np.log2((len(keys) * s + 8) / 2))))  # noqa: E731
    assert sizes[None] == sizes[8] == 16 * groups(8)
    assert [sizes[s] for s in (2, 4, 16)] == [16 * groups(s) for s in (2, 4, 16)]
