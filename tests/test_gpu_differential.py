"""Randomized differential test: GPU (through the C ABI) vs the oracle over random parameters and inputs --
k, w, thresholds, prefix lengths, unit groupings, alphabets (N runs, IUPAC, lower case, newlines), read-length
mixes that cross every tile / flush / ring boundary.  Seeds are fixed; every case prints its parameters on failure."""
import os

import numpy as np
import pytest

from conftest import mutate, random_reads, revcomp

pytestmark = pytest.mark.gpu

KW = [(31, 15), (31, 15), (15, 11), (41, 15), (5, 5), (21, 9), (31, 1), (13, 7), (27, 19), (56, 2), (32, 16)]


def random_case(rng, genome):
    k, w = KW[int(rng.integers(0, len(KW)))]
    n = int(rng.integers(1, 400))
    style = int(rng.integers(0, 4))
    reads = []
    for _ in range(n):
        if style == 0:
            ln = int(rng.integers(0, 320))
        elif style == 1:
            ln = int(rng.choice([k - 1, k, k + w - 2, k + w - 1, k + w, 150, 151, 250]))
        elif style == 2:
            ln = int(min(40_000, max(1, rng.lognormal(6.5, 1.2))))
        else:
            ln = int(rng.integers(100, 3000))
        ln = min(ln, len(genome) - 1)
        r = rng.random()
        if r < 0.45 and ln > 0:
            s = int(rng.integers(0, len(genome) - ln))
            x = mutate(rng, genome[s:s + ln], float(rng.choice([0.0, 0.01, 0.1])))
            if rng.random() < 0.5:
                x = revcomp(x)
        elif r < 0.55 and ln > 0:
            unit = random_reads(rng, 1, 1, 12)[0]
            x = (unit * (ln // len(unit) + 1))[:ln]  # low complexity: ties, re-emitted positions
        else:
            x = random_reads(rng, 1, ln, ln, p_n=float(rng.choice([0.0, 0.001, 0.05])),
                             p_lower=float(rng.choice([0.0, 0.3])),
                             alphabet=b"ACGT" if rng.random() < 0.8 else b"ACGTNRYKMSWBDHV")[0]
        if rng.random() < 0.02 and x:
            x = x + b"\n"
        reads.append(x)
    grouping = int(rng.integers(0, 3))
    if grouping == 0:
        uid = None
    elif grouping == 1:
        uid = (np.arange(n) // 2).astype(np.uint32)
    else:
        uid = np.cumsum(rng.random(n) < 0.6).astype(np.uint32)
        uid -= uid[0]
    params = dict(abs_threshold=int(rng.choice([1, 1, 2, 2, 3, 10])), rel_threshold=float(rng.choice([0.0, 0.01, 0.01, 0.2, 0.5, 1.0])),
                  prefix_length=int(rng.choice([0, 0, 0, 5, 60, 200, 5000])), deplete=bool(rng.integers(0, 2)))
    return k, w, reads, uid, params


@pytest.mark.parametrize("seed", range(int(os.environ.get("DCN_FUZZ_SEEDS", "12"))))  # more seeds for a soak run
def test_differential(oracle, dcn, seed, monkeypatch):
    if seed % 3 == 2:  # (a third of the seeds through the three-stream form of the host entry points: see test_gpu_host_pipeline.py)
        monkeypatch.setenv("DCN_LEAN_MAX_BASES", "0")
    run_seed(oracle, dcn, seed, monkeypatch, 10)


from oracle.oracle import DEFAULT_VARIANT, VARIANTS  # noqa: E402


@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: "rot%d-cmp%d-%s" % v)
def test_differential_under_every_minimizer_variant(oracle, dcn, variant, monkeypatch):
    """The parity-pinning switch (dcn_set_minimizer_variant / dor_set_variant): GPU == oracle under each of the eight
    settings of (ntHash rotation, compared hash bits, fw/rc combination), so that a run of the real crates which
    contradicts the default costs a switch in both, not a rewrite (tests/test_crate_vectors.py)."""
    oracle.set_variant(*variant)
    dcn.set_minimizer_variant(*variant)
    try:
        assert dcn.get_minimizer_variant() == variant
        for seed in (100 + VARIANTS.index(variant), 200 + VARIANTS.index(variant)):
            run_seed(oracle, dcn, seed, monkeypatch, 5)
    finally:
        oracle.set_variant(*DEFAULT_VARIANT)
        dcn.set_minimizer_variant(*DEFAULT_VARIANT)


def test_the_minimizer_variant_travels_with_the_index(oracle, dcn):
    """dcn_set_minimizer_variant is process-wide, but an index captures the setting it was created under: a context
    filters by its index's rule whatever the setting has become since, and set algebra refuses mixed operands."""
    rng = np.random.default_rng(5)
    genome = random_reads(rng, 1, 30_000, 30_000)[0]
    reads = [genome[s:s + 200] for s in range(0, 20_000, 170)] + random_reads(rng, 50, 150, 150)
    b, o = oracle.concat_reads(reads)
    other = (7, 32, "xor")
    try:
        oi_def = oracle.Index.build([genome])
        gi_def = dcn.Index.build([genome], 31, 15)                     # created under the default rules
        want_def = oracle.filter_batch(oi_def, b, o)
        oracle.set_variant(*other)
        dcn.set_minimizer_variant(*other)
        oi_var = oracle.Index.build([genome])
        gi_var = dcn.Index.build([genome], 31, 15)                     # created under (7, 32, xor)
        want_var = oracle.filter_batch(oi_var, b, o)
        assert sorted(gi_var.keys().tolist()) == sorted(oi_var.keys().tolist()) != sorted(oi_def.keys().tolist())
        for gi, want in ((gi_def, want_def), (gi_var, want_var)):      # the process-wide setting is still (7, 32, xor)
            proc = dcn.FilterProcessor(gi, max_batch_bases=len(b) + 64, max_batch_reads=len(reads) + 1)
            got = proc.filter_batch(b, o)
            assert all(g.tolist() == w.tolist() for g, w in zip(got, want))
            proc.close()
        dcn.set_minimizer_variant(*DEFAULT_VARIANT)                     # ... and after switching back as well
        proc = dcn.FilterProcessor(gi_var, max_batch_bases=len(b) + 64, max_batch_reads=len(reads) + 1)
        assert all(g.tolist() == w.tolist() for g, w in zip(proc.filter_batch(b, o), want_var))
        proc.close()
        with pytest.raises(dcn.DeaconHipError, match="different minimizer rules"):
            dcn.Index.union([gi_def, gi_var])
        with pytest.raises(dcn.DeaconHipError, match="different minimizer rules"):
            gi_def.diff(gi_var)
        assert sorted(gi_var.clone(0).keys().tolist()) == sorted(oi_var.keys().tolist())
    finally:
        oracle.set_variant(*DEFAULT_VARIANT)
        dcn.set_minimizer_variant(*DEFAULT_VARIANT)


def run_seed(oracle, dcn, seed, monkeypatch, n_cases):
    rng = np.random.default_rng(1000 + seed)
    genome = random_reads(rng, 1, 60_000, 60_000)[0]
    monkeypatch.setenv("DCN_TILE_WINDOWS", str(int(rng.choice([16, 64, 256, 512, 2048]))))
    # host pipeline geometry: chunk seams anywhere in the batch, staging pieces from tiny to default
    monkeypatch.setenv("DCN_CHUNK_BASES", str(int(rng.choice([1024, 5000, 40_000, 1 << 26]))))
    monkeypatch.setenv("DCN_STAGE_BYTES", str(int(rng.choice([4096, 65536, 32 << 20]))))
    if rng.random() < 0.3:
        monkeypatch.setenv("DCN_NO_HOST_PACK", "1")
    # table load: 2 slots per key makes most probes of a present key walk past a full home group, 8 almost none
    monkeypatch.setenv("DCN_TABLE_SLOTS_PER_KEY", str(int(np.random.default_rng(seed).choice([2, 4, 8]))))
    indexes = {}
    for case in range(n_cases):
        k, w, reads, uid, params = random_case(rng, genome)
        if (k, w) not in indexes:
            oidx = oracle.Index.build([genome], k=k, w=w)
            indexes[(k, w)] = (oidx, dcn.Index.from_keys(oidx.keys(), k, w))
        oidx, gidx = indexes[(k, w)]
        b, o = oracle.concat_reads(reads)
        want = oracle.filter_batch(oidx, b, o, uid, **params)
        proc = dcn.FilterProcessor(gidx, max_batch_bases=max(len(b), 1) + 64, max_batch_reads=len(reads) + 1, **params)
        got = proc.filter_batch(b, o, uid)
        ctx = (seed, case, k, w, len(reads), None if uid is None else "grouped", params)
        assert got[2].tolist() == want[2].tolist(), ("total", ctx)
        assert got[1].tolist() == want[1].tolist(), ("hits", ctx)
        assert got[0].tolist() == want[0].tolist(), ("keep", ctx)
        assert proc.filter_batch(b, o, uid, counts=False).tolist() == want[0].tolist(), ("keep, decisions only", ctx)
        if case % 3 == 0:  # minimizer hashes / positions of every read as well
            off, h, p = proc.minimizer_hashes_batch(b, o)
            for r, s in enumerate(reads):
                wh, wp = oracle.minimizer_hashes_and_positions(s, k, w, prefix_length=params["prefix_length"])
                lo, hi = int(off[r]), int(off[r + 1])
                assert p[lo:hi].tolist() == wp.tolist() and h[lo:hi].tolist() == wh.tolist(), ("minimizers", r, ctx)
        proc.close()
