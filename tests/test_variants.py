"""The parity-pinning switch on the CPU side: the three details of the minimizer rule that the reference's tests
cannot separate (SURVEY.md 8a "Notes on A4": ntHash rotation 1 | 7, 16 | 32 compared hash bits, fw + rc | fw ^ rc)
are run-time switches of the oracle (dor_set_variant) and of the product (dcn_set_minimizer_variant).  Here: every
setting of the oracle agrees with a pure-Python statement of the same setting, the eight settings are pairwise
different on ordinary input, and the reference's behavioural cases hold under every one of them -- which is exactly
why only a run of the real crates (tests/golden/dump_crate_vectors -> tests/test_crate_vectors.py) can choose."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, random_reads
from test_reference_constraints import CASES, run_case


@pytest.fixture
def variant(oracle, request):
    oracle.set_variant(*request.param)
    yield request.param
    oracle.set_variant(*oracle.DEFAULT_VARIANT)


def py_positions(seq, k, w, rot, bits, comb):
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
            fw ^= rotl(F[c[j + i]], rot * (k - 1 - i))
            rc ^= rotl(F[c[j + i] ^ 2], rot * i)
        x = (fw ^ rc) if comb == "xor" else ((fw + rc) & 0xFFFFFFFF)
        h.append(x >> 16 if bits == 16 else x)
    out = []
    for i in range(n - l + 1):
        tg = sum(1 for x in c[i:i + l] if x & 2)
        win = h[i:i + w]
        m = min(win)
        best = i + (win.index(m) if 2 * tg > l else (w - 1 - win[::-1].index(m)))
        if not out or out[-1] != best:
            out.append(best)
    return out


from oracle.oracle import VARIANTS  # noqa: E402


@pytest.mark.parametrize("variant", VARIANTS, indirect=True, ids=lambda v: "rot%d-cmp%d-%s" % v)
def test_oracle_variant_matches_python_statement(oracle, variant):
    rng = np.random.default_rng(99)
    for k, w in [(31, 15), (5, 5), (15, 11), (41, 15), (31, 1), (7, 3)]:
        reads = random_reads(rng, 6, 0, 150, p_n=0.02, p_lower=0.1) + [b"A" * 100, b"ACGT" * 30, b"GCATGCAT" * 15]
        for s in reads:
            a = oracle.canonical_minimizer_positions(s, k, w)
            b = oracle.canonical_minimizer_positions(s, k, w, naive=True)
            assert a.tolist() == b.tolist() == py_positions(s, k, w, *variant)


def test_variants_differ_from_each_other(oracle):
    rng = np.random.default_rng(5)
    seq = random_reads(rng, 1, 400_000, 400_000)[0]  # long enough for 16-bit ties at a window minimum
    seen = {}
    try:
        for v in VARIANTS:
            oracle.set_variant(*v)
            seen[v] = tuple(oracle.canonical_minimizer_positions(seq, 31, 15).tolist())
    finally:
        oracle.set_variant(*oracle.DEFAULT_VARIANT)
    assert len(set(seen.values())) == len(VARIANTS)
    # ... and the committed oracle vectors belong to the default
    vec = json.load(open(os.path.join(GOLDEN, "oracle_vectors.json")))["vectors"][0]
    assert oracle.minimizer_hashes_and_positions(vec["seq"].encode(), vec["k"], vec["w"])[1].tolist() == vec["positions"]


@pytest.mark.parametrize("variant", VARIANTS, indirect=True, ids=lambda v: "rot%d-cmp%d-%s" % v)
def test_reference_cases_hold_under_every_variant(oracle, variant):
    """The reference's behavioural tests do not choose between the eight settings (C-6 only rejects seed-table
    permutations): every case passes under each."""
    for case in CASES["cases"]:
        _, (keep, hits, total) = run_case(oracle, case)
        assert keep.tolist() == case["expect_keep"], (variant, case["id"], hits.tolist(), total.tolist())


def test_tuned_port_equals_port(oracle):
    """bench.py's second CPU leg (dor_filter_batch_tuned_mt) is the same function as the plain port."""
    rng = np.random.default_rng(21)
    genome = random_reads(rng, 1, 120_000, 120_000)[0]
    for k, w in [(31, 15), (15, 11), (5, 5), (31, 1), (41, 15), (32, 14), (56, 2)]:
        idx = oracle.Index.build([genome], k, w)
        reads = []
        for i in range(500):
            ln = int(rng.integers(1, 400)) if i % 10 else int(rng.integers(2000, 12_000))
            if i % 2 == 0:
                s = int(rng.integers(0, len(genome) - ln))
                r = bytearray(genome[s:s + ln])
            else:
                r = bytearray(random_reads(rng, 1, ln, ln, p_n=0.003, p_lower=0.01)[0])
            if i % 7 == 0:
                r += b"\n"
            reads.append(bytes(r))
        reads += [b"ACGT" * 100, b"A" * 300, b""]
        b, o = oracle.concat_reads(reads)
        for uid in (None, (np.arange(len(reads)) // 2).astype(np.uint32), (np.arange(len(reads)) // 3).astype(np.uint32)):
            for pl in (0, 100):
                for dep in (False, True):
                    want = oracle.filter_batch(idx, b, o, uid, 2, 0.01, pl, dep, threads=2)
                    got = oracle.filter_batch(idx, b, o, uid, 2, 0.01, pl, dep, threads=3, tuned=True)
                    for x, y in zip(want, got):
                        assert x.tolist() == y.tolist(), (k, w, pl, dep)
    oracle.set_variant(7, 16, "add")
    try:
        with pytest.raises(ValueError):
            oracle.filter_batch(idx, b, o, None, tuned=True)  # the tuned port knows the default rules only
    finally:
        oracle.set_variant(*oracle.DEFAULT_VARIANT)
