"""The reference's own behavioural tests replayed on the CPU oracle.

These are the only reference-owned checks on the minimizer rule: the reference's tests assert outcomes, never
values.  The sequences and expected outcomes live in tests/golden/reference_cases.json with their file:line."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, "reference_cases.json")))


def run_case(oracle, case):
    idx = oracle.Index.build([s.encode() for s in case["ref"]], k=case["k"], w=case["w"])
    reads, uid = [], []
    for u, unit in enumerate(case["units"]):
        for s in unit:
            reads.append(s.encode())
            uid.append(u)
    b, o = oracle.concat_reads(reads)
    return idx, oracle.filter_batch(idx, b, o, np.array(uid, np.uint32), abs_threshold=case["abs"],
                                    rel_threshold=case["rel"], prefix_length=case.get("prefix_length", 0),
                                    deplete=case["deplete"])


@pytest.mark.parametrize("case", CASES["cases"], ids=[c["id"] for c in CASES["cases"]])
def test_constraint(oracle, case):
    _, (keep, hits, total) = run_case(oracle, case)
    assert keep.tolist() == case["expect_keep"], (case["id"], hits.tolist(), total.tolist())


def test_c2_strand_symmetry(oracle):
    """C-2: a read and its reverse complement select the same canonical k-mers (same hash set)."""
    c = {x["id"]: x for x in CASES["cases"]}
    fwd = c["C-1"]["units"][0][0].encode()
    rev = c["C-2"]["units"][0][0].encode()
    hf, _ = oracle.minimizer_hashes_and_positions(fwd, 31, 15)
    hr, _ = oracle.minimizer_hashes_and_positions(rev, 31, 15)
    assert len(set(hf.tolist())) >= 2
    assert set(hf.tolist()) == set(hr.tolist())


def test_c4_shared_minimizer_counted_once(oracle):
    c = {x["id"]: x for x in CASES["cases"]}["C-4"]
    idx, (keep, hits, total) = run_case(oracle, c)
    assert len(idx) == 1 and hits.tolist() == [1] and total[0] > 2


def test_c6_homopolymer_index_is_hash_of_zero(oracle):
    idx = oracle.Index.build([b"A" * 20], k=5, w=5)
    assert idx.keys().tolist() == [0xC77B3ABB6F87ACD9]  # xxh3_64(0u64): the k-mer AAAAA


def test_c10_index_build_deterministic(oracle):
    c = {x["id"]: x for x in CASES["cases"]}["C-7"]
    a = oracle.Index.build([s.encode() for s in c["ref"]], k=41, w=15)
    b = oracle.Index.build([s.encode() for s in c["ref"]], k=41, w=15)
    assert sorted(a.keys().tolist()) == sorted(b.keys().tolist()) and len(a) > 0
    with pytest.raises(ValueError):
        oracle.Index.build([b"ACGT" * 30], k=31, w=16)  # k + w - 1 even (src/index.rs:186-194)


def test_c11_iupac_and_entropy(oracle):
    for ch, want in CASES["iupac"]["map"].items():
        assert chr(oracle.canonicalise_nucleotide(ord(ch))) == want
    for kmer, k, lo, hi in CASES["entropy"]["bands"]:
        e = oracle.scaled_entropy(kmer.encode(), k)
        assert lo <= e <= hi, (kmer, e)


def test_index_side_skips_non_acgt_and_low_entropy(oracle):
    # src/minimizers.rs:151-168: ACGT test on the original bytes, optional entropy floor
    seq = b"ACGTTGCAAGCTTGCATGCCGATAGCTAGCTAGGATCGATCGNACGATCGATGCTAGCTAGCTAGGCTAGCTAGCTAGCATCGATCGATCGACTAGCTAGC"
    h_all = oracle.index_minimizer_hashes(seq.replace(b"N", b"C"), 31, 15)
    h_n = oracle.index_minimizer_hashes(seq, 31, 15)
    assert 0 < len(h_n) <= len(h_all)
    low = b"A" * 60
    assert len(oracle.index_minimizer_hashes(low, 31, 15, 0.0)) > 0
    assert len(oracle.index_minimizer_hashes(low, 31, 15, 0.5)) == 0
