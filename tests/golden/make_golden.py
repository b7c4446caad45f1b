#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference (Rust, crate deacon 0.10.0) cannot be built or run in this image (no cargo/rustc),
so no fixture here comes from executing it.  Three kinds of data:

  xxh3_kat.json          known answers of XXH3-64 for 8- and 16-byte little-endian inputs, produced by the
                         independent C xxHash library (python `xxhash`) -- pins A6's hash stage.
  reference_cases.json   the sequence literals and expected outcomes of the reference's own behavioural tests
                         (tests/filter_tests.rs, src/minimizers.rs unit tests): DATA transcribed with file:line,
                         the "constraint suite" C-1..C-11 of SURVEY.md 4.1 -- the only reference-owned checks.
  oracle_vectors.json    seeded sequences with the positions / hashes / decisions computed by oracle/ (the CPU
                         restatement).  ORACLE-DERIVED, i.e. "parity unpinned" against the real crates: they
                         guard against regressions and give the GPU tests a committed target.
"""
import json
import os
import struct
import sys

import numpy as np
import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def xxh3_kat():
    rng = np.random.default_rng(20260101)
    v64 = [0, 1, 2, 0x2AA, 0x3FFFFFFFFFFFFFFF, 0x0123456789ABCDEF, 0xFFFFFFFFFFFFFFFF]
    v64 += [int(x) for x in rng.integers(0, 2**63, 64, dtype=np.uint64)]
    v64 += [int(x) & ((1 << 62) - 1) for x in rng.integers(0, 2**63, 64, dtype=np.uint64)]
    v128 = [0, 1, (1 << 82) - 1, 0x0123456789ABCDEF0123456789ABCDEF]
    for _ in range(64):
        a, b = (int(x) for x in rng.integers(0, 2**63, 2, dtype=np.uint64))
        v128.append(((a << 64) | b) & ((1 << 114) - 1))
    return {
        "source": "python xxhash %s (C xxHash), xxh3_64 seed 0" % xxhash.VERSION,
        "u64_le": [[hex(v), hex(xxhash.xxh3_64_intdigest(struct.pack("<Q", v)))] for v in v64],
        "u128_le": [[hex(v), hex(xxhash.xxh3_64_intdigest(v.to_bytes(16, "little")))] for v in v128],
    }


SC2_0_60 = "ATTAAAGGTTTATACCTTCCCAGGTAACAAACCAACCAACTTTCGATCTCTTGTAGATCT"      # tests/filter_tests.rs:43-47
SC2_0_60_REV = "AGATCTACAAGAGATCGAAAGTTGGTTGGTTTGTTACCTGGGAAGGTATAAACCTTTAAT"  # :57-63
SC2_60_120 = "GTTCTCTAAACGAACTTTAAAATCTGTGTGGCTGTCACTCGGCTGCATGCTTAGTGCACT"    # :73-77
SC2_60_120_REV = "AGTGCACTAAGCATGCAGCCGAGTGACAGCCACACAGATTTTAAAGTTCGTTTAGAGAAC"  # :88-92
TEST_FASTA_SEQ1 = ("ACGTGCATAGCTGCATGCATGCATGCATGCATGCATGCAATGCAACGTGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATG"
                   "CATGCATGCATGCATGCATGCATGCATGCATGCATGCA")  # :9
TEST_FASTA_SEQ2 = ("TGCAGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATTGCAGCATGCATG"
                   "CATGCATGCATGCATGCATGCATGCATGCATGCATGC")  # :9


def reference_cases():
    """Each case: index built from `ref` (index-side rule, src/minimizers.rs:125) with (k, w); `units` filtered with
    the given flags; `expect_keep` is what the reference test asserts about the output."""
    return {
        "source": "literals and assertions of /root/reference/tests/filter_tests.rs and src/minimizers.rs (data only)",
        "cases": [
            {"id": "C-1", "cite": "tests/filter_tests.rs:586-622", "k": 31, "w": 15, "ref": [SC2_0_60],
             "units": [[SC2_0_60]], "abs": 1, "rel": 0.01, "deplete": True, "expect_keep": [False]},
            {"id": "C-2", "cite": "tests/filter_tests.rs:625-657", "k": 31, "w": 15, "ref": [SC2_0_60],
             "units": [[SC2_0_60_REV]], "abs": 2, "rel": 0.01, "deplete": True, "expect_keep": [False]},
            {"id": "C-3-fwd", "cite": "tests/filter_tests.rs:660-690", "k": 31, "w": 15, "ref": [SC2_0_60],
             "units": [[SC2_0_60, SC2_60_120]], "abs": 2, "rel": 0.01, "deplete": True, "expect_keep": [False]},
            {"id": "C-3-rev", "cite": "tests/filter_tests.rs:693-723", "k": 31, "w": 15, "ref": [SC2_0_60],
             "units": [[SC2_0_60_REV, SC2_60_120_REV]], "abs": 2, "rel": 0.01, "deplete": True,
             "expect_keep": [False]},
            {"id": "C-4", "cite": "tests/filter_tests.rs:943-1015", "k": 31, "w": 15, "ref": ["ACGT" * 28],
             "units": [["A" * 9 + "ACGT" * 16 + "A" * 10, "T" * 10 + "ACGT" * 16 + "T" * 10]],
             "abs": 2, "rel": 0.01, "deplete": True, "expect_keep": [True]},
            {"id": "C-5", "cite": "tests/filter_tests.rs:1133-1187", "k": 31, "w": 1,
             "ref": ["ACGTTTAAGGCCAACCACACACACACACATT"], "units": [["ACGTTTAAGGCCAACCACACACACACACATT"]],
             "abs": 1, "rel": 0.01, "deplete": False, "expect_keep": [True]},
            {"id": "C-6", "cite": "tests/filter_tests.rs:1190-1251", "k": 5, "w": 5, "ref": ["A" * 20],
             "units": [["AAAAACAAAAACAAAAACAAAAA"]], "abs": 1, "rel": 0.0, "deplete": False,
             "expect_keep": [False]},
            {"id": "C-7", "cite": "tests/filter_tests.rs:1254-1296", "k": 41, "w": 15,
             "ref": [TEST_FASTA_SEQ1, TEST_FASTA_SEQ2], "units": [[TEST_FASTA_SEQ1], [TEST_FASTA_SEQ2]],
             "abs": 1, "rel": 0.0, "deplete": False, "expect_keep": [True, True]},
            {"id": "C-8", "cite": "tests/filter_tests.rs:92-128", "k": 31, "w": 15, "ref": ["A" * 100],
             "units": [[TEST_FASTA_SEQ1], [TEST_FASTA_SEQ2]], "abs": 2, "rel": 0.01, "deplete": False,
             "expect_keep": [False, False]},
            {"id": "C-9", "cite": "tests/filter_tests.rs:315-341", "k": 31, "w": 15,
             "ref": [TEST_FASTA_SEQ1, TEST_FASTA_SEQ2], "units": [[TEST_FASTA_SEQ1], [TEST_FASTA_SEQ2]],
             "abs": 2, "rel": 0.01, "deplete": False, "prefix_length": 6, "expect_keep": [False, False]},
        ],
        "iupac": {"cite": "src/minimizers.rs:197-231",
                  "map": {"A": "A", "C": "C", "G": "G", "T": "T", "a": "A", "c": "C", "R": "G", "Y": "C", "S": "G",
                          "W": "A", "K": "G", "M": "C", "B": "C", "D": "G", "H": "C", "V": "G", "N": "C"}},
        "entropy": {"cite": "src/minimizers.rs:251-386", "bands": [
            ["ACGT", 8, 1.0, 1.0], ["AAAAAAAAAA", 10, 0.0, 0.1], ["ATATATATAT", 10, 0.5, 0.9999],
            ["ACGTACGTAC", 10, 0.9, 1.0], ["ACGTACGTACGTACGTACGTACGTACGTACG", 31, 0.9, 1.0],
            ["AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", 31, 0.0, 0.01], ["AAAAAAAAAAACAAAAAGAAAAATAAAAAAA", 31, 0.25, 0.35],
            ["GCGCGCGCGCGCGCGCGCGCGCGCGCGCGCG", 31, 0.45, 0.55], ["ATATATATATATATATATATATATATATATG", 31, 0.55, 0.65],
            ["ACGACGACGACGACGACGACGACGACGACGA", 31, 0.75, 0.85], ["ACGTACGTACGTAAAACCCGGGTTTACGTAC", 31, 0.8, 1.0],
            ["AACCGGTTAACCGGTTAACCGGTTAACCGGT", 31, 0.95, 1.0], ["AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAT", 31, 0.0, 0.15]]},
    }


def oracle_vectors():
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = {"note": "ORACLE-DERIVED (oracle/deacon_oracle.c); parity with the real crates is unpinned", "vectors": []}
    configs = [(31, 15), (31, 15), (31, 15), (15, 11), (41, 15), (5, 5), (31, 1), (21, 9)]
    for i, (k, w) in enumerate(configs):
        n = int(rng.integers(k + w - 1, 400))
        s = alpha[rng.integers(0, 4, n)].copy()
        if i % 3 == 1:
            s[rng.random(n) < 0.02] = ord("N")
        if i % 3 == 2:
            s[rng.random(n) < 0.3] |= 0x20
        seq = s.tobytes()
        h, p = O.minimizer_hashes_and_positions(seq, k, w)
        out["vectors"].append({"k": k, "w": w, "seq": seq.decode(), "positions": [int(x) for x in p],
                               "hashes": [hex(int(x)) for x in h]})
    # low-complexity stress: periodic reads exercise ties / left-right alternation / re-emitted positions
    for seq in [TEST_FASTA_SEQ1, TEST_FASTA_SEQ2, "ACGT" * 40, "A" * 120, "AT" * 70, "GGGCCC" * 25]:
        h, p = O.minimizer_hashes_and_positions(seq.encode(), 31, 15)
        out["vectors"].append({"k": 31, "w": 15, "seq": seq, "positions": [int(x) for x in p],
                               "hashes": [hex(int(x)) for x in h]})
    return out


def main():
    for name, fn in (("xxh3_kat.json", xxh3_kat), ("reference_cases.json", reference_cases),
                     ("oracle_vectors.json", oracle_vectors)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, indent=1)
        print("wrote", name)


if __name__ == "__main__":
    main()
