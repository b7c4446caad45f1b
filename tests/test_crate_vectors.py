"""Consumes tests/golden/crate_vectors.json -- the output of tests/golden/dump_crate_vectors (a Rust program that
calls the real simd-minimizers 1.3.0 / packed-seq 3.2.1 / xxhash-rust 0.8.15 exactly as src/filter_common.rs:238-307
does).  The build image has no Rust toolchain, so the file is absent here and these tests skip; with the file in
place they are the value-level pin the oracle lacks ("parity unpinned"):

  * the oracle must reproduce every vector under at least one of its eight settings, the test names which, and it
    fails unless that setting is the default (fix = change the default, see dump_crate_vectors/Cargo.toml);
  * on the GPU the product, switched to the same setting, must reproduce every vector as well.
"""
import json
import os

import pytest

from conftest import GOLDEN

PATH = os.path.join(GOLDEN, "crate_vectors.json")
needs_file = pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/crate_vectors.json not generated "
                                "(needs cargo: tests/golden/dump_crate_vectors)")


def load():
    vec = json.load(open(PATH))["vectors"]
    assert len(vec) >= 10
    return vec


def matching_variants(oracle, vec):
    ok = []
    try:
        for v in oracle.VARIANTS:
            oracle.set_variant(*v)
            good = True
            for x in vec:
                seq = x["seq"].encode()
                if "raw_positions" in x and oracle.canonical_minimizer_positions(seq, x["k"], x["w"]).tolist() != x["raw_positions"]:
                    good = False
                    break
                h, p = oracle.minimizer_hashes_and_positions(seq, x["k"], x["w"])
                if p.tolist() != x["positions"] or [hex(int(q)) for q in h] != x["hashes"]:
                    good = False
                    break
            if good:
                ok.append(v)
    finally:
        oracle.set_variant(*oracle.DEFAULT_VARIANT)
    return ok


@needs_file
def test_oracle_reproduces_the_crates(oracle):
    ok = matching_variants(oracle, load())
    assert ok, "no setting of the oracle reproduces the crates' vectors: a rule beyond the three switches differs"
    assert oracle.DEFAULT_VARIANT in ok, (
        f"the crates follow {ok} (rotation, compared bits, combine; every setting listed reproduces all vectors), not "
        f"the default {oracle.DEFAULT_VARIANT}: change the defaults in oracle/deacon_oracle.c and csrc/scan.hip")


@needs_file
@pytest.mark.gpu
def test_gpu_reproduces_the_crates(oracle, dcn):
    import numpy as np
    vec = load()
    ok = matching_variants(oracle, vec)
    assert ok
    v = oracle.DEFAULT_VARIANT if oracle.DEFAULT_VARIANT in ok else ok[0]
    dcn.set_minimizer_variant(*v)
    try:
        by_kw = {}
        for x in vec:
            by_kw.setdefault((x["k"], x["w"]), []).append(x)
        for (k, w), xs in by_kw.items():
            idx = dcn.Index.from_keys(np.array([1], np.uint64), k, w)
            reads = [x["seq"].encode() for x in xs]
            b, o = oracle.concat_reads(reads)
            proc = dcn.FilterProcessor(idx, max_batch_bases=len(b) + 64, max_batch_reads=len(reads) + 1)
            off, h, p = proc.minimizer_hashes_batch(b, o)
            for r, x in enumerate(xs):
                lo, hi = int(off[r]), int(off[r + 1])
                assert p[lo:hi].tolist() == x["positions"] and [hex(int(q)) for q in h[lo:hi]] == x["hashes"], (k, w, r)
            proc.close()
            idx.close()
    finally:
        dcn.set_minimizer_variant(*oracle.DEFAULT_VARIANT)
