"""dcn_pack_ascii (host-side input formatting, no GPU needed): the 2-bit stream and the invalid-base mask it
writes are PackedSeqVec::from_ascii's code (c >> 1) & 3 and the mask loop of src/filter_common.rs:238-258,
stated here in numpy, for every byte value, ragged lengths and sizes that take the threaded path."""
import numpy as np
import pytest


def numpy_pack(b):
    n = len(b)
    g = (n + 31) // 32
    padded = np.full(32 * g, ord("A"), np.uint8)
    padded[:n] = b
    codes = ((padded >> 1) & 3).astype(np.uint32).reshape(-1, 16)
    packed = (codes << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32)
    low = padded | 0x20
    bad = ~((low == ord("a")) | (low == ord("c")) | (low == ord("g")) | (low == ord("t")))
    mask = (bad.reshape(-1, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=1).astype(np.uint32)
    return packed, mask


@pytest.mark.parametrize("n", [0, 1, 31, 32, 33, 63, 64, 1000, 4097, 300_001])
def test_pack_ascii_matches_the_definition(dcn, n):
    rng = np.random.default_rng(n)
    b = rng.integers(0, 256, n, dtype=np.uint8)  # every byte value, not only nucleotides
    if n > 100:
        b[::3] = np.frombuffer(b"ACGTacgtNn\n", np.uint8)[rng.integers(0, 11, len(b[::3]))]
    packed, mask = dcn.pack_ascii(b, allow_newline=True)
    want_p, want_m = numpy_pack(b)
    assert packed.tolist() == want_p.tolist()
    assert mask.tolist() == want_m.tolist()
    if (b == 10).any():  # the flag dcn_host_pack_groups returns used to be dropped
        with pytest.raises(ValueError, match="newline"):
            dcn.pack_ascii(b)
    else:
        dcn.pack_ascii(b)


def test_packed_input_with_a_line_end_is_refused(dcn):
    """A record buffer that still carries its line end must not reach the packed entry points silently: the ASCII
    ones strip it (src/filter_common.rs:229), a packed stream cannot show where reads end."""
    with pytest.raises(ValueError, match="newline"):
        dcn.pack_ascii(b"ACGTACGTAC\n" * 7)
    for n in (1, 31, 32, 33, 5000, 300_000):  # wherever the byte sits, scalar tail and vector body, every worker's share
        b = np.full(n, ord("C"), np.uint8)
        dcn.pack_ascii(b)
        b[n - 1] = 10
        with pytest.raises(ValueError):
            dcn.pack_ascii(b)
        b[n - 1], b[n // 2] = ord("C"), 10
        with pytest.raises(ValueError):
            dcn.pack_ascii(b)


def test_pack_ascii_threaded_path_and_codes(dcn):
    rng = np.random.default_rng(7)
    b = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, 6_000_000)]
    packed, mask = dcn.pack_ascii(b)
    want_p, want_m = numpy_pack(b)
    assert np.array_equal(packed, want_p) and np.array_equal(mask, want_m)
    # A=0 C=1 T=2 G=3
    p, m = dcn.pack_ascii(b"ACTGacgtN")
    assert [(int(p[0]) >> (2 * i)) & 3 for i in range(9)] == [0, 1, 2, 3, 0, 1, 3, 2, 3]
    assert int(m[0]) == 1 << 8


def test_callers_on_several_threads_share_the_host_pool(dcn):
    """Round 4: the pool runs several jobs at a time (the contexts of one process pack side by side): six caller threads,
    each packing its own inputs of its own sizes over and over through the shared workers, must each get exactly the
    single-threaded answer every time -- a slice run twice, skipped, or run with a neighbour's arguments shows here."""
    import threading
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"ACGTN", np.uint8)
    inputs = [alpha[rng.integers(0, 5, n)] for n in (5_000_000, 4_200_001, 37, 3_000_003, 4_194_304, 6_100_000)]
    want = [numpy_pack(b) for b in inputs]
    errors = []

    def work(t):
        try:
            for it in range(12):
                b = inputs[(t + it) % len(inputs)]
                p, m = dcn.pack_ascii(b)
                wp, wm = want[(t + it) % len(inputs)]
                if not (np.array_equal(p, wp) and np.array_equal(m, wm)):
                    errors.append((t, it, "differs"))
        except Exception as ex:  # noqa: BLE001
            errors.append((t, repr(ex)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
        assert not th.is_alive(), "a caller never got its job back from the pool"
    assert not errors, errors


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_pool_under_the_sanitizers(tmp_path, sanitizer):
    """tests/cpp/host_pool_test.cpp: the pool that runs several contexts' jobs side by side (csrc/dcn_host_pool.h, plain C++)
    under ThreadSanitizer and AddressSanitizer / UBSan on the CPU -- six caller threads with 1,500 jobs each (slices claimed by
    compare-exchange, workers going to sleep and waking), then more callers than the pool has slots."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "host_pool_test"
    p = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", f"-fsanitize={sanitizer}", "-fno-sanitize-recover=all",
                        "-I", os.path.join(root, "deacon-server_amd", "csrc"), "-o", str(exe),
                        os.path.join(root, "tests", "cpp", "host_pool_test.cpp")], capture_output=True, text=True)
    if p.returncode != 0 and "sanitize" in p.stderr.lower():
        pytest.skip("no sanitizer runtime for g++ here")
    assert p.returncode == 0, p.stderr[-2000:]
    for env in (dict(os.environ, DCN_HOST_THREADS="6"), dict(os.environ)):
        env.pop("DCN_HOST_SPIN_US", None)
        r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("bad 0"), (r.stdout[-300:], r.stderr[-3000:])
