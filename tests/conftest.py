import os
import sys

import numpy as np
import pytest

try:  # PyTorch bundles its own HIP runtime (same SONAME as /opt/rocm's): whichever library is loaded first
    import torch  # noqa: F401  decides which runtime the process uses, so load torch before libdeacon_hip.so
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _build_if_missing():
    """lib/, bin/ and build/ are build products (git-ignored): a fresh checkout that runs the tests before
    __graft_entry__.build() gets the library and the tool built here, once (make under a file lock).  This is test set-up;
    the product itself still refuses to run without its library (_native.lib())."""
    pkg = os.path.join(ROOT, "deacon-server_amd")
    if os.path.exists(os.path.join(pkg, "lib", "libdeacon_hip.so")) and os.path.exists(os.path.join(pkg, "bin", "deacon-hip")):
        return
    import fcntl
    import subprocess
    with open(os.path.join(pkg, "csrc", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.call(["make", "-C", os.path.join(pkg, "csrc"), "-j8"], stdout=subprocess.DEVNULL)
        except OSError:
            pass  # (no make here: the tests that need the library say what is missing)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _build_if_missing()


def _gpu_count():
    try:
        import deacon_server_amd as d
        import ctypes
        n = ctypes.c_int()
        if d._native.lib().dcn_device_count(ctypes.byref(n)) != 0:
            return 0
        return n.value
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: only unmarked runs skip gpu tests
    if config.getoption("-m") and "gpu" in config.getoption("-m") and "not gpu" not in config.getoption("-m"):
        return
    if _gpu_count() == 0:
        skip = pytest.mark.skip(reason="no GPU in this container (run with -m gpu on the GPU box)")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def dcn():
    import deacon_server_amd as d
    return d


def random_reads(rng, n, min_len, max_len, p_n=0.0, p_lower=0.0, alphabet=b"ACGT"):
    reads = []
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    for _ in range(n):
        ln = int(rng.integers(min_len, max_len + 1))
        s = alpha[rng.integers(0, len(alpha), ln)].copy()
        if p_n > 0 and ln:
            m = rng.random(ln) < p_n
            s[m] = ord("N")
        if p_lower > 0 and ln:
            m = rng.random(ln) < p_lower
            s[m] |= 0x20
        reads.append(s.tobytes())
    return reads


def mutate(rng, seq, rate):
    s = np.frombuffer(seq, dtype=np.uint8).copy()
    m = rng.random(len(s)) < rate
    s[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
    return s.tobytes()


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(seq):
    return bytes(seq).translate(_COMP)[::-1]


# ---- synthetic index remainder with decidable membership (also used by bench.py) ----------------------------------
_M1, _M2 = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def mix64(i):
    """splitmix64's finalizer on a numpy uint64 array: a bijection on u64"""
    z = np.asarray(i, dtype=np.uint64).copy()
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def unmix64(h):
    """its inverse: h is in {mix64(i) : 1 <= i <= n} iff 1 <= unmix64(h) <= n"""
    h = np.asarray(h, dtype=np.uint64)

    def inv_xs(y, s):
        x = y.copy()
        for _ in range(64 // s + 1):
            x = y ^ (x >> np.uint64(s))
        return x
    with np.errstate(over="ignore"):
        z = inv_xs(h, 31) * np.uint64(pow(_M2, -1, 1 << 64))
        z = inv_xs(z, 27) * np.uint64(pow(_M1, -1, 1 << 64))
        return inv_xs(z, 30)
