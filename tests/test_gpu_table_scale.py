"""BASELINE.json configs[4] at its real table size: a 950 M-key device set (2^31 groups of two slots, 34 GB -- the
panmouse-1a u panhuman-1 union), membership checked on sampled present / absent keys, then a mixed long + short
--deplete batch filtered against it and compared with the CPU oracle.

The oracle cannot hold 950 M keys in the time a test has, and does not need to: the index is
host-genome minimizers u {mix64(i) : 1 <= i <= n}, mix64 being a bijection on u64, so whether one of the sample's
minimizer hashes is in the index follows from the index's DEFINITION (sorted host keys + unmix64).  The oracle
then runs on the set of exactly those index keys the sample can touch; every other key of the table is
irrelevant to these reads."""
import time

import numpy as np
import pytest

from conftest import mix64, mutate, random_reads, revcomp, unmix64

N_KEYS = 950_000_000


def test_mix64_is_a_bijection_with_this_inverse():
    i = np.concatenate([np.arange(1, 10_000, dtype=np.uint64), np.array([2**63, 2**64 - 1, 12345678901234567], np.uint64)])
    assert (unmix64(mix64(i)) == i).all()
    assert int(mix64(np.array([1], np.uint64))[0]) == 0x5692161D100B05E5


@pytest.mark.gpu
def test_950m_key_table_membership_and_mixed_deplete_batch(oracle, dcn):
    torch = pytest.importorskip("torch")
    t0 = time.time()
    rng = np.random.default_rng(900)
    genome = random_reads(rng, 1, 2_000_000, 2_000_000)[0]
    host_keys = np.sort(oracle.Index.build([genome], k=31, w=15).keys())
    n_rand = N_KEYS - len(host_keys)
    keys = np.empty(N_KEYS, np.uint64)
    keys[:len(host_keys)] = host_keys
    dev = torch.device("cuda:0")

    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)

    def signed(x):
        return x - (1 << 64) if x >= (1 << 63) else x
    step = 1 << 27
    for a in range(0, n_rand, step):  # mix64(1 + a ..) on the device, as bench.py does
        m = min(step, n_rand - a)
        z = torch.arange(1 + a, 1 + a + m, dtype=torch.int64, device=dev)
        z = (z ^ lsr(z, 30)) * signed(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * signed(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        keys[len(host_keys) + a:len(host_keys) + a + m] = z.cpu().numpy().view(np.uint64)
        del z
    assert (keys[len(host_keys):len(host_keys) + 1000] == mix64(np.arange(1, 1001, dtype=np.uint64))).all()
    torch.cuda.empty_cache()
    t_keys = time.time() - t0
    gidx = dcn.Index.from_keys(keys, 31, 15)
    t_table = time.time() - t0 - t_keys
    # distinct: a host key could coincide with a mix64 key (it does not, in 2^64)
    assert gidx.n_keys == N_KEYS

    # ---- membership: sampled present keys, keys of the bijection just outside the index, perturbed host keys --------
    present = keys[rng.integers(0, N_KEYS, 1_000_000)]
    assert gidx.contains(present).all()
    del keys
    absent = mix64(np.uint64(n_rand) + 1 + rng.integers(0, 1 << 40, 1_000_000).astype(np.uint64))
    perturbed = host_keys ^ np.uint64(1)
    for probe in (absent, perturbed):
        pos = np.minimum(np.searchsorted(host_keys, probe), len(host_keys) - 1)
        want = (host_keys[pos] == probe) | ((unmix64(probe) >= 1) & (unmix64(probe) <= np.uint64(n_rand)))
        assert gidx.contains(probe).tolist() == want.tolist()
        assert want.sum() < 10  # practically all absent

    # ---- a mixed long + short --deplete batch (configs[4]'s stream shape) ---------------------------------------------
    reads = []
    for i in range(20_000):
        ln = int(rng.integers(60, 260)) if i % 60 else int(min(200_000, max(2_000, rng.lognormal(8.9, 0.8))))
        if i % 2:
            s = int(rng.integers(0, len(genome) - ln))
            r = mutate(rng, genome[s:s + ln], 0.03 if ln > 1000 else 0.005)
            reads.append(revcomp(r) if i % 4 == 1 else r)
        else:
            reads.append(random_reads(rng, 1, ln, ln, p_n=0.001)[0])
    reads[11] = genome[100_000:400_000]  # one long exact read: > 2800 hits, the distinct pass's global-set path
    b, o = oracle.concat_reads(reads)
    uid = None
    # index keys the sample can touch
    hs = [np.asarray(oracle.minimizer_hashes_and_positions(r, 31, 15)[0], dtype=np.uint64) for r in reads]
    h = np.unique(np.concatenate([x for x in hs if len(x)]))
    pos = np.minimum(np.searchsorted(host_keys, h), len(host_keys) - 1)
    touch = h[(host_keys[pos] == h) | ((unmix64(h) >= 1) & (unmix64(h) <= np.uint64(n_rand)))]
    small = oracle.Index(touch, 31, 15)
    assert gidx.contains(h).tolist() == np.isin(h, touch).tolist()  # the table agrees hash by hash
    proc = dcn.FilterProcessor(gidx, deplete=True, max_batch_bases=len(b) + 64, max_batch_reads=len(reads))
    want = oracle.filter_batch(small, b, o, uid, deplete=True, threads=8)
    got = proc.filter_batch(b, o, uid)
    assert got[2].tolist() == want[2].tolist()
    assert got[1].tolist() == want[1].tolist()
    assert got[0].tolist() == want[0].tolist()
    assert got[1][11] > 2800 and 0 < got[0].sum() < len(reads)
    assert proc.filter_batch(b, o, uid, counts=False).tolist() == want[0].tolist()
    # the same batch as pairs, packed on the host first
    uid = (np.arange(len(reads)) // 2).astype(np.uint32)
    wantp = oracle.filter_batch(small, b, o, uid, deplete=True, threads=8)
    packed, mask = dcn.pack_ascii(b)
    gotp = proc.filter_batch_packed(packed, mask, o, uid)
    assert gotp[1].tolist() == wantp[1].tolist() and gotp[0].tolist() == wantp[0].tolist()
    proc.close()
    gidx.close()
    print(f"950M-key table: keys {t_keys:.1f} s, table {t_table:.1f} s, whole test {time.time() - t0:.1f} s")
    assert time.time() - t0 < 90
