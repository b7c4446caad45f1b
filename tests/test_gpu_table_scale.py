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


@pytest.mark.gpu
def test_one_batch_beyond_2_to_the_32_bases(oracle, dcn):
    """maximum sizes: ONE device-resident batch of 4.43 Gbp (base offsets, run slots and packed-stream bit positions
    above 2^32), short reads first and 100 kbp host-derived reads at its far end (their hit runs sit in record slots
    above 2^32).  The same reads in five pieces of < 2^30 bases through a small context must give the same keep /
    hits / totals; the reads at the far end are also compared with the oracle."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4321)
    genome = random_reads(rng, 1, 3_000_000, 3_000_000)[0]
    oidx = oracle.Index.build([genome], k=31, w=15)
    gidx = dcn.Index.from_keys(oidx.keys(), 31, 15)
    g_dev = torch.from_numpy(np.frombuffer(genome, np.uint8).copy()).to(dev)
    n_short, L, n_long, LL = 27_500_000, 150, 3_000, 100_000
    n_reads = n_short + n_long
    n_bases = n_short * L + n_long * LL
    assert n_bases > (1 << 32) + (1 << 26)
    d_bases = torch.empty(n_bases, dtype=torch.uint8, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    ar = torch.arange(L, device=dev, dtype=torch.int32)
    step = 2_500_000
    for a in range(0, n_short, step):  # half of the short reads from the genome, half random
        m = min(step, n_short - a)
        st = torch.randint(0, len(genome) - L, (m,), device=dev, generator=gen, dtype=torch.int32)
        mat = g_dev[(st[:, None] + ar[None, :]).long()]
        rnd = acgt[torch.randint(0, 4, (m, L), device=dev, generator=gen)]
        host = torch.rand(m, device=dev, generator=gen) < 0.5
        d_bases[a * L:(a + m) * L] = torch.where(host[:, None], mat, rnd).reshape(-1)
        del st, mat, rnd, host
    long_starts = rng.integers(0, len(genome) - LL, n_long)
    for i, s0 in enumerate(long_starts):  # long reads: verbatim genome segments, every 17th with a few N
        o = n_short * L + i * LL
        d_bases[o:o + LL] = g_dev[s0:s0 + LL]
        if i % 17 == 0:
            d_bases[o + 5_000:o + 5_003] = ord("N")
    offsets = np.concatenate([np.arange(n_short, dtype=np.uint64) * np.uint64(L),
                              np.uint64(n_short * L) + np.arange(n_long + 1, dtype=np.uint64) * np.uint64(LL)])
    assert int(offsets[-1]) == n_bases and len(offsets) == n_reads + 1
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    out = {k_: torch.zeros(n_reads, dtype=t_, device=dev) for k_, t_ in (("keep", torch.uint8), ("hits", torch.int32), ("total", torch.int32))}
    big = dcn.FilterProcessor(gidx, max_batch_bases=n_bases, max_batch_reads=n_reads)
    big.reserve_records(n_long * LL // 4)
    for counts in (True, False):
        big.filter_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, n_bases, out["keep"].data_ptr(),
                                out["hits"].data_ptr() if counts else None, out["total"].data_ptr() if counts else None)
        big.synchronize()
        if counts:
            keep_counting = out["keep"].clone()
        else:
            assert torch.equal(out["keep"], keep_counting)  # decisions-only mode on the same batch
    big.close()
    # the same reads in pieces
    cuts = [0, 6_000_000, 13_000_000, 20_000_000, n_short, n_reads]
    piece_bases = max(int(offsets[cuts[i + 1]] - offsets[cuts[i]]) for i in range(5))
    assert piece_bases < (1 << 30) + (1 << 28)
    small = dcn.FilterProcessor(gidx, max_batch_bases=piece_bases, max_batch_reads=max(cuts[i + 1] - cuts[i] for i in range(5)))
    small.reserve_records(n_long * LL // 4)
    ref = {k_: torch.zeros_like(v) for k_, v in out.items()}
    for i in range(5):
        r0, r1 = cuts[i], cuts[i + 1]
        b0, b1 = int(offsets[r0]), int(offsets[r1])
        off_i = torch.from_numpy((offsets[r0:r1 + 1] - offsets[r0]).view(np.int64)).to(dev)
        small.filter_batch_device(d_bases[b0:b1].data_ptr(), off_i.data_ptr(), r1 - r0, b1 - b0, ref["keep"][r0:].data_ptr(),
                                  ref["hits"][r0:].data_ptr(), ref["total"][r0:].data_ptr())
        small.synchronize()
    small.close()
    for k_ in ("total", "hits", "keep"):
        bad = torch.nonzero(out[k_] != ref[k_]).flatten()
        assert bad.numel() == 0, (k_, bad[:5].tolist(), out[k_][bad[:5]].tolist(), ref[k_][bad[:5]].tolist())
    # far end against the oracle: the last 3000 short reads and the last 40 long reads (all above base 2^32)
    r0 = n_short - 3_000
    assert int(offsets[r0]) > (1 << 32) - (1 << 29)
    for ra, rb in ((r0, n_short), (n_reads - 40, n_reads)):
        b0, b1 = int(offsets[ra]), int(offsets[rb])
        assert rb != n_reads or b0 > (1 << 32)
        hb = d_bases[b0:b1].cpu().numpy()
        want = oracle.filter_batch(oidx, hb, offsets[ra:rb + 1] - offsets[ra], threads=8)
        assert want[2].tolist() == out["total"][ra:rb].cpu().tolist()
        assert want[1].tolist() == out["hits"][ra:rb].cpu().tolist()
        assert want[0].tolist() == out["keep"][ra:rb].cpu().tolist()
    assert int(out["hits"][n_short:].min()) > 1000  # the long reads really are host reads with long runs


@pytest.mark.gpu
def test_one_host_batch_beyond_2_to_the_32_bases(oracle, dcn):
    """the host entry points on ONE call of 4.35 Gbp (pageable ASCII through the chunked pipeline, then the same
    stream 2-bit packed): the batch is a block of 1 M reads repeated 29 times, so every repetition must give the
    block's results, and the block's results are the oracle's."""
    rng = np.random.default_rng(99)
    genome = random_reads(rng, 1, 1_000_000, 1_000_000)[0]
    oidx = oracle.Index.build([genome], k=31, w=15)
    gidx = dcn.Index.from_keys(oidx.keys(), 31, 15)
    nb, L, reps = 1_000_000, 150, 29
    g = np.frombuffer(genome, np.uint8)
    st = rng.integers(0, len(genome) - L, nb)
    block = g[st[:, None] + np.arange(L)[None, :]].copy()
    rnd = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (nb, L))]
    host = rng.random(nb) < 0.5
    block[~host] = rnd[~host]
    block[rng.random(nb) < 0.01, 70] = ord("N")
    block = block.reshape(-1)
    bases = np.tile(block, reps)
    n_reads = nb * reps
    assert len(bases) > (1 << 32)
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
    want = oracle.filter_batch(oidx, block, offsets[:nb + 1], threads=8)
    proc = dcn.FilterProcessor(gidx, max_batch_bases=len(bases), max_batch_reads=n_reads)
    keep, hits, total = proc.filter_batch(bases, offsets)
    for name, got, ref in (("total", total, want[2]), ("hits", hits, want[1]), ("keep", keep, want[0])):
        got = got.reshape(reps, nb)
        bad = np.nonzero((got != np.asarray(ref)[None, :]).any(axis=1))[0]
        assert len(bad) == 0, (name, "repetitions that differ from the oracle's block:", bad[:5].tolist())
    packed, mask = dcn.pack_ascii(bases)
    del bases
    kp = proc.filter_batch_packed(packed, mask, offsets, counts=False)
    assert (np.asarray(kp).reshape(reps, nb) == np.asarray(want[0])[None, :]).all()
    proc.close()


@pytest.mark.gpu
def test_size_independent_properties_at_the_headline_batch(oracle, dcn):
    """BASELINE.json configs[1] at its full size -- 10 M x 150 bp = 1.5 Gbp in one batch against a panhuman-1-sized table
    (409,913,780 keys, 34 GB) -- through properties that need no oracle of that size:
      * strand symmetry: the reverse complement of a read of A/C/G/T has the same minimizer multiset, hence the same total,
        distinct hits and decision (a read with an N is exempt: N packs to G's code (c >> 1) & 3 on both strands --
        SURVEY.md 8a row A2 -- so its two strands are not each other's complement in the 2-bit stream, and the reference's
        choice of minimizers around it is strand-dependent too);
      * idempotence: the same batch twice gives the same arrays;
      * order invariance: the batch with its reads permuted gives the permuted arrays (every wave then holds other reads);
      * decisions only == counting, and the six counters follow from the arrays;
      * a checksum of the per-read results of the first 30 k reads against the CPU oracle on the index keys those reads can
        touch (membership of the synthetic remainder follows from its definition, as in the 950 M-key test above)."""
    torch = pytest.importorskip("torch")
    import bench
    dev = torch.device("cuda:0")
    genome = bench.make_host_genome(64_000_000, 3, dev)
    index, keys, host_keys, n_rand, _ = bench.build_index(genome, bench.PANHUMAN_KEYS, 0)
    del keys
    n = 10_000_000
    fwd = bench.make_reads(genome, n, 5, dev)                              # (n * 150,) ASCII, half host-derived, 0.1 % N
    rc = bench._revcomp_ascii(fwd.reshape(n, 150)).reshape(-1).contiguous()
    off = torch.arange(n + 1, dtype=torch.int64, device=dev) * 150
    proc = dcn.FilterProcessor(index, max_batch_bases=n * 150, max_batch_reads=n)

    def run(bases, counts=True):
        k = torch.zeros(n, dtype=torch.uint8, device=dev)
        h = torch.zeros(n, dtype=torch.int32, device=dev) if counts else None
        t = torch.zeros(n, dtype=torch.int32, device=dev) if counts else None
        torch.cuda.synchronize()
        proc.filter_batch_device(bases.data_ptr(), off.data_ptr(), n, n * 150, k.data_ptr(), h.data_ptr() if counts else None,
                                 t.data_ptr() if counts else None)
        proc.synchronize()
        return k, h, t

    proc.reset_stats()
    k1, h1, t1 = run(fwd)
    st = proc.stats()
    assert st["total_seqs"] == n and st["total_bp"] == n * 150
    assert st["filtered_seqs"] == n - int(k1.sum(dtype=torch.int64).item())            # search mode: kept = matched
    assert st["output_bp"] == 150 * int(k1.sum(dtype=torch.int64).item())
    k2, h2, t2 = run(fwd)
    assert torch.equal(k1, k2) and torch.equal(h1, h2) and torch.equal(t1, t2)        # idempotence
    k3, h3, t3 = run(rc)
    clean = ~(fwd.reshape(n, 150) == ord("N")).any(dim=1)
    assert 0.8 < float(clean.float().mean().item()) < 0.99
    assert torch.equal(t1[clean], t3[clean]) and torch.equal(h1[clean], h3[clean]) and torch.equal(k1[clean], k3[clean])  # strand symmetry
    kd, _, _ = run(fwd, counts=False)
    assert torch.equal(kd, k1)                                                         # decisions only == counting
    perm = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    shuffled = fwd.reshape(n, 150)[perm].reshape(-1).contiguous()
    k4, h4, t4 = run(shuffled)
    assert torch.equal(k4, k1[perm]) and torch.equal(h4, h1[perm]) and torch.equal(t4, t1[perm])   # order invariance
    # plausibility of the whole batch, then exactness on a sample
    assert 0.45 < float(k1.float().mean().item()) < 0.55 and int(t1.min().item()) >= 0 and 12.5 < float(t1.float().mean().item()) < 15.5
    m = 30_000
    b = fwd[:m * 150].cpu().numpy()
    o = np.arange(m + 1, dtype=np.uint64) * np.uint64(150)
    small = bench.touchable_oracle_index(b, o, host_keys, n_rand, 8)
    keep, hits, total = oracle.filter_batch(small, b, o, None, 2, 0.01, 0, False, threads=8)
    assert total.tolist() == t1[:m].cpu().numpy().tolist()
    assert hits.tolist() == h1[:m].cpu().numpy().tolist()
    assert keep.tolist() == k1[:m].cpu().numpy().astype(bool).tolist()
    proc.close()
    index.close()


@pytest.mark.gpu
def test_size_independent_properties_of_long_reads_and_pairs_at_full_size(oracle, dcn):
    """BASELINE.json configs[2] (ONT-style lognormal reads, mean 10 kbp, 1.5 Gbp per batch: every read spans waves, the run
    export and the distinct pass carry the counts) and configs[3] (5 M pairs of 2 x 150 bp, --deplete) at their full sizes
    against the panhuman-1-sized table, through properties:
      long reads -- idempotence; strand symmetry (these reads hold no N); the reads in reverse ORDER give the reversed
        arrays (other tiles share every wave); decisions only == counting; first reads == the CPU oracle;
      pairs -- a pair's total is the sum of its mates' totals as single reads, its distinct hits lie between the larger
        mate's and the sum; swapping the mates of every pair changes nothing (src/filter_common.rs:312-348: the hashes are
        concatenated, hits are distinct across both); decisions only == counting."""
    torch = pytest.importorskip("torch")
    import bench
    dev = torch.device("cuda:0")
    genome = bench.make_host_genome(64_000_000, 3, dev)
    index, keys, host_keys, n_rand, _ = bench.build_index(genome, bench.PANHUMAN_KEYS, 0)
    del keys

    def run(proc, bases, off, n_reads, n_units, uid=None, counts=True):
        k = torch.zeros(n_units, dtype=torch.uint8, device=dev)
        h = torch.zeros(n_units, dtype=torch.int32, device=dev) if counts else None
        t = torch.zeros(n_units, dtype=torch.int32, device=dev) if counts else None
        torch.cuda.synchronize()
        proc.filter_batch_device(bases.data_ptr(), off.data_ptr(), n_reads, int(bases.numel()), k.data_ptr(),
                                 h.data_ptr() if counts else None, t.data_ptr() if counts else None,
                                 d_unit_id=None if uid is None else uid.data_ptr(), n_units=n_units)
        proc.synchronize()
        return k, h, t

    # ---- configs[2] ------------------------------------------------------------------------------------------------
    lb, lo = bench.make_long_reads(genome, 1_500_000_000, 6, dev)
    n = lo.numel() - 1
    proc = dcn.FilterProcessor(index, max_batch_bases=int(lb.numel()), max_batch_reads=n)
    proc.reserve_records(int(lb.numel()) // 6)
    k1, h1, t1 = run(proc, lb, lo, n, n)
    k2, h2, t2 = run(proc, lb, lo, n, n)
    assert torch.equal(k1, k2) and torch.equal(h1, h2) and torch.equal(t1, t2)
    kd, _, _ = run(proc, lb, lo, n, n, counts=False)
    assert torch.equal(kd, k1)
    # reverse complement of every read = the whole stream reverse-complemented, reads in reverse order
    comp = torch.arange(256, dtype=torch.uint8, device=dev)
    for a_, b_ in zip(b"ACGT", b"TGCA"):
        comp[a_] = b_
    rcb = comp[lb.flip(0).long()].contiguous()
    lens = (lo[1:] - lo[:-1]).flip(0)
    rco = torch.zeros_like(lo)
    rco[1:] = torch.cumsum(lens, 0)
    k3, h3, t3 = run(proc, rcb, rco, n, n)
    assert torch.equal(t3.flip(0), t1) and torch.equal(h3.flip(0), h1) and torch.equal(k3.flip(0), k1)
    assert 0.4 < float(k1.float().mean().item()) < 0.6 and float(t1.float().sum().item()) > 0.115 * lb.numel()
    m = int(torch.searchsorted(lo, torch.tensor([30_000_000], device=dev)).item()) - 1
    b = lb[:int(lo[m].item())].cpu().numpy()
    o = lo[:m + 1].cpu().numpy().astype(np.uint64)
    small = bench.touchable_oracle_index(b, o, host_keys, n_rand, 8)
    keep, hits, total = oracle.filter_batch(small, b, o, None, 2, 0.01, 0, False, threads=8)
    assert total.tolist() == t1[:m].cpu().numpy().tolist() and hits.tolist() == h1[:m].cpu().numpy().tolist()
    assert keep.tolist() == k1[:m].cpu().numpy().astype(bool).tolist()
    proc.close()
    del lb, rcb, lo, rco
    torch.cuda.empty_cache()

    # ---- configs[3] ------------------------------------------------------------------------------------------------
    n_pairs = 5_000_000
    pb = bench.make_pairs(genome, n_pairs, 7, dev)                                      # rows 2i, 2i+1 = the mates of pair i
    nr = 2 * n_pairs
    off = torch.arange(nr + 1, dtype=torch.int64, device=dev) * 150
    uid = (torch.arange(nr, dtype=torch.int32, device=dev) // 2).contiguous()
    proc = dcn.FilterProcessor(index, deplete=True, max_batch_bases=nr * 150, max_batch_reads=nr)
    kp, hp, tp = run(proc, pb, off, nr, n_pairs, uid)
    ks, hs, ts = run(proc, pb, off, nr, nr)                                              # the same reads as single units
    assert torch.equal(tp, ts[0::2] + ts[1::2])
    assert bool((hp <= hs[0::2] + hs[1::2]).all()) and bool((hp >= torch.maximum(hs[0::2], hs[1::2])).all())
    swapped = pb.reshape(n_pairs, 2, 150).flip(1).reshape(-1).contiguous()
    kq, hq, tq = run(proc, swapped, off, nr, n_pairs, uid)
    assert torch.equal(kq, kp) and torch.equal(hq, hp) and torch.equal(tq, tp)
    kd, _, _ = run(proc, pb, off, nr, n_pairs, uid, counts=False)
    assert torch.equal(kd, kp)
    assert 0.45 < float(kp.float().mean().item()) < 0.55                                 # --deplete keeps the random half
    proc.close()
    index.close()
