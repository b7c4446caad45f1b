// scan.hip -- K2+K3+K4+K5: canonical minimizer scan, k-mer hash, index probe, distinct-hit count.
//
// Replaces, for a whole batch, the per-read body of get_minimizer_hashes_and_positions
// (src/filter_common.rs:261-307: simd_minimizers::canonical_minimizer_positions, the ACGT `retain`,
// iter_canonical_minimizer_values + xxh3_64) and sequence_matches / pair_matches (:129-198).
//
// Work decomposition (MI355X-first, not the reference's 8-lane SIMD chunking):
//   * a TILE is up to tile_windows consecutive windows of one read; one LANE scans one tile
//     sequentially, so a 64-lane wave (= one workgroup) runs 64 independent rolling scans with no
//     cross-lane traffic in the inner loop.  A window's choice depends only on its own l = k+w-1 bases, so
//     tiles overlap by l-1 bases and are exact; only the consecutive-duplicate rule needs the previous
//     window's choice, which a non-first tile recomputes from one extra "carry" window.
//   * phase A (VALU-bound): per base one rolling ntHash32 step (one 16-byte LDS table read), the
//     two-stack sliding min/max over w keys held in registers (ring index static: the loop is
//     unrolled by W), a rolling TG count, and a predicated append of the chosen position to a per-lane
//     LDS list.  Keys are (h & 0xffff0000) | j: only the top 16 hash bits are compared and ties break
//     on position, leftmost for TG-rich ("canonical") windows, rightmost otherwise.
//   * phase B (latency-bound): the wave flattens the 64 lists and handles one emitted minimizer per
//     lane: ACGT test on the mask bits, canonical k-mer value, XXH3-64, one 32-byte group read of the
//     HBM-resident set.  Hits are appended to an LDS array grouped by unit.
//   * units (reads / pairs) whose tiles all sit in this wave are finished here: exact distinct-hit
//     count over the unit's LDS hits and the -a/-r threshold.  Units spanning waves (long reads)
//     leave (unit, hash) hit records and per-unit totals in global memory for distinct.hip.
#include "dcn_internal.h"
#include "dcn_probe.h"

namespace {

// classic ntHash seeds (low 32 bits, listed A,C,G,T) indexed by the 2-bit code A=0 C=1 T=2 G=3,
// as simd-minimizers 1.x does; complement of a code is code ^ 2.
__device__ __constant__ uint32_t NT_F[4] = {0x95c60474u, 0x62a02b4cu, 0x82572324u, 0x4be24456u};

__device__ inline uint32_t rotl32(uint32_t x, uint32_t r) { return __funnelshift_l(x, x, r); }

__device__ inline uint32_t wave_inclusive_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

__device__ inline uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

struct WaveShared {
    uint4 tab[16];                      // {F[in], rotl(F[in^2],k-1), rotl(F[out],k-1), F[out^2]} at in | out<<2
    uint16_t list[DCN_LCAP][DCN_WAVE];  // per-lane emitted positions (relative to the tile's scan start)
    uint64_t hit_hash[DCN_HCAP];
    uint8_t hit_unit[DCN_HCAP];
    uint32_t total[DCN_WAVE];           // per unit slot: minimizers (after the ACGT filter)
    uint32_t hits[DCN_WAVE];            // per unit slot: distinct hits
    uint32_t unit_of[DCN_WAVE];         // unit slot -> global unit id
    uint16_t start[DCN_WAVE + 1];       // exclusive prefix of the per-lane list lengths
    uint8_t local[DCN_WAVE];            // unit slot is resolved inside this wave
};

// W > 0: window size known at compile time, ring in registers.  W == 0: runtime w, ring in dynamic LDS.
template <int W, bool K128, bool DUMP>
__global__ __launch_bounds__(DCN_WAVE) void scan_kernel(dcn_scan_args a) {
    __shared__ WaveShared sh;
    extern __shared__ uint2 dyn_ring[]; // only for W == 0: [w][64] (lkey, rkey)

    const int lane = threadIdx.x;
    const uint32_t NT = *a.n_tiles;
    const uint32_t wave_first = blockIdx.x * DCN_WAVE;
    if (wave_first >= NT) return;
    const uint32_t k = a.k;
    const uint32_t w = W > 0 ? (uint32_t)W : a.w;
    const uint32_t l = k + w - 1;

    // ---- tile descriptors, unit slots -----------------------------------------------------------------
    const uint32_t tile_idx = wave_first + lane;
    const bool have_tile = tile_idx < NT;
    dcn_tile t = a.tiles[have_tile ? tile_idx : NT - 1];
    if (!have_tile) t.n_windows = 0;
    const uint32_t carry = t.flags & 1u;
    const uint32_t nwc = t.n_windows ? t.n_windows + carry : 0; // windows this lane evaluates
    const uint32_t unit_prev = __shfl_up(t.unit, 1, 64);
    const bool head = lane == 0 || t.unit != unit_prev;
    const unsigned long long head_mask = __ballot(head);
    const uint32_t uslot = (uint32_t)__popcll(head_mask & ((2ull << lane) - 1)) - 1;
    sh.total[lane] = 0;
    sh.hits[lane] = 0;
    if (head) {
        sh.unit_of[uslot] = t.unit;
        bool loc = false;
        if (!DUMP) {
            uint32_t first = a.unit_tile_first[t.unit], last = a.unit_tile_first[t.unit + 1];
            loc = first >= wave_first && last <= wave_first + DCN_WAVE;
        }
        sh.local[uslot] = loc ? 1 : 0;
    }
    if (lane < 16) {
        uint32_t in = lane & 3, out = lane >> 2;
        uint4 e;
        e.x = NT_F[in];
        e.y = rotl32(NT_F[in ^ 2], (k - 1) & 31);
        e.z = rotl32(NT_F[out], (k - 1) & 31);
        e.w = NT_F[out ^ 2];
        sh.tab[lane] = e;
    }
    __syncthreads();

    // ---- stream geometry -----------------------------------------------------------------------------
    const uint32_t *packed = a.packed;
    const int64_t s = (int64_t)t.scan_start;
    const int64_t q_in = s >> 4;
    const uint32_t sh_in = (uint32_t)(s & 15) * 2;
    const int64_t sk = s - (int64_t)(k - 1);
    const int64_t q_k = sk >> 4; // arithmetic shift: floor
    const uint32_t sh_k = (uint32_t)(sk & 15) * 2;
    const int64_t sl = s - (int64_t)(l - 1);
    const int64_t q_l = sl >> 4;
    const uint32_t sh_l = (uint32_t)(sl & 15) * 2;

    // ---- prologue: first k-1 bases (no complete k-mer yet) -------------------------------------------------
    uint32_t fw = 0, rc = 0, tg = 0;
    {
        uint32_t cur = 0;
        for (uint32_t tt = 0; tt + 1 < k; ++tt) {
            if ((tt & 15) == 0) cur = __funnelshift_r(packed[q_in + (tt >> 4)], packed[q_in + (tt >> 4) + 1], sh_in);
            uint32_t c = (cur >> (2 * (tt & 15))) & 3;
            uint4 e = sh.tab[c];
            fw = rotl32(fw, 1) ^ e.x;
            rc = rotl32(rc, 31) ^ e.y;
            tg += c >> 1;
        }
        // the main loop subtracts the TG bit of base t-(l-1) from its first step t=k-1 on; for t < l-1
        // that is one of the w-1 bases in front of the tile: pre-add them so the subtraction cancels.
        for (uint32_t b = 1; b < w; ++b) {
            int64_t p = s - (int64_t)b;
            tg += (packed[p >> 4] >> (2 * (uint32_t)(p & 15) + 1)) & 1u;
        }
    }

    // ---- phase A / phase B state --------------------------------------------------------------------------
    uint32_t cnt = 0;            // entries in this lane's list
    uint32_t emitted_before = 0; // entries already flushed (dump slot numbering)
    uint32_t prev = 0xFFFFFFFFu; // previous window's choice (dedup state)
    uint32_t n_hits = 0;         // wave-uniform: hits held in LDS
    bool go_global = false;      // wave-uniform: hits/totals of every unit go through global memory
    const uint32_t nk = nwc ? nwc + w - 1 : 0;
    const uint32_t jmax = wave_max_u32(nk);

    auto spill_hits = [&]() {
        // move every LDS hit to the global (unit, hash) record list
        if (n_hits > 0) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(&a.status->rec_count, (unsigned long long)n_hits);
            base = __shfl(base, 0, 64);
            for (uint32_t x = lane; x < n_hits; x += DCN_WAVE) {
                unsigned long long r = base + x;
                if (r < a.rec_capacity) {
                    uint32_t u = sh.unit_of[sh.hit_unit[x]];
                    a.rec_unit[r] = u;
                    a.rec_hash[r] = sh.hit_hash[x];
                    atomicAdd(&a.g_hitcnt[u], 1u);
                } else {
                    a.status->rec_overflow = 1;
                }
            }
        }
        __syncthreads();
        n_hits = 0;
    };

    auto flush = [&](bool final_flush) {
        // ---- flatten the 64 lists ------------------------------------------------------------------------
        uint32_t incl = wave_inclusive_scan_u32(cnt, lane);
        uint32_t M = __shfl(incl, 63, 64);
        sh.start[lane] = (uint16_t)(incl - cnt);
        if (lane == 63) sh.start[64] = (uint16_t)M;
        __syncthreads();
        if (!final_flush) go_global = true; // hits of one unit are contiguous only within a flush
        for (uint32_t E = 0; E < M; E += DCN_WAVE) {
            uint32_t e = E + lane;
            bool act = e < M;
            // owner = largest lane whose list starts at or before e
            uint32_t lo = 0, hi = 63;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                uint32_t mid = (lo + hi + 1) >> 1;
                bool le = sh.start[mid] <= e;
                lo = le ? mid : lo;
                hi = le ? hi : mid - 1;
            }
            uint32_t owner = lo;
            uint32_t idx = act ? e - sh.start[owner] : 0;
            uint32_t rel = sh.list[idx][owner];
            long long o_s = __shfl((long long)s, owner, 64);
            uint32_t o_uslot = __shfl(uslot, owner, 64);
            uint64_t p = (uint64_t)(o_s + rel);
            bool valid = act && dcn_kmer_valid(a.invmask, p, k);
            uint64_t hash = 0;
            if (valid) hash = K128 ? dcn_kmer_hash128(packed, p, k) : dcn_kmer_hash64(packed, p, k);
            if (DUMP) {
                uint32_t o_eb = __shfl(emitted_before, owner, 64);
                uint32_t o_rp = __shfl(t.read_pos, owner, 64);
                uint32_t o_carry = __shfl(carry, owner, 64);
                if (act) {
                    uint64_t slot = (uint64_t)o_s + o_carry + o_eb + idx;
                    a.dump_hash[slot] = hash;
                    a.dump_pos[slot] = o_rp + rel;
                    a.dump_valid[slot] = valid ? 1 : 0;
                }
            } else {
                if (valid) atomicAdd(&sh.total[o_uslot], 1u);
                bool hit = valid && dcn_table_contains_dev(a.table, hash);
                unsigned long long hb = __ballot(hit);
                uint32_t nh = (uint32_t)__popcll(hb);
                if (n_hits + nh > DCN_HCAP) { // wave-uniform
                    go_global = true;
                    spill_hits();
                }
                if (hit) {
                    uint32_t x = n_hits + (uint32_t)__popcll(hb & ((1ull << lane) - 1));
                    sh.hit_hash[x] = hash;
                    sh.hit_unit[x] = (uint8_t)o_uslot;
                }
                n_hits += nh;
            }
        }
        emitted_before += cnt;
        cnt = 0;
        __syncthreads();
        if (!DUMP && go_global) spill_hits();
    };

    // ---- phase A: rolling scan ---------------------------------------------------------------------------
    uint32_t ringL[W > 0 ? W : 1], ringR[W > 0 ? W : 1];
    uint32_t pl = 0xFFFFFFFFu, pr = 0;
    if (W > 0) {
#pragma unroll
        for (int r = 0; r < (W > 0 ? W : 1); ++r) {
            ringL[r] = 0xFFFFFFFFu;
            ringR[r] = 0;
        }
    } else {
        for (uint32_t r = 0; r < w; ++r) dyn_ring[r * DCN_WAVE + lane] = make_uint2(0xFFFFFFFFu, 0u);
    }
    uint32_t w_in = 0, w_k = 0, w_l = 0;
    {
        uint32_t c0 = (k - 1) >> 4;
        w_in = __funnelshift_r(packed[q_in + c0], packed[q_in + c0 + 1], sh_in);
        w_k = __funnelshift_r(packed[q_k + c0], packed[q_k + c0 + 1], sh_k);
        w_l = __funnelshift_r(packed[q_l + c0], packed[q_l + c0 + 1], sh_l);
    }
    constexpr uint32_t STEP = W > 0 ? (uint32_t)W : 8u; // steps between flush checks (= ring size for W > 0)
    static_assert(STEP < DCN_LCAP, "list capacity must exceed one step block");
    uint32_t gslot = 0; // W == 0: runtime ring slot, j % w

    for (uint32_t jb = 0; jb < jmax; jb += STEP) {
#pragma unroll
        for (uint32_t r = 0; r < STEP; ++r) {
            const uint32_t j = jb + r;     // k-mer index within the tile scan (wave-uniform)
            const uint32_t tt = j + k - 1; // base index within the tile scan (wave-uniform)
            const uint32_t ti = tt & 15;
            if (ti == 0) {
                uint32_t c = tt >> 4;
                w_in = __funnelshift_r(packed[q_in + c], packed[q_in + c + 1], sh_in);
                w_k = __funnelshift_r(packed[q_k + c], packed[q_k + c + 1], sh_k);
                w_l = __funnelshift_r(packed[q_l + c], packed[q_l + c + 1], sh_l);
            }
            const uint32_t cin = (w_in >> (2 * ti)) & 3;
            const uint32_t ck = (w_k >> (2 * ti)) & 3;
            const uint32_t hl = (w_l >> (2 * ti + 1)) & 1;
            const uint4 e = sh.tab[cin | (ck << 2)];
            const uint32_t fwo = rotl32(fw, 1) ^ e.x;
            const uint32_t rco = rotl32(rc, 31) ^ e.y;
            const uint32_t h = fwo + rco;
            fw = fwo ^ e.z;
            rc = rco ^ e.w;
            const uint32_t lk = (h & 0xFFFF0000u) | j;
            const uint32_t rk = lk ^ 0xFFFF0000u;
            uint32_t lmin, rmax;
            if constexpr (W > 0) {
                // two-stack sliding min/max: ring slot r is static because the loop is unrolled by W
                ringL[r] = lk;
                ringR[r] = rk;
                pl = min(pl, lk);
                pr = max(pr, rk);
                if (r == W - 1) {
#pragma unroll
                    for (int qq = W - 2; qq >= 0; --qq) {
                        ringL[qq] = min(ringL[qq], ringL[qq + 1]);
                        ringR[qq] = max(ringR[qq], ringR[qq + 1]);
                    }
                    lmin = ringL[0];
                    rmax = ringR[0];
                    pl = 0xFFFFFFFFu;
                    pr = 0;
                } else {
                    lmin = min(pl, ringL[(r + 1) % STEP]);
                    rmax = max(pr, ringR[(r + 1) % STEP]);
                }
            } else {
                // generic w: ring of w keys per lane in LDS, O(w) rescan per step
                dyn_ring[gslot * DCN_WAVE + lane] = make_uint2(lk, rk);
                gslot = gslot + 1 == w ? 0 : gslot + 1;
                lmin = 0xFFFFFFFFu;
                rmax = 0;
                for (uint32_t qq = 0; qq < w; ++qq) {
                    uint2 v = dyn_ring[qq * DCN_WAVE + lane];
                    lmin = min(lmin, v.x);
                    rmax = max(rmax, v.y);
                }
            }
            tg += cin >> 1;
            const bool canonical = 2 * tg > l;
            const uint32_t sel = (canonical ? lmin : rmax) & 0xFFFFu;
            tg -= hl;
            const uint32_t i = j - (w - 1); // window index within the tile scan; wraps while j < w-1
            const bool in_range = i < nwc;  // unsigned compare: false while i is "negative"
            const bool emit = in_range && sel != prev && !(carry && i == 0);
            sh.list[cnt][lane] = (uint16_t)sel;
            cnt += emit ? 1u : 0u;
            prev = in_range ? sel : prev;
        }
        if (__any(cnt > DCN_LCAP - STEP)) flush(false);
    }
    flush(true);
    if (DUMP) {
        if (have_tile) a.dump_count[tile_idx] = emitted_before;
        return;
    }

    // ---- finish the units this wave owns -------------------------------------------------------------------
    if (!go_global) {
        unsigned long long base = 0;
        for (uint32_t x0 = 0; x0 < n_hits; x0 += DCN_WAVE) {
            uint32_t x = x0 + lane;
            bool act = x < n_hits;
            uint32_t u = act ? sh.hit_unit[x] : 0;
            bool loc = act && sh.local[u];
            if (loc) {
                uint64_t hv = sh.hit_hash[x];
                bool dup = false;
                for (int y = (int)x - 1; y >= 0 && sh.hit_unit[y] == u; --y)
                    if (sh.hit_hash[y] == hv) {
                        dup = true;
                        break;
                    }
                if (!dup) atomicAdd(&sh.hits[u], 1u);
            }
            bool rec = act && !loc;
            unsigned long long rb = __ballot(rec);
            if (rb) { // wave-uniform
                uint32_t nrec = (uint32_t)__popcll(rb);
                if (lane == 0) base = atomicAdd(&a.status->rec_count, (unsigned long long)nrec);
                base = __shfl(base, 0, 64);
                if (rec) {
                    unsigned long long ridx = base + (uint32_t)__popcll(rb & ((1ull << lane) - 1));
                    if (ridx < a.rec_capacity) {
                        uint32_t gu = sh.unit_of[u];
                        a.rec_unit[ridx] = gu;
                        a.rec_hash[ridx] = sh.hit_hash[x];
                        atomicAdd(&a.g_hitcnt[gu], 1u);
                    } else {
                        a.status->rec_overflow = 1;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (head && have_tile) {
        uint32_t tot = sh.total[uslot];
        if (!go_global && sh.local[uslot]) {
            uint32_t hc = sh.hits[uslot];
            a.keep[t.unit] = dcn_decide(hc, tot, a.abs_threshold, a.rel_threshold, a.deplete) ? 1 : 0;
            if (a.hits) a.hits[t.unit] = hc;
            if (a.total) a.total[t.unit] = tot;
            a.unit_state[t.unit] = 1;
        } else if (tot) {
            atomicAdd(&a.g_total[t.unit], tot);
        }
    }
}

template <int W>
int launch_w(const dcn_scan_args &args, uint32_t blocks, bool dump, bool k128, size_t dyn, hipStream_t stream) {
#define DCN_LAUNCH(K128_, DUMP_)                                                                      \
    do {                                                                                              \
        auto kern = scan_kernel<W, K128_, DUMP_>;                                                     \
        if (dyn > 0) {                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
            if (e != hipSuccess) return dcn_fail(DCN_ERR_HIP, std::string("dyn LDS: ") + hipGetErrorString(e)); \
        }                                                                                             \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(DCN_WAVE), dyn, stream, args);                    \
    } while (0)
    if (k128) {
        if (dump) DCN_LAUNCH(true, true);
        else DCN_LAUNCH(true, false);
    } else {
        if (dump) DCN_LAUNCH(false, true);
        else DCN_LAUNCH(false, false);
    }
#undef DCN_LAUNCH
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

} // namespace

int dcn_launch_scan(const dcn_scan_args &args, uint32_t max_tiles, bool dump, hipStream_t stream) {
    if (max_tiles == 0) return DCN_OK;
    uint32_t blocks = (max_tiles + DCN_WAVE - 1) / DCN_WAVE;
    bool k128 = args.k > 32;
    switch (args.w) {
    case 15: return launch_w<15>(args, blocks, dump, k128, 0, stream);
    case 11: return launch_w<11>(args, blocks, dump, k128, 0, stream);
    case 1: return launch_w<1>(args, blocks, dump, k128, 0, stream);
    default: {
        size_t dyn = (size_t)args.w * DCN_WAVE * sizeof(uint2);
        return launch_w<0>(args, blocks, dump, k128, dyn, stream);
    }
    }
}
