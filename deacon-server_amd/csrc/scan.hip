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
//   * phase A: per base one rolling ntHash32 step (one 16-byte LDS table read, issued one step ahead),
//     the two-stack sliding min/max over w keys held in registers (ring index static: the loop is
//     unrolled by W), a rolling TG count, and a predicated append of the chosen position to a per-lane
//     LDS list.  Keys are (h & 0xffff0000) | j: only the top 16 hash bits are compared and ties break
//     on position, leftmost for TG-rich ("canonical") windows, rightmost otherwise.
//   * phase B: the wave flattens the 64 lists and handles one emitted minimizer per lane and round:
//     ACGT test on the mask bits, canonical k-mer value, XXH3-64, one 32-byte group read of the
//     HBM-resident set.
//   * units (reads / pairs) whose tiles all sit in this wave are finished here: hits go through a small LDS
//     ring in item order (a unit's hits are contiguous), each new hit is compared with the unit's earlier
//     ones, so the exact distinct count needs no per-wave hit table; then the -a/-r threshold.  Units
//     spanning waves (long reads) export (unit, hash) hit records and per-unit totals for plan.hip's
//     distinct pass.
#include "dcn_internal.h"
#include "dcn_probe.h"

#include <atomic>

namespace {

// classic ntHash seeds (low 32 bits, listed A,C,G,T) indexed by the 2-bit code A=0 C=1 T=2 G=3,
// as simd-minimizers 1.x does; complement of a code is code ^ 2.
__device__ __constant__ uint32_t NT_F[4] = {0x95c60474u, 0x62a02b4cu, 0x82572324u, 0x4be24456u};

__device__ inline uint32_t rotl32(uint32_t x, uint32_t r) { return __funnelshift_l(x, x, r); }

// Wave-wide scan / reduction on DPP row shifts and row broadcasts (the gfx9 sequence: row_shr 1, 2, 4, 8 inside each row of
// 16 lanes, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2-3).  No ds_bpermute, hence no per-lane address
// registers: with __shfl_up / __shfl_xor the six address VGPRs of a scan were computed once and stayed live across phase A
// (the decisions-only kernel spilled them to scratch).  `old` = 0 is the identity of both + and unsigned max.
#ifndef DCN_WAVE_OPS_SHFL
#define DCN_WAVE_OPS_SHFL 0 // 1: the round-1..3 form on __shfl, kept for A/B timing
#endif
template <bool MAX>
__device__ inline uint32_t wave_scan_dpp(uint32_t v) {
    auto op = [](uint32_t x, uint32_t y) { return MAX ? max(x, y) : x + y; };
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false)); // row_shr:1
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false)); // row_shr:2
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false)); // row_shr:4
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false)); // row_shr:8
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false)); // row_bcast:15 -> rows 1, 3
    v = op(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false)); // row_bcast:31 -> rows 2, 3
    return v;
}

__device__ inline uint32_t wave_inclusive_scan_u32(uint32_t v, int lane) {
#if DCN_WAVE_OPS_SHFL
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
#else
    (void)lane;
    return wave_scan_dpp<false>(v);
#endif
}

// wave-uniform (the value of lane 63 of the max-scan, read into an SGPR)
__device__ inline uint32_t wave_max_u32(uint32_t v) {
#if DCN_WAVE_OPS_SHFL
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_dpp<true>(v), 63);
#endif
}

// lane 63's value of an inclusive scan = the wave's total, wave-uniform
__device__ inline uint32_t wave_last_u32(uint32_t incl) {
#if DCN_WAVE_OPS_SHFL
    return __shfl(incl, 63, 64);
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
#endif
}

#ifndef DCN_LIST_BY_LANE
#define DCN_LIST_BY_LANE 1 // 0: the round-1 layout list[entry][lane], kept for A/B timing
#endif
#if DCN_LIST_BY_LANE
#define DCN_LIST_AT(lane_, entry_) list[lane_][entry_]
#else
#define DCN_LIST_AT(lane_, entry_) list[entry_][lane_]
#endif
constexpr int DCN_LSTRIDE = 42;  // u16 entries per list row: DCN_LCAP + 2, 21 dwords
static_assert(DCN_LSTRIDE >= DCN_LCAP && (DCN_LSTRIDE / 2) % 2 == 1 && DCN_LSTRIDE % 2 == 0, "odd dword stride");
constexpr int DCN_RCAP = 256; // LDS ring of the most recent hits; a unit resolved in-wave has <= RCAP-64 items

struct WaveShared {
    uint4 tab[16];                      // {F[in], rotl(F[in^2],k-1), rotl(F[out],k), rotr(F[out^2],1)} at in | out<<2
    // per-lane emitted positions (relative to the tile's scan start), one row per lane.  The row stride is an odd
    // number of dwords: phase A's store of entry cnt by every lane, and phase B's read of consecutive entries of one
    // lane by consecutive lanes, both spread over the banks (rows of 64 lanes x u16 put a lane's whole list on one
    // bank: phase B then read it 13-way conflicted on average)
#if DCN_LIST_BY_LANE
    uint16_t list[DCN_WAVE][DCN_LSTRIDE];
#else
    uint16_t list[DCN_LCAP + 2][DCN_WAVE];
#endif
    uint64_t ring_hash[DCN_RCAP];
    uint32_t total[DCN_WAVE];           // per unit slot: emitted minimizers minus those failing the ACGT test
    uint32_t hits[DCN_WAVE];            // per unit slot: distinct hits
    uint32_t items[DCN_WAVE];           // per unit slot: emitted minimizers (bounds the ring span)
    uint16_t hraw[DCN_WAVE];            // per unit slot: hits pushed through the ring so far (at most DCN_RCAP - 64)
    uint32_t ucap[DCN_WAVE];            // per unit slot: first the unit's window span in this wave, then the hits its run holds
    uint32_t unit_of[DCN_WAVE];         // unit slot -> global unit id
    uint16_t start[DCN_WAVE + 2];       // exclusive prefix of the per-lane list lengths
    uint32_t uhits[DCN_WAVE];           // per unit slot: hits written to the unit's run of the record array so far
    uint64_t run_base[DCN_WAVE];        // per unit slot: first slot of the unit's run (scan_start + carry of its first tile here)
    uint8_t local[DCN_WAVE];            // unit slot has all its tiles in this wave
    uint8_t lok[DCN_WAVE];              // unit slot is being resolved inside this wave
};

#ifndef DCN_EXP
#define DCN_EXP 0 // experiment bits (timing-only builds, results wrong): 1 = no set probe, 2 = no phase B,
                  // 4 = no list store, 8 = no duplicate compare, 16 = no mask loads,
                  // 64 = k / l streams reuse the in-stream words (2 instead of 6 stream loads per block),
                  // 128 = no stream loads at all in the main loop (words recycled), 256 = hits of units the wave does not
                  // finish are not written to their runs, 512 = their bookkeeping runs but the store itself is left out,
                  // 1024 = such units are not enrolled for the distinct pass (no tile_hits store, no atomics at the wave's end)
#endif
#ifndef DCN_MIN_WAVES
#define DCN_MIN_WAVES 4
#endif
#ifndef DCN_FAST_EXTRA_ROUNDS
#define DCN_FAST_EXTRA_ROUNDS 4 // own-list rounds beyond abs_threshold before the undecided rest is flattened
#endif
#ifndef DCN_MIN_WAVES_FAST
#define DCN_MIN_WAVES_FAST 4 // decisions-only instantiation
#endif

template <bool B>
struct BoolTag {
    static constexpr bool value = B;
};
template <int N>
struct IntTag {
    static constexpr int value = N;
};
#ifndef DCN_EXPORT_UNIFORM
#define DCN_EXPORT_UNIFORM 0 // 1: rounds whose exported hits all belong to one unit skip the run-head bookkeeping (A/B: profiles/r04_ab.txt)
#endif
#ifndef DCN_PIPE_B
#define DCN_PIPE_B 0 // 1: phase B software-pipelined by one round (the next round's owner search and sequence / mask loads
                     // are issued between this round's set probe and its use); needs DCN_U_FINAL == 1
#endif
#ifndef DCN_U_FINAL
#define DCN_U_FINAL 1 // items per lane per phase-B group in the final flush (measured 1..8: 1 is best)
#endif

// W > 0: window size known at compile time, ring in registers.  W == 0: runtime w, ring in dynamic LDS.
// FAST: the decisions-only instantiation (a.early_out_max_items != 0), kept apart so that the counting kernel's
// register allocation does not carry the early-out path
// VAR: the parity-pinning variant (dcn_set_minimizer_variant): ntHash rotation per base, number of
// hash bits compared and the fw/rc combination are run-time values and the window keys are 64 bits wide (hash bits
// above, position below).  Only instantiated for W == 0; the default rules never take it, so the kernels above are
// the same code with or without it.
template <int W, bool K128, bool DUMP, bool FAST, bool VAR = false>
__global__ __launch_bounds__(DCN_WAVE, FAST ? DCN_MIN_WAVES_FAST : DCN_MIN_WAVES) void scan_kernel(dcn_scan_args a) {
    static_assert(!VAR || (W == 0 && !FAST), "the variant path is the generic-w counting / dump kernel");
    __shared__ WaveShared sh;
    extern __shared__ __align__(16) uint2 dyn_ring[]; // only for W == 0: [w][64] (lkey, rkey); VAR: [w][64] of two u64 keys

    const int lane = threadIdx.x;
    // (bad_offsets: the plan kernel refused the batch's offsets -- api.hip reports it at the next synchronize; its tiles are not looked at)
    const uint32_t NT = a.status->bad_offsets ? 0u : *a.n_tiles;
    const uint32_t wave_first = blockIdx.x * DCN_WAVE;
    if (wave_first >= NT) return;
    const uint32_t k = a.k;
    const uint32_t w = W > 0 ? (uint32_t)W : a.w;
    const uint32_t l = k + w - 1;

    // ---- tile descriptors, unit slots -----------------------------------------------------------------
    const uint32_t tile_idx = wave_first + lane;
    const bool have_tile = tile_idx < NT;
    dcn_tile t = a.tiles[have_tile ? tile_idx : NT - 1];
    if (!have_tile) t.nwf = 0;
    const uint32_t carry = t.carry();
    const uint32_t nwc = t.n_windows() ? t.n_windows() + carry : 0; // windows this lane evaluates
    const uint32_t tile_read_pos = (DUMP && have_tile && a.tile_read_pos) ? a.tile_read_pos[tile_idx] : 0u;
    const uint32_t unit_prev = __shfl_up(t.unit, 1, 64);
    const uint64_t start_prev = (uint64_t)__shfl_up((long long)t.scan_start, 1, 64);
    // A unit slot is a run of adjacent lanes of one unit IN STREAM ORDER.  Tiles follow the stream inside a planning
    // block's range, but the ranges of different blocks are placed in whatever order their cursor atomics ran: a unit
    // of three or more reads cut by a block boundary can meet itself again with the later reads first.  Such a lane
    // starts a new slot (and a new run of the record array: a run may only grow over windows that lie behind it).
    const bool head = lane == 0 || t.unit != unit_prev || t.scan_start < start_prev;
    const unsigned long long head_mask = __ballot(head);
    const uint32_t uslot = (uint32_t)__popcll(head_mask & ((2ull << lane) - 1)) - 1;
    sh.total[lane] = 0;
    sh.hits[lane] = 0;
    sh.items[lane] = 0;
    sh.hraw[lane] = 0;
    sh.lok[lane] = 0;
    sh.uhits[lane] = 0;
    sh.ucap[lane] = 0;
    if (head) {
        sh.unit_of[uslot] = t.unit;
        sh.run_base[uslot] = (t.scan_start + carry) >> a.rec_shift;
        bool loc = false;
        if (!DUMP) {
            if (t.whole_unit()) { // the unit's only tile is this lane's
                loc = true;
            } else {
                uint32_t first = a.unit_tile_first[t.unit], count = a.unit_tile_count[t.unit];
                loc = count != 0xFFFFFFFFu && first >= wave_first && first + count <= wave_first + DCN_WAVE;
            }
        }
        sh.local[uslot] = loc ? 1 : 0;
    }
    const uint32_t ROT = VAR ? a.nt_rot : 1u;          // ntHash rotation per base
    const uint32_t ROTR = (32u - ROT) & 31u;           // the reverse strand rotates the other way
    if (lane < 16) {
        uint32_t in = lane & 3, out = lane >> 2;
        uint4 e;
        e.x = NT_F[in];
        e.y = rotl32(NT_F[in ^ 2], (ROT * (k - 1)) & 31);
        e.z = rotl32(NT_F[out], (ROT * k) & 31);  // rotl(F[out], ROT*(k-1)), pre-rotated by the next step's rotl ROT
        e.w = rotl32(NT_F[out ^ 2], ROTR);        // F[out^2], pre-rotated by the next step's rotr ROT
        sh.tab[lane] = e;
    }
    __syncthreads();
    if (!DUMP) {
        // A unit's run of the record array starts at the slot of its first window in this wave and may grow up to the slot of
        // its last: with one slot per 2^rec_shift windows it holds that many times fewer hits than the unit has windows
        // here, which real sequence never fills (one minimizer per ~8 windows) -- a run that would is refused and the
        // batch comes back with one slot per window (api.hip).
        // (a read's last tile -- the only kind that can be a few windows short of a slot -- also owns the l-1 positions
        // behind its last window, where no window of this read or of the next one starts)
        const uint32_t head_lane = 63u - (uint32_t)__clzll(head_mask & ((2ull << lane) - 1));
        const uint64_t first_win = (uint64_t)__shfl((long long)(t.scan_start + carry), head_lane, 64);
        const uint32_t slack = t.n_windows() < a.tile_windows ? l - 1 : 0u;
        if (have_tile && t.n_windows()) atomicMax(&sh.ucap[uslot], (uint32_t)(t.scan_start + carry + t.n_windows() + slack - first_win));
        __syncthreads();
        if (head) sh.ucap[uslot] = (uint32_t)(((first_win + sh.ucap[uslot]) >> a.rec_shift) - (first_win >> a.rec_shift));
        __syncthreads();
    }

    const uint32_t *packed = a.packed;
    const int64_t s = (int64_t)t.scan_start;

    // ---- prologue: first k-1 bases (no complete k-mer yet) -------------------------------------------------
    // fw / rc hold the hashes BEFORE the removal of the outgoing base; the removal terms of a step are folded
    // into the next step (zprev / wprev), so one step is two rotates, two 3-input xors and one add.
    uint32_t fw = 0, rc = 0, tg = 0;
    {
        const int64_t q_in = s >> 4;
        const uint32_t sh_in = (uint32_t)(s & 15) * 2;
        uint32_t cur = 0;
        for (uint32_t tt = 0; tt + 1 < k; ++tt) {
            if ((tt & 15) == 0) cur = __funnelshift_r(packed[q_in + (tt >> 4)], packed[q_in + (tt >> 4) + 1], sh_in);
            uint32_t c = (cur >> (2 * (tt & 15))) & 3;
            uint4 e = sh.tab[c];
            fw = rotl32(fw, VAR ? ROT : 1u) ^ e.x;
            rc = rotl32(rc, VAR ? ROTR : 31u) ^ e.y;
            tg += c >> 1;
        }
    }
    uint32_t zprev = 0, wprev = 0;

    // ---- phase A / phase B state --------------------------------------------------------------------------
    uint32_t cnt = 0;              // entries in this lane's list
    uint32_t emitted_before = 0;   // entries already flushed (not counting a dropped carry entry)
    bool first_pending = carry;    // a carry tile's first entry only seeds the dedup state: dropped in phase B
    uint32_t prev = 0xFFFFFFFFu;   // previous window's choice (dedup state)
    uint32_t n_ring = 0;           // wave-uniform: hits pushed through the LDS ring so far
    bool go_global = false;        // wave-uniform: no unit of this wave is resolved in-wave any more
    bool any_rec = false;          // wave-uniform: this wave wrote a hit into some tile's run
    const uint32_t nk = nwc ? nwc + w - 1 : 0;
    const uint32_t jmax = wave_max_u32(nk);

    // Phase B.  Flattens the 64 lists (item e belongs to the lane whose prefix range holds e).  A group is U items
    // per lane (item E + u*64 + lane): the sequence/mask words of the whole group are loaded together, then the
    // U set groups, so a lane pays one L2 and one HBM latency per U items.  U is a compile-time tag: the final
    // flush (scan registers dead) uses DCN_U_FINAL, a mid-scan flush 1.
    auto flush = [&](auto u_tag, bool final_flush) {
        constexpr int U = decltype(u_tag)::value;
#ifdef DCN_PHASEB_PRIO
        __builtin_amdgcn_s_setprio(DCN_PHASEB_PRIO); // latency-bound phase: let its few instructions issue first
#endif
        const uint32_t skip0 = (first_pending && cnt > 0) ? 1u : 0u;
        const uint32_t us_skip = uslot | (skip0 << 8); // what phase B needs of an item's owner lane, in one shuffle
        const uint32_t cnt_eff = cnt - skip0;
        uint32_t incl = wave_inclusive_scan_u32(cnt_eff, lane);
        uint32_t M = wave_last_u32(incl);
        sh.start[lane] = (uint16_t)(incl - cnt_eff);
        if (lane == 63) sh.start[64] = (uint16_t)M;
        if (!DUMP && cnt_eff) atomicAdd(&sh.items[uslot], cnt_eff);
        if (!final_flush) go_global = true; // a unit's hits are contiguous only within one flush
        __syncthreads();
        if (!DUMP && head) {
            // resolved in-wave: all tiles here, single flush, and few enough items for the hit ring
            bool ok = sh.local[uslot] && !go_global && sh.items[uslot] <= (uint32_t)(DCN_RCAP - DCN_WAVE);
            sh.lok[uslot] = ok ? 1 : 0;
        }
        __syncthreads();
        constexpr int NPW = K128 ? 5 : 3; // packed words per k-mer
        constexpr int NMW = K128 ? 3 : 2; // mask words per k-mer: its k bits start at bit p % 32 (k <= 32: within two words)
        constexpr bool PIPE = DCN_PIPE_B && U == 1;
        // one item per lane, everything up to and including its sequence / mask loads (the pipelined form issues this for
        // round r+1 while round r's set probe is in flight)
        struct Item {
            bool act;
            uint32_t lo, idx, rel, o_uslot;
            uint64_t p;
            uint32_t mw[NMW], pw[NPW];
        };
        auto fetch = [&](uint32_t E, Item &it) {
            const uint32_t e = E + lane;
            it.act = e < M;
            uint32_t l_ = 0, h_ = 63;
#pragma unroll
            for (int step = 0; step < 6; ++step) {
                const uint32_t mid = (l_ + h_ + 1) >> 1;
                const bool le = sh.start[mid] <= e;
                l_ = le ? mid : l_;
                h_ = le ? h_ : mid - 1;
            }
            it.lo = l_;
            it.idx = it.act ? e - sh.start[l_] : 0;
            const uint32_t o_pk = __shfl(us_skip, l_, 64);
            const uint32_t o_skip = o_pk >> 8;
#ifdef DCN_DEBUG_BOUNDS
            if (it.act && (l_ > 63u || it.idx + o_skip >= (uint32_t)DCN_LCAP + 2u)) {
                a.status->bounds = 1;
                it.act = false;
                it.idx = 0;
            }
#endif
            it.rel = sh.DCN_LIST_AT(l_, it.idx + o_skip);
            const long long o_s = __shfl((long long)s, l_, 64);
            it.o_uslot = o_pk & 0xFFu;
            it.p = (uint64_t)(o_s + it.rel);
#ifdef DCN_DEBUG_BOUNDS
            if (it.act && it.p + k > a.stream_bases) {
                a.status->bounds = 2;
                it.act = false;
                it.p = 0;
            }
#endif
            if (it.act) {
                const uint32_t *mp = a.invmask + (it.p >> 5);
                const uint32_t *pp = packed + (it.p >> 4);
#pragma unroll
                for (int q = 0; q < NMW; ++q) it.mw[q] = mp[q];
#pragma unroll
                for (int q = 0; q < NPW; ++q) it.pw[q] = pp[q];
            }
        };
        Item cur, nxt;
        if (PIPE && M) fetch(0, cur);
        for (uint32_t E = 0; E < ((DCN_EXP & 2) ? 0u : M); E += DCN_WAVE * U) {
            bool act[U];
            uint32_t lo[U], hi[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                act[u] = E + u * DCN_WAVE + lane < M;
                lo[u] = 0;
                hi[u] = 63;
            }
            // owner = largest lane whose list starts at or before e (6 steps, the U searches interleaved)
            if (!PIPE) {
#pragma unroll
            for (int it = 0; it < 6; ++it) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t e = E + u * DCN_WAVE + lane;
                    uint32_t mid = (lo[u] + hi[u] + 1) >> 1;
                    bool le = sh.start[mid] <= e;
                    lo[u] = le ? mid : lo[u];
                    hi[u] = le ? hi[u] : mid - 1;
                }
            }
            }
            uint64_t p[U];
            uint32_t o_uslot[U], idx[U], rel[U];
            uint32_t mw[U][NMW], pw[U][NPW];
            if (PIPE) {
                act[0] = cur.act;
                lo[0] = cur.lo;
                idx[0] = cur.idx;
                rel[0] = cur.rel;
                o_uslot[0] = cur.o_uslot;
                p[0] = cur.p;
#pragma unroll
                for (int q = 0; q < NMW; ++q) mw[0][q] = cur.mw[q];
#pragma unroll
                for (int q = 0; q < NPW; ++q) pw[0][q] = cur.pw[q];
            }
#pragma unroll
            for (int u = 0; u < (PIPE ? 0 : U); ++u) {
                uint32_t e = E + u * DCN_WAVE + lane;
                idx[u] = act[u] ? e - sh.start[lo[u]] : 0;
                const uint32_t o_pk = __shfl(us_skip, lo[u], 64);
                const uint32_t o_skip = o_pk >> 8;
#ifdef DCN_DEBUG_BOUNDS
                // the owner search puts idx inside the owner's list (idx < its entry count <= DCN_LCAP); a build that
                // broke the search (round 1's timing-only "no owner" experiment) read list rows far outside, took the
                // garbage as a position and faulted on the packed stream past its tail pad
                if (act[u] && (lo[u] > 63u || idx[u] + o_skip >= (uint32_t)DCN_LCAP + 2u)) {
                    a.status->bounds = 1;
                    act[u] = false;
                    idx[u] = 0;
                }
#endif
                rel[u] = sh.DCN_LIST_AT(lo[u], idx[u] + o_skip);
                long long o_s = __shfl((long long)s, lo[u], 64);
                o_uslot[u] = o_pk & 0xFFu;
                p[u] = (uint64_t)(o_s + rel[u]);
#ifdef DCN_DEBUG_BOUNDS
                if (act[u] && p[u] + k > a.stream_bases) { // a minimizer's k-mer lies inside its read, hence inside the stream
                    a.status->bounds = 2;
                    act[u] = false;
                    p[u] = 0;
                }
#endif
            }
#pragma unroll
            for (int u = 0; u < (PIPE ? 0 : U); ++u) {
                if (act[u]) {
                    const uint32_t *mp = a.invmask + (p[u] >> 5);
                    const uint32_t *pp = packed + (p[u] >> 4);
#pragma unroll
                    for (int q = 0; q < NMW; ++q) mw[u][q] = (DCN_EXP & 16) ? 0u : mp[q];
#pragma unroll
                    for (int q = 0; q < NPW; ++q) pw[u][q] = pp[q];
                }
            }
            bool valid[U];
            uint64_t hash[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                valid[u] = false;
                hash[u] = 0;
                if (act[u]) {
                    uint32_t msh = (uint32_t)(p[u] & 31);
                    if constexpr (K128) {
                        uint64_t mbits = ((uint64_t)__funnelshift_r(mw[u][1], mw[u][NMW - 1], msh) << 32) |
                                         __funnelshift_r(mw[u][0], mw[u][1], msh);
                        valid[u] = (mbits & ((~0ull) >> (64 - k))) == 0; // src/filter_common.rs:275-286
                    } else {
                        valid[u] = (__funnelshift_r(mw[u][0], mw[u][1], msh) & (0xFFFFFFFFu >> (32 - k))) == 0;
                    }
                    uint32_t psh = (uint32_t)(p[u] & 15) * 2;
                    uint64_t lo64 = ((uint64_t)__funnelshift_r(pw[u][1], pw[u][2], psh) << 32) |
                                    __funnelshift_r(pw[u][0], pw[u][1], psh);
                    if constexpr (K128) {
                        uint64_t hi64 = ((uint64_t)__funnelshift_r(pw[u][3], pw[u][NPW - 1], psh) << 32) |
                                        __funnelshift_r(pw[u][2], pw[u][3], psh);
                        hash[u] = dcn_kmer_hash128_bits(lo64, hi64, k);
                    } else {
                        hash[u] = dcn_kmer_hash64_bits(lo64, k);
                    }
                }
            }
            if (DUMP) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t o_eb = __shfl(emitted_before, lo[u], 64);
                    uint32_t o_rp = __shfl(tile_read_pos, lo[u], 64);
                    uint32_t o_carry = __shfl(carry, lo[u], 64);
                    long long o_s = __shfl((long long)s, lo[u], 64);
                    if (act[u]) {
                        uint64_t slot = (uint64_t)o_s + o_carry + o_eb + idx[u];
                        a.dump_hash[slot] = valid[u] ? hash[u] : 0;
                        a.dump_pos[slot] = a.dump_abs ? (uint32_t)p[u] : o_rp + rel[u];
                        a.dump_valid[slot] = valid[u] ? 1 : 0;
                    }
                }
                if (PIPE) {
                    if (E + DCN_WAVE < M) fetch(E + DCN_WAVE, nxt);
                    cur = nxt;
                }
                continue;
            }
            // set membership: one 32-byte group per item, the U loads in flight together; walking on to the next
            // group only when a group is full without the key (rare at load <= 0.5)
            uint32_t grp[U];
            dcn_group g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                grp[u] = dcn_group_of(hash[u], a.table.group_shift, a.table.group_mask);
                if (valid[u] && !(DCN_EXP & 1)) g[u] = dcn_load_group(a.table, grp[u]);
            }
            if (PIPE && E + DCN_WAVE < M) fetch(E + DCN_WAVE, nxt); // behind the probe, ahead of its use
            bool hit[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                hit[u] = false;
                if (valid[u]) {
                    if (DCN_EXP & 1) {
                        hit[u] = (hash[u] & 1) != 0;
                    } else if (hash[u] == 0) {
                        hit[u] = a.table.has_zero != 0;
                    } else {
                        int r = dcn_group_resolve(g[u], hash[u]);
                        while (r < 0) {
                            grp[u] = (grp[u] + 1) & a.table.group_mask;
                            r = dcn_group_resolve(dcn_load_group(a.table, grp[u]), hash[u]);
                        }
                        hit[u] = r == 1;
                    }
                } else if (act[u]) {
                    atomicSub(&sh.total[o_uslot[u]], 1u); // rare: k-mer with a non-ACGT base
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool lok = hit[u] && sh.lok[o_uslot[u]];
                // hits of units resolved in-wave: through the ring, compared with the unit's earlier hits.  Items
                // are in flat order, so a unit's hits occupy consecutive ring slots: a hit with `run` earlier hits
                // of its unit (earlier rounds: sh.hraw, this round: ballot arithmetic) compares with the `run`
                // slots before it.
                const unsigned long long hb = __ballot(lok);
                if (hb) { // wave-uniform
                    const unsigned long long lt = (1ull << lane) - 1;
                    const uint32_t nh = (uint32_t)__popcll(hb);
                    const uint32_t rank = (uint32_t)__popcll(hb & lt);
                    const uint32_t x = n_ring + rank;
                    if (lok) sh.ring_hash[x & (DCN_RCAP - 1)] = hash[u];
                    const unsigned long long below = hb & lt;
                    const uint32_t prev_lane = below ? 63u - (uint32_t)__clzll(below) : (uint32_t)lane;
                    const uint32_t prev_us = __shfl(o_uslot[u], prev_lane, 64);
                    const bool run_head = lok && (below == 0 || prev_us != o_uslot[u]);
                    const unsigned long long hm_all = __ballot(run_head);
                    const unsigned long long hm = hm_all & (lt | (1ull << lane));
                    const uint32_t head_lane = hm ? 63u - (uint32_t)__clzll(hm) : 0u;
                    const uint32_t rank_head = (uint32_t)__popcll(hb & ((1ull << head_lane) - 1));
                    const uint32_t run = lok ? sh.hraw[o_uslot[u]] + (rank - rank_head) : 0u;
                    __syncthreads();
                    bool dup = false;
                    for (uint32_t d0 = 0; __any(d0 < ((DCN_EXP & 8) ? 0u : run)); d0 += 4) {
                        uint64_t v0 = sh.ring_hash[(x - d0 - 1) & (DCN_RCAP - 1)];
                        uint64_t v1 = sh.ring_hash[(x - d0 - 2) & (DCN_RCAP - 1)];
                        uint64_t v2 = sh.ring_hash[(x - d0 - 3) & (DCN_RCAP - 1)];
                        uint64_t v3 = sh.ring_hash[(x - d0 - 4) & (DCN_RCAP - 1)];
                        dup |= (d0 + 0 < run && v0 == hash[u]) | (d0 + 1 < run && v1 == hash[u]) |
                               (d0 + 2 < run && v2 == hash[u]) | (d0 + 3 < run && v3 == hash[u]);
                    }
                    // a unit's hits of this round are one run of adjacent hit lanes: its first lane books the whole run
                    // (one LDS update per run instead of a same-address atomic per hit lane)
                    const unsigned long long db = __ballot(dup);
                    if (run_head) {
                        const unsigned long long later = hm_all & ~((2ull << lane) - 1);
                        const unsigned long long upto = later ? ((1ull << (__ffsll((long long)later) - 1)) - 1) : ~0ull;
                        const unsigned long long rm = hb & upto & ~lt;
                        sh.hraw[o_uslot[u]] += (uint32_t)__popcll(rm);
                        sh.hits[o_uslot[u]] += (uint32_t)__popcll(rm & ~db);
                    }
                    n_ring += nh;
                    __syncthreads();
                }
                // hits of every other unit: appended to the unit's RUN in the record array.  The array has one slot per
                // 2^rec_shift bases of the batch stream; the run of (this wave, unit) starts at the slot of the first window of
                // the unit's first tile in this wave and holds sh.ucap entries, one per 2^rec_shift windows of the unit's
                // tiles here: it cannot reach another run, and with rec_shift > 0 it CAN fill up (homopolymers and short-
                // period repeats give a hit every window or two) -- then status->run_overflow is raised and the host runs
                // the batch again with one slot per window (api.hip, grow_run_slots).  Hits fill it from the front in item order.  No global
                // atomics: the running length lives in LDS, and since items are in flat order the hits of one unit
                // sit in adjacent lanes of a round.  plan.hip's distinct pass reads the runs back (coalesced).
                // A zero hash (0 marks an empty set slot there) is flagged per unit instead of counted there.
                {
                    const bool rec = hit[u] && !lok;
                    const unsigned long long rb = (DCN_EXP & 256) ? 0ull : __ballot(rec);
#if DCN_EXPORT_UNIFORM
                    // Long reads: a wave's tiles belong to one to three units and items are in flat order, so nearly every
                    // round's hits are one unit's.  Then the run bookkeeping (a shuffle, a second ballot, the head-lane
                    // arithmetic) collapses to one popcount: rank among the hit lanes, one length update by the first.
                    const uint32_t us0 = rb ? (uint32_t)__builtin_amdgcn_readlane((int)o_uslot[u], __ffsll((long long)rb) - 1) : 0u;
                    const bool one_unit = rb && __ballot(rec && o_uslot[u] != us0) == 0;
                    if (one_unit) { // wave-uniform
                        const unsigned long long lt = (1ull << lane) - 1;
                        any_rec = true;
                        const uint32_t base = sh.uhits[us0];
                        if (rec) {
                            const uint32_t at = base + (uint32_t)__popcll(rb & lt);
                            if (at < sh.ucap[us0]) {
                                if (!(DCN_EXP & 512)) a.rec_hash[sh.run_base[us0] + at] = hash[u];
                            } else {
                                a.status->run_overflow = 1;
                            }
                            if (hash[u] == 0) a.g_zero[sh.unit_of[us0]] = 1;
                            if ((rb & lt) == 0) sh.uhits[us0] = base + (uint32_t)__popcll(rb); // (its first hit lane, after the reads)
                        }
                    } else
#endif
                    if (rb) { // wave-uniform
                        const unsigned long long lt = (1ull << lane) - 1;
                        const unsigned long long below = rb & lt;
                        const uint32_t prev_lane = below ? 63u - (uint32_t)__clzll(below) : (uint32_t)lane;
                        const uint32_t prev_us = __shfl(o_uslot[u], prev_lane, 64);
                        const bool run_head = rec && (below == 0 || prev_us != o_uslot[u]);
                        const unsigned long long hm_all = __ballot(run_head);
                        const unsigned long long hm = hm_all & (lt | (1ull << lane));
                        const uint32_t head_lane = hm ? 63u - (uint32_t)__clzll(hm) : 0u;
                        const uint32_t rank = (uint32_t)__popcll(below) - (uint32_t)__popcll(rb & ((1ull << head_lane) - 1));
                        any_rec = true;
                        if (rec) {
                            const uint32_t at = sh.uhits[o_uslot[u]] + rank;
                            if (at < sh.ucap[o_uslot[u]]) {
                                if (!(DCN_EXP & 512)) a.rec_hash[sh.run_base[o_uslot[u]] + at] = hash[u];
                            } else {
                                a.status->run_overflow = 1; // (only with rec_shift > 0; the batch is run again)
                            }
                            if (hash[u] == 0) a.g_zero[sh.unit_of[o_uslot[u]]] = 1;
                        }
                        if (run_head) { // after every lane of the run has read the old length: one update per run
                            const unsigned long long later = hm_all & ~((2ull << lane) - 1);
                            const unsigned long long upto = later ? ((1ull << (__ffsll((long long)later) - 1)) - 1) : ~0ull;
                            sh.uhits[o_uslot[u]] += (uint32_t)__popcll(rb & upto & ~lt);
                        }
                    }
                }
            }
            if (PIPE) cur = nxt;
        }
        if (!DUMP && cnt_eff) atomicAdd(&sh.total[uslot], cnt_eff); // invalid ones were subtracted above
        emitted_before += cnt_eff;
        if (skip0) first_pending = false;
        cnt = 0;
        __syncthreads();
#ifdef DCN_PHASEB_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- phase A: rolling scan in blocks of STEP bases ---------------------------------------------------------
    // Block b covers main steps t = (k-1) + STEP*b + r.  Each of the three base streams (incoming base t,
    // k-mer's first base t-(k-1), window's first base t-(l-1)) is read as one 32-bit word per block holding the
    // block's 2*STEP bits at static offsets 2r; the raw words of block b+1 are loaded while block b computes.
    constexpr uint32_t STEP = W > 0 ? (uint32_t)W : 8u; // steps between flush checks (= ring size for W > 0)
    static_assert(STEP < DCN_LCAP && STEP <= 16, "a block's bits must fit one word and the list must outlast a block");
    uint32_t ringL[STEP], ringR[STEP];
    uint32_t pl = 0xFFFFFFFFu, pr = 0;
    if (W > 0) {
#pragma unroll
        for (uint32_t r = 0; r < STEP; ++r) {
            ringL[r] = 0xFFFFFFFFu;
            ringR[r] = 0;
        }
    } else if (VAR) {
        for (uint32_t r = 0; r < w; ++r)
            reinterpret_cast<ulonglong2 *>(dyn_ring)[r * DCN_WAVE + lane] = make_ulonglong2(~0ull, 0ull);
    } else {
        for (uint32_t r = 0; r < w; ++r) dyn_ring[r * DCN_WAVE + lane] = make_uint2(0xFFFFFFFFu, 0u);
    }
    uint32_t gslot = 0; // W == 0: runtime ring slot, j % w
    uint32_t keymask = 0xFFFF0000u;
    asm volatile("" : "+v"(keymask)); // keep the mask in a VGPR so (h & mask) | j is one v_bfi_b32
    const uint32_t lhalf = l >> 1;    // window is canonical iff tg > l/2 (l odd)

    // bit positions (in the packed stream) of block 0 of the three streams
    int64_t b_in = 2 * (s + (int64_t)(k - 1)), b_k = 2 * s, b_l = 2 * (s - (int64_t)(w - 1));
    int64_t wi_in = b_in >> 5, wi_k = b_k >> 5, wi_l = b_l >> 5;
    uint32_t so_in = (uint32_t)(b_in & 31), so_k = (uint32_t)(b_k & 31), so_l = (uint32_t)(b_l & 31);
    uint32_t ra_in = packed[wi_in], rb_in = packed[wi_in + 1];
    uint32_t ra_k = packed[wi_k], rb_k = packed[wi_k + 1];
    uint32_t ra_l = packed[wi_l], rb_l = packed[wi_l + 1];
    uint32_t w_in = __funnelshift_r(ra_in, rb_in, so_in);
    uint32_t w_k = __funnelshift_r(ra_k, rb_k, so_k);
    uint32_t w_l = __funnelshift_r(ra_l, rb_l, so_l);
    auto advance_streams = [&]() {
        so_in += 2 * STEP; wi_in += so_in >> 5; so_in &= 31;
        so_k += 2 * STEP; wi_k += so_k >> 5; so_k &= 31;
        so_l += 2 * STEP; wi_l += so_l >> 5; so_l &= 31;
        if (DCN_EXP & 128) {
            ra_in = rb_in ^ 0x5A5A5A5Au; rb_in = ra_k + 0x1234567u; ra_k = rb_l; rb_k = ra_l ^ ra_in; ra_l = rb_k; rb_l = rb_in;
            return;
        }
        ra_in = packed[wi_in]; rb_in = packed[wi_in + 1];
        if (DCN_EXP & 64) {
            ra_k = ra_in; rb_k = rb_in; ra_l = ra_in; rb_l = rb_in;
        } else {
            ra_k = packed[wi_k]; rb_k = packed[wi_k + 1];
            ra_l = packed[wi_l]; rb_l = packed[wi_l + 1];
        }
    };
    advance_streams(); // raw words of block 1 in flight
    {
        // the main loop subtracts the TG bit of base t-(l-1) from its first step on; for t < l-1 that is one of
        // the w-1 bases in front of the tile: pre-add them so the subtraction cancels.
        if (W > 0) {
            uint32_t m = (w - 1) >= 16 ? 0xAAAAAAAAu : (0xAAAAAAAAu & ((1u << (2 * (w - 1))) - 1u));
            tg += __popc(w_l & m); // w_l of block 0 starts at base s-(w-1)
        } else {
            for (uint32_t b = 1; b < w; ++b) {
                int64_t p = s - (int64_t)b;
                tg += (packed[p >> 4] >> (2 * (uint32_t)(p & 15) + 1)) & 1u;
            }
        }
    }
    uint4 e_nx = sh.tab[(w_in & 3) | ((w_k & 3) << 2)]; // table entry of the block's first step, one step ahead

    // one block of STEP steps; FIRST: the tile's first block (for W > 0 exactly one window completes, at its end)
    auto block = [&](auto first_tag, const uint32_t jb) {
        constexpr bool FIRST = decltype(first_tag)::value;
        // nibble n of te / to = (k-code << 2 | in-code) of step 2n / 2n+1
        const uint32_t te = (w_in & 0x33333333u) | ((w_k & 0x33333333u) << 2);
        const uint32_t to = ((w_in >> 2) & 0x33333333u) | (w_k & 0xCCCCCCCCu);
        uint32_t nw_in = 0, nw_k = 0, nw_l = 0;
#pragma unroll
        for (uint32_t r = 0; r < STEP; ++r) {
            const uint32_t j = jb + r; // k-mer index within the tile scan (wave-uniform)
            const uint4 e = e_nx;
            if (r + 1 < STEP) {
                const uint32_t tsel = ((r + 1) & 1) ? to : te;
                const uint32_t n = (r + 1) >> 1;
                const uint32_t addr16 = n == 0 ? ((tsel << 4) & 0xF0u) : ((tsel >> (4 * n - 4)) & 0xF0u);
                e_nx = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(sh.tab) + addr16);
            } else {
                // last step: the next block's words are due; fetch its first table entry and the words after it
                nw_in = __funnelshift_r(ra_in, rb_in, so_in);
                nw_k = __funnelshift_r(ra_k, rb_k, so_k);
                nw_l = __funnelshift_r(ra_l, rb_l, so_l);
                e_nx = sh.tab[(nw_in & 3) | ((nw_k & 3) << 2)];
                advance_streams();
            }
            const uint32_t hi_in = (w_in >> (2 * r + 1)) & 1;
            const uint32_t hl = (w_l >> (2 * r + 1)) & 1;
            // (v_bitop3_b32, one instruction each: a ^ b ^ c = table 0x96; (a & b) | c = 0xEA; (~a & b) | c = 0xAE)
            fw = __builtin_amdgcn_bitop3_b32(rotl32(fw, VAR ? ROT : 1u), zprev, e.x, 0x96);
            rc = __builtin_amdgcn_bitop3_b32(rotl32(rc, VAR ? ROTR : 31u), wprev, e.y, 0x96);
            zprev = e.z;
            wprev = e.w;
            const uint32_t h = (VAR && a.nt_combine_xor) ? (fw ^ rc) : (fw + rc);
            const uint32_t jlow = j & 0xFFFFu;
            const uint32_t lk = __builtin_amdgcn_bitop3_b32(h, keymask, jlow, 0xEA);
            const uint32_t rk = __builtin_amdgcn_bitop3_b32(h, keymask, jlow, 0xAE);
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
            } else if constexpr (VAR) {
                // variant rules: a.cmp_mask selects the hash bits that are compared (top 16, or all 32); ties still go
                // to the leftmost / rightmost k-mer, so the position sits below the hash bits of a 64-bit key
                ulonglong2 *ring64 = reinterpret_cast<ulonglong2 *>(dyn_ring);
                ring64[gslot * DCN_WAVE + lane] = make_ulonglong2(((unsigned long long)(h & a.cmp_mask) << 32) | j,
                                                                  ((unsigned long long)(~h & a.cmp_mask) << 32) | j);
                gslot = gslot + 1 == w ? 0 : gslot + 1;
                unsigned long long lmin64 = ~0ull, rmax64 = 0ull;
                for (uint32_t qq = 0; qq < w; ++qq) {
                    ulonglong2 v = ring64[qq * DCN_WAVE + lane];
                    lmin64 = v.x < lmin64 ? v.x : lmin64;
                    rmax64 = v.y > rmax64 ? v.y : rmax64;
                }
                lmin = (uint32_t)lmin64;
                rmax = (uint32_t)rmax64;
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
            tg += hi_in;
            if constexpr (W > 0) {
                if (!FIRST || r == STEP - 1) {
                    const uint32_t sel = ((tg > lhalf) ? lmin : rmax) & 0xFFFFu;
                    // FIRST: window 0, always emitted (a carry tile's copy is dropped in phase B);
                    // later blocks: window j-(w-1) >= 1, emitted when in range and different from its predecessor
                    const bool emit = FIRST ? (nwc > 0) : ((j - (w - 1) < nwc) && sel != prev);
                    if (!(DCN_EXP & 4)) sh.DCN_LIST_AT(lane, cnt) = (uint16_t)sel;
                    cnt += emit ? 1u : 0u;
                    prev = sel;
                }
            } else {
                const uint32_t sel = ((tg > lhalf) ? lmin : rmax) & 0xFFFFu;
                const uint32_t i = j - (w - 1); // window index; wraps while j < w-1
                const bool in_range = i < nwc;  // unsigned compare: false while i is "negative"
                const bool emit = in_range && sel != prev;
                sh.DCN_LIST_AT(lane, cnt) = (uint16_t)sel;
                cnt += emit ? 1u : 0u;
                prev = in_range ? sel : prev;
            }
            tg -= hl;
        }
        w_in = nw_in;
        w_k = nw_k;
        w_l = nw_l;
    };

    if (jmax > 0) {
        block(BoolTag<true>{}, 0);
        if (__any(cnt > DCN_LCAP - STEP)) flush(IntTag<1>{}, false);
        for (uint32_t jb = STEP; jb < jmax; jb += STEP) {
            block(BoolTag<false>{}, jb);
            if (__any(cnt > DCN_LCAP - STEP)) flush(IntTag<1>{}, false);
        }
    }
    // Decision-only fast path (a.early_out_max_items != 0: the caller reads neither hit counts nor totals).  When every
    // tile of the wave is a whole unit of its own, never flushed mid-scan, and short enough that its required hits
    // equal abs_threshold for any number of valid minimizers, each lane walks ITS OWN list (no flattening) and stops
    // at abs_threshold distinct hits: the decision is already fixed (hits >= required, or for --deplete its negation),
    // and the minimizers it did not look at cannot change it.  Reads from the indexed genome stop after their first
    // abs_threshold minimizers; reads without hits are probed in full, exactly as below.  A pair whose two tiles sit
    // in adjacent lanes works the same way, both lanes holding the pair's state.
    if (FAST && !DUMP && !go_global) {
        const uint32_t skip0 = (first_pending && cnt > 0) ? 1u : 0u;
        const uint32_t cnt_eff = cnt - skip0;
        // units of one tile, or of two adjacent tiles (a pair): the two lanes keep identical copies of the unit's state
        // (cross-lane operations stay outside conditional expressions: inside a short-circuit they would only see
        // the lanes that got that far)
        const unsigned long long tile_mask = __ballot(have_tile);
        const unsigned long long up1 = lane < 63 ? 1ull << (lane + 1) : 0ull, down1 = lane > 0 ? 1ull << (lane - 1) : 0ull;
        const bool prev_head = (head_mask & down1) != 0;
        const bool next_mate = (tile_mask & up1) != 0 && (head_mask & up1) == 0;
        const uint32_t uh = head ? (uint32_t)lane : (uint32_t)lane - 1u;                          // the unit's first lane
        const uint32_t partner = head ? (next_mate ? (uint32_t)lane + 1u : (uint32_t)lane) : (uint32_t)lane - 1u;
        const uint32_t partner_cnt = __shfl(cnt_eff, partner, 64);
        const uint32_t cnt_unit = cnt_eff + (partner != (uint32_t)lane ? partner_cnt : 0u);
        const bool own = !have_tile || ((head || (prev_head && a.early_out_pairs)) && sh.local[uslot] && cnt_unit <= a.early_out_max_items);
        if (__all(own)) {
            const uint32_t need = (uint32_t)a.abs_threshold; // 1..4 (host side): need-1 earlier hits to remember
            uint64_t seen0 = 0, seen1 = 0, seen2 = 0;
            uint32_t nh = 0;
            // one minimizer at base position p: mask test, canonical k-mer, XXH3, set probe
            auto probe_item = [&](uint64_t p, bool actv, uint64_t &hash) -> bool {
                bool hit = false;
                hash = 0;
                if (actv) {
                    const uint32_t *mp = a.invmask + (p >> 5);
                    const uint32_t *pp = packed + (p >> 4);
                    const uint32_t m0 = mp[0], m1 = mp[1], m2 = K128 ? mp[2] : 0u;
                    uint32_t pw[K128 ? 5 : 3];
#pragma unroll
                    for (int q = 0; q < (K128 ? 5 : 3); ++q) pw[q] = pp[q];
                    const uint32_t msh = (uint32_t)(p & 31);
                    bool valid;
                    if constexpr (K128) {
                        const uint64_t mbits = ((uint64_t)__funnelshift_r(m1, m2, msh) << 32) | __funnelshift_r(m0, m1, msh);
                        valid = (mbits & ((~0ull) >> (64 - k))) == 0; // src/filter_common.rs:275-286
                    } else {
                        valid = (__funnelshift_r(m0, m1, msh) & (0xFFFFFFFFu >> (32 - k))) == 0;
                    }
                    const uint32_t psh = (uint32_t)(p & 15) * 2;
                    const uint64_t lo64 = ((uint64_t)__funnelshift_r(pw[1], pw[2], psh) << 32) | __funnelshift_r(pw[0], pw[1], psh);
                    if constexpr (K128) {
                        const uint64_t hi64 = ((uint64_t)__funnelshift_r(pw[3], pw[4], psh) << 32) | __funnelshift_r(pw[2], pw[3], psh);
                        hash = dcn_kmer_hash128_bits(lo64, hi64, k);
                    } else {
                        hash = dcn_kmer_hash64_bits(lo64, k);
                    }
                    if (valid) {
                        if (hash == 0) {
                            hit = a.table.has_zero != 0;
                        } else {
                            uint32_t grp = dcn_group_of(hash, a.table.group_shift, a.table.group_mask);
                            int r = dcn_group_resolve(dcn_load_group(a.table, grp), hash);
                            while (r < 0) {
                                grp = (grp + 1) & a.table.group_mask;
                                r = dcn_group_resolve(dcn_load_group(a.table, grp), hash);
                            }
                            hit = r == 1;
                        }
                    }
                }
                return hit;
            };
            auto note_hit = [&](uint64_t hash) { // this lane's unit has a hit on `hash`: count it if it is new
                const bool dup = (nh > 0 && hash == seen0) | (nh > 1 && hash == seen1) | (nh > 2 && hash == seen2);
                if (!dup) {
                    seen2 = nh == 2 ? hash : seen2;
                    seen1 = nh == 1 ? hash : seen1;
                    seen0 = nh == 0 ? hash : seen0;
                    ++nh;
                }
            };
            // rounds 0 .. R1-1: lane = its own list.  Reads from the indexed genome are decided here.
            const uint32_t maxc = wave_max_u32(cnt_eff);
            const uint32_t R1 = min(need + (uint32_t)DCN_FAST_EXTRA_ROUNDS, maxc);
            for (uint32_t j = 0; j < R1; ++j) {
                const bool actv = j < cnt_eff && nh < need;
                if (!__any(actv)) break;
                uint64_t hash;
                const uint32_t rel = sh.DCN_LIST_AT(lane, (actv ? j : 0u) + skip0);
                const bool hit = probe_item((uint64_t)(s + rel), actv, hash);
                // both lanes of a pair apply the unit's hits in the same order (first mate's, then second mate's)
                const bool p_hit = __shfl((int)hit, partner, 64) != 0;
                const uint64_t p_hash = (uint64_t)__shfl((long long)hash, partner, 64);
                const bool paired_lane = partner != (uint32_t)lane;
                const bool h1 = head ? hit : p_hit, h2 = head ? (paired_lane && p_hit) : hit;
                const uint64_t v1 = head ? hash : p_hash, v2 = head ? p_hash : hash;
                if (h1 && nh < need) note_hit(v1);
                if (h2 && nh < need) note_hit(v2);
            }
            // the rest: the undecided lanes' remaining entries, flattened over the wave so that no lane idles
            const uint32_t rem = (nh < need && cnt_eff > R1) ? cnt_eff - R1 : 0u;
            const uint32_t incl = wave_inclusive_scan_u32(rem, lane);
            const uint32_t M = wave_last_u32(incl);
            if (M) {
                sh.start[lane] = (uint16_t)(incl - rem);
                if (lane == 63) sh.start[64] = (uint16_t)M;
                __syncthreads();
                for (uint32_t E = 0; E < M; E += DCN_WAVE) {
                    const uint32_t e = E + lane;
                    const bool act = e < M;
                    uint32_t lo = 0, hi = 63; // owner = largest lane whose range starts at or before e
#pragma unroll
                    for (int it = 0; it < 6; ++it) {
                        const uint32_t mid = (lo + hi + 1) >> 1;
                        const bool le = sh.start[mid] <= e;
                        lo = le ? mid : lo;
                        hi = le ? hi : mid - 1;
                    }
                    const uint32_t idx = act ? e - sh.start[lo] + R1 : 0u;
                    const uint32_t o_skip = __shfl(skip0, lo, 64);
                    const uint32_t o_nh = __shfl(nh, lo, 64);
                    const long long o_s = __shfl((long long)s, lo, 64);
                    const uint32_t rel = sh.DCN_LIST_AT(lo, idx + o_skip);
                    uint64_t hash;
                    const bool hit = probe_item((uint64_t)(o_s + rel), act && o_nh < need, hash);
                    // hand each hit to its owner lane, one at a time (rare for reads that are not from the index)
                    unsigned long long hb = __ballot(hit);
                    while (hb) {
                        const int src = __ffsll((long long)hb) - 1;
                        hb &= hb - 1;
                        const uint64_t h = (uint64_t)__shfl((long long)hash, src, 64);
                        const uint32_t o = __shfl(lo, src, 64);
                        const uint32_t o_uh = __shfl(uh, o, 64);
                        if (uh == o_uh && nh < need) note_hit(h); // every lane of the owner's unit
                    }
                }
            }
            if (have_tile && head) {
                a.keep[t.unit] = (a.deplete ? nh < need : nh >= need) ? 1 : 0;
                a.unit_state[t.unit] = 1;
            }
            return;
        }
    }
    flush(IntTag<DCN_U_FINAL>{}, true);
    if (DUMP) {
        if (have_tile) a.dump_count[tile_idx] = emitted_before;
        return;
    }

    // ---- units left to the distinct pass: run lengths, hit totals, enrolment ------------------------------------------
    if (any_rec && lane == 0) a.status->any_records = 1;
    {
        const bool pending = have_tile && !sh.lok[uslot] && !(DCN_EXP & 1024);
        bool enrol = false;
        if (pending) {
            const uint32_t th = head ? sh.uhits[uslot] : 0u; // the run hangs on the unit's first tile in this wave
            a.tile_hits[tile_idx] = th;
            // one atomic per (wave, unit) with hits; whoever finds the unit's count at zero puts it on the work list
            // (a unit without any hit needs no distinct pass: its count stays 0)
            if (th) enrol = atomicAdd(&a.g_hitcnt[t.unit], th) == 0u;
        }
        const unsigned long long em = __ballot(enrol);
        if (em) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&a.status->n_pending, (uint32_t)__popcll(em));
            base = __shfl(base, 0, 64);
            if (enrol) {
                a.pending[base + (uint32_t)__popcll(em & ((1ull << lane) - 1))] = t.unit;
                if (t.whole_unit()) { // (more hits than the in-wave ring holds) the plan kernel left no tile range for it
                    a.unit_tile_first[t.unit] = tile_idx;
                    a.unit_tile_count[t.unit] = 1;
                }
            }
        }
    }

    // ---- results of the units this wave owns --------------------------------------------------------------
    if (head && have_tile) {
        uint32_t tot = sh.total[uslot];
        if (sh.lok[uslot]) {
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

template <int W, bool VAR = false>
int launch_w(const dcn_scan_args &args, uint32_t blocks, bool dump, bool k128, size_t dyn, hipStream_t stream) {
#define DCN_LAUNCH(K128_, DUMP_, FAST_)                                                               \
    do {                                                                                              \
        auto kern = scan_kernel<W, K128_, DUMP_, FAST_ && !VAR, VAR>;                                 \
        if (dyn > 0) {                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
            if (e != hipSuccess) return dcn_fail(DCN_ERR_HIP, std::string("dyn LDS: ") + hipGetErrorString(e)); \
        }                                                                                             \
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(DCN_WAVE), dyn, stream, args);                    \
    } while (0)
    const bool fast = !dump && args.early_out_max_items != 0;
    if (k128) {
        if (dump) DCN_LAUNCH(true, true, false);
        else if (fast) DCN_LAUNCH(true, false, true);
        else DCN_LAUNCH(true, false, false);
    } else {
        if (dump) DCN_LAUNCH(false, true, false);
        else if (fast) DCN_LAUNCH(false, false, true);
        else DCN_LAUNCH(false, false, false);
    }
#undef DCN_LAUNCH
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

} // namespace

// ---- the parity-pinning variant (process-wide) ---------------------------------------------------
namespace {
std::atomic<uint32_t> g_variant{DCN_VARIANT_DEFAULT}; // rot << 16 | cmp_bits << 8 | combine
}

uint32_t dcn_current_variant() { return g_variant.load(); }

int dcn_set_minimizer_variant(uint32_t nt_rot, uint32_t cmp_bits, uint32_t combine) {
    if (nt_rot < 1 || nt_rot > 31) return dcn_fail(DCN_ERR_ARG, "minimizer variant: rotation must be 1..31");
    if (cmp_bits != 16 && cmp_bits != 32) return dcn_fail(DCN_ERR_ARG, "minimizer variant: 16 or 32 hash bits are compared");
    if (combine > 1) return dcn_fail(DCN_ERR_ARG, "minimizer variant: combine is 0 (fw + rc) or 1 (fw ^ rc)");
    g_variant.store((nt_rot << 16) | (cmp_bits << 8) | combine);
    return DCN_OK;
}

int dcn_get_minimizer_variant(uint32_t *nt_rot, uint32_t *cmp_bits, uint32_t *combine) {
    const uint32_t v = g_variant.load();
    if (nt_rot) *nt_rot = v >> 16;
    if (cmp_bits) *cmp_bits = (v >> 8) & 0xFF;
    if (combine) *combine = v & 0xFF;
    return DCN_OK;
}

int dcn_launch_scan(const dcn_scan_args &args_in, uint32_t max_tiles, bool dump, hipStream_t stream) {
    if (max_tiles == 0) return DCN_OK;
    uint32_t blocks = (max_tiles + DCN_WAVE - 1) / DCN_WAVE;
    bool k128 = args_in.k > 32;
    const uint32_t v = args_in.variant ? args_in.variant : DCN_VARIANT_DEFAULT; // the index's rule, not the process's
    if (v != DCN_VARIANT_DEFAULT) {
        // not the rules of SURVEY.md 8a row A4: one generic kernel with the three choices as run-time values
        dcn_scan_args args = args_in;
        args.nt_rot = v >> 16;
        args.cmp_mask = ((v >> 8) & 0xFF) == 32 ? 0xFFFFFFFFu : 0xFFFF0000u;
        args.nt_combine_xor = v & 0xFF;
        args.early_out_max_items = 0;
        if (args.w > 128) return dcn_fail(DCN_ERR_ARG, "minimizer variant: w <= 128 (two u64 keys per window slot in LDS)");
        size_t dyn = (size_t)args.w * DCN_WAVE * sizeof(ulonglong2);
        return launch_w<0, true>(args, blocks, dump, k128, dyn, stream);
    }
    const dcn_scan_args &args = args_in;
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
