// plan.hip -- the small kernels around the scan: tile planning (A1 of SURVEY.md section 8a), the exact
// distinct-hit pass for units that span several waves, the per-unit decision and the six counters.
#include "dcn_internal.h"
#include "dcn_plan.h"
#include "dcn_probe.h"

#include <algorithm>

namespace {

// ---- A1: effective sequence -> number of windows (src/filter_common.rs:217-229) -------------------------
__device__ inline uint32_t effective_windows(const uint8_t *ascii, uint64_t off, uint64_t len, uint64_t prefix,
                                             uint32_t k, uint32_t l) {
    if (len < k) return 0;                      // :217 -- tested on the full read, before the prefix cut
    uint64_t n = (prefix > 0 && len > prefix) ? prefix : len; // :222-226
    if (ascii && n > 0 && ascii[off + n - 1] == '\n') n -= 1; // :229 (ascii == null: no read of the batch ends in one)
    return n >= l ? (uint32_t)(n - l + 1) : 0;
}

__device__ inline void write_tile(const dcn_plan_args &a, uint32_t first, uint32_t j, uint64_t off, uint32_t nwin,
                                  uint32_t unit, bool whole_unit = false) {
    uint32_t wstart = j * a.tile_windows;
    uint32_t carry = j > 0 ? 1u : 0u;
    if (a.check_offsets && first + j >= a.max_tiles) { // reads that overlap (bad offsets that pass one by one): more tiles than
        a.status->bad_offsets = 1;                      // the batch's bases can have; the scan kernel does not run then
        return;
    }
    dcn_tile t;
    t.scan_start = off + wstart - carry;
    t.unit = unit;
    t.nwf = min(a.tile_windows, nwin - wstart) | (carry << 31) | (whole_unit ? 1u << 30 : 0u);
    a.tiles[first + j] = t;
    if (a.tile_read_pos) a.tile_read_pos[first + j] = wstart - carry;
}

// One launch plans the whole batch.  A workgroup takes PLAN_READS consecutive reads (8 per thread): effective lengths
// -> tile counts, a block-level exclusive sum, ONE atomicAdd on the global tile cursor for the block's range (a single
// word takes only ~88 atomics/us, hence the large block), then the tile descriptors.  Tiles of one read are
// consecutive, and so are the tiles of a unit whose reads sit in one workgroup (pairs always do: the block size is
// even); the order of the blocks' ranges is arbitrary, which nothing depends on: units carry (first tile, tile
// count).  A unit cut by a workgroup boundary is marked non-contiguous.
// PLAN_CH = reads per thread: 8 for batches of many short reads (few cursor atomics), 1 when reads are few / long
// (more workgroups, and the per-workgroup loop over long reads stays short).
//
// Two classes inside a workgroup's range (batches without unit ids, where a read is a unit): first the tiles of the
// reads that need several tiles, then the single-tile reads.  A stream that mixes long and short reads (BASELINE
// configs[4]) otherwise puts both kinds into most scan waves: the short reads' lanes idle through 60 % of the wave's
// steps, and -- worse -- the mid-scan flush a long tile forces takes the in-wave resolution away from every short read
// of the wave, whose hits then all go through the distinct pass (measured: 157 Gbp/s against 250 / 380 Gbp/s for the
// long / short reads alone).  Reads keep their order inside a class, and a read's tiles stay consecutive.
template <uint32_t PLAN_CH>
__global__ __launch_bounds__(256) void plan_kernel(dcn_plan_args a) {
    constexpr uint32_t PLAN_READS = 256 * PLAN_CH;
    constexpr uint32_t OWN = 4; // tiles a thread writes itself; longer reads are finished by the whole workgroup
    __shared__ uint32_t s_tiles[PLAN_READS];
    __shared__ uint32_t s_first[PLAN_READS];
    __shared__ uint32_t s_wave[4], s_wave1[4];
    __shared__ uint32_t s_base;
    __shared__ uint32_t long_reads[PLAN_READS];
    __shared__ uint32_t n_long;
    const uint32_t tid = threadIdx.x;
    const uint32_t block_first = blockIdx.x * PLAN_READS;
    const uint32_t block_end = min(a.n_reads, block_first + PLAN_READS);
    if (tid == 0) n_long = 0;
    // the one-byte probe for a trailing '\n' costs a 64-byte sector per read: skipped when the batch came packed
    // (a.ascii == null) or when the pack kernel, which reads every byte anyway, saw no '\n' at all
    const uint8_t *ascii = (a.ascii && (a.newline_flag ? *a.newline_flag : a.status->any_newline)) ? a.ascii : nullptr;
    uint32_t nwin[PLAN_CH], nt[PLAN_CH], loc[PLAN_CH];
    const bool two_classes = a.unit_id == nullptr;
    uint32_t carry = 0;  // tiles of the chunks before this one: all of them, or (two classes) those of multi-tile reads
    uint32_t carry1 = 0; // two classes: single-tile reads of the chunks before this one
#pragma unroll
    for (uint32_t c = 0; c < PLAN_CH; ++c) {
        const uint32_t r = block_first + c * 256 + tid;
        nwin[c] = 0;
        nt[c] = 0;
        if (r < a.n_reads) {
            uint64_t off = a.offsets[r], end = a.offsets[r + 1];
            // (device-pointer API: nobody has looked at this array before -- an array that was not written yet when the
            // kernel ran, say, must end in an error code, not in tiles that point outside the batch)
            if (a.check_offsets && (end < off || end > a.stream_bases)) {
                a.status->bad_offsets = 1;
                end = off = 0;
            }
            nwin[c] = effective_windows(ascii, off, end - off, a.prefix_length, a.k, a.k + a.w - 1);
            nt[c] = (nwin[c] + a.tile_windows - 1) / a.tile_windows;
        }
        const bool single = two_classes && nt[c] == 1;
        const uint32_t own = single ? 0u : nt[c], own1 = single ? 1u : 0u;
        uint32_t inc = own, inc1 = own1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(inc, d, 64), o1 = __shfl_up(inc1, d, 64);
            if ((int)(tid & 63) >= d) {
                inc += o;
                inc1 += o1;
            }
        }
        __syncthreads(); // s_wave free again
        if ((tid & 63) == 63) {
            s_wave[tid >> 6] = inc;
            s_wave1[tid >> 6] = inc1;
        }
        __syncthreads();
        uint32_t wave_base = 0, chunk_total = 0, wave_base1 = 0, chunk_total1 = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            if (i < (tid >> 6)) {
                wave_base += s_wave[i];
                wave_base1 += s_wave1[i];
            }
            chunk_total += s_wave[i];
            chunk_total1 += s_wave1[i];
        }
        // (a single-tile read's place is counted from the start of its class here; the class's start is added below)
        loc[c] = single ? (0x80000000u | (carry1 + wave_base1 + inc1 - 1u)) : (carry + wave_base + inc - own);
        carry += chunk_total;
        carry1 += chunk_total1;
        s_tiles[c * 256 + tid] = nt[c];
    }
    if (tid == 0) s_base = (carry + carry1) ? atomicAdd(a.tile_cursor, carry + carry1) : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t c = 0; c < PLAN_CH; ++c) {
        if (loc[c] & 0x80000000u) loc[c] = carry + (loc[c] & 0x7FFFFFFFu); // single-tile reads follow the multi-tile ones
        s_first[c * 256 + tid] = s_base + loc[c];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t c = 0; c < PLAN_CH; ++c) {
        const uint32_t r = block_first + c * 256 + tid;
        if (r >= a.n_reads) continue;
        const uint32_t first = s_base + loc[c];
        if (a.read_tiles) { // only the minimizer dump reads these back
            a.read_tiles[r] = nt[c];
            a.read_tile_first[r] = first;
        }
        uint32_t u = r;
        bool unit_head = true;
        if (a.unit_id) {
            u = a.unit_id[r] - a.unit_base;
            if (a.check_offsets) { // (as for the offsets: ids that are not 0, 0|1, ... n_units - 1 end in an error code --
                // an id outside the batch's units, and also a unit WITHOUT a read: its entry of unit_first_read would be a
                // previous batch's, and the finish kernel would follow it into this batch's offsets)
                const uint32_t prev = r ? a.unit_id[r - 1] : a.unit_base - 1u;
                if (u >= a.n_units || a.unit_id[r] - prev > 1u || (r == a.n_reads - 1 && u != a.n_units - 1)) {
                    a.status->bad_offsets = 1;
                    if (u >= a.n_units) u = 0;
                }
            }
            unit_head = r == 0 || a.unit_id[r - 1] != a.unit_id[r];
            if (unit_head) a.unit_first_read[u] = r;
            if (r == a.n_reads - 1) a.unit_first_read[a.n_units] = a.n_reads;
        }
        if (unit_head) {
            uint32_t count = nt[c];
            if (a.unit_id) {
                // the unit's other reads follow; they are contiguous in tile space only inside this workgroup
                uint32_t q = r + 1;
                while (q < a.n_reads && a.unit_id[q] - a.unit_base == u) {
                    if (q >= block_end) {
                        count = 0xFFFFFFFFu;
                        break;
                    }
                    count += s_tiles[q - block_first];
                    ++q;
                }
            }
            // a read that is a unit of its own with a single tile (every short read) carries that fact in its tile:
            // 8 bytes per unit less to write here, and to read for the scan kernel
            if (a.unit_id || nt[c] != 1) {
                a.unit_tile_first[u] = first;
                a.unit_tile_count[u] = count;
            }
            // (the four scratch words per unit -- g_total, g_hitcnt, g_distinct, g_zero -- are zero between batches:
            // finish_kernel puts back to zero what a batch touched, which for short reads is next to nothing, instead of
            // 16 bytes per unit being written here)
            if (a.unit_state) a.unit_state[u] = 0; // null for the minimizer dump / index build, which keep no per-unit state
        }
        const uint64_t off = a.offsets[r];
        for (uint32_t j = 0; j < min(nt[c], OWN); ++j) write_tile(a, first, j, off, nwin[c], u, !a.unit_id && nt[c] == 1);
        if (nt[c] > OWN) long_reads[atomicAdd(&n_long, 1u)] = c * 256 + tid;
    }
    __syncthreads();
    for (uint32_t q = 0; q < n_long; ++q) {
        const uint32_t li = long_reads[q], lr = block_first + li;
        const uint32_t lnt = s_tiles[li], lfirst = s_first[li];
        uint32_t lu = a.unit_id ? a.unit_id[lr] - a.unit_base : lr;
        if (a.check_offsets && lu >= a.n_units) lu = 0; // (flagged above)
        const uint64_t loff = a.offsets[lr];
        const uint32_t lnwin = effective_windows(ascii, loff, a.offsets[lr + 1] - loff, a.prefix_length, a.k, a.k + a.w - 1);
        for (uint32_t j = OWN + tid; j < lnt; j += blockDim.x) write_tile(a, lfirst, j, loff, lnwin, lu);
    }
}

// ---- exact distinct-hit count of the units the scan kernel did not finish ------------------------------------
// (long reads, pairs cut by a wave boundary, units with more hits than the in-wave ring holds.)  The scan kernel left
// every such unit's hits as runs in the record array (scan.hip): the run of (wave, unit) starts at slot
// scan_start + carry of the unit's first tile in that wave, tile_hits[that tile] hashes long (every other tile of the
// unit has tile_hits 0); a 0 entry stands for "nothing" (a zero-hash hit is flagged in g_zero).  g_hitcnt[u] is the
// unit's total run length, status->n_pending / pending[] the work list.
//
// Pass A, one wave per enrolled unit: up to DCN_LDS_SET_MAX hits are deduplicated in an LDS hash set right here -- no
// global atomics at all, which is where the old (unit, hash) CAS pass spent 0.64 ms per 600 Mbp of long reads.  A
// larger unit (a long read from the indexed genome) gets a region of the global set scratch instead, so that pass B
// can spread its runs over the whole chip.
#ifndef DCN_LDS_SET_LOG2
#define DCN_LDS_SET_LOG2 11 // 2048 slots = 16 KB: ten waves per CU (12 = 32 KB measured in profiles/r02_ab.txt)
#endif
constexpr uint32_t DCN_LDS_SET_SLOTS = 1u << DCN_LDS_SET_LOG2;
constexpr uint32_t DCN_LDS_SET_MAX = DCN_LDS_SET_SLOTS * 7 / 10 - 33; // hits deduplicated in LDS (load <= 0.68): 1400 for 2048 slots
// pass A walks a unit's tiles 64 at a time with one wave, a dependent pair of loads per step: a unit of more tiles than
// this (a read beyond ~260 kbp at 256 windows per tile) goes to pass B's one-wave-per-64-tiles form however few hits it
// has, so that a chromosome-sized read with a handful of hits is not one wave's serial tail
constexpr uint32_t DCN_LDS_WALK_MAX_TILES = 1024;

__device__ inline uint32_t set_slot_of(uint64_t h, uint32_t cap) {
    uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
    return ((lo ^ ((hi << 13) | (hi >> 19))) * 0x85EBCA6Bu) & (cap - 1);
}

__global__ __launch_bounds__(64) void unit_distinct_kernel(dcn_distinct_args a) {
    __shared__ unsigned long long set[DCN_LDS_SET_SLOTS];
    if (a.status->run_overflow) return; // a run was refused hits: the lengths no longer describe the runs, the batch is re-run
    const uint32_t NP = a.status->n_pending;
    const uint32_t lane = threadIdx.x;
    // the next unit's descriptor is fetched while this one is counted
    uint32_t nu = 0, nfirst = 0, ncount = 0, nH = 0, ntot = 0;
    auto fetch = [&](uint32_t i) {
        if (i < NP) {
            nu = a.pending[i];
            nfirst = a.unit_tile_first[nu];
            ncount = a.unit_tile_count[nu];
            nH = a.g_hitcnt[nu];
            ntot = a.g_total ? a.g_total[nu] : 0u;
        }
    };
    fetch(blockIdx.x);
    for (uint32_t i = blockIdx.x; i < NP; i += gridDim.x) {
        const uint32_t u = nu, first = nfirst, count = ncount, H = nH, tot = ntot;
        fetch(i + gridDim.x);
        if (H == 0) {
            if (lane == 0) a.caps[u] = 0;
            continue;
        }
        // Decisions only: the unit is kept or dropped by `distinct hits >= required`, and required is known now (the scan
        // has finished, so the unit's minimizer total is complete).  Fewer hits than required, duplicates or not: decided,
        // nothing to count.  Otherwise the count may stop as soon as it reaches required -- a read of the indexed genome has
        // hundreds of hits and needs a dozen.  (Counting mode reports the exact number and takes neither shortcut.)
        uint32_t enough = 0xFFFFFFFFu;
        if (a.g_total) {
            const uint64_t req = dcn_required_hits(a.abs_threshold, a.rel_threshold, tot);
            if ((uint64_t)H < req) {
                if (lane == 0) {
                    a.g_distinct[u] = 0; // any value below `required` gives the same decision
                    a.caps[u] = 0;
                }
                continue;
            }
            enough = req > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)req;
        }
        // (a count that may stop at `enough` never holds more than that many keys: the LDS set serves units of any hit count then)
        const uint32_t Hset = H < enough ? H : enough;
        if (count == 0xFFFFFFFFu || Hset > DCN_LDS_SET_MAX || count > DCN_LDS_WALK_MAX_TILES) {
            // global set: a power-of-two region of >= 2x the hit count, handed out from one cursor (all regions are
            // cleared by one small kernel between the passes)
            uint32_t cap = 64;
            while (cap < 2u * H && cap < (1u << 31)) cap <<= 1;
            unsigned long long off = 0;
            if (lane == 0) off = atomicAdd(&a.status->set_cursor, (unsigned long long)cap);
            off = (unsigned long long)__shfl((long long)off, 0, 64);
            const bool fits = off + cap <= a.set_capacity;
            // work items of pass B: one per 64 tiles of the unit (a run hangs on the first tile a scan wave held, so a
            // slice holds one or two runs), which spreads a chromosome-sized read over as many waves as it has runs
            const uint32_t n_items = count == 0xFFFFFFFFu ? 0u : (count + 63) / 64;
            uint32_t item0 = 0;
            if (lane == 0) {
                a.set_off[u] = (uint32_t)off;
                a.caps[u] = fits ? cap : 0u;
                if (!fits) a.status->rec_overflow = 1;
                else if (n_items) item0 = atomicAdd(&a.status->n_big, n_items);
                if (fits && count == 0xFFFFFFFFu) a.status->any_scattered = 1;
            }
            item0 = __shfl(item0, 0, 64);
            if (fits)
                for (uint32_t q = lane; q < n_items; q += 64) a.big[item0 + q] = make_uint2(u, 64 * q);
            continue;
        }
        // (the stop is tested once per 256 entries: up to 255 keys beyond `enough` may have gone in by then)
        const uint32_t most = enough == 0xFFFFFFFFu ? Hset : (Hset + 256u < H ? Hset + 256u : H);
        uint32_t cap = 64;
        while (cap < 2u * most && cap < DCN_LDS_SET_SLOTS) cap <<= 1;
        // the unit's first 64 tiles (usually all of them): both loads in flight while the set is cleared
        auto tile_run = [&](uint32_t t, uint32_t &n, uint64_t &slot0) {
            n = 0;
            slot0 = 0;
            if (t < count) {
                n = a.tile_hits[first + t];
                const dcn_tile tl = a.tiles[first + t];
                slot0 = (tl.scan_start + tl.carry()) >> a.rec_shift;
            }
        };
        uint32_t n;
        uint64_t slot0;
        tile_run(lane, n, slot0);
        for (uint32_t q = lane; q < cap; q += 64) set[q] = 0;
        __syncthreads();
        uint32_t distinct = 0;
        for (uint32_t t0 = 0; t0 < count && distinct < enough; t0 += 64) {
            if (t0) tile_run(t0 + lane, n, slot0);
            unsigned long long runs = __ballot(n != 0);
            while (runs && distinct < enough) { // one or two per unit: a run per wave that held tiles of it
                const int r = __ffsll((long long)runs) - 1;
                runs &= runs - 1;
                const uint32_t rn = __shfl(n, r, 64);
                const uint64_t rs = (uint64_t)__shfl((long long)slot0, r, 64);
                for (uint32_t j0 = 0; j0 < rn && distinct < enough; j0 += 256) { // four loads in flight per lane
                    uint64_t h[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t j = j0 + q * 64 + lane;
                        h[q] = j < rn ? a.rec_hash[rs + j] : 0ull;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bool fresh = false;
                        if (h[q] != 0) {
                            uint32_t sl = set_slot_of(h[q], cap);
                            for (;;) {
                                unsigned long long old = atomicCAS(&set[sl], 0ull, (unsigned long long)h[q]);
                                if (old == 0) {
                                    fresh = true;
                                    break;
                                }
                                if (old == h[q]) break;
                                sl = (sl + 1) & (cap - 1);
                            }
                        }
                        distinct += (uint32_t)__popcll(__ballot(fresh));
                    }
                }
            }
        }
        __syncthreads();
        if (lane == 0) {
            a.g_distinct[u] = distinct; // the zero-hash flag is added by the finish kernel
            a.caps[u] = 0;
        }
    }
}

// CAS-insert the run of `n` hashes at rec_hash[slot0 ..] into `region`; returns the number of new keys (wave-wide)
__device__ inline uint32_t insert_run(const uint64_t *rec_hash, uint64_t slot0, uint32_t n, unsigned long long *region,
                                      uint32_t cap, uint32_t lane) {
    uint32_t fresh_n = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 256) { // four keys per lane: their loads, then their first CAS, in flight together
        uint64_t h[4];
        uint32_t sl[4];
        unsigned long long old[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t j = j0 + q * 64 + lane;
            h[q] = j < n ? rec_hash[slot0 + j] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sl[q] = set_slot_of(h[q], cap);
            old[q] = h[q] ? atomicCAS(&region[sl[q]], 0ull, (unsigned long long)h[q]) : h[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bool fresh = false;
            if (h[q] != 0) {
                unsigned long long o = old[q];
                for (;;) {
                    if (o == 0) {
                        fresh = true;
                        break;
                    }
                    if (o == h[q]) break;
                    sl[q] = (sl[q] + 1) & (cap - 1);
                    o = atomicCAS(&region[sl[q]], 0ull, (unsigned long long)h[q]);
                }
            }
            fresh_n += (uint32_t)__popcll(__ballot(fresh));
        }
    }
    return fresh_n;
}

__global__ __launch_bounds__(256) void distinct_clear_kernel(uint64_t *set_slots, uint64_t capacity, const dcn_status *status) {
    if (status->set_cursor == 0 || status->rec_overflow || status->run_overflow) return;
    uint64_t total = status->set_cursor;
    if (total > capacity) total = capacity;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) set_slots[i] = 0;
}

// Pass B, one wave per work item (64 consecutive tiles of a unit with a global set): find the runs hanging on those
// tiles and CAS-insert them; one atomicAdd of the number of new keys per wave.  Units whose tiles are not contiguous
// in the tile array have no tile list: if there is one, every tile of the batch is looked at.  That is a unit of two or
// more reads cut by a planning block's boundary: never a single read, never a pair of a batch made of pairs only (the
// block size is even), but a pair CAN be cut when a batch mixes unit sizes (units of one and of two reads).
__global__ __launch_bounds__(256) void big_insert_kernel(dcn_distinct_args a) {
    if (a.status->set_cursor == 0 || a.status->rec_overflow || a.status->run_overflow) return;
    const uint32_t NB = a.status->n_big;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t gwave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t bi = gwave; bi < NB; bi += n_waves) {
        const uint2 item = a.big[bi];
        const uint32_t u = item.x;
        const uint32_t first = a.unit_tile_first[u], count = a.unit_tile_count[u];
        const uint32_t cap = a.caps[u];
        unsigned long long *region = (unsigned long long *)(a.set_slots + a.set_off[u]);
        const uint32_t t = item.y + lane;
        uint32_t n = 0;
        uint64_t slot0 = 0;
        if (t < count) {
            n = a.tile_hits[first + t];
            const dcn_tile tl = a.tiles[first + t];
            slot0 = (tl.scan_start + tl.carry()) >> a.rec_shift;
        }
        uint32_t fresh_n = 0;
        unsigned long long runs = __ballot(n != 0);
        while (runs) {
            const int r = __ffsll((long long)runs) - 1;
            runs &= runs - 1;
            fresh_n += insert_run(a.rec_hash, (uint64_t)__shfl((long long)slot0, r, 64), __shfl(n, r, 64), region, cap, lane);
        }
        if (lane == 0 && fresh_n) atomicAdd(&a.g_distinct[u], fresh_n);
    }
    if (a.status->any_scattered == 0) return;
    const uint32_t NT = *a.n_tiles;
    for (uint32_t t = gwave; t < NT; t += n_waves) {
        const dcn_tile tl = a.tiles[t];
        if (a.unit_state[tl.unit] || a.unit_tile_count[tl.unit] != 0xFFFFFFFFu) continue;
        const uint32_t cap = a.caps[tl.unit];
        const uint32_t n = a.tile_hits[t];
        if (!cap || !n) continue;
        const uint32_t fresh_n = insert_run(a.rec_hash, (tl.scan_start + tl.carry()) >> a.rec_shift, n,
                                            (unsigned long long *)(a.set_slots + a.set_off[tl.unit]), cap, lane);
        if (lane == 0 && fresh_n) atomicAdd(&a.g_distinct[tl.unit], fresh_n);
    }
}

// ---- A7/A10: decision for units not resolved by the scan kernel + the six counters ----------------------------
__global__ __launch_bounds__(256) void finish_kernel(dcn_finish_args a) {
    unsigned long long st[DCN_N_STATS] = {0, 0, 0, 0, 0, 0};
    // grid-stride: few blocks, so the six counters see a few thousand atomics instead of one set per 256 units; four
    // units per thread and step, their loads issued together (the loop is latency-bound: 0.036 -> 0.02 ms at 4 M units)
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u0 = blockIdx.x * blockDim.x + threadIdx.x; u0 < a.n_units; u0 += 4 * stride) {
        uint8_t state[4], kept[4];
        uint32_t r0[4], r1[4];
        uint64_t o0[4], o1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t u = u0 + q * stride;
            state[q] = 2; // 2 = no unit
            if (u < a.n_units) {
                state[q] = a.unit_state[u];
                kept[q] = a.keep[u];
                r0[q] = a.unit_first_read ? a.unit_first_read[u] : u;
                r1[q] = a.unit_first_read ? a.unit_first_read[u + 1] : u + 1;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o0[q] = o1[q] = 0;
            // (bad_offsets: the read ranges of the units may be as wrong as the offsets; the counters are not written then)
            if (state[q] != 2 && a.offsets && !a.status->bad_offsets) {
                o0[q] = a.offsets[r0[q]];
                o1[q] = a.offsets[r1[q]];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (state[q] == 2) continue;
            const uint32_t u = u0 + q * stride;
            bool keep;
            if (state[q]) {
                keep = kept[q] != 0;
            } else {
                uint32_t tot = a.g_total[u], hc = a.g_distinct[u] + (a.g_zero[u] ? 1u : 0u);
                // this unit went through the global scratch words: leave them zero for the next batch
                a.g_total[u] = 0;
                a.g_hitcnt[u] = 0;
                a.g_distinct[u] = 0;
                a.g_zero[u] = 0;
                keep = dcn_decide(hc, tot, a.abs_threshold, a.rel_threshold, a.deplete);
                a.keep[u] = keep ? 1 : 0;
                if (a.hits) a.hits[u] = hc;
                if (a.total) a.total[u] = tot;
            }
            if (a.offsets) {
                // ProcessingStats, src/local_filter.rs:346-371 (single) / :417-445 (pair)
                unsigned long long nseq = r1[q] - r0[q], bp = o1[q] - o0[q];
                st[DCN_STAT_TOTAL_SEQS] += nseq;
                st[DCN_STAT_TOTAL_BP] += bp;
                if (keep) {
                    st[DCN_STAT_OUTPUT_BP] += bp;
                    st[DCN_STAT_OUTPUT_SEQ_COUNTER] += nseq;
                } else {
                    st[DCN_STAT_FILTERED_SEQS] += nseq;
                    st[DCN_STAT_FILTERED_BP] += bp;
                }
            }
        }
    }
    if (a.status->bounds && blockIdx.x == 0 && threadIdx.x == 0) a.report->bounds = a.status->bounds;
    if (a.status->bad_offsets) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.report->bad_offsets = 1;
        return; // (the counters would be sums over those offsets)
    }
    if (a.status->rec_overflow || a.status->run_overflow) {
        // sticky: the status words are cleared before the next chunk, the report is read when the batch is waited for
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicOr(&a.report->overflow, (a.status->rec_overflow ? 1u : 0u) | (a.status->run_overflow ? 2u : 0u));
            if (a.status->rec_overflow) atomicMax(&a.report->need, (a.status->set_cursor + 3) / 4);
        }
        return; // an overflowed attempt is re-run: do not count it
    }
    if (!a.offsets) return;
    __shared__ unsigned long long red[DCN_N_STATS][4];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < DCN_N_STATS; ++c) {
        unsigned long long v = st[c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0) red[c][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < DCN_N_STATS) {
        unsigned long long v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        if (v) atomicAdd(&a.report->stats[threadIdx.x], v);
    }
}

// ---- server seam: probe precomputed hashes (src/remote_filter.rs:230-301) ------------------------------------
// Every unit becomes one pseudo-tile whose run is its own slice of the hash array: the probe kernel overwrites each
// hash that is not in the index with 0 ("no entry"), and the distinct pass above does the rest.
__global__ __launch_bounds__(256) void probe_hashes_kernel(dcn_probe_hashes_args a) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n_hashes; i += stride) {
        uint64_t h = a.hashes[i];
        if (h == 0) {
            if (a.table.has_zero) {
                // unit = largest u with hash_offsets[u] <= i
                uint32_t lo = 0, hi = a.n_units - 1;
                while (lo < hi) {
                    uint32_t mid = lo + (hi - lo + 1) / 2;
                    if (a.hash_offsets[mid] <= i) lo = mid;
                    else hi = mid - 1;
                }
                a.g_zero[lo] = 1;
            }
        } else if (!dcn_table_contains_dev(a.table, h)) {
            a.hashes[i] = 0;
        }
    }
}

__global__ __launch_bounds__(256) void hash_units_kernel(dcn_probe_hashes_args a) {
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u == 0) {
        *a.n_tiles = a.n_units;
        a.status->any_records = 1;
        a.status->n_pending = a.n_units;
    }
    if (u >= a.n_units) return;
    const uint64_t n = a.hash_offsets[u + 1] - a.hash_offsets[u];
    dcn_tile t;
    t.scan_start = a.hash_offsets[u];
    t.unit = u;
    t.nwf = (uint32_t)n & 0x3FFFFFFFu; // (the run length travels in tile_hits; no scan ever reads this field here)
    a.tiles[u] = t;
    a.tile_hits[u] = (uint32_t)n;
    a.unit_tile_first[u] = u;
    a.unit_tile_count[u] = 1;
    a.unit_state[u] = 0;
    a.pending[u] = u;
    a.g_total[u] = (uint32_t)n;
    a.g_hitcnt[u] = (uint32_t)n;
    a.g_distinct[u] = 0;
    a.g_zero[u] = 0;
}

inline uint32_t blocks_for(uint64_t n, uint32_t cap = 256 * 16) {
    uint64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    return (uint32_t)(b > cap ? cap : b);
}

} // namespace

int dcn_launch_plan(const dcn_plan_args &a, hipStream_t stream) {
    if (a.n_reads >= (1u << 19))
        hipLaunchKernelGGL(plan_kernel<8>, dim3((a.n_reads + 2047) / 2048), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(plan_kernel<1>, dim3((a.n_reads + 255) / 256), dim3(256), 0, stream, a);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

int dcn_launch_distinct(const dcn_distinct_args &a, hipStream_t stream) {
    hipLaunchKernelGGL(unit_distinct_kernel, dim3(std::max(1u, std::min(a.n_units, 256u * 10u))), dim3(64), 0, stream, a);
    hipLaunchKernelGGL(distinct_clear_kernel, dim3(1024), dim3(256), 0, stream, a.set_slots, a.set_capacity, a.status);
    hipLaunchKernelGGL(big_insert_kernel, dim3(2048), dim3(256), 0, stream, a);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

int dcn_launch_finish(const dcn_finish_args &a, hipStream_t stream) {
    hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(a.n_units, 1024)), dim3(256), 0, stream, a);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

int dcn_launch_probe_hashes(const dcn_probe_hashes_args &a, hipStream_t stream) {
    hipLaunchKernelGGL(hash_units_kernel, dim3((a.n_units + 255) / 256), dim3(256), 0, stream, a);
    if (a.n_hashes)
        hipLaunchKernelGGL(probe_hashes_kernel, dim3(blocks_for(a.n_hashes)), dim3(256), 0, stream, a);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}
