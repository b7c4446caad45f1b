// plan.hip -- the small kernels around the scan: tile planning (A1 of SURVEY.md section 8a), the exact
// distinct-hit pass for units that span several waves, the per-unit decision and the six counters.
#include "dcn_internal.h"
#include "dcn_plan.h"
#include "dcn_probe.h"

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
                                  uint32_t unit) {
    uint32_t wstart = j * a.tile_windows;
    uint32_t carry = j > 0 ? 1u : 0u;
    dcn_tile t;
    t.scan_start = off + wstart - carry;
    t.read_pos = wstart - carry;
    t.unit = unit;
    t.n_windows = min(a.tile_windows, nwin - wstart);
    t.flags = carry;
    a.tiles[first + j] = t;
}

// One launch plans the whole batch.  A workgroup takes PLAN_READS consecutive reads (8 per thread): effective lengths
// -> tile counts, a block-level exclusive sum, ONE atomicAdd on the global tile cursor for the block's range (a single
// word takes only ~88 atomics/us, hence the large block), then the tile descriptors.  Tiles of one read are
// consecutive, and so are the tiles of a unit whose reads sit in one workgroup (pairs always do: the block size is
// even); the order of the blocks' ranges is arbitrary, which nothing depends on: units carry (first tile, tile
// count).  A unit cut by a workgroup boundary is marked non-contiguous.
// PLAN_CH = reads per thread: 8 for batches of many short reads (few cursor atomics), 1 when reads are few / long
// (more workgroups, and the per-workgroup loop over long reads stays short).
template <uint32_t PLAN_CH>
__global__ __launch_bounds__(256) void plan_kernel(dcn_plan_args a) {
    constexpr uint32_t PLAN_READS = 256 * PLAN_CH;
    constexpr uint32_t OWN = 4; // tiles a thread writes itself; longer reads are finished by the whole workgroup
    __shared__ uint32_t s_tiles[PLAN_READS];
    __shared__ uint32_t s_first[PLAN_READS];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    __shared__ uint32_t long_reads[PLAN_READS];
    __shared__ uint32_t n_long;
    const uint32_t tid = threadIdx.x;
    const uint32_t block_first = blockIdx.x * PLAN_READS;
    const uint32_t block_end = min(a.n_reads, block_first + PLAN_READS);
    if (tid == 0) n_long = 0;
    // the one-byte probe for a trailing '\n' costs a 64-byte sector per read: skipped when the batch came packed
    // (a.ascii == null) or when the pack kernel, which reads every byte anyway, saw no '\n' at all
    const uint8_t *ascii = (a.ascii && a.status->any_newline) ? a.ascii : nullptr;
    uint32_t nwin[PLAN_CH], nt[PLAN_CH], loc[PLAN_CH];
    uint32_t carry = 0; // tiles of the chunks before this one
#pragma unroll
    for (uint32_t c = 0; c < PLAN_CH; ++c) {
        const uint32_t r = block_first + c * 256 + tid;
        nwin[c] = 0;
        nt[c] = 0;
        if (r < a.n_reads) {
            uint64_t off = a.offsets[r];
            nwin[c] = effective_windows(ascii, off, a.offsets[r + 1] - off, a.prefix_length, a.k, a.k + a.w - 1);
            nt[c] = (nwin[c] + a.tile_windows - 1) / a.tile_windows;
        }
        uint32_t inc = nt[c];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(inc, d, 64);
            if ((int)(tid & 63) >= d) inc += o;
        }
        __syncthreads(); // s_wave free again
        if ((tid & 63) == 63) s_wave[tid >> 6] = inc;
        __syncthreads();
        uint32_t wave_base = 0, chunk_total = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            if (i < (tid >> 6)) wave_base += s_wave[i];
            chunk_total += s_wave[i];
        }
        loc[c] = carry + wave_base + inc - nt[c];
        carry += chunk_total;
        s_tiles[c * 256 + tid] = nt[c];
    }
    if (tid == 0) s_base = carry ? atomicAdd(a.tile_cursor, carry) : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t c = 0; c < PLAN_CH; ++c) s_first[c * 256 + tid] = s_base + loc[c];
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
            a.unit_tile_first[u] = first;
            a.unit_tile_count[u] = count;
            // per-unit scratch of this batch starts from zero (saves two memset launches per batch)
            if (a.unit_state) { // null for the minimizer dump / index build, which keep no per-unit state
                a.unit_state[u] = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) a.unit_scratch[(uint64_t)q * a.scratch_stride + u] = 0;
            }
        }
        const uint64_t off = a.offsets[r];
        for (uint32_t j = 0; j < min(nt[c], OWN); ++j) write_tile(a, first, j, off, nwin[c], u);
        if (nt[c] > OWN) long_reads[atomicAdd(&n_long, 1u)] = c * 256 + tid;
    }
    __syncthreads();
    for (uint32_t q = 0; q < n_long; ++q) {
        const uint32_t li = long_reads[q], lr = block_first + li;
        const uint32_t lnt = s_tiles[li], lfirst = s_first[li];
        const uint32_t lu = a.unit_id ? a.unit_id[lr] - a.unit_base : lr;
        const uint64_t loff = a.offsets[lr];
        const uint32_t lnwin = effective_windows(ascii, loff, a.offsets[lr + 1] - loff, a.prefix_length, a.k, a.k + a.w - 1);
        for (uint32_t j = OWN + tid; j < lnt; j += blockDim.x) write_tile(a, lfirst, j, loff, lnwin, lu);
    }
}

// ---- exact distinct-hit count from (unit, hash) records ------------------------------------------------------
// (every kernel of the distinct pass returns at once when no hit record was appended in this batch)
// A unit with hit records gets a power-of-two region of >= 2x its record count; regions are handed out from one
// cursor in whatever order the units arrive (only units spanning waves have records, so the atomics are few).
__global__ __launch_bounds__(256) void distinct_cap_kernel(const uint32_t *g_hitcnt, uint32_t n_units, uint32_t *caps,
                                                          uint32_t *set_off, dcn_status *status) {
    if (status->any_records == 0) return;
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_units) return;
    uint32_t hc = g_hitcnt[u];
    uint32_t cap = 0;
    if (hc) {
        cap = 2;
        while (cap < 2u * hc && cap < (1u << 31)) cap <<= 1;
        set_off[u] = (uint32_t)atomicAdd(&status->set_cursor, (unsigned long long)cap);
    }
    caps[u] = cap;
}

__global__ __launch_bounds__(256) void distinct_clear_kernel(uint64_t *set_slots, uint64_t capacity, const dcn_status *status) {
    if (status->any_records == 0) return;
    uint64_t total = status->set_cursor;
    if (total > capacity) total = capacity;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) set_slots[i] = 0;
}

__global__ __launch_bounds__(256) void distinct_insert_kernel(dcn_distinct_args a) {
    if (a.status->any_records == 0) return;
    uint64_t total_slots = a.status->set_cursor;
    if (total_slots > a.set_capacity) {
        if (blockIdx.x == 0 && threadIdx.x == 0) a.status->rec_overflow = 1;
        return;
    }
    const unsigned long long seg = a.rec_capacity / DCN_REC_SHARDS;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1;
    // blocks are dealt round-robin over the shards; within a shard a grid-stride loop in whole waves
    for (uint32_t shard = blockIdx.x % DCN_REC_SHARDS; shard < DCN_REC_SHARDS; shard += DCN_REC_SHARDS) {
        unsigned long long n = a.status->rec_count[shard];
        if (n > seg) n = seg;
        const uint32_t blocks_per_shard = gridDim.x / DCN_REC_SHARDS;
        const uint64_t stride = (uint64_t)blocks_per_shard * blockDim.x;
        const uint64_t first = (uint64_t)(blockIdx.x / DCN_REC_SHARDS) * blockDim.x + threadIdx.x;
        for (uint64_t i0 = first - lane; i0 < n; i0 += stride) { // i0: wave-uniform base
            const uint64_t i = i0 + lane;
            bool fresh = false;
            uint32_t u = 0xFFFFFFFFu;
            if (i < n) {
                u = a.rec_unit[shard * seg + i];
                uint64_t h = a.rec_hash[shard * seg + i];
                if (h == 0) { // 0 marks an empty slot: a zero hash is tracked by a per-unit flag
                    fresh = atomicExch(&a.g_zero[u], 1u) == 0u;
                } else {
                    uint32_t cap = a.caps[u];
                    unsigned long long *region = (unsigned long long *)(a.set_slots + a.set_off[u]);
                    uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
                    uint32_t slot = ((lo ^ ((hi << 13) | (hi >> 19))) * 0x85EBCA6Bu) & (cap - 1);
                    for (;;) {
                        unsigned long long old = atomicCAS(&region[slot], 0ull, (unsigned long long)h);
                        if (old == 0) {
                            fresh = true;
                            break;
                        }
                        if (old == h) break;
                        slot = (slot + 1) & (cap - 1);
                    }
                }
            }
            // one atomicAdd per run of equal unit among the lanes with a new key (records of a wave round are
            // adjacent and mostly of one unit)
            const unsigned long long fb = __ballot(fresh);
            if (fb) {
                const unsigned long long below = fb & lt;
                const uint32_t prev_lane = below ? 63u - (uint32_t)__clzll(below) : (uint32_t)lane;
                const uint32_t prev_u = __shfl(u, prev_lane, 64);
                const bool run_head = fresh && (below == 0 || prev_u != u);
                const unsigned long long hm = __ballot(run_head);
                if (run_head) {
                    const unsigned long long later = hm & ~((2ull << lane) - 1);
                    const unsigned long long upto = later ? ((1ull << (__ffsll((long long)later) - 1)) - 1) : ~0ull;
                    atomicAdd(&a.g_distinct[u], (uint32_t)__popcll(fb & upto & ~lt));
                }
            }
        }
    }
}

// ---- A7/A10: decision for units not resolved by the scan kernel + the six counters ----------------------------
__global__ __launch_bounds__(256) void finish_kernel(dcn_finish_args a) {
    unsigned long long st[DCN_N_STATS] = {0, 0, 0, 0, 0, 0};
    // grid-stride: few blocks, so the six counters see a few thousand atomics instead of one set per 256 units
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < a.n_units; u += gridDim.x * blockDim.x) {
        uint32_t r0 = a.unit_first_read ? a.unit_first_read[u] : u;
        uint32_t r1 = a.unit_first_read ? a.unit_first_read[u + 1] : u + 1;
        bool keep;
        if (a.unit_state[u]) {
            keep = a.keep[u] != 0;
        } else {
            uint32_t tot = a.g_total[u], hc = a.g_distinct[u];
            keep = dcn_decide(hc, tot, a.abs_threshold, a.rel_threshold, a.deplete);
            a.keep[u] = keep ? 1 : 0;
            if (a.hits) a.hits[u] = hc;
            if (a.total) a.total[u] = tot;
        }
        if (a.offsets) {
            // ProcessingStats, src/local_filter.rs:346-371 (single) / :417-445 (pair)
            unsigned long long nseq = r1 - r0, bp = a.offsets[r1] - a.offsets[r0];
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
    if (a.status->rec_overflow) {
        // sticky: the status words are cleared before the next chunk, the report is read when the batch is waited for
        if (blockIdx.x == 0 && threadIdx.x < 64) {
            unsigned long long need = 0;
            for (uint32_t sidx = threadIdx.x; sidx < DCN_REC_SHARDS; sidx += 64) need = max(need, a.status->rec_count[sidx]);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) need = max(need, (unsigned long long)__shfl_xor((long long)need, d, 64));
            if (threadIdx.x == 0) {
                a.report->overflow = 1;
                atomicMax(&a.report->need, need * DCN_REC_SHARDS);
            }
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
__global__ __launch_bounds__(256) void probe_hashes_kernel(dcn_probe_hashes_args a) {
    uint64_t n = a.n_hashes;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t shard = blockIdx.x % DCN_REC_SHARDS;
    const unsigned long long seg = a.rec_capacity / DCN_REC_SHARDS;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = a.hashes[i];
        if (!dcn_table_contains_dev(a.table, h)) continue;
        // unit = largest u with hash_offsets[u] <= i
        uint32_t lo = 0, hi = a.n_units - 1;
        while (lo < hi) {
            uint32_t mid = lo + (hi - lo + 1) / 2;
            if (a.hash_offsets[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        unsigned long long r = atomicAdd(&a.status->rec_count[shard], 1ull);
        a.status->any_records = 1;
        if (r < seg) {
            a.rec_unit[shard * seg + r] = lo;
            a.rec_hash[shard * seg + r] = h;
        } else {
            a.status->rec_overflow = 1;
        }
        atomicAdd(&a.g_hitcnt[lo], 1u);
    }
}

__global__ __launch_bounds__(256) void hash_totals_kernel(const uint64_t *hash_offsets, uint32_t n_units,
                                                         uint32_t *g_total) {
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n_units) g_total[u] = (uint32_t)(hash_offsets[u + 1] - hash_offsets[u]);
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
    hipLaunchKernelGGL(distinct_cap_kernel, dim3((a.n_units + 255) / 256), dim3(256), 0, stream, a.g_hitcnt,
                       a.n_units, a.caps, a.set_off, a.status);
    hipLaunchKernelGGL(distinct_clear_kernel, dim3(2048), dim3(256), 0, stream, a.set_slots, a.set_capacity, a.status);
    hipLaunchKernelGGL(distinct_insert_kernel, dim3(DCN_REC_SHARDS * 32), dim3(256), 0, stream, a);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

int dcn_launch_finish(const dcn_finish_args &a, hipStream_t stream) {
    hipLaunchKernelGGL(finish_kernel, dim3(blocks_for(a.n_units, 1024)), dim3(256), 0, stream, a);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

int dcn_launch_probe_hashes(const dcn_probe_hashes_args &a, hipStream_t stream) {
    hipLaunchKernelGGL(hash_totals_kernel, dim3((a.n_units + 255) / 256), dim3(256), 0, stream, a.hash_offsets,
                       a.n_units, a.g_total);
    if (a.n_hashes)
        hipLaunchKernelGGL(probe_hashes_kernel, dim3(blocks_for(a.n_hashes)), dim3(256), 0, stream, a);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}
