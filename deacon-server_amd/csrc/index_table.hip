// index_table.hip -- device-resident minimizer set (K4 of SURVEY.md section 2.3).
//
// Replaces the reference's FxHashSet<u64> index (src/index.rs:98-105 insert loop,
// src/filter_common.rs:144 `contains`).  Only membership is ever observed on the filter path, so the
// layout is free: open addressing over small groups of u64 slots (DCN_GROUP_SLOTS, dcn_internal.h), linear
// probing group by group.  One probe = one aligned group read inside one 64-byte HBM sector.
#include "dcn_internal.h"
#include "dcn_probe.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

// SKIP_ZERO: `keys` is an old slot array (0 = empty slot, not a key) being re-inserted into a larger table
template <bool SKIP_ZERO>
__global__ void table_insert_kernel(uint64_t *slots, uint32_t group_shift, uint32_t group_mask,
                                    const uint64_t *keys, uint64_t n, unsigned long long *n_new,
                                    uint32_t *has_zero) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long fresh = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t key = keys[i];
        if (key == 0) {
            if (!SKIP_ZERO && atomicExch(has_zero, 1u) == 0u) fresh++;
            continue;
        }
        uint32_t g = dcn_group_of(key, group_shift, group_mask);
        bool done = false;
        while (!done) {
            unsigned long long *grp = (unsigned long long *)(slots + (uint64_t)g * DCN_GROUP_SLOTS);
            for (int s = 0; s < DCN_GROUP_SLOTS && !done; ++s) {
                unsigned long long cur = __hip_atomic_load(&grp[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == key) {
                    done = true;
                } else if (cur == 0) {
                    unsigned long long old = atomicCAS(&grp[s], 0ull, (unsigned long long)key);
                    if (old == 0) {
                        fresh++;
                        done = true;
                    } else if (old == key) {
                        done = true;
                    }
                    // else: another key claimed the slot first -> keep walking
                }
            }
            g = (g + 1) & group_mask;
        }
    }
    if (fresh) atomicAdd(n_new, fresh);
}

// Keys straight from the bytes of an index file whose hashes are all 9-byte varints (0xFD + u64 LE, which is
// every hash >= 2^32): record i sits at raw[9 i].  `raw` is 8-byte aligned and padded by 16 readable bytes.
__global__ void table_insert_varint9_kernel(uint64_t *slots, uint32_t group_shift, uint32_t group_mask,
                                            const uint64_t *raw, uint64_t n, unsigned long long *n_new,
                                            uint32_t *has_zero, uint32_t *bad_marker) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long fresh = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t byte = 9 * i;
        uint64_t w0 = raw[byte >> 3], w1 = raw[(byte >> 3) + 1];
        uint32_t sh = (uint32_t)(byte & 7) * 8;
        uint32_t marker = (uint32_t)(w0 >> sh) & 0xFFu;
        // the 8 value bytes start one byte after the marker
        uint64_t key = sh == 56 ? w1 : (w0 >> (sh + 8)) | (w1 << (56 - sh));
        if (marker != 0xFDu) {
            atomicExch(bad_marker, 1u);
            continue;
        }
        if (key == 0) {
            if (atomicExch(has_zero, 1u) == 0u) fresh++;
            continue;
        }
        uint32_t g = dcn_group_of(key, group_shift, group_mask);
        bool done = false;
        while (!done) {
            unsigned long long *grp = (unsigned long long *)(slots + (uint64_t)g * DCN_GROUP_SLOTS);
            for (int s = 0; s < DCN_GROUP_SLOTS && !done; ++s) {
                unsigned long long cur = __hip_atomic_load(&grp[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == key) {
                    done = true;
                } else if (cur == 0) {
                    unsigned long long old = atomicCAS(&grp[s], 0ull, (unsigned long long)key);
                    if (old == 0) {
                        fresh++;
                        done = true;
                    } else if (old == key) {
                        done = true;
                    }
                }
            }
            g = (g + 1) & group_mask;
        }
    }
    if (fresh) atomicAdd(n_new, fresh);
}

// four keys per thread and step, their home groups in flight together (one dwordx4 each); the rare key whose home
// group is full without it walks on afterwards
__global__ void table_contains_kernel(dcn_table_view t, const uint64_t *keys, uint64_t n, uint8_t *out) {
    constexpr int U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * U) {
        uint64_t key[U];
        uint32_t grp[U];
        dcn_group g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + u * stride;
            key[u] = i < n ? keys[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            grp[u] = dcn_group_of(key[u], t.group_shift, t.group_mask);
            g[u] = dcn_load_group(t, grp[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + u * stride;
            if (i >= n) continue;
            bool hit;
            if (key[u] == 0) {
                hit = t.has_zero != 0;
            } else {
                int r = dcn_group_resolve(g[u], key[u]);
                while (r < 0) {
                    grp[u] = (grp[u] + 1) & t.group_mask;
                    r = dcn_group_resolve(dcn_load_group(t, grp[u]), key[u]);
                }
                hit = r == 1;
            }
            out[i] = hit ? 1 : 0;
        }
    }
}
} // namespace

// The table's allocation.  DCN_TABLE_CONTIGUOUS=1 (experiment, profiles/r04_ab.txt section 8) asks the runtime for physically
// contiguous memory -- the scan kernel's time moves by up to 8 % with where the 34 GB table landed (profiles/r03_placement.txt) --
// and falls back to the plain allocation when that is refused.
hipError_t dcn_table_malloc(uint64_t **p, uint64_t bytes) {
    static const bool contiguous = getenv("DCN_TABLE_CONTIGUOUS") != nullptr;
    if (contiguous) {
        if (hipExtMallocWithFlags((void **)p, bytes, hipDeviceMallocContiguous) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();
        static bool said = false;
        if (!said) fprintf(stderr, "deacon-hip: no contiguous allocation of %llu bytes for the table; plain hipMalloc\n", (unsigned long long)bytes);
        said = true;
    }
    return hipMalloc((void **)p, bytes);
}

int dcn_table_build(dcn_index *idx, const uint64_t *host_keys, uint64_t n) {
    DCN_HIP(hipSetDevice(idx->device));
    uint64_t groups = dcn_table_groups_for(n);
    if (groups > (1ull << 32)) return dcn_fail(DCN_ERR_CAPACITY, "index too large for 2^32 groups");
    idx->n_groups = groups;
    DCN_HIP(dcn_table_malloc(&idx->d_slots, groups * DCN_GROUP_SLOTS * sizeof(uint64_t)));
    DCN_HIP(hipMemset(idx->d_slots, 0, groups * DCN_GROUP_SLOTS * sizeof(uint64_t)));
    unsigned long long *d_new = nullptr;
    uint32_t *d_zero = nullptr;
    DCN_HIP(hipMalloc((void **)&d_new, sizeof(unsigned long long)));
    DCN_HIP(hipMalloc((void **)&d_zero, sizeof(uint32_t)));
    DCN_HIP(hipMemset(d_new, 0, sizeof(unsigned long long)));
    DCN_HIP(hipMemset(d_zero, 0, sizeof(uint32_t)));
    dcn_table_view v = idx->view();
    const uint64_t CHUNK = 1ull << 25; // 32 Mi keys = 256 MiB staged per step
    uint64_t *d_keys = nullptr;
    uint64_t chunk_cap = std::min<uint64_t>(std::max<uint64_t>(n, 1), CHUNK);
    DCN_HIP(hipMalloc((void **)&d_keys, chunk_cap * sizeof(uint64_t)));
    int rc = DCN_OK;
    for (uint64_t off = 0; off < n && rc == DCN_OK; off += CHUNK) {
        uint64_t m = std::min<uint64_t>(CHUNK, n - off);
        hipError_t e = hipMemcpy(d_keys, host_keys + off, m * sizeof(uint64_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rc = dcn_fail(DCN_ERR_HIP, std::string("hipMemcpy keys: ") + hipGetErrorString(e));
            break;
        }
        uint32_t blocks = (uint32_t)std::min<uint64_t>((m + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(table_insert_kernel<false>, dim3(blocks), dim3(256), 0, 0, idx->d_slots, v.group_shift,
                           v.group_mask, d_keys, m, d_new, d_zero);
        e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = dcn_fail(DCN_ERR_HIP, std::string("table insert: ") + hipGetErrorString(e));
    }
    unsigned long long h_new = 0;
    uint32_t h_zero = 0;
    if (rc == DCN_OK) {
        hipMemcpy(&h_new, d_new, sizeof(h_new), hipMemcpyDeviceToHost);
        hipMemcpy(&h_zero, d_zero, sizeof(h_zero), hipMemcpyDeviceToHost);
        idx->n_keys = h_new;
        idx->has_zero = h_zero != 0;
    }
    if (rc == DCN_OK && hipDeviceSynchronize() != hipSuccess) rc = dcn_fail(DCN_ERR_HIP, "table build: device synchronize failed");
    hipFree(d_keys);
    hipFree(d_new);
    hipFree(d_zero);
    return rc;
}

// one chunk of 9-byte varint records already on the device; counters accumulate over chunks
int dcn_table_insert_varint9(dcn_index *idx, const uint64_t *d_raw, uint64_t n, unsigned long long *d_new,
                             uint32_t *d_zero, uint32_t *d_bad, hipStream_t stream) {
    if (n == 0) return DCN_OK;
    dcn_table_view v = idx->view();
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(table_insert_varint9_kernel, dim3(blocks), dim3(256), 0, stream, idx->d_slots, v.group_shift,
                       v.group_mask, d_raw, n, d_new, d_zero, d_bad);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

int dcn_table_contains(const dcn_index *idx, const uint64_t *host_keys, uint64_t n, uint8_t *out) {
    DCN_HIP(hipSetDevice(idx->device));
    if (n == 0) return DCN_OK;
    uint64_t *d_keys = nullptr;
    uint8_t *d_out = nullptr;
    DCN_HIP(hipMalloc((void **)&d_keys, n * sizeof(uint64_t)));
    hipError_t e = hipMalloc((void **)&d_out, n);
    if (e != hipSuccess) {
        hipFree(d_keys);
        return dcn_fail(DCN_ERR_NOMEM, "hipMalloc contains out");
    }
    int rc = DCN_OK;
    e = hipMemcpy(d_keys, host_keys, n * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(table_contains_kernel, dim3(blocks), dim3(256), 0, 0, idx->view(), d_keys, n, d_out);
        e = hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) rc = dcn_fail(DCN_ERR_HIP, std::string("contains: ") + hipGetErrorString(e));
    hipFree(d_keys);
    hipFree(d_out);
    return rc;
}

// ---- the measured ceiling of the probe stage --------------------------------------------------------------------------
// Nothing but the home-group read of every key (one 16-byte request each), U of them in flight per lane, the results
// folded into a register: what the memory system gives THIS table for THIS key stream when no other work competes.
// d_keys == null: pseudo-random keys made in the kernel (uniform over the table: every probe an L2 miss).
namespace {
__device__ inline uint64_t probe_mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void probe_ceiling_kernel(dcn_table_view t, const uint64_t *keys, uint64_t n, uint64_t salt,
                                                            uint64_t *sink) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * U) {
        uint64_t key[U];
        u64x2 g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + u * stride;
            key[u] = keys ? (i < n ? keys[i] : 0) : probe_mix64(i + salt);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t grp = dcn_group_of(key[u], t.group_shift, t.group_mask);
            const u64x2 *p = reinterpret_cast<const u64x2 *>(t.slots + (uint64_t)grp * DCN_GROUP_SLOTS);
            g[u] = (i0 + u * stride < n) ? (NT ? __builtin_nontemporal_load(p) : *p) : u64x2{0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += g[u].x ^ g[u].y ^ key[u];
    }
    if (acc == 0x123456789ABCDEFull) sink[0] = acc; // keeps the loads alive
}

template <int U, bool NT>
int time_probe_ceiling(const dcn_index *idx, const uint64_t *d_keys, uint64_t n, uint32_t reps, uint32_t blocks,
                       uint64_t *d_sink, hipStream_t stream, double *rate) {
    struct Events { // destroyed on every way out, the DCN_HIP early returns included
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() {
            if (a) hipEventDestroy(a);
            if (b) hipEventDestroy(b);
        }
    } ev;
    DCN_HIP(hipEventCreate(&ev.a));
    DCN_HIP(hipEventCreate(&ev.b));
    hipLaunchKernelGGL((probe_ceiling_kernel<U, NT>), dim3(blocks), dim3(256), 0, stream, idx->view(), d_keys, n, 1ull, d_sink);
    DCN_HIP(hipEventRecord(ev.a, stream));
    for (uint32_t r = 0; r < reps; ++r)  // another salt per repetition: a generated stream never re-probes the lines of the last one
        hipLaunchKernelGGL((probe_ceiling_kernel<U, NT>), dim3(blocks), dim3(256), 0, stream, idx->view(), d_keys, n,
                           0x9E3779B97F4A7C15ull * (r + 2), d_sink);
    DCN_HIP(hipEventRecord(ev.b, stream));
    DCN_HIP(hipEventSynchronize(ev.b));
    float ms = 0;
    DCN_HIP(hipEventElapsedTime(&ms, ev.a, ev.b));
    DCN_HIP(hipGetLastError());
    *rate = ms > 0 ? (double)n * reps / (ms * 1e-3) : 0.0;
    return DCN_OK;
}
} // namespace

int dcn_table_probe_ceiling(const dcn_index *idx, const uint64_t *d_keys, uint64_t n, uint32_t reps, double *best,
                            hipStream_t stream) {
    DCN_HIP(hipSetDevice(idx->device));
    *best = 0;
    if (n == 0 || reps == 0) return DCN_OK;
    uint64_t *d_sink = nullptr;
    DCN_HIP(hipMalloc(&d_sink, sizeof(uint64_t)));
    double r[6] = {0, 0, 0, 0, 0, 0};
    int rc = DCN_OK;
    // the forms the microbenchmarks found within a few per cent of each other (profiles/microbench/probe_patterns.hip): the
    // best of them is the ceiling
    const uint32_t wide = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 32), narrow = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 8);
    if (rc == DCN_OK) rc = time_probe_ceiling<4, false>(idx, d_keys, n, reps, wide, d_sink, stream, &r[0]);
    if (rc == DCN_OK) rc = time_probe_ceiling<4, true>(idx, d_keys, n, reps, wide, d_sink, stream, &r[1]);
    if (rc == DCN_OK) rc = time_probe_ceiling<1, false>(idx, d_keys, n, reps, wide, d_sink, stream, &r[2]);
    if (rc == DCN_OK) rc = time_probe_ceiling<1, true>(idx, d_keys, n, reps, wide, d_sink, stream, &r[3]);
    if (rc == DCN_OK) rc = time_probe_ceiling<8, false>(idx, d_keys, n, reps, narrow, d_sink, stream, &r[4]);
    if (rc == DCN_OK) rc = time_probe_ceiling<8, true>(idx, d_keys, n, reps, narrow, d_sink, stream, &r[5]);
    hipFree(d_sink);
    for (double x : r) *best = std::max(*best, x);
    return rc;
}

int dcn_table_contains_device(const dcn_index *idx, const uint64_t *d_keys, uint64_t n, uint8_t *d_out,
                              hipStream_t stream) {
    DCN_HIP(hipSetDevice(idx->device));
    if (n == 0) return DCN_OK;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(table_contains_kernel, dim3(blocks), dim3(256), 0, stream, idx->view(), d_keys, n, d_out);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}

// ----------------------------------------------------------------------------------------------------
// index build support (f1 of SURVEY.md 8f): growable table, insertion of dumped minimizer hashes with the
// index-side filters (src/minimizers.rs:151-168), key export (src/index.rs:130-164 iterates the set)
// ----------------------------------------------------------------------------------------------------
namespace {

// p * log2(p) for p = count/total, computed on the HOST in f32 exactly as calculate_scaled_entropy does
// (src/minimizers.rs:110-116), so the device only subtracts table entries in the reference's order
constexpr int ENT_MAX = 57;
__device__ float g_plogp[ENT_MAX][ENT_MAX];

__device__ inline float scaled_entropy_dev(const uint8_t *kmer, uint32_t k) { // src/minimizers.rs:73-121
    if (k < 10) return 1.0f;
    uint32_t cnt[4] = {0, 0, 0, 0};
    uint32_t total = 0;
    for (uint32_t i = 0; i < k; ++i) {
        uint32_t c = kmer[i] | 0x20u;
        int j = c == 'a' ? 0 : c == 'c' ? 1 : c == 'g' ? 2 : c == 't' ? 3 : -1;
        if (j >= 0) {
            cnt[j]++;
            total++;
        }
    }
    if (total == 0) return 1.0f;
    float entropy = 0.0f;
    for (int j = 0; j < 4; ++j)
        if (cnt[j] > 0) entropy = __fsub_rn(entropy, g_plogp[total][cnt[j]]);
    return __fdiv_rn(entropy, 2.0f);
}

__global__ void insert_dump_kernel(uint64_t *slots, uint32_t group_shift, uint32_t group_mask, const uint64_t *hash,
                                   const uint8_t *valid, const uint32_t *abs_pos, uint64_t n, const uint8_t *ascii,
                                   uint32_t k, float entropy_threshold, unsigned long long *n_new, uint32_t *has_zero) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long fresh = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (valid[i] != 1) continue;
        if (entropy_threshold != 0.0f && scaled_entropy_dev(ascii + abs_pos[i], k) < entropy_threshold) continue;
        uint64_t key = hash[i];
        if (key == 0) {
            if (atomicExch(has_zero, 1u) == 0u) fresh++;
            continue;
        }
        uint32_t g = dcn_group_of(key, group_shift, group_mask);
        bool done = false;
        while (!done) {
            unsigned long long *grp = (unsigned long long *)(slots + (uint64_t)g * DCN_GROUP_SLOTS);
            for (int s = 0; s < DCN_GROUP_SLOTS && !done; ++s) {
                unsigned long long cur = __hip_atomic_load(&grp[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == key) {
                    done = true;
                } else if (cur == 0) {
                    unsigned long long old = atomicCAS(&grp[s], 0ull, (unsigned long long)key);
                    if (old == 0) {
                        fresh++;
                        done = true;
                    } else if (old == key) {
                        done = true;
                    }
                }
            }
            g = (g + 1) & group_mask;
        }
    }
    if (fresh) atomicAdd(n_new, fresh);
}

__global__ void count_valid_kernel(const uint8_t *valid, uint64_t n, unsigned long long *count) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += valid[i] == 1;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

__global__ void export_keys_kernel(const uint64_t *slots, uint64_t n_slots, uint64_t *out, uint64_t capacity,
                                   unsigned long long *cursor) {
    // A wave takes SPAN x 64 consecutive slots at a time and asks the cursor ONCE for all the keys in them: the table is
    // sparse (8-40 slots per key) and large, and one atomic per 64 slots on one address was the whole kernel (a 34 GB table:
    // 67 M of them, 0.2 s; the slots themselves stream by in 10 ms).
    constexpr int SPAN = 16;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t w0 = wave * (64ull * SPAN); w0 < n_slots; w0 += n_waves * (64ull * SPAN)) {
        uint64_t key[SPAN];
        unsigned long long m[SPAN];
        unsigned total = 0;
#pragma unroll
        for (int j = 0; j < SPAN; ++j) {
            const uint64_t i = w0 + 64ull * j + lane;
            key[j] = i < n_slots ? slots[i] : 0;
            m[j] = __ballot(key[j] != 0);
            total += (unsigned)__popcll(m[j]);
        }
        if (total == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(cursor, (unsigned long long)total);
        base = __shfl(base, 0, 64);
#pragma unroll
        for (int j = 0; j < SPAN; ++j) {
            if (key[j]) {
                const unsigned long long at = base + (unsigned long long)__popcll(m[j] & ((1ull << lane) - 1));
                if (at < capacity) out[at] = key[j];
            }
            base += (unsigned long long)__popcll(m[j]);
        }
    }
}

} // namespace

// capacity rule: power-of-two number of groups, at least 64, >= S slots per key.  S = DCN_SLOTS_PER_KEY_ROOMY when
// that table is at most DCN_ROOMY_MAX_GROUPS groups and a third of the device's free memory, DCN_SLOTS_PER_KEY
// otherwise; DCN_TABLE_SLOTS_PER_KEY=<n> fixes S (>= 2).  Call with the index's device current.
uint64_t dcn_table_groups_for(uint64_t n_keys) {
    auto groups_at = [&](uint64_t s) {
        uint64_t groups = 64;
        while (groups * DCN_GROUP_SLOTS < n_keys * s + 8) groups <<= 1;
        return groups;
    };
    if (const char *e = getenv("DCN_TABLE_SLOTS_PER_KEY")) {
        long s = atol(e);
        if (s >= 2 && s <= 64) return groups_at((uint64_t)s);
    }
    uint64_t roomy = groups_at(DCN_SLOTS_PER_KEY_ROOMY);
    if (roomy <= DCN_ROOMY_MAX_GROUPS) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && roomy * DCN_GROUP_SLOTS * sizeof(uint64_t) <= free_b / 3)
            return roomy;
        (void)hipGetLastError();
    }
    return groups_at(DCN_SLOTS_PER_KEY);
}

int dcn_table_alloc(dcn_index *idx, uint64_t n_keys_capacity) {
    DCN_HIP(hipSetDevice(idx->device));
    uint64_t groups = dcn_table_groups_for(n_keys_capacity);
    if (groups > (1ull << 32)) return dcn_fail(DCN_ERR_CAPACITY, "index too large for 2^32 groups");
    idx->n_groups = groups;
    DCN_HIP(dcn_table_malloc(&idx->d_slots, groups * DCN_GROUP_SLOTS * sizeof(uint64_t)));
    DCN_HIP(hipMemset(idx->d_slots, 0, groups * DCN_GROUP_SLOTS * sizeof(uint64_t)));
    // the memset runs on the null stream without waiting for the host, and the callers' streams are non-blocking
    // (they do not wait for the null stream): the table must be clear before anyone inserts or probes
    DCN_HIP(hipDeviceSynchronize());
    idx->n_keys = 0;
    idx->has_zero = false;
    return DCN_OK;
}

// make room for n_keys_capacity keys: allocate a larger table and re-insert every stored key
int dcn_table_reserve(dcn_index *idx, uint64_t n_keys_capacity) {
    DCN_HIP(hipSetDevice(idx->device));
    uint64_t want = dcn_table_groups_for(n_keys_capacity);
    if (want <= idx->n_groups) return DCN_OK;
    if (want > (1ull << 32)) return dcn_fail(DCN_ERR_CAPACITY, "index too large for 2^32 groups");
    uint64_t *old_slots = idx->d_slots;
    uint64_t old_n = idx->n_groups * DCN_GROUP_SLOTS;
    uint64_t keep_keys = idx->n_keys;
    bool keep_zero = idx->has_zero;
    idx->d_slots = nullptr;
    idx->n_groups = want;
    DCN_HIP(dcn_table_malloc(&idx->d_slots, want * DCN_GROUP_SLOTS * sizeof(uint64_t)));
    DCN_HIP(hipMemset(idx->d_slots, 0, want * DCN_GROUP_SLOTS * sizeof(uint64_t)));
    unsigned long long *d_new = nullptr;
    uint32_t *d_zero = nullptr;
    DCN_HIP(hipMalloc((void **)&d_new, sizeof(unsigned long long)));
    DCN_HIP(hipMalloc((void **)&d_zero, sizeof(uint32_t)));
    DCN_HIP(hipMemset(d_new, 0, sizeof(unsigned long long)));
    DCN_HIP(hipMemset(d_zero, 0, sizeof(uint32_t)));
    dcn_table_view v = idx->view();
    // the old slot array is a key list with holes (0 = empty slot)
    uint32_t blocks = (uint32_t)std::min<uint64_t>((old_n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(table_insert_kernel<true>, dim3(blocks), dim3(256), 0, 0, idx->d_slots, v.group_shift, v.group_mask,
                       old_slots, old_n, d_new, d_zero);
    hipError_t e = hipDeviceSynchronize();
    hipFree(old_slots);
    hipFree(d_new);
    hipFree(d_zero);
    if (e != hipSuccess) return dcn_fail(DCN_ERR_HIP, std::string("table rehash: ") + hipGetErrorString(e));
    idx->n_keys = keep_keys;
    idx->has_zero = keep_zero;
    return DCN_OK;
}

int dcn_table_count_valid(const uint8_t *d_valid, uint64_t n, uint64_t *count, hipStream_t stream) {
    unsigned long long *d_c = nullptr;
    DCN_HIP(hipMalloc((void **)&d_c, sizeof(unsigned long long)));
    DCN_HIP(hipMemsetAsync(d_c, 0, sizeof(unsigned long long), stream));
    if (n) {
        uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(count_valid_kernel, dim3(blocks), dim3(256), 0, stream, d_valid, n, d_c);
    }
    unsigned long long h = 0;
    hipError_t e = hipMemcpyAsync(&h, d_c, sizeof(h), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(d_c);
    if (e != hipSuccess) return dcn_fail(DCN_ERR_HIP, std::string("count valid: ") + hipGetErrorString(e));
    *count = h;
    return DCN_OK;
}

int dcn_table_insert_dump(dcn_index *idx, const uint64_t *d_hash, const uint8_t *d_valid, const uint32_t *d_abs_pos,
                          uint64_t n_slots, const uint8_t *d_ascii, float entropy_threshold, hipStream_t stream) {
    if (n_slots == 0) return DCN_OK;
    static bool table_ready = false; // the p*log2(p) table is the same for every build
    if (entropy_threshold != 0.0f && !table_ready) {
        static float host_tab[ENT_MAX][ENT_MAX];
        for (int t = 1; t < ENT_MAX; ++t)
            for (int c = 1; c <= t; ++c) {
                float p = (float)c / (float)t;
                host_tab[t][c] = p * log2f(p);
            }
        DCN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_plogp), host_tab, sizeof(host_tab)));
        table_ready = true;
    }
    unsigned long long *d_new = nullptr;
    uint32_t *d_zero = nullptr;
    DCN_HIP(hipMalloc((void **)&d_new, sizeof(unsigned long long)));
    DCN_HIP(hipMalloc((void **)&d_zero, sizeof(uint32_t)));
    DCN_HIP(hipMemsetAsync(d_new, 0, sizeof(unsigned long long), stream));
    DCN_HIP(hipMemsetAsync(d_zero, 0, sizeof(uint32_t), stream));
    dcn_table_view v = idx->view();
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n_slots + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(insert_dump_kernel, dim3(blocks), dim3(256), 0, stream, idx->d_slots, v.group_shift, v.group_mask,
                       d_hash, d_valid, d_abs_pos, n_slots, d_ascii, (uint32_t)idx->k, entropy_threshold, d_new, d_zero);
    unsigned long long h_new = 0;
    uint32_t h_zero = 0;
    hipError_t e = hipMemcpyAsync(&h_new, d_new, sizeof(h_new), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_zero, d_zero, sizeof(h_zero), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(d_new);
    hipFree(d_zero);
    if (e != hipSuccess) return dcn_fail(DCN_ERR_HIP, std::string("insert dump: ") + hipGetErrorString(e));
    // the kernel counts the zero key once per call: not new if an earlier chunk already had it
    idx->n_keys += h_new - ((h_zero && idx->has_zero) ? 1 : 0);
    idx->has_zero = idx->has_zero || h_zero != 0;
    return DCN_OK;
}

int dcn_table_export(const dcn_index *idx, uint64_t *host_out, uint64_t capacity, uint64_t *n_out) {
    DCN_HIP(hipSetDevice(idx->device));
    *n_out = idx->n_keys;
    if (capacity < idx->n_keys) return dcn_fail(DCN_ERR_CAPACITY, "output capacity too small: need " + std::to_string(idx->n_keys));
    uint64_t n_nonzero = idx->n_keys - (idx->has_zero ? 1 : 0);
    uint64_t at = 0;
    if (idx->has_zero) host_out[at++] = 0;
    if (n_nonzero == 0) return DCN_OK;
    uint64_t *d_out = nullptr;
    unsigned long long *d_cur = nullptr;
    DCN_HIP(hipMalloc((void **)&d_out, n_nonzero * sizeof(uint64_t)));
    hipError_t e = hipMalloc((void **)&d_cur, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(d_cur, 0, sizeof(unsigned long long));
    if (e == hipSuccess) {
        uint64_t n_slots = idx->n_groups * DCN_GROUP_SLOTS;
        uint32_t blocks = (uint32_t)std::min<uint64_t>((n_slots + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(export_keys_kernel, dim3(blocks), dim3(256), 0, 0, idx->d_slots, n_slots, d_out, n_nonzero, d_cur);
        // One hipMemcpy into the caller's pageable (and usually untouched) array moved 2 GB/s: the runtime stages it on one
        // thread, first-touch faults included -- 0.2 s of a 0.7 s index build for 50 M keys, seconds for a panhuman-sized
        // union.  Pieces of 32 MB through two page-locked buffers instead, each copied out by the host threads while the next
        // one crosses the link.
        const uint64_t bytes = n_nonzero * sizeof(uint64_t), PIECE = 32ull << 20;
        void *pin[2] = {nullptr, nullptr};
        hipStream_t st = nullptr;
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool piped = bytes > PIECE && hipHostMalloc(&pin[0], PIECE, hipHostMallocDefault) == hipSuccess &&
                     hipHostMalloc(&pin[1], PIECE, hipHostMallocDefault) == hipSuccess &&
                     hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) == hipSuccess;
        if (piped) {
            e = hipDeviceSynchronize();  // (the export kernel ran on the null stream)
            const uint64_t n_pieces = (bytes + PIECE - 1) / PIECE;
            auto issue = [&](uint64_t i) {
                const uint64_t off = i * PIECE, len = std::min(PIECE, bytes - off);
                hipError_t r = hipMemcpyAsync(pin[i & 1], (const uint8_t *)d_out + off, len, hipMemcpyDeviceToHost, st);
                return r == hipSuccess ? hipEventRecord(ev[i & 1], st) : r;
            };
            if (e == hipSuccess) e = issue(0);
            for (uint64_t i = 0; i < n_pieces && e == hipSuccess; ++i) {
                e = hipEventSynchronize(ev[i & 1]);
                if (e != hipSuccess) break;
                const uint64_t off = i * PIECE, len = std::min(PIECE, bytes - off);
                // (piece i + 1 goes into the other buffer, which piece i - 1 has left)
                if (i + 1 < n_pieces) e = issue(i + 1);
                dcn_host_parallel_copy((uint8_t *)(host_out + at) + off, pin[i & 1], len);
            }
            if (e != hipSuccess) (void)hipStreamSynchronize(st);
        } else {
            (void)hipGetLastError();
            e = hipMemcpy(host_out + at, d_out, bytes, hipMemcpyDeviceToHost);
        }
        for (int i = 0; i < 2; ++i) {
            if (ev[i]) hipEventDestroy(ev[i]);
            if (pin[i]) hipHostFree(pin[i]);
        }
        if (st) hipStreamDestroy(st);
    }
    hipFree(d_out);
    if (d_cur) hipFree(d_cur);
    if (e != hipSuccess) return dcn_fail(DCN_ERR_HIP, std::string("export keys: ") + hipGetErrorString(e));
    return DCN_OK;
}

// A replica on another GPU made from the KEYS instead of from the table: the table is sparse by design
// (8-40 slots per key: 34 GB for panhuman-1's 3.3 GB of keys), and an xGMI link is what a peer copy is bound by.  The source
// compacts its keys (the export kernel: the table streams by once, 10 ms), the compact array crosses the link (3.3 GB instead
// of 34), the target inserts them into an empty table of the same geometry (one launch).  `dst` arrives with every field of
// `src` copied, its own device set and d_slots null.  Membership, n_keys and has_zero are the source's; which slot of a group
// a key sits in is not, and nothing observes that.
int dcn_table_clone_by_keys(const dcn_index *src, dcn_index *dst) {
    const uint64_t n_nonzero = src->n_keys - (src->has_zero ? 1 : 0);
    const uint64_t table_bytes = src->n_groups * DCN_GROUP_SLOTS * sizeof(uint64_t);
    uint64_t *d_src_keys = nullptr, *d_dst_keys = nullptr;
    unsigned long long *d_cur = nullptr, *d_new = nullptr;
    uint32_t *d_zero = nullptr;
    hipError_t e = hipSetDevice(src->device);
    if (e == hipSuccess && n_nonzero) {
        e = hipMalloc((void **)&d_src_keys, n_nonzero * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&d_cur, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMemset(d_cur, 0, sizeof(unsigned long long));
        if (e == hipSuccess) {
            const uint64_t n_slots = src->n_groups * DCN_GROUP_SLOTS;
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_slots + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(export_keys_kernel, dim3(blocks), dim3(256), 0, 0, src->d_slots, n_slots, d_src_keys, n_nonzero, d_cur);
            e = hipGetLastError();
        }
        unsigned long long found = 0;
        if (e == hipSuccess) e = hipMemcpy(&found, d_cur, sizeof(found), hipMemcpyDeviceToHost);  // (also the kernel's end)
        if (e == hipSuccess && found != n_nonzero) {
            hipFree(d_src_keys);
            hipFree(d_cur);
            return dcn_fail(DCN_ERR_HIP, "index clone: the table holds " + std::to_string(found) + " keys, its header says " + std::to_string(n_nonzero));
        }
    }
    if (e == hipSuccess) e = hipSetDevice(dst->device);
    if (e == hipSuccess) e = dcn_table_malloc(&dst->d_slots, std::max<uint64_t>(table_bytes, 16));
    if (e == hipSuccess && table_bytes) e = hipMemsetAsync(dst->d_slots, 0, table_bytes, 0);
    if (e == hipSuccess && n_nonzero) {
        e = hipMalloc((void **)&d_dst_keys, n_nonzero * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemcpyPeer(d_dst_keys, dst->device, d_src_keys, src->device, n_nonzero * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMalloc((void **)&d_new, sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc((void **)&d_zero, sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_new, 0, sizeof(unsigned long long), 0);
        if (e == hipSuccess) e = hipMemsetAsync(d_zero, 0, sizeof(uint32_t), 0);
        if (e == hipSuccess) {
            const dcn_table_view v = dst->view();
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_nonzero + 255) / 256, 256 * 16);
            hipLaunchKernelGGL(table_insert_kernel<false>, dim3(blocks), dim3(256), 0, 0, dst->d_slots, v.group_shift, v.group_mask,
                               d_dst_keys, n_nonzero, d_new, d_zero);
            e = hipGetLastError();
        }
        unsigned long long inserted = 0;
        if (e == hipSuccess) e = hipMemcpy(&inserted, d_new, sizeof(inserted), hipMemcpyDeviceToHost);
        if (e == hipSuccess && inserted != n_nonzero) {
            e = hipErrorUnknown;
            (void)dcn_fail(DCN_ERR_HIP, "index clone: " + std::to_string(inserted) + " of " + std::to_string(n_nonzero) + " keys arrived");
        }
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (d_dst_keys) hipFree(d_dst_keys);
    if (d_new) hipFree(d_new);
    if (d_zero) hipFree(d_zero);
    (void)hipSetDevice(src->device);
    if (d_src_keys) hipFree(d_src_keys);
    if (d_cur) hipFree(d_cur);
    if (e != hipSuccess) {
        (void)hipSetDevice(dst->device);
        if (dst->d_slots) hipFree(dst->d_slots);
        dst->d_slots = nullptr;
        if (e == hipErrorUnknown) return DCN_ERR_HIP;  // (message set above)
        return dcn_fail(e == hipErrorOutOfMemory ? DCN_ERR_NOMEM : DCN_ERR_HIP, std::string("index clone by keys: ") + hipGetErrorString(e));
    }
    return DCN_OK;
}

// ----------------------------------------------------------------------------------------------------
// set algebra on device tables (f4 of SURVEY.md 8f): index::union (src/index.rs:563-664), index::diff (:421-536)
// ----------------------------------------------------------------------------------------------------
namespace {
// insert every key of `src` that is NOT in `minus` (minus.slots == nullptr: no filter) into dst
__global__ void table_copy_filtered_kernel(uint64_t *dst, uint32_t dst_shift, uint32_t dst_mask, const uint64_t *src,
                                           uint64_t src_slots, dcn_table_view minus, unsigned long long *n_new) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long fresh = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < src_slots; i += stride) {
        uint64_t key = src[i];
        if (key == 0) continue;
        if (minus.slots && dcn_table_contains_dev(minus, key)) continue;
        uint32_t g = dcn_group_of(key, dst_shift, dst_mask);
        bool done = false;
        while (!done) {
            unsigned long long *grp = (unsigned long long *)(dst + (uint64_t)g * DCN_GROUP_SLOTS);
            for (int s = 0; s < DCN_GROUP_SLOTS && !done; ++s) {
                unsigned long long cur = __hip_atomic_load(&grp[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur == key) {
                    done = true;
                } else if (cur == 0) {
                    unsigned long long old = atomicCAS(&grp[s], 0ull, (unsigned long long)key);
                    if (old == 0) {
                        fresh++;
                        done = true;
                    } else if (old == key) {
                        done = true;
                    }
                }
            }
            g = (g + 1) & dst_mask;
        }
    }
    if (fresh) atomicAdd(n_new, fresh);
}
} // namespace

// dst (freshly allocated, large enough) += keys of src that are not in `minus` (may be null)
int dcn_table_merge(dcn_index *dst, const dcn_index *src, const dcn_index *minus) {
    DCN_HIP(hipSetDevice(dst->device));
    unsigned long long *d_new = nullptr;
    DCN_HIP(hipMalloc((void **)&d_new, sizeof(unsigned long long)));
    DCN_HIP(hipMemset(d_new, 0, sizeof(unsigned long long)));
    dcn_table_view dv = dst->view();
    dcn_table_view mv;
    mv.slots = nullptr;
    mv.group_shift = 32;
    mv.group_mask = 0;
    mv.has_zero = 0;
    if (minus) mv = minus->view();
    uint64_t src_slots = src->n_groups * DCN_GROUP_SLOTS;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((src_slots + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(table_copy_filtered_kernel, dim3(blocks), dim3(256), 0, 0, dst->d_slots, dv.group_shift,
                       dv.group_mask, src->d_slots, src_slots, mv, d_new);
    unsigned long long h_new = 0;
    hipError_t e = hipMemcpy(&h_new, d_new, sizeof(h_new), hipMemcpyDeviceToHost);
    hipFree(d_new);
    if (e != hipSuccess) return dcn_fail(DCN_ERR_HIP, std::string("table merge: ") + hipGetErrorString(e));
    dst->n_keys += h_new;
    if (src->has_zero && !(minus && minus->has_zero) && !dst->has_zero) {
        dst->has_zero = true;
        dst->n_keys += 1;
    }
    return DCN_OK;
}
