// index_table.hip -- device-resident minimizer set (K4 of SURVEY.md section 2.3).
//
// Replaces the reference's FxHashSet<u64> index (src/index.rs:98-105 insert loop,
// src/filter_common.rs:144 `contains`).  Only membership is ever observed on the filter path, so the
// layout is free: open addressing over small groups of u64 slots (DCN_GROUP_SLOTS, dcn_internal.h), linear
// probing group by group.  One probe = one aligned group read inside one 64-byte HBM sector.
#include "dcn_internal.h"
#include "dcn_probe.h"

#include <algorithm>
#include <vector>

namespace {

__global__ void table_insert_kernel(uint64_t *slots, uint32_t group_shift, uint32_t group_mask,
                                    const uint64_t *keys, uint64_t n, unsigned long long *n_new,
                                    uint32_t *has_zero) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long fresh = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t key = keys[i];
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
                    // else: another key claimed the slot first -> keep walking
                }
            }
            g = (g + 1) & group_mask;
        }
    }
    if (fresh) atomicAdd(n_new, fresh);
}

__global__ void table_contains_kernel(dcn_table_view t, const uint64_t *keys, uint64_t n, uint8_t *out) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = dcn_table_contains_dev(t, keys[i]) ? 1 : 0;
}
} // namespace

int dcn_table_build(dcn_index *idx, const uint64_t *host_keys, uint64_t n) {
    DCN_HIP(hipSetDevice(idx->device));
    // capacity: >= DCN_SLOTS_PER_KEY slots per key, power-of-two number of groups, at least 64 groups
    uint64_t groups = 64;
    while (groups * DCN_GROUP_SLOTS < n * DCN_SLOTS_PER_KEY + 8) groups <<= 1;
    if (groups > (1ull << 32)) return dcn_fail(DCN_ERR_CAPACITY, "index too large for 2^32 groups");
    idx->n_groups = groups;
    DCN_HIP(hipMalloc((void **)&idx->d_slots, groups * DCN_GROUP_SLOTS * sizeof(uint64_t)));
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
        hipLaunchKernelGGL(table_insert_kernel, dim3(blocks), dim3(256), 0, 0, idx->d_slots, v.group_shift,
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
    hipFree(d_keys);
    hipFree(d_new);
    hipFree(d_zero);
    return rc;
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

int dcn_table_contains_device(const dcn_index *idx, const uint64_t *d_keys, uint64_t n, uint8_t *d_out,
                              hipStream_t stream) {
    DCN_HIP(hipSetDevice(idx->device));
    if (n == 0) return DCN_OK;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(table_contains_kernel, dim3(blocks), dim3(256), 0, stream, idx->view(), d_keys, n, d_out);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}
