// prefix_scan.hip -- exclusive prefix sum of u32 (plumbing for the tile planner and the distinct pass).
// Three launches: per-block sums, a single-block scan of the block sums, per-block rescan + offset.
#include "dcn_internal.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr uint32_t SCAN_TILE = SCAN_THREADS * SCAN_ITEMS; // 1024 values per block

__device__ inline uint32_t wave_inclusive_scan(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d, 64);
        if ((int)(threadIdx.x & 63) >= d) v += o;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, *total = block sum
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *total) {
    __shared__ uint32_t wave_sums[SCAN_THREADS / 64];
    uint32_t inc = wave_inclusive_scan(v);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads(); // protect wave_sums against the previous call
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / 64; ++i) {
        uint32_t s = wave_sums[i];
        if (i < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums(const uint32_t *in, uint32_t n, uint32_t *sums) {
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) v += in[base + i];
    uint32_t tot;
    block_exclusive_scan(v, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single block: exclusive scan of sums[0..nb) in place; sums[nb] = grand total
__global__ __launch_bounds__(SCAN_THREADS) void scan_sums_inplace(uint32_t *sums, uint32_t nb) {
    uint32_t carry = 0;
    for (uint32_t start = 0; start < nb; start += SCAN_THREADS) {
        uint32_t i = start + threadIdx.x;
        uint32_t v = i < nb ? sums[i] : 0;
        uint32_t tot;
        uint32_t ex = block_exclusive_scan(v, &tot);
        if (i < nb) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) sums[nb] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const uint32_t *in, uint32_t n, const uint32_t *sums,
                                                           uint32_t nb, uint32_t *out) {
    uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t vals[SCAN_ITEMS];
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        vals[i] = (base + i < n) ? in[base + i] : 0;
        v += vals[i];
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan(v, &tot) + sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = ex;
        ex += vals[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nb];
}

} // namespace

uint32_t dcn_scan_tmp_words(uint32_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 2; }

int dcn_launch_exclusive_scan(const uint32_t *d_in, uint32_t *d_out, uint32_t n, uint32_t *d_tmp,
                              hipStream_t stream) {
    if (n == 0) {
        DCN_HIP(hipMemsetAsync(d_out, 0, sizeof(uint32_t), stream));
        return DCN_OK;
    }
    uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(SCAN_THREADS), 0, stream, d_in, n, d_tmp);
    hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(SCAN_THREADS), 0, stream, d_tmp, nb);
    hipLaunchKernelGGL(scan_apply, dim3(nb), dim3(SCAN_THREADS), 0, stream, d_in, n, d_tmp, nb, d_out);
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}
