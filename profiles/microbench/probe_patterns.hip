// Microbenchmark: scattered reads of an HBM-resident table, varying the per-probe access shape.
// build: hipcc -O3 --offload-arch=gfx950 probe_patterns.hip -o probe_patterns ; run: ./probe_patterns [log2_slots] [n_probes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
__device__ inline ulonglong2 nt_load(const ulonglong2 *p) {
    ull2 v = __builtin_nontemporal_load(reinterpret_cast<const ull2 *>(p));
    return make_ulonglong2(v.x, v.y);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

// MODE 0: two dwordx4 (32 B) per probe        MODE 1: one dwordx4 (16 B) per probe
// MODE 2: one dwordx2 (8 B) per probe         MODE 3: lane pairs share a probe, 16 B each (32 B per probe)
// MODE 4: MODE 0 with nontemporal loads       MODE 5: MODE 1 with nontemporal loads
// MODE 6: four lanes share a probe, 16 B each (64 B per probe, one full sector)
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void probe(const uint64_t *__restrict__ table, uint64_t mask32 /* #32B groups - 1 */,
                                             uint64_t n, uint64_t *__restrict__ out) {
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (uint64_t i = tid; i < n; i += stride * UNROLL) {
        ulonglong2 v[UNROLL][2];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            uint64_t item = i + (uint64_t)u * stride;
            uint64_t key = item;
            int sub = 0;
            if (MODE == 3) { key = item >> 1; sub = item & 1; }
            if (MODE == 6) { key = item >> 2; sub = item & 3; }
            uint64_t g = mix(key) & mask32;
            const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(table + g * 4);
            v[u][1] = make_ulonglong2(0, 0);
            if (MODE == 0) { v[u][0] = p[0]; v[u][1] = p[1]; }
            else if (MODE == 1) { v[u][0] = p[0]; }
            else if (MODE == 2) { v[u][0].x = table[g * 4]; v[u][0].y = 0; }
            else if (MODE == 3) { v[u][0] = p[sub]; }
            else if (MODE == 4) { v[u][0] = nt_load(&p[0]); v[u][1] = nt_load(&p[1]); }
            else if (MODE == 5) { v[u][0] = nt_load(&p[0]); }
            else if (MODE == 6) { const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(table + (g & ~1ull) * 4); v[u][0] = q[sub]; }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u][0].x ^ v[u][0].y ^ v[u][1].x ^ v[u][1].y;
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int MODE, int UNROLL>
void run(const char *name, const uint64_t *table, uint64_t mask32, uint64_t n_probes, uint64_t *out, int blocks) {
    uint64_t items = n_probes * (MODE == 3 ? 2 : MODE == 6 ? 4 : 1);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<MODE, UNROLL>), dim3(blocks), dim3(256), 0, 0, table, mask32, items, out);
    CK(hipEventRecord(a));
    const int R = 5;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((probe<MODE, UNROLL>), dim3(blocks), dim3(256), 0, 0, table, mask32, items, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= R;
    printf("%-44s unroll=%d blocks=%5d  %.3f ms  %.1f G probes/s\n", name, UNROLL, blocks, ms, n_probes / (ms * 1e-3) / 1e9);
}

int main(int argc, char **argv) {
    int lg = argc > 1 ? atoi(argv[1]) : 30;               // slots (8 B each)
    uint64_t n = argc > 2 ? strtoull(argv[2], 0, 10) : 64000000ull;
    uint64_t slots = 1ull << lg;
    uint64_t *table, *out;
    CK(hipMalloc(&table, slots * 8)); CK(hipMemset(table, 0, slots * 8)); CK(hipMalloc(&out, 64));
    uint64_t mask32 = slots / 4 - 1;
    printf("table: 2^%d slots = %.2f GB, %llu probes\n", lg, slots * 8 / 1e9, (unsigned long long)n);
    for (int blocks : {2048, 8192}) {
        run<0, 1>("2 x 16 B per probe (32 B group)", table, mask32, n, out, blocks);
        run<0, 4>("2 x 16 B per probe (32 B group)", table, mask32, n, out, blocks);
        run<1, 1>("1 x 16 B per probe", table, mask32, n, out, blocks);
        run<1, 4>("1 x 16 B per probe", table, mask32, n, out, blocks);
        run<1, 8>("1 x 16 B per probe", table, mask32, n, out, blocks);
        run<2, 4>("1 x 8 B per probe", table, mask32, n, out, blocks);
        run<3, 4>("lane pair: 2 lanes x 16 B = 32 B per probe", table, mask32, n, out, blocks);
        run<6, 4>("lane quad: 4 lanes x 16 B = 64 B per probe", table, mask32, n, out, blocks);
        run<4, 4>("2 x 16 B nontemporal", table, mask32, n, out, blocks);
        run<5, 4>("1 x 16 B nontemporal", table, mask32, n, out, blocks);
    }
    return 0;
}
