// pack.hip -- K1: ASCII -> 2-bit stream + invalid-base bitmask, for the whole batch buffer at once.
//
// Replaces packed_seq::PackedSeqVec::from_ascii and the invalid-mask loop of
// get_minimizer_hashes_and_positions (src/filter_common.rs:238-258):
//   code = (c >> 1) & 3 for EVERY byte (A=0 C=1 T=2 G=3; non-ACGT mapped the same lossy way);
//   mask bit = 1 iff the byte is not one of ACGTacgt.
// Reads are concatenated with no separators, so the batch is packed as one dense stream: base i of the
// batch is bits [2(i%16), 2(i%16)+2) of packed[i/16] and bit i%32 of invmask[i/32]; no kernel here
// needs to know where reads begin.  HBM-bound: 1 B/bp read, 0.375 B/bp written.
#include "dcn_internal.h"

#include <algorithm>

namespace {

__device__ inline uint32_t pack4(uint32_t x) {
    // 4 ASCII bytes -> 8 bits of 2-bit codes (byte 0 in bits 0..1)
    uint32_t y = (x >> 1) & 0x03030303u;
    return (y * ((1u << 24) | (1u << 18) | (1u << 12) | (1u << 6))) >> 24;
}

// index-side code (src/minimizers.rs:24-43 canonicalise_nucleotide, then the same (c >> 1) & 3):
// A,W -> A=0; T -> T=2; G,R,S,K,D,V -> G=3; everything else (C,Y,M,B,H,N,...) -> C=1; case-insensitive
__device__ inline uint32_t pack4_index_side(uint32_t x) {
    uint32_t out = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // (byte | 0x20) == 'a' holds exactly for 'A' and 'a' (same for every letter)
        uint32_t c = ((x >> (8 * j)) & 0xFFu) | 0x20u;
        bool is_a = (c == 'a') | (c == 'w');
        bool is_t = (c == 't');
        bool is_g = (c == 'g') | (c == 'r') | (c == 's') | (c == 'k') | (c == 'd') | (c == 'v');
        uint32_t code = is_a ? 0u : is_t ? 2u : is_g ? 3u : 1u;
        out |= code << (2 * j);
    }
    return out;
}

__device__ inline uint32_t invalid4(uint32_t x) {
    // 4 ASCII bytes -> 4 bits, bit j = byte j is not in ACGTacgt.  All four bytes at once: with y = byte | 0x20 and
    // T = y.bit2 & ~y.bit1 (set for 't' only among a c g t), a valid byte is 0 1 1 T 0 y2 y1 ~T -- bits 1 and 2 are
    // free (they are the 2-bit code), every other bit is fixed by them.  17 instructions instead of 4 x 10: this
    // kernel was bound by its compares (12 VALU instructions per base), not by HBM.
    const uint32_t y = x | 0x20202020u;
    const uint32_t T = (y >> 2) & ~(y >> 1) & 0x01010101u;
    const uint32_t expect = 0x60606060u | (T << 4) | (T ^ 0x01010101u);
    const uint32_t diff = (y & 0xF9F9F9F9u) ^ expect;                       // non-zero byte <=> invalid byte
    const uint32_t nz = ((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff;        // bit 7 of each byte: byte != 0 (no carries across bytes)
    return (((nz >> 7) & 0x01010101u) * 0x01020408u) >> 24;                 // bits 0,8,16,24 -> bits 0..3
}

// each thread packs 32 bases: two packed words and one mask word.  Groups [g_first, g_first + n_chunks) of the
// stream are packed (a chunk of a larger batch starts at any group); bytes at or past n_bases read as 'A'.
template <bool ALIGNED, bool INDEX_SIDE>
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ ascii, uint64_t n_bases,
                                                   uint32_t *__restrict__ packed,
                                                   uint32_t *__restrict__ invmask, uint64_t g_first, uint64_t n_chunks,
                                                   dcn_status *status) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t nl = 0;
    for (uint64_t t = g_first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < g_first + n_chunks; t += stride) {
        uint64_t base = t * 32;
        uint32_t w[8];
        if (ALIGNED && base + 32 <= n_bases) {
            const uint4 *p = reinterpret_cast<const uint4 *>(ascii + base);
            uint4 a = p[0], b = p[1];
            w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
            w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                uint32_t v = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint64_t i = base + 4 * q + j;
                    // bases past the end are packed as 'A' and flagged valid; no window ever reaches them
                    uint32_t c = i < n_bases ? ascii[i] : (uint32_t)'A';
                    v |= c << (8 * j);
                }
                w[q] = v;
            }
        }
        uint32_t p0 = 0, p1 = 0, m = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p0 |= (INDEX_SIDE ? pack4_index_side(w[q]) : pack4(w[q])) << (8 * q);
            p1 |= (INDEX_SIDE ? pack4_index_side(w[q + 4]) : pack4(w[q + 4])) << (8 * q);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) m |= invalid4(w[q]) << (4 * q);
        if (!INDEX_SIDE && m) {
            // any '\n' byte in the batch?  It is an invalid byte, so only groups with one are looked at:
            // (x ^ 0x0A..) has a zero byte <=> some byte of x is '\n'
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                uint32_t z = w[q] ^ 0x0A0A0A0Au;
                nl |= (z - 0x01010101u) & ~z & 0x80808080u;
            }
        }
        packed[2 * t] = p0;
        packed[2 * t + 1] = p1;
        invmask[t] = m;
    }
    if (!INDEX_SIDE && nl && status) status->any_newline = 1; // rare; plan_kernel then probes the read ends
}


// The same for a batch that is packed BESIDE another kernel's waves (api.hip packs batch i+1 while the scan kernel of batch i
// runs): the scan kernel holds 4 waves x 120 VGPRs of a SIMD's 512, so only a wave of <= 32 VGPRs fits next to them.  Each
// thread packs 16 bases (one uint4 in, one packed word and half a mask word out), one-wave workgroups, whole 16-base pieces
// of a 16-byte-aligned stream only (dcn_launch_pack_beside leaves the rest to the general kernel).
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) void pack16_kernel(
    const uint8_t *__restrict__ ascii, uint32_t *__restrict__ packed, uint32_t *__restrict__ invmask, uint32_t h_first,
    uint32_t h_end, dcn_status *status) {
    const uint32_t stride = gridDim.x * 64u;
    uint32_t nl = 0;
    uint16_t *mask16 = reinterpret_cast<uint16_t *>(invmask);
    for (uint32_t t = h_first + blockIdx.x * 64u + threadIdx.x; t < h_end; t += stride) { // whole 16-base pieces only
        const uint4 a = *reinterpret_cast<const uint4 *>(ascii + (uint64_t)t * 16);
        const uint32_t p = pack4(a.x) | (pack4(a.y) << 8) | (pack4(a.z) << 16) | (pack4(a.w) << 24);
        const uint32_t m = invalid4(a.x) | (invalid4(a.y) << 4) | (invalid4(a.z) << 8) | (invalid4(a.w) << 12);
        if (m) {
            uint32_t z = a.x ^ 0x0A0A0A0Au;
            nl |= (z - 0x01010101u) & ~z & 0x80808080u;
            z = a.y ^ 0x0A0A0A0Au;
            nl |= (z - 0x01010101u) & ~z & 0x80808080u;
            z = a.z ^ 0x0A0A0A0Au;
            nl |= (z - 0x01010101u) & ~z & 0x80808080u;
            z = a.w ^ 0x0A0A0A0Au;
            nl |= (z - 0x01010101u) & ~z & 0x80808080u;
        }
        packed[t] = p;
        mask16[t] = (uint16_t)m;
    }
    if (nl && status) status->any_newline = 1;
}

} // namespace

int dcn_launch_pack_beside(const uint8_t *d_ascii, uint64_t base_begin, uint64_t base_end, uint32_t *d_packed, uint32_t *d_invmask,
                           dcn_status *status, hipStream_t stream) {
    if (base_end <= base_begin) return DCN_OK;
    // whole 32-base groups by the small kernel, the (at most one) group that the stream's end cuts by the general one
    const uint64_t whole_end = base_end / 32 * 32;
    if ((reinterpret_cast<uintptr_t>(d_ascii) & 15) != 0 || (base_begin & 31) != 0 || whole_end / 16 > 0xFFFFFFFFull || whole_end <= base_begin)
        return dcn_launch_pack(d_ascii, base_begin, base_end, d_packed, d_invmask, status, stream, false, 64);
    const uint32_t h_first = (uint32_t)(base_begin / 16), h_end = (uint32_t)(whole_end / 16);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)(h_end - h_first) + 63) / 64, 256ull * 4 * 16);
    hipLaunchKernelGGL(pack16_kernel, dim3(blocks), dim3(64), 0, stream, d_ascii, d_packed, d_invmask, h_first, h_end, status);
    DCN_HIP(hipGetLastError());
    if (whole_end < base_end) return dcn_launch_pack(d_ascii, whole_end, base_end, d_packed, d_invmask, status, stream, false, 64);
    return DCN_OK;
}

int dcn_launch_pack(const uint8_t *d_ascii, uint64_t base_begin, uint64_t base_end, uint32_t *d_packed,
                    uint32_t *d_invmask, dcn_status *status, hipStream_t stream, bool index_side, uint32_t block_threads) {
    if (base_end <= base_begin) return DCN_OK;
    const uint64_t g_first = base_begin / 32, n_bases = base_end;
    uint64_t n_chunks = (base_end + 31) / 32 - g_first;
    // block_threads = 64: one-wave workgroups, which fit into any single wave slot another kernel's waves leave behind
    // (api.hip packs the next batch beside the running scan kernel); 256 otherwise
    const uint32_t bt = block_threads == 64 ? 64u : 256u;
    uint32_t blocks = (uint32_t)((n_chunks + bt - 1) / bt);
    if (blocks > 256 * 32 * (256 / bt)) blocks = 256 * 32 * (256 / bt);
    bool aligned = (reinterpret_cast<uintptr_t>(d_ascii) & 15) == 0;
#define DCN_PACK(AL, IX) \
    hipLaunchKernelGGL((pack_kernel<AL, IX>), dim3(blocks), dim3(bt), 0, stream, d_ascii, n_bases, d_packed, d_invmask, g_first, n_chunks, status)
    if (index_side) {
        if (aligned) DCN_PACK(true, true);
        else DCN_PACK(false, true);
    } else {
        if (aligned) DCN_PACK(true, false);
        else DCN_PACK(false, false);
    }
#undef DCN_PACK
    DCN_HIP(hipGetLastError());
    return DCN_OK;
}
