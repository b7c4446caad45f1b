// dcn_internal.h -- shared declarations of the HIP filter pipeline (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/deacon_hip.h"

// ----------------------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------------------
void dcn_set_error(const std::string &msg);
int dcn_fail(int code, const std::string &msg);

#define DCN_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return dcn_fail(_e == hipErrorOutOfMemory ? DCN_ERR_NOMEM : DCN_ERR_HIP,               \
                            std::string(#expr) + ": " + hipGetErrorString(_e));                    \
    } while (0)

// memcpy on the host pool's threads (api.hip): large copies into memory nobody has touched yet are first-touch bound on one thread
void dcn_host_parallel_copy(void *dst, const void *src, size_t n);

// ----------------------------------------------------------------------------------------------------
// device-resident index: open-addressing set over groups of DCN_GROUP_SLOTS u64 slots, linear probing group by
// group.  Slot value 0 = empty; key 0 is tracked by `has_zero`.
//   DCN_GROUP_SLOTS 2: 16-byte groups, one dwordx4 request per probe, >= 4 slots per key (load <= 0.25)
//   DCN_GROUP_SLOTS 4: 32-byte groups, two dwordx4 requests per probe, >= 2 slots per key (load <= 0.5)
// A random probe costs one 64-byte HBM sector either way; what differs is the number of requests the memory
// pipeline handles per probe (measured: 41 G probes/s with two requests, 47-49 G/s with one).
// ----------------------------------------------------------------------------------------------------
#ifndef DCN_GROUP_SLOTS
#define DCN_GROUP_SLOTS 2
#endif
// table size = the smallest power-of-two number of groups with >= S slots per key.  S is at least
// DCN_SLOTS_PER_KEY; dcn_table_groups_for (index_table.hip) doubles it while the table stays small against the
// device's 288 GB (every probe that finds its home group full is one more scattered request, and the scan
// kernel's time follows the request count: 1.39 -> 1.35 ms on the headline batch for 17 -> 34 GB of table).
constexpr int DCN_SLOTS_PER_KEY = DCN_GROUP_SLOTS == 2 ? 4 : 2;
constexpr int DCN_SLOTS_PER_KEY_ROOMY = 2 * DCN_SLOTS_PER_KEY;
constexpr uint64_t DCN_ROOMY_MAX_GROUPS = 1ull << 31; // 34 GB of 16-byte groups

struct dcn_table_view {
    const uint64_t *slots; // n_groups * DCN_GROUP_SLOTS
    uint32_t group_shift;  // 32 - log2(n_groups)
    uint32_t group_mask;   // n_groups - 1
    uint32_t has_zero;
};

// the minimizer rule in force (dcn_set_minimizer_variant): rotation << 16 | compared bits << 8 | combine
constexpr uint32_t DCN_VARIANT_DEFAULT = (1u << 16) | (16u << 8) | 0u;
uint32_t dcn_current_variant();

struct dcn_index {
    // captured when the index is created (built, loaded, merged, cloned): the rule its keys were selected by travels
    // with it, and every context filters by its index's rule whatever the process-wide setting has become since
    uint32_t variant = dcn_current_variant();
    int device = 0;
    uint8_t k = 0, w = 0;
    uint64_t n_keys = 0; // distinct
    uint64_t n_groups = 0;
    uint64_t *d_slots = nullptr;
    bool has_zero = false;
    dcn_table_view view() const {
        dcn_table_view v;
        v.slots = d_slots;
        uint32_t bits = 0;
        while ((1ull << bits) < n_groups) ++bits;
        v.group_shift = 32 - bits;
        v.group_mask = (uint32_t)(n_groups - 1);
        v.has_zero = has_zero ? 1u : 0u;
        return v;
    }
};

// group index of a key: all 64 bits feed a 32-bit multiplicative hash, top bits select the group
__host__ __device__ inline uint32_t dcn_group_of(uint64_t key, uint32_t group_shift, uint32_t group_mask) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t x = (lo ^ ((hi << 15) | (hi >> 17))) * 0x9E3779B1u;
    return group_shift >= 32 ? 0u : ((x >> group_shift) & group_mask);
}

// ----------------------------------------------------------------------------------------------------
// pipeline geometry
// ----------------------------------------------------------------------------------------------------
constexpr int DCN_WAVE = 64;           // one wave per workgroup in the scan kernel: 64 tiles
constexpr int DCN_LCAP = 40;           // per-lane emitted-position list capacity between flushes
constexpr uint32_t DCN_FRONT_PAD = 64; // u32 words of zero padding in front of the packed stream
constexpr uint32_t DCN_TAIL_PAD = 256; // u32 words after it (lanes over-read past short tiles)
// A lane scans as many steps as the longest tile of its wave, plus k-1 bases and one prefetched block: a short last
// tile next to a full one over-reads by up to tile_windows + l + 32 bases.  The 256-word tail pad (4096 bases of the
// packed stream, 8192 of the mask) covers that up to 2048 windows per tile.
constexpr uint32_t DCN_MAX_TILE_WINDOWS = 2048;

// one tile = up to `tile_windows` consecutive windows of one read, scanned by one lane
// (16 bytes: one dwordx4 per lane of the scan kernel, whose time is its L2 misses -- DESIGN.md section 6.2.  The
// position of scan_start inside its read, which only the minimizer dump reports, lives in a side array.)
struct dcn_tile {
    uint64_t scan_start; // absolute base index (in the batch stream) of the first base to scan
    uint32_t unit;       // global unit id
    uint32_t nwf;        // bits 0..29: windows whose minimizers this tile emits; bit 31: has a carry window (the first
                         // scanned window only seeds the dedup state); bit 30: this tile is its unit's only tile and
                         // unit == read -- such a unit has no entry in unit_tile_first / unit_tile_count unless the
                         // scan kernel hands it to the distinct pass (it then writes the entry itself)
    __host__ __device__ uint32_t n_windows() const { return nwf & 0x3FFFFFFFu; }
    __host__ __device__ uint32_t carry() const { return nwf >> 31; }
    __host__ __device__ bool whole_unit() const { return (nwf >> 30) & 1u; }
};
static_assert(sizeof(dcn_tile) == 16, "tile descriptor");

// status words written by the device pipeline (one per ctx, zeroed before every enqueued batch / chunk)
struct dcn_status {
    uint32_t rec_overflow;         // the global hash sets of the distinct pass did not fit their scratch
    uint32_t n_tiles;
    uint32_t any_records;          // the scan wrote a hit into some tile's run: the distinct pass has work
    uint32_t n_big;                // work items of units with more hits than the LDS set holds (one per 64 tiles)
    uint32_t any_scattered;        // ... one of them with tiles that are not contiguous (found by a sweep over all tiles)
    uint32_t n_pending;            // units enrolled for the distinct pass (scan.hip)
    uint32_t any_newline;          // the pack kernel saw a '\n' byte: only then does planning probe read ends
    uint32_t bounds;               // DCN_DEBUG_BOUNDS builds: phase B met an index outside its list / its stream
    uint32_t run_overflow;         // a unit had more hits in one wave than its run of the record array holds (rec_shift > 0)
    uint32_t bad_offsets;          // the plan kernel met offsets[r] > offsets[r+1] or offsets[r+1] > n_bases: those reads were planned as empty
    unsigned long long set_cursor; // distinct pass: slots handed out to the global per-unit hash sets
};

// What survives the per-chunk clearing of dcn_status: one per host batch in flight, one per context for the
// device-pointer API.  Read back when the batch is waited for / the context is synchronised.
struct dcn_batch_report {
    uint32_t overflow;                     // a chunk dropped hit records: its multi-wave units were decided from
                                           // truncated records and its counters were skipped.  Bit 0: the global sets'
                                           // scratch was too small; bit 1: a run of the record array was (rec_shift > 0)
    uint32_t bounds;                       // DCN_DEBUG_BOUNDS builds only: the scan kernel refused an out-of-range index
    uint32_t bad_offsets;                  // the offsets array was not non-decreasing within [0, n_bases] when the plan kernel read it
    uint32_t reserved_;
    unsigned long long need;               // record capacity (set slots / 4) that would have sufficed
    unsigned long long stats[DCN_N_STATS]; // the six ProcessingStats counters
};

struct dcn_scan_args {
    const uint32_t *packed; // 2-bit stream, already offset by DCN_FRONT_PAD words
    const uint32_t *invmask; // 1 bit per base, same padding (in 32-bit words)
    const dcn_tile *tiles;
    const uint32_t *tile_read_pos; // dump mode with read-relative positions: position of scan_start in its read
    const uint32_t *n_tiles; // device-side tile count
    uint32_t *unit_tile_first; // n_units: first tile index of each unit (no entry for whole-unit tiles, see dcn_tile)
    uint32_t *unit_tile_count; // n_units: number of tiles, 0xFFFFFFFF when they are not contiguous
    dcn_table_view table;
    uint32_t k, w;
    uint64_t stream_bases; // bases of the packed stream (checked by DCN_DEBUG_BOUNDS builds only)
    // thresholds
    uint64_t abs_threshold;
    double rel_threshold;
    uint32_t deplete;
    // decisions only (caller takes neither hit counts nor totals): a tile that is a whole unit with at most this many
    // list entries has required hits == abs_threshold whatever its valid-minimizer total is, so its lane may stop
    // probing at abs_threshold distinct hits.  0 = always count everything.
    uint32_t early_out_max_items;
    uint32_t early_out_pairs; // 1: two-tile units in adjacent lanes may take that path too
    // outputs for units resolved inside one wave
    uint8_t *keep;
    uint32_t *hits, *total;
    uint8_t *unit_state; // 1 = resolved by the scan kernel
    // outputs for units spanning several waves (or too large for the in-wave hit ring)
    uint32_t *g_total;   // per unit, atomically accumulated
    uint32_t *g_hitcnt;  // per unit: hits written to its runs (one atomicAdd per wave holding tiles of the unit)
    uint32_t *g_zero;    // per unit: the zero hash was a hit
    uint64_t *rec_hash;  // one slot per 2^rec_shift bases of the batch stream; a run starts at (scan_start + carry of its
                         // first tile) >> rec_shift and holds as many hits as its unit's windows in this wave >> rec_shift
    uint32_t rec_shift;
    uint32_t tile_windows; // windows of a full tile (a shorter tile is its read's last)
    uint32_t *tile_hits; // per tile: length of the run that starts at this tile (0: none)
    uint32_t *pending;   // units the scan did not finish, in no particular order (status->n_pending of them)
    dcn_status *status;
    // dump mode (dcn_minimizer_hashes_batch): per emitted minimizer
    uint64_t *dump_hash;
    uint32_t *dump_pos;
    uint8_t *dump_valid;
    uint32_t *dump_count; // per tile
    uint32_t dump_abs;    // 1: dump_pos holds the low 32 bits of the absolute base index instead of the read position
    // parity-pinning variant (scan_kernel<..., VAR>; filled in by dcn_launch_scan from dcn_set_minimizer_variant)
    uint32_t variant;        // the index's dcn_index::variant (0: the default rules)
    uint32_t nt_rot;         // ntHash rotation per base (1)
    uint32_t cmp_mask;       // hash bits that are compared (0xFFFF0000)
    uint32_t nt_combine_xor; // 0: fw + rc, 1: fw ^ rc
};

// ---- kernels launched by api.hip -------------------------------------------------------------------
// packs bases [base_begin, base_end) of the stream (whole 32-base groups; bytes at or past base_end read as 'A');
// status (may be null) receives any_newline
int dcn_launch_pack(const uint8_t *d_ascii, uint64_t base_begin, uint64_t base_end, uint32_t *d_packed,
                    uint32_t *d_invmask, dcn_status *status, hipStream_t stream, bool index_side = false,
                    uint32_t block_threads = 256);
// ... as a kernel of <= 32 VGPRs in one-wave workgroups, to run beside another kernel's waves (pack.hip)
int dcn_launch_pack_beside(const uint8_t *d_ascii, uint64_t base_begin, uint64_t base_end, uint32_t *d_packed, uint32_t *d_invmask,
                           dcn_status *status, hipStream_t stream);
int dcn_launch_scan(const dcn_scan_args &args, uint32_t max_tiles, bool dump, hipStream_t stream);

int dcn_table_build(dcn_index *idx, const uint64_t *host_keys, uint64_t n);
int dcn_table_contains(const dcn_index *idx, const uint64_t *host_keys, uint64_t n, uint8_t *out);
int dcn_table_contains_device(const dcn_index *idx, const uint64_t *d_keys, uint64_t n, uint8_t *d_out,
                              hipStream_t stream);
int dcn_table_probe_ceiling(const dcn_index *idx, const uint64_t *d_keys, uint64_t n, uint32_t reps, double *best,
                            hipStream_t stream);
int dcn_table_insert_varint9(dcn_index *idx, const uint64_t *d_raw, uint64_t n, unsigned long long *d_new,
                             uint32_t *d_zero, uint32_t *d_bad, hipStream_t stream); // 9-byte varint records
uint64_t dcn_table_groups_for(uint64_t n_keys);                                  // sizing rule (see DCN_SLOTS_PER_KEY)
int dcn_table_alloc(dcn_index *idx, uint64_t n_keys_capacity);                 // empty table for >= that many keys
int dcn_table_reserve(dcn_index *idx, uint64_t n_keys_capacity);               // grow + rehash if needed
int dcn_table_insert_dump(dcn_index *idx, const uint64_t *d_hash, const uint8_t *d_valid, const uint32_t *d_abs_pos,
                          uint64_t n_slots, const uint8_t *d_ascii, float entropy_threshold, hipStream_t stream);
int dcn_table_count_valid(const uint8_t *d_valid, uint64_t n, uint64_t *count, hipStream_t stream);
int dcn_table_export(const dcn_index *idx, uint64_t *host_out, uint64_t capacity, uint64_t *n_out);
int dcn_table_merge(dcn_index *dst, const dcn_index *src, const dcn_index *minus);
hipError_t dcn_table_malloc(uint64_t **p, uint64_t bytes);                            // the table's allocation (DCN_TABLE_CONTIGUOUS=1: physically contiguous)
int dcn_table_clone_by_keys(const dcn_index *src, dcn_index *dst);              // replica on dst->device from the compacted keys


// required = max(abs, total==0 ? 0 : max(1, round_half_away(rel*total)))  (src/filter_common.rs:84-96)
__host__ __device__ inline uint64_t dcn_required_hits(uint64_t abs_threshold, double rel_threshold,
                                                      uint64_t total) {
    uint64_t rel_required = 0;
    if (total != 0) {
        double r = round(rel_threshold * (double)total); // round(): half away from zero, as f64::round
        if (!(r > 0.0)) rel_required = 0;                // Rust `as usize`: negative / NaN -> 0
        else if (r >= 18446744073709551616.0) rel_required = ~0ull;
        else rel_required = (uint64_t)r;
        if (rel_required < 1) rel_required = 1;
    }
    return abs_threshold > rel_required ? abs_threshold : rel_required;
}

__host__ __device__ inline bool dcn_decide(uint64_t hits, uint64_t total, uint64_t abs_threshold,
                                           double rel_threshold, uint32_t deplete) {
    uint64_t required = dcn_required_hits(abs_threshold, rel_threshold, total);
    return deplete ? (hits < required) : (hits >= required);
}
