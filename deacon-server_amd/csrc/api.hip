// api.hip -- C ABI of include/deacon_hip.h: context / buffer management, host staging, and the order in
// which the kernels of the batch pipeline are enqueued.
//
// One batch on the context's compute stream:
//   memset(per-unit scratch) -> pack (K1) -> plan (one launch) -> scan (K2-K5, fused)
//   -> distinct pass for multi-wave units -> finish (decision + six counters, K6)
// Host batches are staged through two pinned buffers and copied with hipMemcpyAsync on a side stream while
// the host fills the other buffer; the compute stream waits on the copy's event.
#include "dcn_internal.h"
#include "dcn_plan.h"
#include "dcn_host_pool.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define DCN_VERSION_STRING "deacon-hip 0.4.0 (gfx950)"

// ----------------------------------------------------------------------------------------------------
// errors
// ----------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void dcn_set_error(const std::string &msg) { g_last_error = msg; }
int dcn_fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

extern "C" const char *dcn_last_error(void) { return g_last_error.c_str(); }
extern "C" const char *dcn_version(void) { return DCN_VERSION_STRING; }

extern "C" int dcn_abi_version(uint32_t *major, uint32_t *minor) {
    if (!major || !minor) return dcn_fail(DCN_ERR_ARG, "dcn_abi_version: NULL output");
    *major = DCN_ABI_MAJOR;
    *minor = DCN_ABI_MINOR;
    return DCN_OK;
}

extern "C" int dcn_device_count(int *count) {
    if (!count) return dcn_fail(DCN_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return dcn_fail(DCN_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return DCN_OK;
}

// ----------------------------------------------------------------------------------------------------
// index
// ----------------------------------------------------------------------------------------------------
int dcn_read_index_file(const char *path, uint8_t *k, uint8_t *w, std::vector<uint64_t> *keys); // index_file.cpp

#define DCN_TRY_EARLY(expr)             \
    do {                               \
        int _rc = (expr);              \
        if (_rc != DCN_OK) return _rc; \
    } while (0)

static int check_kw(uint8_t k, uint8_t w) {
    if (k < 1 || k > 56) return dcn_fail(DCN_ERR_ARG, "k must be in 1..=56 (src/filter_common.rs:269-272)");
    if (w < 1) return dcn_fail(DCN_ERR_ARG, "w must be >= 1");
    if (((uint32_t)k + w - 1) % 2 == 0)
        return dcn_fail(DCN_ERR_ARG, "Constraint violated: k + w - 1 must be odd (src/index.rs:186-194)");
    // every index passes through here when it is made, and captures the process's rule then: a rule other than the default
    // runs the generic kernel, whose window ring holds two u64 keys per slot in LDS (scan.hip) -- refused now, not at the
    // first filter call
    if (dcn_current_variant() != DCN_VARIANT_DEFAULT && w > 128)
        return dcn_fail(DCN_ERR_ARG, "minimizer variant: w <= 128 under a non-default rule (dcn_set_minimizer_variant)");
    return DCN_OK;
}

extern "C" int dcn_index_from_keys(const uint64_t *keys, uint64_t n, uint8_t k, uint8_t w, int device,
                                   dcn_index **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n > 0 && !keys) return dcn_fail(DCN_ERR_ARG, "keys is NULL");
    int rc = check_kw(k, w);
    if (rc != DCN_OK) return rc;
    int ndev = 0;
    rc = dcn_device_count(&ndev);
    if (rc != DCN_OK) return rc;
    if (device < 0 || device >= ndev) return dcn_fail(DCN_ERR_ARG, "no such HIP device");
    dcn_index *idx = new (std::nothrow) dcn_index();
    if (!idx) return dcn_fail(DCN_ERR_NOMEM, "host allocation failed");
    idx->device = device;
    idx->k = k;
    idx->w = w;
    rc = dcn_table_build(idx, keys, n);
    if (rc != DCN_OK) {
        if (idx->d_slots) hipFree(idx->d_slots);
        delete idx;
        return rc;
    }
    *out = idx;
    return DCN_OK;
}

int dcn_load_index_fixed9(const char *path, int device, dcn_index **out, bool *handled); // below (needs the copy pool)

extern "C" int dcn_index_from_file(const char *path, int device, dcn_index **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!path) return dcn_fail(DCN_ERR_ARG, "path is NULL");
    // The reference's file format has no room for the rule its keys were selected by (src/index.rs:17-31: version, k, w):
    // a file is taken to have been built under the rule in force now.  Under the default that is what every existing
    // file was built by; under any other setting (the parity-pinning switch) say so once, since a mismatch would probe
    // with the wrong minimizers and raise no error.
    if (dcn_current_variant() != DCN_VARIANT_DEFAULT && !getenv("DCN_QUIET")) {
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            std::fprintf(stderr, "deacon-hip: loading an index file under a non-default minimizer rule (dcn_set_minimizer_variant): "
                                 "the file carries no marker of the rule it was built by and is assumed to match\n");
    }
    // files whose hashes are all 9-byte varints (every hash >= 2^32: all of them, in practice) are decoded on
    // the device while they stream in; anything else takes the host decoder below
    bool handled = false;
    int rc = dcn_load_index_fixed9(path, device, out, &handled);
    if (rc != DCN_OK || handled) return rc;
    uint8_t k = 0, w = 0;
    std::vector<uint64_t> keys;
    rc = dcn_read_index_file(path, &k, &w, &keys);
    if (rc != DCN_OK) return rc;
    return dcn_index_from_keys(keys.data(), keys.size(), k, w, device, out);
}

int dcn_write_index_file(const char *path, uint8_t k, uint8_t w, const uint64_t *keys, uint64_t n); // index_file.cpp
int dcn_build_index_impl(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seqs, float entropy_threshold,
                         uint64_t capacity_keys, dcn_index *idx); // below, needs dcn_ctx

extern "C" int dcn_index_build(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seqs, uint8_t k, uint8_t w,
                               float entropy_threshold, uint64_t capacity_keys, int device, dcn_index **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    int rc = check_kw(k, w);
    if (rc != DCN_OK) return rc;
    if (n_seqs > 0 && !offsets) return dcn_fail(DCN_ERR_ARG, "offsets is NULL");
    if (n_seqs > 0 && offsets[n_seqs] > 0 && !bases) return dcn_fail(DCN_ERR_ARG, "bases is NULL");
    if (!(entropy_threshold >= 0.0f && entropy_threshold <= 1.0f)) return dcn_fail(DCN_ERR_ARG, "entropy_threshold must be in [0, 1]");
    int ndev = 0;
    rc = dcn_device_count(&ndev);
    if (rc != DCN_OK) return rc;
    if (device < 0 || device >= ndev) return dcn_fail(DCN_ERR_ARG, "no such HIP device");
    dcn_index *idx = new (std::nothrow) dcn_index();
    if (!idx) return dcn_fail(DCN_ERR_NOMEM, "host allocation failed");
    idx->device = device;
    idx->k = k;
    idx->w = w;
    rc = dcn_table_alloc(idx, std::max<uint64_t>(capacity_keys, 1024));
    if (rc == DCN_OK) rc = dcn_build_index_impl(bases, offsets, n_seqs, entropy_threshold, capacity_keys, idx);
    if (rc != DCN_OK) {
        if (idx->d_slots) hipFree(idx->d_slots);
        delete idx;
        return rc;
    }
    *out = idx;
    return DCN_OK;
}

static int same_params(const dcn_index *a, const dcn_index *b) {
    if (a->k != b->k || a->w != b->w)
        return dcn_fail(DCN_ERR_ARG, "Incompatible headers: k=" + std::to_string((int)b->k) + ", w=" + std::to_string((int)b->w) +
                                         " vs k=" + std::to_string((int)a->k) + ", w=" + std::to_string((int)a->w));
    if (a->device != b->device) return dcn_fail(DCN_ERR_ARG, "indexes live on different devices");
    if (a->variant != b->variant)
        return dcn_fail(DCN_ERR_ARG, "indexes were created under different minimizer rules (dcn_set_minimizer_variant)");
    return DCN_OK;
}

extern "C" int dcn_index_union(const dcn_index *const *inputs, uint32_t n, dcn_index **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!inputs || n == 0 || !inputs[0]) return dcn_fail(DCN_ERR_ARG, "at least one input index is required");
    uint64_t sum = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (!inputs[i]) return dcn_fail(DCN_ERR_ARG, "input index is NULL");
        int rc = same_params(inputs[0], inputs[i]);
        if (rc != DCN_OK) return rc;
        sum += inputs[i]->n_keys;  // worst-case capacity, as the reference pre-allocates (src/index.rs:579-594)
    }
    dcn_index *idx = new (std::nothrow) dcn_index();
    if (!idx) return dcn_fail(DCN_ERR_NOMEM, "host allocation failed");
    idx->device = inputs[0]->device;
    idx->variant = inputs[0]->variant;
    idx->k = inputs[0]->k;
    idx->w = inputs[0]->w;
    int rc = dcn_table_alloc(idx, std::max<uint64_t>(sum, 16));
    for (uint32_t i = 0; i < n && rc == DCN_OK; ++i) rc = dcn_table_merge(idx, inputs[i], nullptr);
    if (rc != DCN_OK) {
        if (idx->d_slots) hipFree(idx->d_slots);
        delete idx;
        return rc;
    }
    *out = idx;
    return DCN_OK;
}

extern "C" int dcn_index_diff(const dcn_index *first, const dcn_index *second, dcn_index **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!first || !second) return dcn_fail(DCN_ERR_ARG, "index is NULL");
    int rc = same_params(first, second);
    if (rc != DCN_OK) return rc;
    dcn_index *idx = new (std::nothrow) dcn_index();
    if (!idx) return dcn_fail(DCN_ERR_NOMEM, "host allocation failed");
    idx->device = first->device;
    idx->variant = first->variant;
    idx->k = first->k;
    idx->w = first->w;
    rc = dcn_table_alloc(idx, std::max<uint64_t>(first->n_keys, 16));
    if (rc == DCN_OK) rc = dcn_table_merge(idx, first, second);
    if (rc != DCN_OK) {
        if (idx->d_slots) hipFree(idx->d_slots);
        delete idx;
        return rc;
    }
    *out = idx;
    return DCN_OK;
}

extern "C" int dcn_index_keys(const dcn_index *index, uint64_t *out, uint64_t capacity, uint64_t *n) {
    if (!index || !n) return dcn_fail(DCN_ERR_ARG, "index/n is NULL");
    if (capacity > 0 && !out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    return dcn_table_export(index, out, capacity, n);
}

extern "C" int dcn_index_write_file(const dcn_index *index, const char *path) {
    if (!index || !path) return dcn_fail(DCN_ERR_ARG, "index/path is NULL");
    // (not a std::vector: its resize() writes 8 bytes of zero per key before the keys are copied over them)
    std::unique_ptr<uint64_t[]> keys(new (std::nothrow) uint64_t[std::max<uint64_t>(index->n_keys, 1)]);
    if (!keys) return dcn_fail(DCN_ERR_NOMEM, "index too large for host memory");
    uint64_t n = 0;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = dcn_table_export(index, keys.get(), index->n_keys, &n);
    if (rc != DCN_OK) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    rc = dcn_write_index_file(path, index->k, index->w, keys.get(), n);
    if (getenv("DCN_INDEX_TIMING"))
        fprintf(stderr, "index write timing: keys out of the table %.3f s, encoded and written %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    return rc;
}

extern "C" int dcn_index_header(const dcn_index *index, uint8_t *k, uint8_t *w, uint64_t *n_keys) {
    if (!index) return dcn_fail(DCN_ERR_ARG, "index is NULL");
    if (k) *k = index->k;
    if (w) *w = index->w;
    if (n_keys) *n_keys = index->n_keys;
    return DCN_OK;
}

extern "C" int dcn_index_memory(const dcn_index *index, uint64_t *table_bytes) {
    if (!index || !table_bytes) return dcn_fail(DCN_ERR_ARG, "index/table_bytes is NULL");
    *table_bytes = index->n_groups * DCN_GROUP_SLOTS * sizeof(uint64_t);
    return DCN_OK;
}

extern "C" int dcn_index_device(const dcn_index *index, int *device) {
    if (!index || !device) return dcn_fail(DCN_ERR_ARG, "index/device is NULL");
    *device = index->device;
    return DCN_OK;
}

extern "C" int dcn_index_contains(const dcn_index *index, const uint64_t *keys, uint64_t n, uint8_t *out) {
    if (!index) return dcn_fail(DCN_ERR_ARG, "index is NULL");
    if (n > 0 && (!keys || !out)) return dcn_fail(DCN_ERR_ARG, "keys/out is NULL");
    return dcn_table_contains(index, keys, n, out);
}

extern "C" int dcn_index_contains_device(const dcn_index *index, const uint64_t *d_keys, uint64_t n, uint8_t *d_out,
                                         void *stream) {
    if (!index) return dcn_fail(DCN_ERR_ARG, "index is NULL");
    if (n > 0 && (!d_keys || !d_out)) return dcn_fail(DCN_ERR_ARG, "d_keys/d_out is NULL");
    return dcn_table_contains_device(index, d_keys, n, d_out, (hipStream_t)stream);
}

extern "C" int dcn_index_probe_ceiling(const dcn_index *index, const uint64_t *d_keys, uint64_t n, uint32_t reps,
                                       double *probes_per_s) {
    if (!index || !probes_per_s) return dcn_fail(DCN_ERR_ARG, "index/probes_per_s is NULL");
    return dcn_table_probe_ceiling(index, d_keys, n, reps, probes_per_s, nullptr);
}

extern "C" int dcn_index_clone(const dcn_index *index, int device, dcn_index **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!index) return dcn_fail(DCN_ERR_ARG, "index is NULL");
    int ndev = 0;
    DCN_TRY_EARLY(dcn_device_count(&ndev));
    if (device < 0 || device >= ndev) return dcn_fail(DCN_ERR_ARG, "no such HIP device");
    dcn_index *idx = new (std::nothrow) dcn_index(*index);
    if (!idx) return dcn_fail(DCN_ERR_NOMEM, "host allocation failed");
    idx->device = device;
    idx->d_slots = nullptr;
    const uint64_t bytes = idx->n_groups * DCN_GROUP_SLOTS * sizeof(uint64_t);
    // Another GPU: the keys cross the link, not the table (a tenth of the bytes at the default 8 slots per key; dcn_table_clone_by_keys).
    // The same GPU: a device-to-device copy of the table at HBM's pace.  DCN_CLONE_BY_KEYS=1 / DCN_CLONE_BY_COPY=1 force one form
    // (the first is how the one-GPU tests reach the cross-device code).
    const bool by_keys = std::getenv("DCN_CLONE_BY_KEYS") || (device != index->device && !std::getenv("DCN_CLONE_BY_COPY"));
    if (by_keys) {
        if (device != index->device) {
            int can = 0;
            if (hipSetDevice(device) == hipSuccess && hipDeviceCanAccessPeer(&can, device, index->device) == hipSuccess && can)
                if (hipDeviceEnablePeerAccess(index->device, 0) != hipSuccess) (void)hipGetLastError(); // already enabled
        }
        const int rc = dcn_table_clone_by_keys(index, idx);
        if (rc != DCN_OK) {
            delete idx;
            return rc;
        }
        *out = idx;
        return DCN_OK;
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = dcn_table_malloc(&idx->d_slots, std::max<uint64_t>(bytes, 16));
    if (e == hipSuccess && bytes) {
        if (device == index->device) {
            e = hipMemcpy(idx->d_slots, index->d_slots, bytes, hipMemcpyDeviceToDevice);
        } else {
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, device, index->device);
            if (can) {
                hipError_t pe = hipDeviceEnablePeerAccess(index->device, 0);
                if (pe != hipSuccess) (void)hipGetLastError(); // already enabled
            }
            e = hipMemcpyPeer(idx->d_slots, device, index->d_slots, index->device, bytes);
        }
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (idx->d_slots) hipFree(idx->d_slots);
        delete idx;
        return dcn_fail(e == hipErrorOutOfMemory ? DCN_ERR_NOMEM : DCN_ERR_HIP, std::string("index clone: ") + hipGetErrorString(e));
    }
    *out = idx;
    return DCN_OK;
}

extern "C" void dcn_index_destroy(dcn_index *index) {
    if (!index) return;
    hipSetDevice(index->device);
    if (index->d_slots) hipFree(index->d_slots);
    delete index;
}

// ----------------------------------------------------------------------------------------------------
// context
// ----------------------------------------------------------------------------------------------------
// One pipeline step of a host batch: reads [r0, r1) = units [u0, u1) = bases [b0, b1) of the batch stream.
// groups [g0, g1) of the invalid-base mask that crossed the link as their non-zero words only: n (group, word) pairs at
// pairs_off of the slot's pair buffer; the range is cleared and the pairs scattered in front of the chunk's kernels
struct dcn_mask_range {
    uint64_t g0 = 0, g1 = 0, pairs_off = 0;
    uint32_t n = 0;
};

struct dcn_chunk {
    uint32_t r0 = 0, r1 = 0, u0 = 0, u1 = 0;
    uint64_t b0 = 0, b1 = 0;
    uint64_t max_len = 0; // longest read of the chunk
    std::vector<dcn_mask_range> mask_ranges;
};

// One host batch in flight (dcn_filter_batch_submit .. dcn_filter_batch_wait).  Everything a later batch's copies
// could overwrite while this batch's kernels still read it is per slot; the compute scratch (tiles, per-unit state,
// hit records) is shared, because all kernels of a context run on one stream.  Slot 0 uses the context's own
// buffers, slot 1 is allocated when a second batch is first submitted while slot 0 is busy.
struct dcn_slot {
    bool allocated = false, busy = false, owns_buffers = false;
    uint64_t ticket = 0;
    // device inputs / outputs
    uint8_t *d_ascii = nullptr;
    uint32_t *d_packed = nullptr, *d_invmask = nullptr;
    uint64_t *d_offsets = nullptr;
    // offsets of a batch of < 2^32 bases cross the link as u32 (4 instead of 8 bytes per read: 6 % of a packed call's bytes)
    // and are widened into d_offsets by a kernel in front of each chunk's own kernels
    uint32_t *d_off32 = nullptr, *h_off32 = nullptr;
    bool off32 = false;
    bool lean = false; // submitted on one stream, plain forms of everything (see submit_impl)
    // The invalid-base mask of a packed stream is a third of its bytes and almost all zero (a word per 32 bases, non-zero
    // only where a base is not ACGT): its non-zero words cross the link as (group, word) pairs, the rest is a memset on the
    // device.  A chunk whose pairs do not fit (one group in 16 non-zero, over the batch) goes whole.
    uint2 *d_mask_pairs = nullptr, *h_mask_pairs = nullptr;
    uint64_t mask_pairs_cap = 0, mask_pairs_used = 0;
    uint32_t *d_unit_id = nullptr;
    uint8_t *d_keep = nullptr;
    uint32_t *d_hits = nullptr, *d_total = nullptr;
    dcn_batch_report *d_report = nullptr;
    // page-locked result staging (used when the caller's output arrays are pageable)
    uint8_t *h_keep = nullptr;
    uint32_t *h_hits = nullptr, *h_total = nullptr;
    dcn_batch_report *h_report = nullptr;
    hipEvent_t done = nullptr;
    std::vector<hipEvent_t> ev_h2d, ev_comp; // one pair per chunk, grown on demand
    // the submitted batch, kept for result delivery and for the re-run after a record overflow
    dcn_params params = {};
    bool device_pack = false; // ASCII crossed the link: the pack kernel runs, read ends are probed for a newline
    bool has_units = false, counts = false;
    uint32_t n_reads = 0, n_units = 0;
    uint64_t n_bases = 0;
    uint8_t *u_keep = nullptr;
    uint32_t *u_hits = nullptr, *u_total = nullptr;
    bool keep_direct = false, hits_direct = false, total_direct = false; // caller's arrays are page-locked: copied into directly
    std::vector<dcn_chunk> chunks;
};

struct dcn_ctx {
    const dcn_index *index = nullptr;
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr, d2h_stream = nullptr;
    // device-pointer API, pack one batch ahead (ensure_pack_ahead): a second packed stream + mask, the pack kernel's own
    // status words and stream, and per buffer "packed" / "free again" events
    hipStream_t pack_stream = nullptr;
    uint32_t *d_packed_b = nullptr, *d_invmask_b = nullptr;
    dcn_status *d_pack_status = nullptr; // [2]
    hipEvent_t pack_done[2] = {}, buf_free[2] = {}, plan_done = nullptr;
    int pack_buf = 0, pack_ahead_state = 0; // 0 = not tried yet, 1 = ready, -1 = off (DCN_NO_PACK_AHEAD, or no memory for it)
    static constexpr int N_STAGE = 3, N_EV = 8, N_SLOTS = 2;
    hipEvent_t copy_done = nullptr, stage_free[N_STAGE] = {};
    hipEvent_t ev_h2d[N_EV] = {}, ev_comp[N_EV] = {};
    int ev_next = 0, stage_next = 0;
    uint64_t max_bases = 0;
    uint32_t max_reads = 0;
    uint32_t tile_windows = 256; // long reads: 12 % faster scan than 512 (fewer mid-scan flushes per wave), 128 and 1024 slower (profiles/r02_tile_sweep.txt)
    uint32_t max_tiles = 0;
    uint64_t chunk_bases = 0; // pipeline granularity of the host API (DCN_CHUNK_BASES)
    // device inputs (host API staging targets of slot 0; also used by the minimizer dump and the index build)
    uint8_t *d_ascii = nullptr;
    uint64_t *d_offsets = nullptr;
    uint32_t *d_unit_id = nullptr;
    // packed stream
    uint32_t *d_packed = nullptr, *d_invmask = nullptr; // allocations (views skip DCN_FRONT_PAD words)
    // plan
    uint32_t *d_read_tiles = nullptr, *d_read_tile_first = nullptr;
    uint32_t *d_unit_first_read = nullptr, *d_unit_tile_first = nullptr, *d_unit_tile_count = nullptr;
    dcn_tile *d_tiles = nullptr;
    // per-unit results / scratch
    uint8_t *d_keep = nullptr, *d_unit_state = nullptr;
    uint32_t *d_hits = nullptr, *d_total = nullptr;
    uint32_t *d_unit_scratch = nullptr; // g_total | g_hitcnt | g_distinct | g_zero, max_reads each; zero between batches
    bool scratch_dirty = false;         // a run was enqueued up to, but not including, its finish kernel
    uint32_t *d_caps = nullptr, *d_set_off = nullptr;
    // hit records + distinct sets
    // hit runs of the units the scan does not finish (one slot per base, see scan.hip) + global sets of the few
    // units whose hits do not fit the LDS set of the distinct pass (4 slots per record of capacity)
    uint64_t *d_rec_hash = nullptr;
    uint32_t rec_shift = 0;      // one slot of d_rec_hash per 2^rec_shift bases (dcn_scan_args::rec_shift)
    char *d_slab = nullptr;      // DCN_CTX_SLAB: one allocation behind the fixed-size buffers above
    uint64_t slab_bytes = 0;
    uint32_t *d_tile_hits = nullptr, *d_pending = nullptr;
    uint2 *d_big = nullptr; // work items of the distinct pass B: at most one per 64 tiles + one per unit
    uint64_t rec_capacity = 0;
    uint64_t *d_set_slots = nullptr;
    dcn_status *d_status = nullptr;
    dcn_batch_report *d_report = nullptr; // device-pointer API: counters + sticky overflow since the last synchronize
    dcn_batch_report *h_report = nullptr; // page-locked
    uint64_t host_stats[DCN_N_STATS] = {}; // counters of completed host batches
    // pinned host staging (pageable inputs)
    uint8_t *h_stage[N_STAGE] = {};
    uint64_t stage_bytes = 0;
    dcn_slot slots[N_SLOTS];
    uint64_t next_ticket = 1;
    // dump mode buffers (lazy)
    uint64_t *d_dump_hash = nullptr;
    uint32_t *d_dump_pos = nullptr, *d_dump_count = nullptr, *d_tile_read_pos = nullptr;
    uint8_t *d_dump_valid = nullptr;
    // deferred state of the last enqueued device-API batch
    bool batch_pending = false;
    bool lean = false; // a small host batch is being submitted: copies and result copies go on `stream` itself (submit_impl)
    // optional per-stage timing: a ring of event sets, one per batch in flight
    static constexpr int PROF_RING = 64;
    int profiling = 0; // 0 off, 1 every stage, 2 the scan stage only (two events per run instead of six)
    hipEvent_t prof_ev[PROF_RING][DCN_N_STAGES + 1] = {};
    bool prof_used[PROF_RING] = {}, prof_scan_only[PROF_RING] = {};
    int prof_next = 0;
    double prof_ms[DCN_N_STAGES] = {};
    uint64_t prof_batches = 0;
};

namespace {

template <typename T>
int dev_alloc(T **p, uint64_t count, const char *what) {
    hipError_t e = hipMalloc((void **)p, std::max<uint64_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) {
        *p = nullptr;
        return dcn_fail(DCN_ERR_NOMEM, std::string("hipMalloc ") + what + ": " + hipGetErrorString(e));
    }
    return DCN_OK;
}

#define DCN_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc != DCN_OK) return _rc; \
    } while (0)

uint64_t packed_words(uint64_t max_bases) { return DCN_FRONT_PAD + 2 * ((max_bases + 31) / 32) + DCN_TAIL_PAD; }
uint64_t mask_words(uint64_t max_bases) { return DCN_FRONT_PAD + (max_bases + 31) / 32 + DCN_TAIL_PAD; }

int alloc_records(dcn_ctx *c, uint64_t n_records) {
    n_records = (n_records + 63) / 64 * 64;
    if (n_records > (1ull << 29)) return dcn_fail(DCN_ERR_CAPACITY, "more than 2^29 hit records in global sets per batch: use smaller batches");
    if (c->d_set_slots) hipFree(c->d_set_slots);
    c->d_set_slots = nullptr;
    c->rec_capacity = 0;
    DCN_TRY(dev_alloc(&c->d_set_slots, 4 * n_records + 64, "set_slots"));
    c->rec_capacity = n_records;
    return DCN_OK;
}

void free_slot_buffers(dcn_slot &sl) {
    if (sl.owns_buffers) {
        void *dev[] = {sl.d_ascii, sl.d_packed, sl.d_invmask, sl.d_offsets, sl.d_unit_id, sl.d_keep, sl.d_hits, sl.d_total};
        for (void *p : dev)
            if (p) hipFree(p);
    }
    if (sl.d_report) hipFree(sl.d_report);
    if (sl.d_off32) hipFree(sl.d_off32);
    if (sl.d_mask_pairs) hipFree(sl.d_mask_pairs);
    void *host[] = {sl.h_keep, sl.h_hits, sl.h_total, sl.h_report, sl.h_off32, sl.h_mask_pairs};
    for (void *p : host)
        if (p) hipHostFree(p);
    if (sl.done) hipEventDestroy(sl.done);
    for (hipEvent_t e : sl.ev_h2d) hipEventDestroy(e);
    for (hipEvent_t e : sl.ev_comp) hipEventDestroy(e);
    sl = dcn_slot();
}

void free_ctx(dcn_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    if (c->d2h_stream) hipStreamSynchronize(c->d2h_stream);
    if (c->pack_stream) {
        hipStreamSynchronize(c->pack_stream);
        hipStreamDestroy(c->pack_stream);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->pack_done[i]) hipEventDestroy(c->pack_done[i]);
        if (c->buf_free[i]) hipEventDestroy(c->buf_free[i]);
    }
    if (c->plan_done) hipEventDestroy(c->plan_done);
    if (c->d_packed_b) hipFree(c->d_packed_b);
    if (c->d_invmask_b) hipFree(c->d_invmask_b);
    if (c->d_pack_status) hipFree(c->d_pack_status);
    for (auto &sl : c->slots) free_slot_buffers(sl);
    void *dev[] = {c->d_ascii, c->d_offsets, c->d_unit_id, c->d_packed, c->d_invmask,
                   c->d_read_tiles, c->d_read_tile_first, c->d_unit_first_read, c->d_unit_tile_first, c->d_unit_tile_count, c->d_tiles,
                   c->d_keep, c->d_unit_state, c->d_hits, c->d_total, c->d_unit_scratch, c->d_caps,
                   c->d_set_off, c->d_tile_hits, c->d_pending, c->d_big, c->d_rec_hash, c->d_set_slots, c->d_status, c->d_report, c->d_dump_hash,
                   c->d_dump_pos, c->d_dump_count, c->d_dump_valid, c->d_tile_read_pos};
    for (void *p : dev)
        if (p && !((char *)p >= c->d_slab && (char *)p < c->d_slab + c->slab_bytes)) hipFree(p);
    if (c->d_slab) hipFree(c->d_slab);
    for (int i = 0; i < dcn_ctx::N_STAGE; ++i) {
        if (c->h_stage[i]) hipHostFree(c->h_stage[i]);
        if (c->stage_free[i]) hipEventDestroy(c->stage_free[i]);
    }
    for (int i = 0; i < dcn_ctx::N_EV; ++i) {
        if (c->ev_h2d[i]) hipEventDestroy(c->ev_h2d[i]);
        if (c->ev_comp[i]) hipEventDestroy(c->ev_comp[i]);
    }
    if (c->h_report) hipHostFree(c->h_report);
    for (int i = 0; i < dcn_ctx::PROF_RING; ++i)
        for (int j = 0; j <= DCN_N_STAGES; ++j)
            if (c->prof_ev[i][j]) hipEventDestroy(c->prof_ev[i][j]);
    if (c->copy_done) hipEventDestroy(c->copy_done);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->d2h_stream) hipStreamDestroy(c->d2h_stream);
    delete c;
}

// fold the event pairs of every completed batch into the per-stage accumulators
int prof_harvest(dcn_ctx *c, int only_slot = -1) {
    for (int i = 0; i < dcn_ctx::PROF_RING; ++i) {
        if (!c->prof_used[i] || (only_slot >= 0 && i != only_slot)) continue;
        const int first = c->prof_scan_only[i] ? DCN_STAGE_SCAN : 0, last = c->prof_scan_only[i] ? DCN_STAGE_SCAN : DCN_N_STAGES - 1;
        DCN_HIP(hipEventSynchronize(c->prof_ev[i][last + 1]));
        for (int j = first; j <= last; ++j) {
            float ms = 0.f;
            DCN_HIP(hipEventElapsedTime(&ms, c->prof_ev[i][j], c->prof_ev[i][j + 1]));
            c->prof_ms[j] += ms;
        }
        c->prof_batches++;
        c->prof_used[i] = false;
    }
    return DCN_OK;
}

// returns the event slot for this batch (or -1 when profiling is off) after recording its first event
int prof_begin(dcn_ctx *c, int *slot) {
    *slot = -1;
    if (!c->profiling) return DCN_OK;
    int i = c->prof_next;
    c->prof_next = (i + 1) % dcn_ctx::PROF_RING;
    if (c->prof_used[i]) DCN_TRY(prof_harvest(c, i));
    for (int j = 0; j <= DCN_N_STAGES; ++j)
        if (!c->prof_ev[i][j]) DCN_HIP(hipEventCreate(&c->prof_ev[i][j]));
    c->prof_scan_only[i] = c->profiling == 2;
    if (c->profiling == 1) DCN_HIP(hipEventRecord(c->prof_ev[i][0], c->stream));
    *slot = i;
    return DCN_OK;
}

// (a timed event is a marker packet the stream stops at: six per run cost the headline step ~2.5 %, which is why
// the scan-only level exists: the end of the plan stage is the start of the scan stage)
#define DCN_PROF_MARK(stage)                                                                                      \
    do {                                                                                                          \
        if (prof_slot >= 0 && (c->profiling == 1 || (stage) == DCN_STAGE_PLAN || (stage) == DCN_STAGE_SCAN))      \
            DCN_HIP(hipEventRecord(c->prof_ev[prof_slot][(stage) + 1], st));                                      \
    } while (0)

int check_params(const dcn_params *p) {
    if (!p) return dcn_fail(DCN_ERR_ARG, "params is NULL");
    if (p->reserved != 0) return dcn_fail(DCN_ERR_ARG, "params.reserved must be 0");
    if (p->deplete > 1) return dcn_fail(DCN_ERR_ARG, "params.deplete must be 0 or 1");
    return DCN_OK;
}

// What one run of the device pipeline works on: a whole batch of the device-pointer API, or one chunk of a host
// batch.  Base offsets in d_offsets are positions in the batch stream (d_ascii / d_packed are the stream's
// origin); read and unit indices are local to the view (arrays already point at the view's first entry).
struct BatchView {
    const uint8_t *d_ascii = nullptr;  // null: the stream arrived packed (no pack kernel, no newline probe)
    uint32_t *d_packed = nullptr, *d_invmask = nullptr; // allocation starts (DCN_FRONT_PAD words in front of base 0)
    const uint64_t *d_offsets = nullptr;
    const uint32_t *d_unit_id = nullptr;
    uint32_t unit_base = 0;
    uint32_t n_reads = 0, n_units = 0;
    uint64_t b0 = 0, b1 = 0; // bases of the stream this view covers
    uint64_t stream_bases = 0; // bases of the whole stream
    uint8_t *d_keep = nullptr;
    uint32_t *d_hits = nullptr, *d_total = nullptr;
    dcn_batch_report *d_report = nullptr;
};

// Device-pointer API, experiment kept behind DCN_PACK_AHEAD=1: the pack kernel of batch i+1 runs beside the scan kernel of
// batch i.  The pack is a streaming kernel (1 B/bp in, 0.375 out: 0.39 ms of a 3.7 ms step at 1.5 Gbp) and the scan kernel
// moves only 38 % of HBM's peak, so batch i+1's stream is packed into a SECOND buffer on a side stream once the batch that
// last read that buffer (i-1) has finished and batch i's plan kernel is through, instead of in front of its own scan.
// Costs 0.375 B per base of context; everything else of a batch stays in order on the context's stream.
bool ensure_pack_ahead(dcn_ctx *c) {
    if (c->pack_ahead_state != 0) return c->pack_ahead_state > 0;
    c->pack_ahead_state = -1;
    // OFF unless asked for (DCN_PACK_AHEAD=1): measured in round 4 (profiles/r04_ab.txt section 4), it buys nothing.  The two
    // kernels do run side by side (kernel trace), and the scan kernel then takes longer by exactly the pack's time (3.30 ->
    // 3.67 ms, step 3.85 -> 3.88): what the scan kernel leaves of HBM's bandwidth is not spare -- its scattered sectors and
    // the pack's stream wait for the same DRAM cycles.
    if (!getenv("DCN_PACK_AHEAD") || getenv("DCN_NO_PACK_AHEAD")) return false;
    // (highest priority: the scan kernel's grid is 150 k workgroups deep, and a queue of ordinary priority only gets its
    // turn when that grid has drained)
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    bool ok = hipStreamCreateWithPriority(&c->pack_stream, hipStreamNonBlocking, prio_high) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->plan_done, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < 2; ++i)
        ok = hipEventCreateWithFlags(&c->pack_done[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->buf_free[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void **)&c->d_packed_b, packed_words(c->max_bases) * sizeof(uint32_t)) == hipSuccess &&
         hipMalloc((void **)&c->d_invmask_b, mask_words(c->max_bases) * sizeof(uint32_t)) == hipSuccess &&
         hipMalloc((void **)&c->d_pack_status, 2 * sizeof(dcn_status)) == hipSuccess;
    if (ok) { // (pads in front of and behind the stream are read by the scan kernel: zero, as in the first buffer)
        ok = hipMemset(c->d_packed_b, 0, packed_words(c->max_bases) * sizeof(uint32_t)) == hipSuccess &&
             hipMemset(c->d_invmask_b, 0, mask_words(c->max_bases) * sizeof(uint32_t)) == hipSuccess &&
             hipDeviceSynchronize() == hipSuccess; // null-stream memsets must not overtake the first pack
    }
    if (!ok) {
        (void)hipGetLastError(); // no memory for a second stream: batches are packed in line, as before
        if (c->d_packed_b) hipFree(c->d_packed_b);
        if (c->d_invmask_b) hipFree(c->d_invmask_b);
        if (c->d_pack_status) hipFree(c->d_pack_status);
        c->d_packed_b = c->d_invmask_b = nullptr;
        c->d_pack_status = nullptr;
        return false;
    }
    c->pack_ahead_state = 1;
    return true;
}

// enqueue the whole device pipeline for one view on the context's compute stream
int enqueue_batch(dcn_ctx *c, const BatchView &v, const dcn_params *params, bool pack_ahead = false) {
    hipStream_t st = c->stream;
    const dcn_index *idx = c->index;
    // the per-unit scratch words are zero between batches (finish_kernel leaves them so); a run that did not get as
    // far as enqueueing its finish kernel may have left some behind
    if (c->scratch_dirty) DCN_HIP(hipMemsetAsync(c->d_unit_scratch, 0, (uint64_t)c->max_reads * 4 * sizeof(uint32_t), st));
    c->scratch_dirty = true;
    // per-run scratch: the status words (the per-unit state and scratch words are cleared by the plan kernel)
    DCN_HIP(hipMemsetAsync(c->d_status, 0, sizeof(dcn_status), st));

    int prof_slot = -1;
    DCN_TRY(prof_begin(c, &prof_slot));
    uint32_t *packed = v.d_packed + DCN_FRONT_PAD, *invmask = v.d_invmask + DCN_FRONT_PAD;
    int ahead_buf = -1;
    const uint32_t *newline_flag = nullptr;
    // (per-stage profiling wants the stages one after the other on one stream: in line then)
    if (v.d_ascii && pack_ahead && c->profiling != 1 && ensure_pack_ahead(c)) {
        ahead_buf = c->pack_buf;
        c->pack_buf ^= 1;
        if (ahead_buf == 1) {
            packed = c->d_packed_b + DCN_FRONT_PAD;
            invmask = c->d_invmask_b + DCN_FRONT_PAD;
        }
        dcn_status *ps = c->d_pack_status + ahead_buf;
        DCN_HIP(hipStreamWaitEvent(c->pack_stream, c->buf_free[ahead_buf], 0)); // (never recorded yet: no wait)
        // ... and not before the previous batch's plan kernel is through: its buffer is free from the moment the batch
        // before that one finished, which is just when the previous batch's (small, latency-bound) plan kernel starts --
        // packing beside THAT only delays the scan kernel behind it (kernel trace: plan 0.10 -> 0.47 ms)
        if (!getenv("DCN_PACK_AHEAD_EARLY")) DCN_HIP(hipStreamWaitEvent(c->pack_stream, c->plan_done, 0));
        DCN_HIP(hipMemsetAsync(ps, 0, sizeof(dcn_status), c->pack_stream));
        DCN_TRY(dcn_launch_pack_beside(v.d_ascii, v.b0, v.b1, packed, invmask, ps, c->pack_stream));
        DCN_HIP(hipEventRecord(c->pack_done[ahead_buf], c->pack_stream));
        DCN_HIP(hipStreamWaitEvent(st, c->pack_done[ahead_buf], 0));
        newline_flag = &ps->any_newline;
    } else if (v.d_ascii) {
        DCN_TRY(dcn_launch_pack(v.d_ascii, v.b0, v.b1, packed, invmask, c->d_status, st));
    }
    DCN_PROF_MARK(DCN_STAGE_PACK);

    const uint32_t n_reads = v.n_reads, n_units = v.n_units;
    dcn_plan_args pa = {};
    pa.ascii = v.d_ascii;
    pa.offsets = v.d_offsets;
    pa.unit_id = v.d_unit_id;
    pa.unit_base = v.unit_base;
    pa.n_reads = n_reads;
    pa.n_units = n_units;
    pa.k = idx->k;
    pa.w = idx->w;
    pa.prefix_length = params->prefix_length;
    pa.tile_windows = c->tile_windows;
    pa.read_tiles = nullptr; // per-read tile ranges are only needed by the minimizer dump
    pa.read_tile_first = nullptr;
    pa.unit_first_read = c->d_unit_first_read;
    pa.unit_state = c->d_unit_state;
    pa.unit_scratch = c->d_unit_scratch;
    pa.scratch_stride = c->max_reads;
    pa.unit_tile_first = c->d_unit_tile_first;
    pa.unit_tile_count = c->d_unit_tile_count;
    pa.tile_cursor = &c->d_status->n_tiles;
    pa.tiles = c->d_tiles;
    pa.status = c->d_status;
    pa.newline_flag = newline_flag;
    pa.stream_bases = v.stream_bases;
    pa.check_offsets = 1;
    pa.max_tiles = c->max_tiles;
    DCN_TRY(dcn_launch_plan(pa, st));
    if (pack_ahead && c->pack_ahead_state == 1) DCN_HIP(hipEventRecord(c->plan_done, st));
    DCN_PROF_MARK(DCN_STAGE_PLAN);

    uint32_t *g_total = c->d_unit_scratch, *g_hitcnt = g_total + c->max_reads, *g_distinct = g_hitcnt + c->max_reads,
             *g_zero = g_distinct + c->max_reads;
    dcn_scan_args sa;
    memset(&sa, 0, sizeof(sa));
    sa.packed = packed;
    sa.invmask = invmask;
    sa.tiles = c->d_tiles;
    sa.n_tiles = &c->d_status->n_tiles;
    sa.unit_tile_first = c->d_unit_tile_first;
    sa.unit_tile_count = c->d_unit_tile_count;
    sa.table = idx->view();
    sa.k = idx->k;
    sa.variant = idx->variant;
    sa.w = idx->w;
    sa.stream_bases = v.stream_bases;
    sa.abs_threshold = params->abs_threshold;
    sa.rel_threshold = params->rel_threshold;
    sa.deplete = params->deplete;
    // decisions only: largest list length whose required hits still equal abs_threshold (dcn_required_hits is
    // monotone in the total); the scan kernel's lanes then stop at abs_threshold distinct hits (scan.hip)
    sa.early_out_max_items = 0;
    static const bool no_early_out = getenv("DCN_NO_EARLY_OUT") != nullptr, no_early_out_pairs = getenv("DCN_NO_EARLY_OUT_PAIRS") != nullptr;
    if (!v.d_hits && !v.d_total && params->abs_threshold >= 1 && params->abs_threshold <= 4 && !no_early_out) {
        uint32_t lo = 0, hi = 65535; // required(lo) == abs always holds for lo = 0
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) / 2;
            if (dcn_required_hits(params->abs_threshold, params->rel_threshold, mid) == params->abs_threshold) lo = mid;
            else hi = mid - 1;
        }
        sa.early_out_max_items = lo;
        sa.early_out_pairs = no_early_out_pairs ? 0u : 1u;
    }
    sa.keep = v.d_keep;
    sa.hits = v.d_hits;
    sa.total = v.d_total;
    sa.unit_state = c->d_unit_state;
    sa.g_total = g_total;
    sa.g_hitcnt = g_hitcnt;
    sa.g_zero = g_zero;
    sa.rec_hash = c->d_rec_hash;
    sa.rec_shift = c->rec_shift;
    sa.tile_windows = c->tile_windows;
    sa.tile_hits = c->d_tile_hits;
    sa.pending = c->d_pending;
    sa.status = c->d_status;
    uint64_t tile_bound = (uint64_t)n_reads + (v.b1 - v.b0) / c->tile_windows + 1;
    if (tile_bound > c->max_tiles) tile_bound = c->max_tiles;
    DCN_TRY(dcn_launch_scan(sa, (uint32_t)tile_bound, false, st));
    DCN_PROF_MARK(DCN_STAGE_SCAN);

    dcn_distinct_args da;
    da.tiles = c->d_tiles;
    da.n_tiles = &c->d_status->n_tiles;
    da.unit_tile_first = c->d_unit_tile_first;
    da.unit_tile_count = c->d_unit_tile_count;
    da.unit_state = c->d_unit_state;
    da.tile_hits = c->d_tile_hits;
    da.pending = c->d_pending;
    da.rec_hash = c->d_rec_hash;
    da.rec_shift = c->rec_shift;
    da.g_hitcnt = g_hitcnt;
    da.g_distinct = g_distinct;
    da.set_off = c->d_set_off;
    da.set_slots = c->d_set_slots;
    da.set_capacity = 4 * c->rec_capacity + 64;
    da.n_units = n_units;
    da.status = c->d_status;
    da.caps = c->d_caps;
    da.big = c->d_big;
    da.g_total = (!v.d_hits && !v.d_total && !getenv("DCN_NO_EARLY_OUT")) ? g_total : nullptr;
    da.abs_threshold = params->abs_threshold;
    da.rel_threshold = params->rel_threshold;
    DCN_TRY(dcn_launch_distinct(da, st));
    DCN_PROF_MARK(DCN_STAGE_DISTINCT);

    dcn_finish_args fa;
    fa.n_units = n_units;
    fa.unit_first_read = v.d_unit_id ? c->d_unit_first_read : nullptr;
    fa.offsets = v.d_offsets;
    fa.unit_state = c->d_unit_state;
    fa.g_total = g_total;
    fa.g_hitcnt = g_hitcnt;
    fa.g_distinct = g_distinct;
    fa.g_zero = g_zero;
    fa.abs_threshold = params->abs_threshold;
    fa.rel_threshold = params->rel_threshold;
    fa.deplete = params->deplete;
    fa.keep = v.d_keep;
    fa.hits = v.d_hits;
    fa.total = v.d_total;
    fa.report = v.d_report;
    fa.status = c->d_status;
    DCN_TRY(dcn_launch_finish(fa, st));
    c->scratch_dirty = false;
    // the next pack into the buffer this batch read may start (a batch packed in line between packed-ahead ones -- per-stage
    // profiling -- read the first buffer)
    if (pack_ahead && c->pack_ahead_state == 1) DCN_HIP(hipEventRecord(c->buf_free[ahead_buf >= 0 ? ahead_buf : 0], st));
    DCN_PROF_MARK(DCN_STAGE_FINISH);
    if (prof_slot >= 0) c->prof_used[prof_slot] = true;
    return DCN_OK;
}

// a run overflowed (dcn_status::run_overflow): from now on the context keeps one slot of the record array per window
int grow_run_slots(dcn_ctx *c) {
    // (a batch that was in flight beside the one that made the context switch over reports the same overflow, from its
    // run under the old geometry: it is simply run again)
    if (c->rec_shift == 0) return DCN_OK;
    // The old array goes first (the caller has synchronized the stream: nothing reads it any more), so the peak is the
    // new 8 B per base and not 8 + 2: on a context sized close to the card the larger array alone may still fit.
    const bool in_slab = c->d_rec_hash && (char *)c->d_rec_hash >= c->d_slab && (char *)c->d_rec_hash < c->d_slab + c->slab_bytes;
    if (c->d_rec_hash && !in_slab) hipFree(c->d_rec_hash);
    c->d_rec_hash = nullptr;
    const uint32_t old_shift = c->rec_shift;
    c->rec_shift = 0;
    int rc = dev_alloc(&c->d_rec_hash, c->max_bases + 128, "rec_hash (one slot per window)");
    if (rc != DCN_OK) { // back to an array of the old geometry, so that the context stays usable for batches that fit it
        c->rec_shift = old_shift;
        int rc2 = dev_alloc(&c->d_rec_hash, (c->max_bases >> old_shift) + 256, "rec_hash");
        return rc2 != DCN_OK ? rc2 : rc;
    }
    static std::atomic<bool> said{false};
    if (!said.exchange(true) && getenv("DCN_QUIET") == nullptr)
        std::fprintf(stderr, "deacon-hip: a unit's hits outgrew its run of the record array; this context now keeps one slot per window "
                             "(%.2f GB instead of %.2f GB of device memory)\n", (c->max_bases + 128) * 8e-9,
                     ((c->max_bases >> old_shift) + 256) * 8e-9);
    return DCN_OK;
}

int overflow_error(const dcn_ctx *c, uint64_t need) {
    return dcn_fail(DCN_ERR_CAPACITY, "hit-record scratch overflow: need " + std::to_string(need) +
                                          " records, have " + std::to_string(c->rec_capacity) +
                                          " (dcn_ctx_reserve_records)");
}

// Wait for the compute stream and surface deferred errors of the device-pointer API.  The overflow word is
// sticky across batches (the per-run status words are not): an overflow in ANY batch enqueued since the last
// synchronize is reported here, however many smaller batches followed it.
int sync_and_check(dcn_ctx *c, uint64_t *needed_records) {
    DCN_HIP(hipStreamSynchronize(c->stream));
    if (needed_records) *needed_records = 0;
    if (c->profiling) DCN_TRY(prof_harvest(c));
    if (!c->batch_pending) return DCN_OK;
    c->batch_pending = false;
    DCN_HIP(hipMemcpy(c->h_report, c->d_report, sizeof(dcn_batch_report), hipMemcpyDeviceToHost));
    if (c->h_report->bounds) {
        DCN_HIP(hipMemsetAsync(c->d_report, 0, offsetof(dcn_batch_report, stats), c->stream));
        return dcn_fail(DCN_ERR_INTERNAL, "scan kernel: index out of range in phase B (DCN_DEBUG_BOUNDS build)");
    }
    if (c->h_report->bad_offsets) {
        DCN_HIP(hipMemsetAsync(c->d_report, 0, offsetof(dcn_batch_report, stats), c->stream));
        return dcn_fail(DCN_ERR_ARG, "d_offsets of a batch since the last synchronize were not non-decreasing within [0, n_bases] when the "
                                     "device read them (were they written, and ordered before the context's stream, when "
                                     "dcn_filter_batch_device was called?): the outputs and counters of those batches are undefined");
    }
    if (c->h_report->overflow) {
        const uint64_t need = c->h_report->need;
        const bool runs = (c->h_report->overflow & 2u) != 0, sets = (c->h_report->overflow & 1u) != 0;
        DCN_HIP(hipMemsetAsync(c->d_report, 0, offsetof(dcn_batch_report, stats), c->stream)); // re-arm, ordered before the next batch
        if (runs) DCN_TRY(grow_run_slots(c)); // (the stream is idle: nothing reads the old array any more)
        if (needed_records) *needed_records = sets ? need : 0;
        if (!sets)
            return dcn_fail(DCN_ERR_CAPACITY, "a unit had more hits in one wave than its run of the record array holds; the context "
                                              "now keeps one slot per window: enqueue the batches since the last synchronize again");
        return overflow_error(c, need);
    }
    return DCN_OK;
}

// (class HostPool: dcn_host_pool.h, included at the top of this file)
using dcn_host::HostPool;

// page-locked host memory (hipHostMalloc / hipHostRegister, e.g. from dcn_host_alloc) needs no staging
bool is_pinned_host(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// Copy `bytes` to the device on the copy stream.  Page-locked sources go straight over the link; anything else is
// cut into pieces that pass through the ring of pinned staging buffers (the host fills one while earlier ones are
// in flight).  `fill(dst, first, n)` produces bytes [first, first + n) of the payload in the staging buffer:
// a (threaded) memcpy, or the host-side 2-bit pack.
template <typename Fill>
int staged_h2d_fill(dcn_ctx *c, void *d_dst, uint64_t bytes, uint64_t piece_align, Fill fill) {
    uint8_t *dst = (uint8_t *)d_dst;
    const uint64_t piece = c->stage_bytes / piece_align * piece_align;
    for (uint64_t off = 0; off < bytes; off += piece) {
        const int which = c->stage_next;
        c->stage_next = (which + 1) % dcn_ctx::N_STAGE;
        const uint64_t m = std::min<uint64_t>(piece, bytes - off);
        DCN_HIP(hipEventSynchronize(c->stage_free[which])); // previous copy out of this buffer finished
        fill(c->h_stage[which], off, m);
        DCN_HIP(hipMemcpyAsync(dst + off, c->h_stage[which], m, hipMemcpyHostToDevice, c->copy_stream));
        DCN_HIP(hipEventRecord(c->stage_free[which], c->copy_stream));
    }
    return DCN_OK;
}

int staged_h2d(dcn_ctx *c, void *d_dst, const void *h_src, uint64_t bytes, int pinned = -1) {
    if (bytes == 0) return DCN_OK;
    if (pinned < 0) pinned = is_pinned_host(h_src) ? 1 : 0;
    if (pinned) {
        DCN_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->copy_stream));
        return DCN_OK;
    }
    const uint8_t *src = (const uint8_t *)h_src;
    return staged_h2d_fill(c, d_dst, bytes, 64, [&](uint8_t *stage, uint64_t first, uint64_t n) {
        HostPool::get().copy(stage, src + first, n);
    });
}

} // namespace
void dcn_host_parallel_copy(void *dst, const void *src, size_t n) { HostPool::get().copy(dst, src, n); }
namespace {

int slots_busy(const dcn_ctx *c) {
    int n = 0;
    for (const auto &sl : c->slots) n += sl.busy ? 1 : 0;
    return n;
}

} // namespace

// Index file whose remaining bytes after the count are exactly 9 per hash: every hash is `0xFD + u64 LE`
// (nothing shorter fits, 9 is the longest u64 varint), so record i is at a fixed offset.  The file is mapped,
// copied chunk by chunk into pinned memory by the host copy threads, and decoded + inserted by
// table_insert_varint9_kernel while the next chunk is being copied (load_minimizer_hashes, src/index.rs:80-107).
// *handled stays false when the file is not of that shape (or cannot be mapped): the caller then runs the
// general host decoder, which also produces the reference's error messages.
int dcn_load_index_fixed9(const char *path, int device, dcn_index **out, bool *handled) {
    *handled = false;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return DCN_OK;
    struct stat st;
    uint8_t head[12];
    ssize_t got = 0;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || (got = pread(fd, head, sizeof head, 0)) < 4 || head[0] != 2) {
        close(fd);
        return DCN_OK;
    }
    const uint8_t k = head[1], w = head[2], b = head[3];
    size_t len = b < 251 ? 1 : b == 0xFB ? 3 : b == 0xFC ? 5 : b == 0xFD ? 9 : 0;
    uint64_t count = b;
    if (len == 0 || (size_t)got < 3 + len) {
        close(fd);
        return DCN_OK;
    }
    if (len > 1) {
        count = 0;
        memcpy(&count, head + 4, len - 1);
    }
    const uint64_t pos = 3 + len, size = (uint64_t)st.st_size;
    int ndev = 0;
    // DCN_LOAD_TIMING=1: where the load's wall time goes, one line on stderr (runtime start-up = the first HIP call)
    const bool timing = getenv("DCN_LOAD_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    double t_marks[6] = {0, 0, 0, 0, 0, 0};
    auto mark = [&](int i) { t_marks[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    if (count == 0 || count > (1ull << 40) || size - pos != 9 * count || check_kw(k, w) != DCN_OK ||
        dcn_device_count(&ndev) != DCN_OK || device < 0 || device >= ndev) {
        close(fd);
        return DCN_OK;
    }
    mark(0);  // runtime up (dcn_device_count)
    // The host threads pread() their slices of a chunk straight into the page-locked buffer.  (Until round 3 the file was
    // mapped and copied out of the mapping: unmapping its 900 k pages and releasing two 72 MB staging buffers took
    // 0.06-0.11 s of a 0.32-0.43 s load of panhuman-1's 3.7 GB, pinning the buffers 0.04-0.06 s; now 0.00 and 0.02-0.03 s.
    // DCN_LOAD_TIMING=1 prints the split.)
    (void)posix_fadvise(fd, 0, 0, POSIX_FADV_SEQUENTIAL);
    const uint64_t CH = std::min<uint64_t>(count, 7ull << 19); // records per chunk (33 MB)
    const uint64_t ch_bytes = 9 * CH + 16;
    dcn_index *idx = new (std::nothrow) dcn_index();
    uint8_t *h_buf[2] = {nullptr, nullptr};
    uint64_t *d_raw[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t st_ = nullptr;
    unsigned long long *d_new = nullptr;
    uint32_t *d_flags = nullptr; // [0] has_zero, [1] bad marker
    int rc = idx ? DCN_OK : dcn_fail(DCN_ERR_NOMEM, "host allocation failed");
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == DCN_OK) rc = dcn_fail(DCN_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (rc == DCN_OK) {
        idx->device = device;
        idx->k = k;
        idx->w = w;
        hip_ok(hipSetDevice(device), "hipSetDevice");
    }
    if (rc == DCN_OK) rc = dcn_table_alloc(idx, count);
    if (rc == DCN_OK) {
        hip_ok(hipDeviceSynchronize(), "table clear"); // the table's memset ran on the null stream
        mark(1);  // table allocated and cleared
        hip_ok(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking), "stream");
        for (int i = 0; i < 2 && rc == DCN_OK; ++i) {
            hip_ok(hipHostMalloc((void **)&h_buf[i], ch_bytes, hipHostMallocDefault), "hipHostMalloc");
            hip_ok(hipMalloc((void **)&d_raw[i], ch_bytes), "hipMalloc");
            hip_ok(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming), "event");
            if (rc == DCN_OK) hip_ok(hipMemsetAsync(d_raw[i], 0, ch_bytes, st_), "memset");
        }
        hip_ok(hipMalloc((void **)&d_new, sizeof(unsigned long long)), "hipMalloc");
        hip_ok(hipMalloc((void **)&d_flags, 2 * sizeof(uint32_t)), "hipMalloc");
        if (rc == DCN_OK) {
            hip_ok(hipMemsetAsync(d_new, 0, sizeof(unsigned long long), st_), "memset");
            hip_ok(hipMemsetAsync(d_flags, 0, 2 * sizeof(uint32_t), st_), "memset");
        }
    }
    mark(2);  // staging buffers
    bool read_failed = false;  // (the general decoder then reports what is wrong with the file)
    int which = 0;
    for (uint64_t off = 0; off < count && rc == DCN_OK; off += CH, which ^= 1) {
        const uint64_t m = std::min<uint64_t>(CH, count - off);
        if (!hip_ok(hipEventSynchronize(ev[which]), "event wait")) break; // the copy out of this buffer is done
        std::atomic<int> bad{0};
        HostPool::get().run([&](int i, int nt) {
            const uint64_t n = 9 * m, per = ((n / nt) + 4095) & ~4095ull;
            uint64_t lo = std::min(n, per * i);
            const uint64_t hi = i == nt - 1 ? n : std::min(n, per * (i + 1));
            while (lo < hi) {
                const ssize_t g = pread(fd, h_buf[which] + lo, hi - lo, (off_t)(pos + 9 * off + lo));
                if (g < 0 && errno == EINTR) continue;
                if (g <= 0) {
                    bad.store(1);
                    return;
                }
                lo += (uint64_t)g;
            }
        });
        if (bad.load()) {
            read_failed = true;
            break;
        }
        if (!hip_ok(hipMemcpyAsync(d_raw[which], h_buf[which], 9 * m, hipMemcpyHostToDevice, st_), "hipMemcpyAsync")) break;
        rc = dcn_table_insert_varint9(idx, d_raw[which], m, d_new, d_flags, d_flags + 1, st_);
        if (rc == DCN_OK) hip_ok(hipEventRecord(ev[which], st_), "event record");
    }
    unsigned long long h_new = 0;
    uint32_t h_flags[2] = {0, 0};
    close(fd);
    if (rc == DCN_OK && !read_failed) {
        hip_ok(hipStreamSynchronize(st_), "index load");
        hip_ok(hipMemcpy(&h_new, d_new, sizeof h_new, hipMemcpyDeviceToHost), "hipMemcpy");
        hip_ok(hipMemcpy(h_flags, d_flags, sizeof h_flags, hipMemcpyDeviceToHost), "hipMemcpy");
    }
    if (rc == DCN_OK && h_flags[1]) rc = dcn_fail(DCN_ERR_FORMAT, "Failed to deserialise minimizer hash");
    mark(3);  // every chunk copied, decoded and inserted
    if (st_) hipStreamSynchronize(st_);
    for (int i = 0; i < 2; ++i) {
        if (h_buf[i]) hipHostFree(h_buf[i]);
        if (d_raw[i]) hipFree(d_raw[i]);
        if (ev[i]) hipEventDestroy(ev[i]);
    }
    if (d_new) hipFree(d_new);
    if (d_flags) hipFree(d_flags);
    if (st_) hipStreamDestroy(st_);
    if (rc != DCN_OK || read_failed) {
        if (idx && idx->d_slots) hipFree(idx->d_slots);
        delete idx;
        return rc;
    }
    idx->n_keys = h_new;
    idx->has_zero = h_flags[0] != 0;
    *out = idx;
    *handled = true;
    mark(4);
    if (timing)
        fprintf(stderr, "load timing: runtime up %.3f s, table of %.1f GB allocated + cleared %.3f, staging buffers %.3f, %.2f GB copied / "
                        "decoded / inserted %.3f, buffers released %.3f\n",
                t_marks[0], (double)idx->n_groups * 16 / 1e9, t_marks[1] - t_marks[0], t_marks[2] - t_marks[1], 9.0 * count / 1e9,
                t_marks[3] - t_marks[2], t_marks[4] - t_marks[3]);
    return DCN_OK;
}

namespace {
// devices of this process that have a live context: the host pool is sized by them (HostPool::ensure_devices)
std::mutex g_ctx_devices_mu;
int g_ctx_per_device[64] = {0};
void note_ctx_device(int device, int delta) {
    int n_devices = 0;
    {
        std::lock_guard<std::mutex> g(g_ctx_devices_mu);
        if (device >= 0 && device < 64) g_ctx_per_device[device] += delta;
        for (int d = 0; d < 64; ++d) n_devices += g_ctx_per_device[d] > 0;
    }
    if (delta > 0) HostPool::get().ensure_devices(n_devices);
}
} // namespace

extern "C" int dcn_ctx_create(const dcn_index *index, uint64_t max_batch_bases, uint32_t max_batch_reads,
                              dcn_ctx **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!index) return dcn_fail(DCN_ERR_ARG, "index is NULL");
    if (max_batch_bases == 0 || max_batch_reads == 0) return dcn_fail(DCN_ERR_ARG, "batch limits must be > 0");
    if (max_batch_reads > 0xFFFFFF00u) return dcn_fail(DCN_ERR_ARG, "max_batch_reads too large");
    dcn_ctx *c = new (std::nothrow) dcn_ctx();
    if (!c) return dcn_fail(DCN_ERR_NOMEM, "host allocation failed");
    c->index = index;
    c->device = index->device;
    c->max_bases = max_batch_bases;
    c->max_reads = max_batch_reads;
    if (const char *tw = getenv("DCN_TILE_WINDOWS")) {
        long v = strtol(tw, nullptr, 10);
        if (v >= 16 && v <= (long)DCN_MAX_TILE_WINDOWS) c->tile_windows = (uint32_t)v;
    }
    uint64_t mt = (uint64_t)max_batch_reads + max_batch_bases / c->tile_windows + 1;
    if (mt > 0xFFFFFF00ull) {
        delete c;
        return dcn_fail(DCN_ERR_ARG, "batch limits imply more than 2^32 tiles");
    }
    c->max_tiles = (uint32_t)mt;
    c->chunk_bases = 64ull << 20;
    if (const char *cb = getenv("DCN_CHUNK_BASES")) {
        long long v = atoll(cb);
        if (v >= 1024) c->chunk_bases = (uint64_t)v;
    }
    int rc = DCN_OK;
    auto fail = [&](int code) {
        free_ctx(c);
        return code;
    };
    if (hipSetDevice(c->device) != hipSuccess) return fail(dcn_fail(DCN_ERR_HIP, "hipSetDevice failed"));
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&c->d2h_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < dcn_ctx::N_STAGE; ++i)
        ok = hipEventCreateWithFlags(&c->stage_free[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < dcn_ctx::N_EV; ++i)
        ok = hipEventCreateWithFlags(&c->ev_h2d[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_comp[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) return fail(dcn_fail(DCN_ERR_HIP, "stream/event creation failed"));
    // Runs of the record array: one slot per four windows (2 B per base of the batch instead of 8).  A unit would need a hit
    // in more than every fourth window of a wave to fill its run -- real sequence has a minimizer in every eighth -- and a
    // batch that does (w = 1, say) is run again with one slot per window (grow_run_slots).  DCN_REC_SHIFT = 0..3 fixes it.
    c->rec_shift = 2;
    // ... unless the index's window makes that likely from the start: the density of minimizers is 2 / (w + 1), and low-
    // complexity sequence ties its way to a hit in every window or two (leftmost / rightmost alternate); from w <= 7 on
    // (2 / (w + 1) >= 1/4) the context starts with one slot per window instead of finding out in mid-run
    if (index->w <= 7) c->rec_shift = 0;
    if (const char *rs = getenv("DCN_REC_SHIFT")) c->rec_shift = (uint32_t)std::min(3, std::max(0, atoi(rs)));
    uint64_t MR = max_batch_reads;
    // DCN_CTX_SLAB=1 (experiment, profiles/placement_order.py): the fixed-size buffers below come out of ONE allocation,
    // each on a 2 MB boundary, instead of 24 separate ones
    const bool slab = getenv("DCN_CTX_SLAB") != nullptr;
    uint64_t slab_off = 0;
    for (int pass = slab ? 0 : 1; pass < 2; ++pass) {
        if (slab && pass == 1) {
            c->slab_bytes = slab_off;
            if (hipMalloc((void **)&c->d_slab, c->slab_bytes) != hipSuccess) {
                c->d_slab = nullptr;
                c->slab_bytes = 0;
                return fail(dcn_fail(DCN_ERR_NOMEM, "hipMalloc of the context slab failed"));
            }
            slab_off = 0;
        }
#define A(ptr, count, what)                                                                                   \
    if (slab) {                                                                                               \
        if (pass == 1) c->ptr = reinterpret_cast<decltype(c->ptr)>(c->d_slab + slab_off);                     \
        slab_off += (std::max<uint64_t>((count), 1) * sizeof(*c->ptr) + (2u << 20) - 1) / (2u << 20) * (2u << 20); \
    } else if ((rc = dev_alloc(&c->ptr, (count), what)) != DCN_OK)                                            \
        return fail(rc)
    A(d_ascii, max_batch_bases + 64, "ascii");
    A(d_offsets, MR + 1, "offsets");
    A(d_unit_id, MR, "unit_id");
    A(d_packed, packed_words(max_batch_bases), "packed");
    A(d_invmask, mask_words(max_batch_bases), "invmask");
    A(d_read_tiles, MR, "read_tiles");
    A(d_read_tile_first, MR + 1, "read_tile_first");
    A(d_unit_first_read, MR + 1, "unit_first_read");
    A(d_unit_tile_first, MR + 1, "unit_tile_first");
    A(d_unit_tile_count, MR + 1, "unit_tile_count");
    A(d_tiles, mt, "tiles");
    A(d_keep, MR, "keep");
    A(d_unit_state, MR, "unit_state");
    A(d_hits, MR, "hits");
    A(d_total, MR, "total");
    A(d_unit_scratch, MR * 4, "unit_scratch");
    A(d_caps, MR, "caps");
    A(d_set_off, MR + 1, "set_off");
    A(d_tile_hits, mt, "tile_hits");
    A(d_pending, MR, "pending");
    A(d_big, MR + mt / 64 + 1, "big");
    A(d_rec_hash, (max_batch_bases >> c->rec_shift) + 256, "rec_hash");
    A(d_status, 1, "status");
    A(d_report, 1, "report");
#undef A
    }
    // global sets of the distinct pass (only units with more hits than its LDS set holds use them): sized for the
    // expected long-read density, grown on demand by the host API / dcn_ctx_reserve_records
    uint64_t recs = std::min<uint64_t>(std::max<uint64_t>(max_batch_bases / 16, 1u << 16), 1ull << 29);
    if (const char *rc_env = getenv("DCN_RECORD_CAPACITY")) recs = std::min<uint64_t>(std::max<uint64_t>(strtoull(rc_env, nullptr, 10), 64), 1ull << 29); // tests: force the growth path
    if ((rc = alloc_records(c, recs)) != DCN_OK) return fail(rc);
    c->stage_bytes = std::min<uint64_t>(std::max<uint64_t>(max_batch_bases + 64, 4096), 32ull << 20);
    if (const char *sb = getenv("DCN_STAGE_BYTES")) // tests: small staging buffers, so that one chunk needs many pieces
        c->stage_bytes = std::min<uint64_t>(std::max<uint64_t>(strtoull(sb, nullptr, 10), 4096), 32ull << 20);
    for (int i = 0; i < dcn_ctx::N_STAGE; ++i)
        if (hipHostMalloc((void **)&c->h_stage[i], c->stage_bytes, hipHostMallocDefault) != hipSuccess)
            return fail(dcn_fail(DCN_ERR_NOMEM, "pinned staging allocation failed"));
    if (hipHostMalloc((void **)&c->h_report, sizeof(dcn_batch_report), hipHostMallocDefault) != hipSuccess)
        return fail(dcn_fail(DCN_ERR_NOMEM, "pinned status allocation failed"));
    // zero padding in front of / behind the packed stream is written once; pack only touches the middle
    if (hipMemset(c->d_packed, 0, packed_words(max_batch_bases) * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(c->d_invmask, 0, mask_words(max_batch_bases) * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(c->d_status, 0, sizeof(dcn_status)) != hipSuccess ||
        hipMemset(c->d_unit_scratch, 0, (uint64_t)max_batch_reads * 4 * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(c->d_report, 0, sizeof(dcn_batch_report)) != hipSuccess ||
        // hipMemset runs on the null stream and does not wait for the host; the context's streams are non-blocking
        // and do not wait for the null stream: without this, the first batch's copies into the packed stream can
        // be overtaken by the memset above (seen as an all-'A' first chunk, once in a few hundred runs)
        hipDeviceSynchronize() != hipSuccess)
        return fail(dcn_fail(DCN_ERR_HIP, "hipMemset failed"));
    note_ctx_device(c->device, +1);
    *out = c;
    return DCN_OK;
}

extern "C" void dcn_ctx_destroy(dcn_ctx *ctx) {
    if (ctx) note_ctx_device(ctx->device, -1);
    free_ctx(ctx);
}

extern "C" void *dcn_ctx_stream(dcn_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int dcn_ctx_reserve_records(dcn_ctx *ctx, uint64_t n_records) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    if (slots_busy(ctx)) return dcn_fail(DCN_ERR_ARG, "host batches are in flight: wait for them first");
    DCN_HIP(hipSetDevice(ctx->device));
    DCN_HIP(hipStreamSynchronize(ctx->stream));
    if (n_records <= ctx->rec_capacity) return DCN_OK;
    return alloc_records(ctx, n_records);
}

extern "C" int dcn_ctx_synchronize(dcn_ctx *ctx) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    DCN_HIP(hipSetDevice(ctx->device));
    return sync_and_check(ctx, nullptr);
}

extern "C" int dcn_filter_batch_device(dcn_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets,
                                       const uint32_t *d_unit_id, uint32_t n_reads, uint64_t n_bases,
                                       uint32_t n_units, const dcn_params *params, uint8_t *d_keep, uint32_t *d_hits,
                                       uint32_t *d_total) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    DCN_TRY(check_params(params));
    if (n_reads == 0) return DCN_OK;
    if (!d_bases || !d_offsets || !d_keep) return dcn_fail(DCN_ERR_ARG, "d_bases/d_offsets/d_keep is NULL");
    if (n_reads > ctx->max_reads) return dcn_fail(DCN_ERR_CAPACITY, "n_reads exceeds the context's max_batch_reads");
    if (n_bases > ctx->max_bases) return dcn_fail(DCN_ERR_CAPACITY, "n_bases exceeds the context's max_batch_bases");
    if (n_units == 0 || n_units > n_reads || (!d_unit_id && n_units != n_reads))
        return dcn_fail(DCN_ERR_ARG, "n_units inconsistent with n_reads / d_unit_id");
    // the packed stream of slot 0 is this path's pack target and a host batch's copy target
    if (ctx->slots[0].busy) return dcn_fail(DCN_ERR_ARG, "a host batch is in flight on this context: wait for it first");
    DCN_HIP(hipSetDevice(ctx->device));
    BatchView v;
    v.d_ascii = d_bases;
    v.d_packed = ctx->d_packed;
    v.d_invmask = ctx->d_invmask;
    v.d_offsets = d_offsets;
    v.d_unit_id = d_unit_id;
    v.n_reads = n_reads;
    v.n_units = n_units;
    v.b0 = 0;
    v.b1 = n_bases;
    v.stream_bases = n_bases;
    v.d_keep = d_keep;
    v.d_hits = d_hits;
    v.d_total = d_total;
    v.d_report = ctx->d_report;
    DCN_TRY(enqueue_batch(ctx, v, params, /*pack_ahead=*/true));
    ctx->batch_pending = true;
    return DCN_OK;
}

// ----------------------------------------------------------------------------------------------------
// host batches: submit / wait
// ----------------------------------------------------------------------------------------------------
bool dcn_host_pack_is_wide();
bool dcn_host_pack_groups(const uint8_t *ascii, uint64_t n_bases, uint64_t g0, uint64_t g1, uint32_t *packed,
                          uint32_t *mask); // host_pack.cpp; true: a '\n' byte was seen

namespace {

enum class Transport {
    AsciiDirect, // page-locked ASCII: DMA as it is, pack on the device
    AsciiStaged, // pageable ASCII copied into the pinned ring, pack on the device
    HostPacked,  // pageable ASCII packed by the host threads INTO the pinned ring: 0.375 B/bp on the link
    Packed,      // the caller hands over the 2-bit stream + mask
};

struct HostInput {
    const uint8_t *bases = nullptr;    // ASCII, or null
    const uint32_t *packed = nullptr;  // caller-packed stream (Transport::Packed)
    const uint32_t *invmask = nullptr;
    const uint64_t *offsets = nullptr;
    const uint32_t *unit_id = nullptr;
    uint32_t n_reads = 0;
};

int alloc_slot_impl(dcn_ctx *c, int si);

// a slot is either complete or empty: a failure half-way (slot 1 duplicates every max_bases-sized buffer, so it is the
// allocation most likely to fail) releases what it got, and the next submit starts from null pointers again
int alloc_slot(dcn_ctx *c, int si) {
    if (c->slots[si].allocated) return DCN_OK;
    const int rc = alloc_slot_impl(c, si);
    if (rc != DCN_OK) free_slot_buffers(c->slots[si]);
    return rc;
}

int alloc_slot_impl(dcn_ctx *c, int si) {
    dcn_slot &sl = c->slots[si];
    const uint64_t MR = c->max_reads;
    if (si == 0) { // the context's own buffers
        sl.d_ascii = c->d_ascii;
        sl.d_packed = c->d_packed;
        sl.d_invmask = c->d_invmask;
        sl.d_offsets = c->d_offsets;
        sl.d_unit_id = c->d_unit_id;
        sl.d_keep = c->d_keep;
        sl.d_hits = c->d_hits;
        sl.d_total = c->d_total;
        sl.owns_buffers = false;
    } else {
        sl.owns_buffers = true;
        DCN_TRY(dev_alloc(&sl.d_ascii, c->max_bases + 64, "slot ascii"));
        DCN_TRY(dev_alloc(&sl.d_packed, packed_words(c->max_bases), "slot packed"));
        DCN_TRY(dev_alloc(&sl.d_invmask, mask_words(c->max_bases), "slot invmask"));
        DCN_TRY(dev_alloc(&sl.d_offsets, MR + 1, "slot offsets"));
        DCN_TRY(dev_alloc(&sl.d_unit_id, MR, "slot unit_id"));
        DCN_TRY(dev_alloc(&sl.d_keep, MR, "slot keep"));
        DCN_TRY(dev_alloc(&sl.d_hits, MR, "slot hits"));
        DCN_TRY(dev_alloc(&sl.d_total, MR, "slot total"));
        DCN_HIP(hipMemset(sl.d_packed, 0, packed_words(c->max_bases) * sizeof(uint32_t)));
        DCN_HIP(hipMemset(sl.d_invmask, 0, mask_words(c->max_bases) * sizeof(uint32_t)));
        DCN_HIP(hipDeviceSynchronize()); // the null-stream memsets must not overtake this slot's first copies
    }
    DCN_TRY(dev_alloc(&sl.d_report, 1, "slot report"));
    DCN_TRY(dev_alloc(&sl.d_off32, MR + 1, "slot offsets (u32)"));
    sl.mask_pairs_cap = std::max<uint64_t>(4096, mask_words(c->max_bases) / 16);
    if (const char *e = getenv("DCN_SPARSE_MASK_CAP")) sl.mask_pairs_cap = std::max<uint64_t>(1, strtoull(e, nullptr, 10)); // (tests: force the whole-mask path)
    DCN_TRY(dev_alloc(&sl.d_mask_pairs, sl.mask_pairs_cap, "slot mask pairs"));
    DCN_HIP(hipHostMalloc((void **)&sl.h_report, sizeof(dcn_batch_report), hipHostMallocDefault));
    DCN_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    sl.allocated = true;
    return DCN_OK;
}

template <typename T>
int ensure_pinned(T **p, uint64_t count) {
    if (*p) return DCN_OK;
    hipError_t e = hipHostMalloc((void **)p, std::max<uint64_t>(count, 1) * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) {
        *p = nullptr;
        return dcn_fail(DCN_ERR_NOMEM, std::string("pinned result staging: ") + hipGetErrorString(e));
    }
    return DCN_OK;
}

// End of the chunk that starts at read r0: the first unit boundary at which the chunk holds its target number of
// bases (or the end of the batch).  (Tapering the chunks towards the end of the batch, so that less kernel time is
// left uncovered behind the last copy, was measured slower: 107 vs 114 Gbp/s packed -- every extra chunk costs more
// in copy commands and launches than the shorter tail gives back.)  Found by bisection on offsets that have NOT been
// validated yet (any answer in (r0, n_reads] is safe; validate_chunk runs while the
// chunk's payload is already on its way).
uint32_t find_cut(const dcn_ctx *c, const HostInput &in, uint32_t r0, uint64_t chunk_bases) {
    const uint64_t *off = in.offsets;
    const uint64_t b0 = off[r0];
    const uint64_t target = b0 + chunk_bases;
    uint32_t lo = r0 + 1, hi = in.n_reads; // smallest r in [lo, hi] with off[r] >= target, else n_reads
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (off[mid] >= target) hi = mid;
        else lo = mid + 1;
    }
    uint32_t r = lo;
    if (in.unit_id)
        while (r < in.n_reads && in.unit_id[r] == in.unit_id[r - 1]) ++r; // mates stay together
    return r;
}

// offsets / unit ids of reads [r0, r1): the checks of the ABI's contract, and the chunk's longest read
__global__ __launch_bounds__(256) void widen_offsets_kernel(const uint32_t *__restrict__ in32, uint64_t *__restrict__ out64, uint32_t n) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out64[i] = in32[i];
}

__global__ __launch_bounds__(256) void scatter_mask_kernel(const uint2 *__restrict__ pairs, uint32_t n, uint32_t *__restrict__ invmask) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) invmask[pairs[i].x] = pairs[i].y;
}

// The non-zero words of mask[0, m) (group g_base + i of the stream) appended to the slot's page-locked pair buffer, found by
// the pool.  false: sparse form off, or no room left -- the caller sends the words whole.
bool sparse_mask_pairs(dcn_slot &sl, const uint32_t *mask, uint64_t g_base, uint64_t m, dcn_mask_range *out) {
    if (sl.lean) return false; // (a small batch: the mask words themselves are one short copy, the pairs a memset and a kernel more)
    if (getenv("DCN_NO_SPARSE_MASK") || g_base + m > 0xFFFFFFFFull) return false; // (read per chunk: tests switch it)
    if (ensure_pinned(&sl.h_mask_pairs, sl.mask_pairs_cap) != DCN_OK) return false;
    std::vector<std::vector<uint2>> found((size_t)std::max(1, HostPool::get().width()));
    HostPool::get().run([&](int i, int nt) {
        const uint64_t per = (m + nt - 1) / nt, lo = std::min<uint64_t>(m, per * i), hi = std::min<uint64_t>(m, lo + per);
        std::vector<uint2> &v = found[(size_t)i];
        uint64_t j = lo;
        for (; j + 8 <= hi; j += 8) { // (the OR of eight words first: zero nearly always)
            const uint32_t *q = mask + j;
            if ((q[0] | q[1] | q[2] | q[3] | q[4] | q[5] | q[6] | q[7]) == 0) continue;
            for (int t = 0; t < 8; ++t)
                if (q[t]) v.push_back(make_uint2((uint32_t)(g_base + j + t), q[t]));
        }
        for (; j < hi; ++j)
            if (mask[j]) v.push_back(make_uint2((uint32_t)(g_base + j), mask[j]));
    }, m < (1u << 16));
    uint64_t total = 0;
    for (const auto &v : found) total += v.size();
    if (sl.mask_pairs_used + total > sl.mask_pairs_cap) return false;
    out->g0 = g_base;
    out->g1 = g_base + m;
    out->pairs_off = sl.mask_pairs_used;
    out->n = (uint32_t)total;
    for (const auto &v : found) {
        if (!v.empty()) memcpy(sl.h_mask_pairs + sl.mask_pairs_used, v.data(), v.size() * sizeof(uint2));
        sl.mask_pairs_used += v.size();
    }
    return true;
}

// off32_out (may be null): the chunk's offsets [r0, r1] narrowed to u32, written to off32_out[r0 .. r1]
int validate_chunk(const HostInput &in, uint32_t r0, uint32_t r1, uint64_t n_bases_total, uint64_t *max_len_out,
                   uint32_t *off32_out = nullptr) {
    // One pass over the chunk's offsets (and unit ids) on the host threads: at 10 M reads per batch the plain loop cost the
    // submitting thread 4-5 ms of a 12 ms call, next to the pack it also waits for when the bases are pageable.
    const uint64_t *off = in.offsets;
    const uint32_t *uid = in.unit_id;
    const uint32_t n = r1 - r0, n_reads = in.n_reads;
    std::atomic<uint32_t> bad{0};
    std::atomic<uint64_t> max_len{0};
    static const bool serial = getenv("DCN_SERIAL_VALIDATE") != nullptr; // (A/B: the submitting thread alone)
    HostPool::get().run([&](int i, int nt) {
        const uint32_t per = (n + (uint32_t)nt - 1) / (uint32_t)nt;
        const uint32_t a = r0 + std::min<uint64_t>(n, (uint64_t)per * (uint32_t)i), b = r0 + std::min<uint64_t>(n, (uint64_t)per * ((uint32_t)i + 1));
        uint64_t m = 0;
        uint32_t e = 0;
        for (uint32_t r = a; r < b; ++r) { // (no early exit: the loop vectorises)
            const uint64_t lo = off[r], hi = off[r + 1];
            e |= (uint32_t)(hi < lo) | (uint32_t)(hi > n_bases_total);
            m = std::max(m, hi - lo);
        }
        if (off32_out) {
            for (uint32_t r = a; r < b; ++r) off32_out[r] = (uint32_t)off[r];
            if (b == r1) off32_out[r1] = (uint32_t)off[r1]; // (every slice that ends at r1 writes the same value)
        }
        if (uid)
            for (uint32_t r = a + 1; r <= b && r < n_reads; ++r) e |= (uid[r] != uid[r - 1] && uid[r] != uid[r - 1] + 1) ? 2u : 0u;
        if (e) bad.fetch_or(e);
        uint64_t cur = max_len.load();
        while (m > cur && !max_len.compare_exchange_weak(cur, m)) {
        }
    }, n < (1u << 16) || serial);
    const uint32_t e = bad.load();
    if (e & 1u) return dcn_fail(DCN_ERR_ARG, "offsets must be non-decreasing");
    if (max_len.load() > 0xFFFFFFF0ull) return dcn_fail(DCN_ERR_ARG, "read longer than 2^32 bases");
    if (e & 2u) return dcn_fail(DCN_ERR_ARG, "unit_id must stay equal or grow by one");
    *max_len_out = max_len.load();
    return DCN_OK;
}

int chunk_events(dcn_slot &sl, size_t n) {
    while (sl.ev_h2d.size() < n) {
        hipEvent_t a = nullptr, b = nullptr;
        DCN_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        sl.ev_h2d.push_back(a);
        DCN_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        sl.ev_comp.push_back(b);
    }
    return DCN_OK;
}

// kernels + result copies of one chunk (its inputs are on the device, or on their way on the copy stream)
// n_known: the batch's chunks are all in sl.chunks (false while a host-bound submission is still cutting them: the
// decisions then go back in one copy behind the last chunk)
int enqueue_chunk(dcn_ctx *c, dcn_slot &sl, size_t ci, bool wait_h2d, bool n_known = true, bool is_last = false) {
    const dcn_chunk &ch = sl.chunks[ci];
    if (wait_h2d && !c->lean) DCN_HIP(hipStreamWaitEvent(c->stream, sl.ev_h2d[ci], 0)); // (lean: the copies are on this stream)
    if (sl.off32) {
        const uint32_t n = ch.r1 - ch.r0 + 1;
        hipLaunchKernelGGL(widen_offsets_kernel, dim3(std::min<uint32_t>((n + 255) / 256, 1024)), dim3(256), 0, c->stream,
                           sl.d_off32 + ch.r0, sl.d_offsets + ch.r0, n);
        DCN_HIP(hipGetLastError());
    }
    for (const dcn_mask_range &mr : ch.mask_ranges) { // mask words that crossed the link as their non-zero ones only
        DCN_HIP(hipMemsetAsync(sl.d_invmask + DCN_FRONT_PAD + mr.g0, 0, (mr.g1 - mr.g0) * sizeof(uint32_t), c->stream));
        if (mr.n) {
            hipLaunchKernelGGL(scatter_mask_kernel, dim3(std::min<uint32_t>((mr.n + 255) / 256, 1024)), dim3(256), 0, c->stream,
                               sl.d_mask_pairs + mr.pairs_off, mr.n, sl.d_invmask + DCN_FRONT_PAD);
            DCN_HIP(hipGetLastError());
        }
    }
    BatchView v;
    v.d_ascii = sl.device_pack ? sl.d_ascii : nullptr;
    v.d_packed = sl.d_packed;
    v.d_invmask = sl.d_invmask;
    v.d_offsets = sl.d_offsets + ch.r0;
    v.d_unit_id = sl.has_units ? sl.d_unit_id + ch.r0 : nullptr;
    v.unit_base = ch.u0;
    v.n_reads = ch.r1 - ch.r0;
    v.n_units = ch.u1 - ch.u0;
    // ASCII chunks are copied and packed in whole 32-base groups (see submit_impl)
    v.b0 = ch.b0 / 32 * 32;
    v.b1 = std::min<uint64_t>((ch.b1 + 31) / 32 * 32, sl.n_bases);
    v.stream_bases = sl.n_bases;
    v.d_keep = sl.d_keep + ch.u0;
    v.d_hits = sl.counts ? sl.d_hits + ch.u0 : nullptr;
    v.d_total = sl.counts ? sl.d_total + ch.u0 : nullptr;
    v.d_report = sl.d_report;
    DCN_TRY(enqueue_batch(c, v, &sl.params));
    // Results travel back per chunk when hit counts were asked for (8 bytes per unit: worth overlapping).  When only the
    // decisions are (1 byte per unit) a copy per chunk is three runtime calls per chunk for nothing: they go back in one
    // copy behind the last chunk -- or, for a batch of many chunks, in two: everything up to the last chunk but one while
    // the last chunk is still on the link, and the last chunk's own (a 10 M-read call otherwise ends with 10 MB crossing
    // the link back after everything else is done: 0.2 ms of its 12.4 ms).
    const size_t n_ch = sl.chunks.size();
    const bool last = n_known ? ci + 1 == n_ch : is_last;
    const bool split = n_known && !sl.counts && n_ch >= 4, early = split && ci + 2 == n_ch;
    if (!sl.counts && !last && !early) return DCN_OK;
    if (!c->lean) { // (lean: d2h_stream IS the compute stream for this submission)
        DCN_HIP(hipEventRecord(sl.ev_comp[ci], c->stream));
        DCN_HIP(hipStreamWaitEvent(c->d2h_stream, sl.ev_comp[ci], 0));
    }
    const uint32_t k0 = sl.counts ? ch.u0 : (split && last ? sl.chunks[n_ch - 2].u1 : 0u);
    const uint32_t nu = ch.u1 - k0;
    DCN_HIP(hipMemcpyAsync((sl.keep_direct ? sl.u_keep : sl.h_keep) + k0, sl.d_keep + k0, nu, hipMemcpyDeviceToHost,
                           c->d2h_stream));
    if (sl.u_hits)
        DCN_HIP(hipMemcpyAsync((sl.hits_direct ? sl.u_hits : sl.h_hits) + ch.u0, sl.d_hits + ch.u0,
                               (uint64_t)nu * sizeof(uint32_t), hipMemcpyDeviceToHost, c->d2h_stream));
    if (sl.u_total)
        DCN_HIP(hipMemcpyAsync((sl.total_direct ? sl.u_total : sl.h_total) + ch.u0, sl.d_total + ch.u0,
                               (uint64_t)nu * sizeof(uint32_t), hipMemcpyDeviceToHost, c->d2h_stream));
    return DCN_OK;
}

int finish_submission(dcn_ctx *c, dcn_slot &sl) {
    // the report is written on the compute stream (cleared at submission, filled by the finish kernels): the copy must
    // come behind all of it, also for a batch without any chunk
    if (!c->lean) {
        const int e = c->ev_next;
        c->ev_next = (e + 1) % dcn_ctx::N_EV;
        DCN_HIP(hipEventRecord(c->ev_comp[e], c->stream));
        DCN_HIP(hipStreamWaitEvent(c->d2h_stream, c->ev_comp[e], 0));
    }
    DCN_HIP(hipMemcpyAsync(sl.h_report, sl.d_report, sizeof(dcn_batch_report), hipMemcpyDeviceToHost, c->d2h_stream));
    DCN_HIP(hipEventRecord(sl.done, c->d2h_stream));
    return DCN_OK;
}

// after a failure in the middle of a submission: nothing of this context may still be running when the caller's
// buffers go away
void drain(dcn_ctx *c) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->d2h_stream);
}

int submit_impl(dcn_ctx *c, const HostInput &in, const dcn_params *params, uint8_t *keep, uint32_t *hits,
                uint32_t *total, uint64_t *ticket) {
    if (!c) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    if (!ticket) return dcn_fail(DCN_ERR_ARG, "ticket is NULL");
    *ticket = 0;
    DCN_TRY(check_params(params));
    if (in.n_reads > 0 && (!in.offsets || !keep)) return dcn_fail(DCN_ERR_ARG, "offsets/keep is NULL");
    if (in.n_reads > c->max_reads) return dcn_fail(DCN_ERR_CAPACITY, "n_reads exceeds the context's max_batch_reads");
    if (c->batch_pending) return dcn_fail(DCN_ERR_ARG, "device-pointer batches are pending: dcn_ctx_synchronize first");
    int si = -1;
    for (int i = 0; i < dcn_ctx::N_SLOTS && si < 0; ++i)
        if (!c->slots[i].busy) si = i;
    if (si < 0) return dcn_fail(DCN_ERR_CAPACITY, "two batches are already in flight: dcn_filter_batch_wait first");
    DCN_HIP(hipSetDevice(c->device));
    DCN_TRY(alloc_slot(c, si));
    dcn_slot &sl = c->slots[si];
    const uint32_t n_reads = in.n_reads;
    uint64_t n_bases = 0;
    uint32_t n_units = 0;
    if (n_reads) {
        if (in.offsets[0] != 0) return dcn_fail(DCN_ERR_ARG, "offsets[0] must be 0");
        n_bases = in.offsets[n_reads];
        if (n_bases > c->max_bases) return dcn_fail(DCN_ERR_CAPACITY, "batch exceeds the context's max_batch_bases");
        if (in.unit_id && in.unit_id[0] != 0) return dcn_fail(DCN_ERR_ARG, "unit_id[0] must be 0");
        const bool packed_in = in.packed != nullptr;
        if (n_bases > 0 && !packed_in && !in.bases) return dcn_fail(DCN_ERR_ARG, "bases is NULL");
        if (n_bases > 0 && packed_in && !in.invmask) return dcn_fail(DCN_ERR_ARG, "invmask is NULL");
    }
    const bool host_pack_ok = !getenv("DCN_NO_HOST_PACK"); // read per call: tests switch it
    // Page-locked ASCII can cross the link as it is (1 byte per base, no host work: 52-53 Gbp/s) or be packed by the host
    // threads like pageable ASCII (0.375 bytes per base: 90-110 Gbp/s where they pack with AVX-512, csrc/host_pack.cpp).
    // The faster one is taken; DCN_PINNED_ASCII_DMA=1 keeps the host out of it.
    const bool bases_pinned = is_pinned_host(in.bases);
    const bool pinned_dma = bases_pinned && (!host_pack_ok || !dcn_host_pack_is_wide() || getenv("DCN_PINNED_ASCII_DMA"));
    Transport tr;
    if (in.packed) tr = Transport::Packed;
    else if (pinned_dma) tr = Transport::AsciiDirect;
    else tr = host_pack_ok ? Transport::HostPacked : Transport::AsciiStaged;

    // A SMALL batch is some twenty GPU commands -- copies, memsets, a handful of kernels, event records and waits between three
    // streams -- whatever it carries, and those, not its kernels, are what it costs (130 us for 1,024 reads as for 16,384), and
    // what several contexts calling at once queue up behind (eight threads: 1.2 Gbp/s at 1,024 reads per call, 17 at 16,384;
    // profiles/r04_small_calls.txt).  Up to DCN_LEAN_MAX_BASES (16 Mbp; 0: never) a batch is submitted in its plain form on ONE
    // stream: copies, kernels and result copies in order on `stream` (no events between streams; what a batch of this size
    // could overlap inside itself is tens of microseconds), 64-bit offsets as they are (no narrowing and widening kernel), the
    // mask words themselves (no pairs, memset and scatter kernel).  Same box: 1,024 reads per call 1.2 -> 1.5 Gbp/s from one
    // thread and 1.2 -> 4.0 from eight, 16,384: 15 -> 18 and 17 -> 45, 65,536: 37 -> 40 and 52 -> 79; at 262,144 (39 Mbp) the
    // three-stream form wins again (98 against 67 from two threads), hence the limit.
    const char *lean_env = getenv("DCN_LEAN_MAX_BASES"); // (read per call: tests run both forms in one process)
    const uint64_t lean_max_bases = lean_env ? strtoull(lean_env, nullptr, 10) : (16ull << 20);
    struct LeanScope {
        dcn_ctx *c;
        hipStream_t copy, d2h;
        bool on;
        ~LeanScope() {
            if (!on) return;
            c->copy_stream = copy;
            c->d2h_stream = d2h;
            c->lean = false;
        }
    } lean_scope{c, c->copy_stream, c->d2h_stream, n_reads > 0 && n_bases <= lean_max_bases && n_bases <= c->chunk_bases};
    if (lean_scope.on) {
        c->copy_stream = c->stream;
        c->d2h_stream = c->stream;
        c->lean = true;
    }
    sl.lean = lean_scope.on;
    sl.params = *params;
    sl.counts = hits || total;
    sl.has_units = in.unit_id != nullptr;
    sl.n_reads = n_reads;
    sl.n_bases = n_bases;
    sl.u_keep = keep;
    sl.u_hits = hits;
    sl.u_total = total;
    sl.keep_direct = is_pinned_host(keep);
    sl.hits_direct = hits && is_pinned_host(hits);
    sl.total_direct = total && is_pinned_host(total);
    if (!sl.keep_direct) DCN_TRY(ensure_pinned(&sl.h_keep, c->max_reads));
    if (hits && !sl.hits_direct) DCN_TRY(ensure_pinned(&sl.h_hits, c->max_reads));
    if (total && !sl.total_direct) DCN_TRY(ensure_pinned(&sl.h_total, c->max_reads));
    const int off_pinned = is_pinned_host(in.offsets) ? 1 : 0, uid_pinned = is_pinned_host(in.unit_id) ? 1 : 0;
    static const bool no_off32 = getenv("DCN_NO_OFF32") != nullptr; // (A/B)
    sl.off32 = n_reads > 0 && n_bases < (1ull << 32) && !no_off32 && !sl.lean;
    if (sl.off32) DCN_TRY(ensure_pinned(&sl.h_off32, c->max_reads + 1));
    const int pk_pinned = in.packed ? ((is_pinned_host(in.packed) && is_pinned_host(in.invmask)) ? 1 : 0) : 0;

    // With another batch already in flight the kernels of this batch's last chunk are covered by the next batch's
    // copies, so nothing argues for small chunks any more, and every chunk costs the host ~0.2 ms of runtime calls:
    // twice the chunk size then (packed input, two in flight: 105 -> 120 Gbp/s when the host was the limit).
    static const uint64_t inflight_factor = getenv("DCN_INFLIGHT_CHUNK_FACTOR") ? strtoull(getenv("DCN_INFLIGHT_CHUNK_FACTOR"), nullptr, 10) : 2;
    const uint64_t chunk_bases = c->chunk_bases * (slots_busy(c) ? std::max<uint64_t>(inflight_factor, 1) : 1);
    // DCN_SUBMIT_TIMING=1: where the submitting thread's time goes, one line per call on stderr
    static const bool submit_timing = getenv("DCN_SUBMIT_TIMING") != nullptr;
    double tm[6] = {0, 0, 0, 0, 0, 0}; // stage wait, pack, copy calls, offsets / unit ids, validate, kernels
    const auto t_submit0 = std::chrono::steady_clock::now();
    auto lap = [&](int which, std::chrono::steady_clock::time_point &t) {
        if (!submit_timing) return;
        const auto now = std::chrono::steady_clock::now();
        tm[which] += std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
    };
    for (int attempt = 0;; ++attempt) {
        sl.device_pack = tr == Transport::AsciiDirect || tr == Transport::AsciiStaged;
        sl.chunks.clear();
        sl.mask_pairs_used = 0;
        DCN_HIP(hipMemsetAsync(sl.d_report, 0, sizeof(dcn_batch_report), c->stream));
        bool saw_newline = false;
        int rc = DCN_OK;
        if (n_reads && off_pinned && !sl.off32) rc = staged_h2d(c, sl.d_offsets, in.offsets, (uint64_t)(n_reads + 1) * sizeof(uint64_t), 1);
        if (rc == DCN_OK && n_reads && in.unit_id && uid_pinned)
            rc = staged_h2d(c, sl.d_unit_id, in.unit_id, (uint64_t)n_reads * sizeof(uint32_t), 1);
        uint32_t r0 = 0, u0 = 0;
        uint64_t groups_done = 0; // 32-base groups of the stream already sent (HostPacked / Packed)
        static const bool no_interleave = getenv("DCN_NO_INTERLEAVE") != nullptr, no_ride = getenv("DCN_NO_RIDE") != nullptr; // (A/B)
        const bool interleave = !no_interleave;
        while (r0 < n_reads && rc == DCN_OK) {
            dcn_chunk ch;
            ch.r0 = r0;
            ch.u0 = u0;
            ch.r1 = find_cut(c, in, r0, chunk_bases);
            ch.u1 = in.unit_id ? (ch.r1 == n_reads ? in.unit_id[n_reads - 1] + 1 : in.unit_id[ch.r1]) : ch.r1;
            ch.b0 = in.offsets[ch.r0];
            ch.b1 = in.offsets[ch.r1];
            if (ch.b1 < ch.b0 || ch.b1 > n_bases) {
                rc = dcn_fail(DCN_ERR_ARG, "offsets must be non-decreasing");
                break;
            }
            bool rode = false; // this chunk's pageable offsets / unit ids went with its packed piece
            auto copies = [&]() -> int {
                if (ch.b1 > ch.b0) {
                    if (sl.device_pack) {
                        // whole 32-base groups, so that the device pack of the groups two chunks share is right
                        // whichever of them runs last
                        const uint64_t a0 = ch.b0 / 32 * 32, a1 = std::min<uint64_t>((ch.b1 + 31) / 32 * 32, n_bases);
                        DCN_TRY(staged_h2d(c, sl.d_ascii + a0, in.bases + a0, a1 - a0, tr == Transport::AsciiDirect ? 1 : 0));
                    } else {
                        // groups not sent yet, up to the one holding this chunk's last base
                        const uint64_t g0 = groups_done, g1 = (ch.b1 + 31) / 32;
                        if (g1 > g0) {
                            uint32_t *dp = sl.d_packed + DCN_FRONT_PAD + 2 * g0, *dm = sl.d_invmask + DCN_FRONT_PAD + g0;
                            if (tr == Transport::Packed) {
                                DCN_TRY(staged_h2d(c, dp, in.packed + 2 * g0, (g1 - g0) * 8, pk_pinned));
                                dcn_mask_range mr;
                                if (sparse_mask_pairs(sl, in.invmask + g0, g0, g1 - g0, &mr)) {
                                    if (mr.n)
                                        DCN_HIP(hipMemcpyAsync(sl.d_mask_pairs + mr.pairs_off, sl.h_mask_pairs + mr.pairs_off, (uint64_t)mr.n * sizeof(uint2),
                                                               hipMemcpyHostToDevice, c->copy_stream));
                                    ch.mask_ranges.push_back(mr);
                                } else {
                                    DCN_TRY(staged_h2d(c, dm, in.invmask + g0, (g1 - g0) * 4, pk_pinned));
                                }
                            } else {
                                // pieces of whole groups: 8 bytes of stream + 4 of mask per group, side by side in a
                                // staging buffer, packed there by the host threads
                                const uint64_t per_piece = c->stage_bytes / 12 / 64 * 64;
                                // pageable offsets (and unit ids) of the chunk ride in the same staging buffer when they fit
                                // behind its one piece, copied by the threads that pack it: no ring slot, no job and no
                                // copy by the submitting thread of their own (3.5 MB per 64 Mbp chunk of 150 bp reads:
                                // 3.7 ms of a 10 M-read call)
                                const uint64_t n_off = sl.off32 ? 0 : (uint64_t)(ch.r1 - ch.r0 + 1) * sizeof(uint64_t); // (u32 offsets go by themselves, below)
                                const uint64_t n_uid = (in.unit_id && !uid_pinned) ? (uint64_t)(ch.r1 - ch.r0) * sizeof(uint32_t) : 0;
                                const bool ride = (!off_pinned || sl.off32) && (n_off || n_uid) && !no_ride && g1 - g0 <= per_piece &&
                                                  12 * (g1 - g0) + 16 + n_off + n_uid <= c->stage_bytes;
                                for (uint64_t g = g0; g < g1; g += per_piece) {
                                    const uint64_t m = std::min<uint64_t>(per_piece, g1 - g);
                                    const int which = c->stage_next;
                                    c->stage_next = (which + 1) % dcn_ctx::N_STAGE;
                                    auto tl = std::chrono::steady_clock::now();
                                    DCN_HIP(hipEventSynchronize(c->stage_free[which]));
                                    lap(0, tl);
                                    uint32_t *hp = (uint32_t *)c->h_stage[which], *hm = hp + 2 * m;
                                    uint8_t *ho = (uint8_t *)(((uintptr_t)(hm + m) + 7) & ~(uintptr_t)7), *hu = ho + n_off;
                                    std::atomic<bool> nl(false);
                                    HostPool::get().run([&](int i, int nt) {
                                        if (ride) {
                                            const uint64_t o0 = n_off * i / nt, o1 = n_off * (i + 1) / nt, q0 = n_uid * i / nt, q1 = n_uid * (i + 1) / nt;
                                            if (n_off) memcpy(ho + o0, (const uint8_t *)(in.offsets + ch.r0) + o0, o1 - o0);
                                            if (n_uid) memcpy(hu + q0, (const uint8_t *)(in.unit_id + ch.r0) + q0, q1 - q0);
                                        }
                                        const uint64_t per = (m + nt - 1) / nt, lo = std::min<uint64_t>(m, per * i),
                                                       hi = std::min<uint64_t>(m, lo + per);
                                        if (hi <= lo) return;
                                        // a '\n' anywhere means some read may end in one (src/filter_common.rs:229 strips
                                        // it): only the device path probes read ends, so the batch is sent again as ASCII
                                        if (dcn_host_pack_groups(in.bases, n_bases, g + lo, g + hi, hp + 2 * lo, hm + lo))
                                            nl.store(true);
                                    }, m < 4096);
                                    lap(1, tl);
                                    saw_newline = saw_newline || nl.load();
                                    DCN_HIP(hipMemcpyAsync(dp + 2 * (g - g0), hp, m * 8, hipMemcpyHostToDevice, c->copy_stream));
                                    dcn_mask_range mr;
                                    if (sparse_mask_pairs(sl, hm, g, m, &mr)) {
                                        if (mr.n)
                                            DCN_HIP(hipMemcpyAsync(sl.d_mask_pairs + mr.pairs_off, sl.h_mask_pairs + mr.pairs_off, (uint64_t)mr.n * sizeof(uint2),
                                                                   hipMemcpyHostToDevice, c->copy_stream));
                                        ch.mask_ranges.push_back(mr);
                                    } else {
                                        DCN_HIP(hipMemcpyAsync(dm + (g - g0), hm, m * 4, hipMemcpyHostToDevice, c->copy_stream));
                                    }
                                    if (ride) {
                                        if (n_off) DCN_HIP(hipMemcpyAsync(sl.d_offsets + ch.r0, ho, n_off, hipMemcpyHostToDevice, c->copy_stream));
                                        if (n_uid) DCN_HIP(hipMemcpyAsync(sl.d_unit_id + ch.r0, hu, n_uid, hipMemcpyHostToDevice, c->copy_stream));
                                        rode = true;
                                    }
                                    DCN_HIP(hipEventRecord(c->stage_free[which], c->copy_stream));
                                    lap(2, tl);
                                }
                            }
                            groups_done = g1;
                        }
                    }
                }
                // page-locked offsets / unit ids went over in one copy each before the first chunk (two runtime calls
                // less per chunk); pageable ones are staged chunk by chunk
                auto to = std::chrono::steady_clock::now();
                if (!off_pinned && !rode && !sl.off32)
                    DCN_TRY(staged_h2d(c, sl.d_offsets + ch.r0, in.offsets + ch.r0, (uint64_t)(ch.r1 - ch.r0 + 1) * sizeof(uint64_t), 0));
                if (in.unit_id && !uid_pinned && !rode)
                    DCN_TRY(staged_h2d(c, sl.d_unit_id + ch.r0, in.unit_id + ch.r0, (uint64_t)(ch.r1 - ch.r0) * sizeof(uint32_t), 0));
                lap(3, to);
                return DCN_OK;
            };
            if ((rc = copies()) != DCN_OK) break;
            if (saw_newline) break; // this attempt is abandoned
            // validated while the copies above are in flight; the kernels are only queued in the second pass
            auto tv = std::chrono::steady_clock::now();
            rc = validate_chunk(in, ch.r0, ch.r1, n_bases, &ch.max_len, sl.off32 ? sl.h_off32 : nullptr);
            lap(4, tv);
            if (rc != DCN_OK) break;
            if (sl.off32) { // narrowed by the check's own pass over them, into the slot's page-locked copy
                hipError_t he2 = hipMemcpyAsync(sl.d_off32 + ch.r0, sl.h_off32 + ch.r0, (uint64_t)(ch.r1 - ch.r0 + 1) * sizeof(uint32_t),
                                                hipMemcpyHostToDevice, c->copy_stream);
                if (he2 != hipSuccess) {
                    rc = dcn_fail(DCN_ERR_HIP, std::string("hipMemcpyAsync (offsets): ") + hipGetErrorString(he2));
                    break;
                }
            }
            if (ch.u1 < ch.u0 || (uint64_t)ch.u1 - ch.u0 > (uint64_t)ch.r1 - ch.r0) {
                rc = dcn_fail(DCN_ERR_ARG, "unit_id must stay equal or grow by one");
                break;
            }
            sl.chunks.push_back(ch);
            if ((rc = chunk_events(sl, sl.chunks.size())) != DCN_OK) break;
            hipError_t he = c->lean ? hipSuccess : hipEventRecord(sl.ev_h2d[sl.chunks.size() - 1], c->copy_stream);
            if (he != hipSuccess) {
                rc = dcn_fail(DCN_ERR_HIP, std::string("hipEventRecord: ") + hipGetErrorString(he));
                break;
            }
            r0 = ch.r1;
            u0 = ch.u1;
            if (interleave && sl.chunks.size() >= 2) { // the chunk before this one: its copies are queued, it is not the last
                auto tk = std::chrono::steady_clock::now();
                rc = enqueue_chunk(c, sl, sl.chunks.size() - 2, true, false, false);
                lap(5, tk);
            }
        }
        // The kernels of a chunk are queued as soon as the next chunk's copies are, one chunk behind: they then run under
        // the host's work on the chunks that follow (packing pageable bases: a blocking 10 M-read call 19.3 -> 15 ms; scanning a
        // caller's mask for its non-zero words: 10.5 -> 8.8 ms), not after it.  (Until late in round 3 all copies were queued
        // first and all kernels after them -- right while a chunk's runtime calls cost the host as much as its copy took on
        // the link; a chunk's copy is three times that now.  DCN_NO_INTERLEAVE=1 brings the two passes back.)
        auto tk = std::chrono::steady_clock::now();
        if (interleave) {
            if (rc == DCN_OK && !saw_newline && !sl.chunks.empty()) {
                const size_t n_ch = sl.chunks.size();
                if (!sl.counts && n_ch >= 4) { // the decisions of every chunk but the last go back while the last one's kernels run
                    hipError_t he = hipEventRecord(sl.ev_comp[n_ch - 2], c->stream);
                    if (he == hipSuccess) he = hipStreamWaitEvent(c->d2h_stream, sl.ev_comp[n_ch - 2], 0);
                    if (he == hipSuccess)
                        he = hipMemcpyAsync(sl.keep_direct ? sl.u_keep : sl.h_keep, sl.d_keep, sl.chunks[n_ch - 2].u1, hipMemcpyDeviceToHost, c->d2h_stream);
                    if (he != hipSuccess) rc = dcn_fail(DCN_ERR_HIP, std::string("early result copy: ") + hipGetErrorString(he));
                }
                if (rc == DCN_OK) rc = enqueue_chunk(c, sl, n_ch - 1, true); // (all chunks known now: it copies its own decisions only)
            }
        } else {
            for (size_t ci = 0; rc == DCN_OK && !saw_newline && ci < sl.chunks.size(); ++ci) rc = enqueue_chunk(c, sl, ci, true);
        }
        lap(5, tk);

        n_units = u0;
        if (rc == DCN_OK && saw_newline && attempt == 0) {
            drain(c);
            tr = bases_pinned ? Transport::AsciiDirect : Transport::AsciiStaged;
            continue;
        }
        if (rc != DCN_OK) {
            drain(c);
            return rc;
        }
        break;
    }
    sl.n_units = n_units;
    int rc = finish_submission(c, sl);
    if (rc != DCN_OK) {
        drain(c);
        return rc;
    }
    if (submit_timing)
        fprintf(stderr, "submit timing: %.2f ms, %zu chunks: stage wait %.2f, pack %.2f, copy calls %.2f, offsets / unit ids %.2f, validate %.2f, "
                        "kernels %.2f\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_submit0).count(),
                sl.chunks.size(), tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]);
    sl.busy = true;
    sl.ticket = c->next_ticket++;
    *ticket = sl.ticket;
    return DCN_OK;
}

int wait_impl(dcn_ctx *c, uint64_t ticket) {
    if (!c) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    dcn_slot *slp = nullptr;
    for (auto &s : c->slots)
        if (s.busy && s.ticket == ticket) slp = &s;
    if (!slp) return dcn_fail(DCN_ERR_ARG, "no batch with this ticket is in flight");
    dcn_slot &sl = *slp;
    DCN_HIP(hipSetDevice(c->device));
    auto fail = [&](int rc) {
        drain(c);
        sl.busy = false;
        return rc;
    };
    for (int attempt = 0;; ++attempt) {
        hipError_t e = hipEventSynchronize(sl.done);
        if (e != hipSuccess) return fail(dcn_fail(DCN_ERR_HIP, std::string("hipEventSynchronize: ") + hipGetErrorString(e)));
        if (sl.h_report->bounds) return fail(dcn_fail(DCN_ERR_INTERNAL, "scan kernel: index out of range in phase B (DCN_DEBUG_BOUNDS build)"));
        if (sl.h_report->bad_offsets) return fail(dcn_fail(DCN_ERR_INTERNAL, "the device met offsets the host had validated as decreasing or beyond the batch"));
        if (!sl.h_report->overflow) break;
        // Some chunk dropped hit records.  Grow the scratch and run the batch's kernels again: its inputs are still
        // resident in the slot.  Everything else in flight is drained first, since the scratch is shared.
        const uint64_t need = sl.h_report->need;
        if (attempt >= 4) return fail(overflow_error(c, need));
        drain(c);
        int rc = DCN_OK;
        if (sl.h_report->overflow & 2u) rc = grow_run_slots(c); // one slot per window from now on
        if (rc == DCN_OK && (sl.h_report->overflow & 1u)) {
            uint64_t want = std::max<uint64_t>(need + need / 8 + 1024, c->rec_capacity * 2);
            rc = alloc_records(c, std::min<uint64_t>(want, 1ull << 29));
        }
        if (rc != DCN_OK) return fail(rc);
        if (hipMemsetAsync(sl.d_report, 0, sizeof(dcn_batch_report), c->stream) != hipSuccess)
            return fail(dcn_fail(DCN_ERR_HIP, "hipMemsetAsync failed"));
        for (size_t ci = 0; ci < sl.chunks.size(); ++ci)
            if ((rc = enqueue_chunk(c, sl, ci, false)) != DCN_OK) return fail(rc);
        if ((rc = finish_submission(c, sl)) != DCN_OK) return fail(rc);
    }
    if (!sl.keep_direct && sl.n_units) memcpy(sl.u_keep, sl.h_keep, sl.n_units);
    if (sl.u_hits && !sl.hits_direct && sl.n_units) memcpy(sl.u_hits, sl.h_hits, (uint64_t)sl.n_units * sizeof(uint32_t));
    if (sl.u_total && !sl.total_direct && sl.n_units) memcpy(sl.u_total, sl.h_total, (uint64_t)sl.n_units * sizeof(uint32_t));
    for (int i = 0; i < DCN_N_STATS; ++i) c->host_stats[i] += sl.h_report->stats[i];
    sl.busy = false;
    return DCN_OK;
}

} // namespace

extern "C" int dcn_filter_batch_submit(dcn_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, const uint32_t *unit_id,
                                       uint32_t n_reads, const dcn_params *params, uint8_t *keep, uint32_t *hits,
                                       uint32_t *total, uint64_t *ticket) {
    HostInput in;
    in.bases = bases;
    in.offsets = offsets;
    in.unit_id = unit_id;
    in.n_reads = n_reads;
    return submit_impl(ctx, in, params, keep, hits, total, ticket);
}

extern "C" int dcn_filter_batch_packed_submit(dcn_ctx *ctx, const uint32_t *packed, const uint32_t *invmask,
                                              const uint64_t *offsets, const uint32_t *unit_id, uint32_t n_reads,
                                              const dcn_params *params, uint8_t *keep, uint32_t *hits, uint32_t *total,
                                              uint64_t *ticket) {
    if (n_reads > 0 && !packed) return dcn_fail(DCN_ERR_ARG, "packed is NULL");
    HostInput in;
    in.packed = packed;
    in.invmask = invmask;
    in.offsets = offsets;
    in.unit_id = unit_id;
    in.n_reads = n_reads;
    return submit_impl(ctx, in, params, keep, hits, total, ticket);
}

extern "C" int dcn_filter_batch_wait(dcn_ctx *ctx, uint64_t ticket) { return wait_impl(ctx, ticket); }

extern "C" int dcn_filter_batch(dcn_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, const uint32_t *unit_id,
                                uint32_t n_reads, const dcn_params *params, uint8_t *keep, uint32_t *hits,
                                uint32_t *total) {
    uint64_t ticket = 0;
    DCN_TRY(dcn_filter_batch_submit(ctx, bases, offsets, unit_id, n_reads, params, keep, hits, total, &ticket));
    return wait_impl(ctx, ticket);
}

extern "C" int dcn_filter_batch_packed(dcn_ctx *ctx, const uint32_t *packed, const uint32_t *invmask,
                                       const uint64_t *offsets, const uint32_t *unit_id, uint32_t n_reads,
                                       const dcn_params *params, uint8_t *keep, uint32_t *hits, uint32_t *total) {
    uint64_t ticket = 0;
    DCN_TRY(dcn_filter_batch_packed_submit(ctx, packed, invmask, offsets, unit_id, n_reads, params, keep, hits, total,
                                           &ticket));
    return wait_impl(ctx, ticket);
}

extern "C" int dcn_pack_ascii(const uint8_t *bases, uint64_t n_bases, uint32_t *packed, uint32_t *invmask,
                              uint32_t *saw_newline) {
    if (saw_newline) *saw_newline = 0;
    if (n_bases > 0 && (!bases || !packed || !invmask)) return dcn_fail(DCN_ERR_ARG, "bases/packed/invmask is NULL");
    const uint64_t G = (n_bases + 31) / 32;
    std::atomic<uint32_t> nl{0};
    HostPool::get().run([&](int i, int nt) {
        const uint64_t per = (G + nt - 1) / nt, lo = std::min<uint64_t>(G, per * i), hi = std::min<uint64_t>(G, lo + per);
        if (dcn_host_pack_groups(bases, n_bases, lo, hi, packed + 2 * lo, invmask + lo)) nl.store(1, std::memory_order_relaxed);
    }, G < 4096);
    if (saw_newline) *saw_newline = nl.load();
    return DCN_OK;
}

extern "C" int dcn_host_alloc(uint64_t bytes, void **out) {
    if (!out) return dcn_fail(DCN_ERR_ARG, "out is NULL");
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, std::max<uint64_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        *out = nullptr;
        return dcn_fail(DCN_ERR_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    return DCN_OK;
}

extern "C" void dcn_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

extern "C" int dcn_ctx_set_profiling(dcn_ctx *ctx, int enable) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    DCN_HIP(hipSetDevice(ctx->device));
    DCN_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < dcn_ctx::PROF_RING; ++i) ctx->prof_used[i] = false;
    for (int j = 0; j < DCN_N_STAGES; ++j) ctx->prof_ms[j] = 0.0;
    ctx->prof_batches = 0;
    ctx->profiling = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return DCN_OK;
}

extern "C" int dcn_ctx_profile(dcn_ctx *ctx, double stage_ms[DCN_N_STAGES], uint64_t *n_batches) {
    if (!ctx || !stage_ms) return dcn_fail(DCN_ERR_ARG, "ctx/stage_ms is NULL");
    if (ctx->profiling) { // waits for the runs marked so far
        DCN_HIP(hipSetDevice(ctx->device));
        DCN_TRY(prof_harvest(ctx));
    }
    for (int j = 0; j < DCN_N_STAGES; ++j) stage_ms[j] = ctx->prof_ms[j];
    if (n_batches) *n_batches = ctx->prof_batches;
    return DCN_OK;
}

// counters = completed host batches (summed on the host when each batch is waited for) + everything the
// device-pointer API has enqueued (accumulated on the device)
extern "C" int dcn_ctx_stats(dcn_ctx *ctx, uint64_t counters[DCN_N_STATS]) {
    if (!ctx || !counters) return dcn_fail(DCN_ERR_ARG, "ctx/counters is NULL");
    DCN_HIP(hipSetDevice(ctx->device));
    DCN_HIP(hipStreamSynchronize(ctx->stream));
    DCN_HIP(hipMemcpy(ctx->h_report, ctx->d_report, sizeof(dcn_batch_report), hipMemcpyDeviceToHost));
    for (int i = 0; i < DCN_N_STATS; ++i) counters[i] = ctx->h_report->stats[i] + ctx->host_stats[i];
    return DCN_OK;
}

extern "C" int dcn_ctx_reset_stats(dcn_ctx *ctx) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    DCN_HIP(hipSetDevice(ctx->device));
    DCN_HIP(hipStreamSynchronize(ctx->stream));
    DCN_HIP(hipMemsetAsync(ctx->d_report->stats, 0, sizeof(unsigned long long) * DCN_N_STATS, ctx->stream));
    DCN_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < DCN_N_STATS; ++i) ctx->host_stats[i] = 0;
    return DCN_OK;
}

// Sum of the six counters over several contexts: the merge the reference does when its worker threads finish
// (ProcessingStats, src/local_filter.rs:388-396).  All contexts live in this process, one per device or several per
// device, so the sum is taken on the host; ranks in separate processes reduce with RCCL instead.
extern "C" int dcn_stats_allreduce(dcn_ctx *const *ctxs, int n_ctx, uint64_t counters[DCN_N_STATS]) {
    if (!counters || (n_ctx > 0 && !ctxs) || n_ctx < 0) return dcn_fail(DCN_ERR_ARG, "ctxs/counters is NULL");
    for (int i = 0; i < DCN_N_STATS; ++i) counters[i] = 0;
    for (int j = 0; j < n_ctx; ++j) {
        uint64_t one[DCN_N_STATS];
        DCN_TRY(dcn_ctx_stats(ctxs[j], one));
        for (int i = 0; i < DCN_N_STATS; ++i) counters[i] += one[i];
    }
    return DCN_OK;
}

namespace {
int validate_host_batch(const dcn_ctx *c, const uint64_t *offsets, uint32_t n_reads) {
    if (n_reads > c->max_reads) return dcn_fail(DCN_ERR_CAPACITY, "n_reads exceeds the context's max_batch_reads");
    if (slots_busy(c) || c->batch_pending) return dcn_fail(DCN_ERR_ARG, "batches are in flight on this context: wait for them first");
    if (offsets[0] != 0) return dcn_fail(DCN_ERR_ARG, "offsets[0] must be 0");
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (offsets[r + 1] < offsets[r]) return dcn_fail(DCN_ERR_ARG, "offsets must be non-decreasing");
        if (offsets[r + 1] - offsets[r] > 0xFFFFFFF0ull) return dcn_fail(DCN_ERR_ARG, "read longer than 2^32 bases");
    }
    if (offsets[n_reads] > c->max_bases) return dcn_fail(DCN_ERR_CAPACITY, "batch exceeds the context's max_batch_bases");
    return DCN_OK;
}
} // namespace

// ----------------------------------------------------------------------------------------------------
// minimizer dump (parity / debugging seam)
// ----------------------------------------------------------------------------------------------------
extern "C" int dcn_minimizer_hashes_batch(dcn_ctx *ctx, const uint8_t *bases, const uint64_t *offsets,
                                          uint32_t n_reads, uint64_t prefix_length, uint64_t *out_offsets,
                                          uint64_t *out_hashes, uint32_t *out_positions, uint64_t capacity) {
    if (!ctx || !out_offsets) return dcn_fail(DCN_ERR_ARG, "ctx/out_offsets is NULL");
    out_offsets[0] = 0;
    if (n_reads == 0) return DCN_OK;
    if (!offsets) return dcn_fail(DCN_ERR_ARG, "offsets is NULL");
    DCN_TRY(validate_host_batch(ctx, offsets, n_reads));
    uint64_t n_bases = offsets[n_reads];
    if (n_bases > 0 && !bases) return dcn_fail(DCN_ERR_ARG, "bases is NULL");
    dcn_ctx *c = ctx;
    DCN_HIP(hipSetDevice(c->device));
    if (!c->d_dump_hash) {
        DCN_TRY(dev_alloc(&c->d_dump_hash, c->max_bases + 2, "dump_hash"));
        DCN_TRY(dev_alloc(&c->d_dump_pos, c->max_bases + 2, "dump_pos"));
        DCN_TRY(dev_alloc(&c->d_dump_valid, c->max_bases + 2, "dump_valid"));
        DCN_TRY(dev_alloc(&c->d_dump_count, c->max_tiles, "dump_count"));
    }
    if (!c->d_tile_read_pos) DCN_TRY(dev_alloc(&c->d_tile_read_pos, c->max_tiles, "tile_read_pos"));
    DCN_TRY(staged_h2d(c, c->d_ascii, bases, n_bases));
    DCN_TRY(staged_h2d(c, c->d_offsets, offsets, (uint64_t)(n_reads + 1) * sizeof(uint64_t)));
    DCN_HIP(hipEventRecord(c->copy_done, c->copy_stream));
    DCN_HIP(hipStreamWaitEvent(c->stream, c->copy_done, 0));
    hipStream_t st = c->stream;
    DCN_HIP(hipMemsetAsync(c->d_status, 0, sizeof(dcn_status), st));
    uint32_t *packed = c->d_packed + DCN_FRONT_PAD, *invmask = c->d_invmask + DCN_FRONT_PAD;
    DCN_TRY(dcn_launch_pack(c->d_ascii, 0, n_bases, packed, invmask, c->d_status, st));
    dcn_plan_args pa = {};
    pa.ascii = c->d_ascii;
    pa.offsets = c->d_offsets;
    pa.unit_id = nullptr;
    pa.n_reads = n_reads;
    pa.n_units = n_reads;
    pa.k = c->index->k;
    pa.w = c->index->w;
    pa.prefix_length = prefix_length;
    pa.tile_windows = c->tile_windows;
    pa.read_tiles = c->d_read_tiles;
    pa.read_tile_first = c->d_read_tile_first;
    pa.unit_first_read = c->d_unit_first_read;
    pa.unit_tile_first = c->d_unit_tile_first;
    pa.unit_tile_count = c->d_unit_tile_count;
    pa.tile_cursor = &c->d_status->n_tiles;
    pa.tiles = c->d_tiles;
    pa.tile_read_pos = c->d_tile_read_pos;
    pa.status = c->d_status;
    DCN_TRY(dcn_launch_plan(pa, st));
    dcn_scan_args sa;
    memset(&sa, 0, sizeof(sa));
    sa.packed = packed;
    sa.invmask = invmask;
    sa.tiles = c->d_tiles;
    sa.tile_read_pos = c->d_tile_read_pos;
    sa.n_tiles = &c->d_status->n_tiles;
    sa.table = c->index->view();
    sa.k = c->index->k;
    sa.variant = c->index->variant;
    sa.w = c->index->w;
    sa.stream_bases = n_bases;
    sa.status = c->d_status;
    sa.dump_hash = c->d_dump_hash;
    sa.dump_pos = c->d_dump_pos;
    sa.dump_valid = c->d_dump_valid;
    sa.dump_count = c->d_dump_count;
    uint64_t tile_bound = std::min<uint64_t>((uint64_t)n_reads + n_bases / c->tile_windows + 1, c->max_tiles);
    DCN_TRY(dcn_launch_scan(sa, (uint32_t)tile_bound, true, st));
    DCN_HIP(hipStreamSynchronize(st));
    // gather on the host: tiles are in read order, a tile's entries sit at [first own window's absolute
    // base index ...) in emit order; entries failing the ACGT test are dropped (src/filter_common.rs:275-286)
    std::vector<uint32_t> rtf(n_reads), rtn(n_reads);
    DCN_HIP(hipMemcpy(rtf.data(), c->d_read_tile_first, (uint64_t)n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost));
    DCN_HIP(hipMemcpy(rtn.data(), c->d_read_tiles, (uint64_t)n_reads * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint32_t nt = 0;
    DCN_HIP(hipMemcpy(&nt, &c->d_status->n_tiles, sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<dcn_tile> tiles(nt);
    std::vector<uint32_t> tcount(nt);
    std::vector<uint64_t> h(n_bases + 2);
    std::vector<uint32_t> p(n_bases + 2);
    std::vector<uint8_t> v(n_bases + 2);
    if (nt) {
        DCN_HIP(hipMemcpy(tiles.data(), c->d_tiles, (uint64_t)nt * sizeof(dcn_tile), hipMemcpyDeviceToHost));
        DCN_HIP(hipMemcpy(tcount.data(), c->d_dump_count, (uint64_t)nt * sizeof(uint32_t), hipMemcpyDeviceToHost));
        DCN_HIP(hipMemcpy(h.data(), c->d_dump_hash, (n_bases + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
        DCN_HIP(hipMemcpy(p.data(), c->d_dump_pos, (n_bases + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        DCN_HIP(hipMemcpy(v.data(), c->d_dump_valid, (n_bases + 1), hipMemcpyDeviceToHost));
    }
    uint64_t n_out = 0;
    for (uint32_t r = 0; r < n_reads; ++r) {
        for (uint32_t t = rtf[r]; t < rtf[r] + rtn[r]; ++t) {
            uint64_t base = tiles[t].scan_start + tiles[t].carry();
            for (uint32_t e = 0; e < tcount[t]; ++e) {
                if (!v[base + e]) continue;
                if (n_out < capacity) {
                    if (out_hashes) out_hashes[n_out] = h[base + e];
                    if (out_positions) out_positions[n_out] = p[base + e];
                }
                n_out++;
            }
        }
        out_offsets[r + 1] = n_out;
    }
    if (n_out > capacity) return dcn_fail(DCN_ERR_CAPACITY, "output capacity too small: need " + std::to_string(n_out));
    return DCN_OK;
}

// ----------------------------------------------------------------------------------------------------
// server batch seam: hashes precomputed (src/remote_filter.rs:230-301)
// ----------------------------------------------------------------------------------------------------
extern "C" int dcn_should_keep_hashes(dcn_ctx *ctx, const uint64_t *hashes, const uint64_t *hash_offsets,
                                      uint32_t n_units, const dcn_params *params, uint8_t *keep, uint32_t *hits,
                                      uint32_t *total) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    DCN_TRY(check_params(params));
    if (n_units == 0) return DCN_OK;
    if (!hash_offsets || !keep) return dcn_fail(DCN_ERR_ARG, "hash_offsets/keep is NULL");
    if (n_units > ctx->max_reads) return dcn_fail(DCN_ERR_CAPACITY, "n_units exceeds the context's max_batch_reads");
    if (slots_busy(ctx) || ctx->batch_pending) return dcn_fail(DCN_ERR_ARG, "batches are in flight on this context: wait for them first");
    if (hash_offsets[0] != 0) return dcn_fail(DCN_ERR_ARG, "hash_offsets[0] must be 0");
    for (uint32_t u = 0; u < n_units; ++u) {
        if (hash_offsets[u + 1] < hash_offsets[u]) return dcn_fail(DCN_ERR_ARG, "hash_offsets must be non-decreasing");
        if (hash_offsets[u + 1] - hash_offsets[u] > 0xFFFFFFF0ull) return dcn_fail(DCN_ERR_ARG, "unit has more than 2^32 hashes");
    }
    uint64_t n_hashes = hash_offsets[n_units];
    if (n_hashes > 0 && !hashes) return dcn_fail(DCN_ERR_ARG, "hashes is NULL");
    dcn_ctx *c = ctx;
    DCN_HIP(hipSetDevice(c->device));
    // a unit with more hashes than the LDS set of the distinct pass holds takes a global set of <= 4 slots per hash
    if (n_hashes > c->rec_capacity) DCN_TRY(dcn_ctx_reserve_records(c, std::min<uint64_t>(n_hashes, 1ull << 29)));
    if (n_hashes > c->rec_capacity) return dcn_fail(DCN_ERR_CAPACITY, "too many hashes in one call");
    uint64_t *d_hashes = nullptr, *d_hoff = nullptr;
    DCN_TRY(dev_alloc(&d_hashes, n_hashes, "hashes"));
    int rc = dev_alloc(&d_hoff, (uint64_t)n_units + 1, "hash_offsets");
    if (rc != DCN_OK) {
        hipFree(d_hashes);
        return rc;
    }
    auto body = [&]() -> int {
        hipStream_t st = c->stream;
        DCN_HIP(hipMemcpyAsync(d_hashes, hashes, n_hashes * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        DCN_HIP(hipMemcpyAsync(d_hoff, hash_offsets, ((uint64_t)n_units + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        DCN_HIP(hipMemsetAsync(c->d_status, 0, sizeof(dcn_status), st));
        uint32_t *g_total = c->d_unit_scratch, *g_hitcnt = g_total + c->max_reads,
                 *g_distinct = g_hitcnt + c->max_reads, *g_zero = g_distinct + c->max_reads;
        dcn_probe_hashes_args pa;
        pa.table = c->index->view();
        pa.hashes = d_hashes;
        pa.hash_offsets = d_hoff;
        pa.n_hashes = n_hashes;
        pa.n_units = n_units;
        pa.tiles = c->d_tiles;
        pa.n_tiles = &c->d_status->n_tiles;
        pa.tile_hits = c->d_tile_hits;
        pa.unit_tile_first = c->d_unit_tile_first;
        pa.unit_tile_count = c->d_unit_tile_count;
        pa.pending = c->d_pending;
        pa.unit_state = c->d_unit_state;
        pa.g_total = g_total;
        pa.g_hitcnt = g_hitcnt;
        pa.g_distinct = g_distinct;
        pa.g_zero = g_zero;
        pa.status = c->d_status;
        DCN_TRY(dcn_launch_probe_hashes(pa, st));
        dcn_distinct_args da;
        da.tiles = c->d_tiles;
        da.n_tiles = &c->d_status->n_tiles;
        da.unit_tile_first = c->d_unit_tile_first;
        da.unit_tile_count = c->d_unit_tile_count;
        da.unit_state = c->d_unit_state;
        da.tile_hits = c->d_tile_hits;
        da.pending = c->d_pending;
        da.rec_hash = d_hashes;
        da.rec_shift = 0;
        da.g_hitcnt = g_hitcnt;
        da.g_distinct = g_distinct;
        da.set_off = c->d_set_off;
        da.set_slots = c->d_set_slots;
        da.set_capacity = 4 * c->rec_capacity + 64;
        da.n_units = n_units;
        da.status = c->d_status;
        da.caps = c->d_caps;
        da.big = c->d_big;
        da.g_total = nullptr; // (the server's answer carries the hit count: src/server_common.rs:54-58)
        da.abs_threshold = params->abs_threshold;
        da.rel_threshold = params->rel_threshold;
        DCN_TRY(dcn_launch_distinct(da, st));
        dcn_finish_args fa;
        fa.n_units = n_units;
        fa.unit_first_read = nullptr;
        fa.offsets = nullptr; // no read lengths here: the counters are untouched
        fa.unit_state = c->d_unit_state;
        fa.g_total = g_total;
        fa.g_hitcnt = g_hitcnt;
        fa.g_distinct = g_distinct;
        fa.g_zero = g_zero;
        fa.abs_threshold = params->abs_threshold;
        fa.rel_threshold = params->rel_threshold;
        fa.deplete = params->deplete;
        fa.keep = c->d_keep;
        fa.hits = c->d_hits;
        fa.total = c->d_total;
        fa.report = c->d_report;
        fa.status = c->d_status;
        DCN_TRY(dcn_launch_finish(fa, st));
        c->batch_pending = true;
        DCN_TRY(sync_and_check(c, nullptr));
        DCN_HIP(hipMemcpy(keep, c->d_keep, n_units, hipMemcpyDeviceToHost));
        if (hits) DCN_HIP(hipMemcpy(hits, c->d_hits, (uint64_t)n_units * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (total) DCN_HIP(hipMemcpy(total, c->d_total, (uint64_t)n_units * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return DCN_OK;
    };
    rc = body();
    hipStreamSynchronize(c->stream);
    hipFree(d_hashes);
    hipFree(d_hoff);
    return rc;
}

// ----------------------------------------------------------------------------------------------------
// index build (f1): chunks of sequence pieces -> pack (index-side codes) -> plan -> scan in dump mode -> insert
// ----------------------------------------------------------------------------------------------------
int dcn_build_index_impl(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seqs, float entropy_threshold,
                         uint64_t capacity_keys, dcn_index *idx) {
    (void)capacity_keys;
    if (n_seqs == 0) return DCN_OK;
    if (offsets[0] != 0) return dcn_fail(DCN_ERR_ARG, "offsets[0] must be 0");
    const uint32_t k = idx->k, l = (uint32_t)idx->k + idx->w - 1;
    // a piece is a range of one sequence; a sequence longer than the chunk is cut into pieces overlapping by
    // l-1 bases, which yields every window exactly once (an extra duplicate at a seam merges in the set)
    uint64_t chunk_bases = 1ull << 27;
    if (const char *cb = getenv("DCN_BUILD_CHUNK_BASES")) {
        long long v = atoll(cb);
        if (v >= 4096) chunk_bases = (uint64_t)v;
    }
    const uint32_t max_pieces = 1u << 16;
    dcn_ctx *c = nullptr;
    DCN_TRY(dcn_ctx_create(idx, chunk_bases, max_pieces, &c));
    int rc = DCN_OK;
    auto body = [&]() -> int {
        DCN_TRY(dev_alloc(&c->d_dump_hash, c->max_bases + 2, "dump_hash"));
        DCN_TRY(dev_alloc(&c->d_dump_pos, c->max_bases + 2, "dump_pos"));
        DCN_TRY(dev_alloc(&c->d_dump_valid, c->max_bases + 2, "dump_valid"));
        DCN_TRY(dev_alloc(&c->d_dump_count, c->max_tiles, "dump_count"));
        std::vector<uint64_t> p_off;   // offsets of the pieces inside the chunk buffer
        std::vector<const uint8_t *> p_src;
        std::vector<uint64_t> p_len;
        auto run_chunk = [&]() -> int {
            if (p_len.empty()) return DCN_OK;
            uint32_t np = (uint32_t)p_len.size();
            p_off.assign(np + 1, 0);
            for (uint32_t i = 0; i < np; ++i) p_off[i + 1] = p_off[i] + p_len[i];
            uint64_t nb = p_off[np];
            for (uint32_t i = 0; i < np; ++i)
                DCN_TRY(staged_h2d(c, c->d_ascii + p_off[i], p_src[i], p_len[i]));
            DCN_TRY(staged_h2d(c, c->d_offsets, p_off.data(), (uint64_t)(np + 1) * sizeof(uint64_t)));
            DCN_HIP(hipEventRecord(c->copy_done, c->copy_stream));
            DCN_HIP(hipStreamWaitEvent(c->stream, c->copy_done, 0));
            hipStream_t st = c->stream;
            DCN_HIP(hipMemsetAsync(c->d_status, 0, sizeof(dcn_status), st));
            DCN_HIP(hipMemsetAsync(c->d_dump_valid, 0, nb + 2, st));
            uint32_t *packed = c->d_packed + DCN_FRONT_PAD, *invmask = c->d_invmask + DCN_FRONT_PAD;
            DCN_TRY(dcn_launch_pack(c->d_ascii, 0, nb, packed, invmask, nullptr, st, /*index_side=*/true));
            dcn_plan_args pa = {};
            pa.ascii = c->d_ascii;
            pa.offsets = c->d_offsets;
            pa.unit_id = nullptr;
            pa.n_reads = np;
            pa.n_units = np;
            pa.k = idx->k;
            pa.w = idx->w;
            pa.prefix_length = 0;
            pa.tile_windows = c->tile_windows;
            pa.read_tiles = c->d_read_tiles;
            pa.read_tile_first = c->d_read_tile_first;
            pa.unit_first_read = c->d_unit_first_read;
            pa.unit_tile_first = c->d_unit_tile_first;
            pa.unit_tile_count = c->d_unit_tile_count;
            pa.tile_cursor = &c->d_status->n_tiles;
            pa.tiles = c->d_tiles;
            pa.status = c->d_status;
            DCN_TRY(dcn_launch_plan(pa, st));
            dcn_scan_args sa;
            memset(&sa, 0, sizeof(sa));
            sa.packed = packed;
            sa.invmask = invmask;
            sa.tiles = c->d_tiles;
            sa.n_tiles = &c->d_status->n_tiles;
            sa.table = idx->view();
            sa.k = idx->k;
            sa.variant = idx->variant;
            sa.w = idx->w;
            sa.stream_bases = nb;
            sa.status = c->d_status;
            sa.dump_hash = c->d_dump_hash;
            sa.dump_pos = c->d_dump_pos;
            sa.dump_valid = c->d_dump_valid;
            sa.dump_count = c->d_dump_count;
            sa.dump_abs = 1;
            uint64_t tile_bound = std::min<uint64_t>((uint64_t)np + nb / c->tile_windows + 1, c->max_tiles);
            DCN_TRY(dcn_launch_scan(sa, (uint32_t)tile_bound, true, st));
            uint64_t n_valid = 0;
            DCN_TRY(dcn_table_count_valid(c->d_dump_valid, nb, &n_valid, st));
            DCN_TRY(dcn_table_reserve(idx, idx->n_keys + n_valid));
            DCN_TRY(dcn_table_insert_dump(idx, c->d_dump_hash, c->d_dump_valid, c->d_dump_pos, nb, c->d_ascii,
                                          entropy_threshold, st));
            p_src.clear();
            p_len.clear();
            return DCN_OK;
        };
        uint64_t used = 0;
        for (uint32_t sidx = 0; sidx < n_seqs; ++sidx) {
            if (offsets[sidx + 1] < offsets[sidx]) return dcn_fail(DCN_ERR_ARG, "offsets must be non-decreasing");
            const uint8_t *seq = bases + offsets[sidx];
            uint64_t len = offsets[sidx + 1] - offsets[sidx];
            if (len < k || len < l) continue; // src/minimizers.rs:135; fewer than l bases have no window
            uint64_t a = 0;
            while (a + l <= len) {
                uint64_t room = chunk_bases - used;
                if (room < l || p_len.size() >= max_pieces) {
                    DCN_TRY(run_chunk());
                    used = 0;
                    room = chunk_bases;
                }
                uint64_t take = std::min<uint64_t>(room, len - a);
                if (take > 0xFFFFFF00ull) take = 0xFFFFFF00ull;
                p_src.push_back(seq + a);
                p_len.push_back(take);
                used += take;
                if (a + take >= len) break;
                a += take - (l - 1); // next piece starts l-1 bases before the cut
            }
        }
        return run_chunk();
    };
    rc = body();
    dcn_ctx_destroy(c);
    return rc;
}
