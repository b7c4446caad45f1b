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

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define DCN_VERSION_STRING "deacon-hip 0.1.0 (gfx950)"

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

static int check_kw(uint8_t k, uint8_t w) {
    if (k < 1 || k > 56) return dcn_fail(DCN_ERR_ARG, "k must be in 1..=56 (src/filter_common.rs:269-272)");
    if (w < 1) return dcn_fail(DCN_ERR_ARG, "w must be >= 1");
    if (((uint32_t)k + w - 1) % 2 == 0)
        return dcn_fail(DCN_ERR_ARG, "Constraint violated: k + w - 1 must be odd (src/index.rs:186-194)");
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
    std::vector<uint64_t> keys;
    try {
        keys.resize(index->n_keys);
    } catch (...) {
        return dcn_fail(DCN_ERR_NOMEM, "index too large for host memory");
    }
    uint64_t n = 0;
    int rc = dcn_table_export(index, keys.data(), keys.size(), &n);
    if (rc != DCN_OK) return rc;
    return dcn_write_index_file(path, index->k, index->w, keys.data(), n);
}

extern "C" int dcn_index_header(const dcn_index *index, uint8_t *k, uint8_t *w, uint64_t *n_keys) {
    if (!index) return dcn_fail(DCN_ERR_ARG, "index is NULL");
    if (k) *k = index->k;
    if (w) *w = index->w;
    if (n_keys) *n_keys = index->n_keys;
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

extern "C" void dcn_index_destroy(dcn_index *index) {
    if (!index) return;
    hipSetDevice(index->device);
    if (index->d_slots) hipFree(index->d_slots);
    delete index;
}

// ----------------------------------------------------------------------------------------------------
// context
// ----------------------------------------------------------------------------------------------------
struct dcn_ctx {
    const dcn_index *index = nullptr;
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    hipEvent_t copy_done = nullptr, stage_free[2] = {nullptr, nullptr};
    uint64_t max_bases = 0;
    uint32_t max_reads = 0;
    uint32_t tile_windows = 512;
    uint32_t max_tiles = 0;
    // device inputs (host API staging targets)
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
    uint32_t *d_unit_scratch = nullptr; // g_total | g_hitcnt | g_distinct | g_zero, max_reads each
    uint32_t *d_caps = nullptr, *d_set_off = nullptr;
    // hit records + distinct sets
    uint64_t rec_capacity = 0;
    uint32_t *d_rec_unit = nullptr;
    uint64_t *d_rec_hash = nullptr;
    uint64_t *d_set_slots = nullptr;
    dcn_status *d_status = nullptr;
    // pinned host staging
    uint8_t *h_stage[2] = {nullptr, nullptr};
    uint64_t stage_bytes = 0;
    dcn_status *h_status = nullptr;
    // dump mode buffers (lazy)
    uint64_t *d_dump_hash = nullptr;
    uint32_t *d_dump_pos = nullptr, *d_dump_count = nullptr;
    uint8_t *d_dump_valid = nullptr;
    // deferred state of the last enqueued batch
    bool batch_pending = false;
    // optional per-stage timing: a ring of event sets, one per batch in flight
    static constexpr int PROF_RING = 32;
    bool profiling = false;
    hipEvent_t prof_ev[PROF_RING][DCN_N_STAGES + 1] = {};
    bool prof_used[PROF_RING] = {};
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
    n_records = (n_records + DCN_REC_SHARDS - 1) / DCN_REC_SHARDS * DCN_REC_SHARDS;
    if (n_records > (1ull << 29)) return dcn_fail(DCN_ERR_CAPACITY, "more than 2^29 hit records per batch: use smaller batches");
    if (c->d_rec_unit) hipFree(c->d_rec_unit);
    if (c->d_rec_hash) hipFree(c->d_rec_hash);
    if (c->d_set_slots) hipFree(c->d_set_slots);
    c->d_rec_unit = nullptr;
    c->d_rec_hash = nullptr;
    c->d_set_slots = nullptr;
    c->rec_capacity = 0;
    DCN_TRY(dev_alloc(&c->d_rec_unit, n_records, "rec_unit"));
    DCN_TRY(dev_alloc(&c->d_rec_hash, n_records, "rec_hash"));
    DCN_TRY(dev_alloc(&c->d_set_slots, 4 * n_records + 64, "set_slots"));
    c->rec_capacity = n_records;
    return DCN_OK;
}

void free_ctx(dcn_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    void *dev[] = {c->d_ascii, c->d_offsets, c->d_unit_id, c->d_packed, c->d_invmask,
                   c->d_read_tiles, c->d_read_tile_first, c->d_unit_first_read, c->d_unit_tile_first, c->d_unit_tile_count, c->d_tiles,
                   c->d_keep, c->d_unit_state, c->d_hits, c->d_total, c->d_unit_scratch, c->d_caps,
                   c->d_set_off, c->d_rec_unit, c->d_rec_hash, c->d_set_slots, c->d_status, c->d_dump_hash,
                   c->d_dump_pos, c->d_dump_count, c->d_dump_valid};
    for (void *p : dev)
        if (p) hipFree(p);
    for (int i = 0; i < 2; ++i) {
        if (c->h_stage[i]) hipHostFree(c->h_stage[i]);
        if (c->stage_free[i]) hipEventDestroy(c->stage_free[i]);
    }
    if (c->h_status) hipHostFree(c->h_status);
    for (int i = 0; i < dcn_ctx::PROF_RING; ++i)
        for (int j = 0; j <= DCN_N_STAGES; ++j)
            if (c->prof_ev[i][j]) hipEventDestroy(c->prof_ev[i][j]);
    if (c->copy_done) hipEventDestroy(c->copy_done);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    delete c;
}

// fold the event pairs of every completed batch into the per-stage accumulators
int prof_harvest(dcn_ctx *c, int only_slot = -1) {
    for (int i = 0; i < dcn_ctx::PROF_RING; ++i) {
        if (!c->prof_used[i] || (only_slot >= 0 && i != only_slot)) continue;
        DCN_HIP(hipEventSynchronize(c->prof_ev[i][DCN_N_STAGES]));
        for (int j = 0; j < DCN_N_STAGES; ++j) {
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
    DCN_HIP(hipEventRecord(c->prof_ev[i][0], c->stream));
    *slot = i;
    return DCN_OK;
}

#define DCN_PROF_MARK(stage)                                                        \
    do {                                                                            \
        if (prof_slot >= 0) DCN_HIP(hipEventRecord(c->prof_ev[prof_slot][(stage) + 1], st)); \
    } while (0)

int check_params(const dcn_params *p) {
    if (!p) return dcn_fail(DCN_ERR_ARG, "params is NULL");
    if (p->reserved != 0) return dcn_fail(DCN_ERR_ARG, "params.reserved must be 0");
    if (p->deplete > 1) return dcn_fail(DCN_ERR_ARG, "params.deplete must be 0 or 1");
    return DCN_OK;
}

// enqueue the whole device pipeline for one batch whose ASCII / offsets / unit ids are in device memory
int enqueue_batch(dcn_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, const uint32_t *d_unit_id,
                  uint32_t n_reads, uint64_t n_bases, uint32_t n_units, const dcn_params *params, uint8_t *d_keep,
                  uint32_t *d_hits, uint32_t *d_total) {
    hipStream_t st = c->stream;
    const dcn_index *idx = c->index;
    // per-batch scratch: status header (not the counters), per-unit state
    // (the per-unit state and scratch words are cleared by the plan kernel)
    DCN_HIP(hipMemsetAsync(c->d_status, 0, offsetof(dcn_status, stats), st));

    int prof_slot = -1;
    DCN_TRY(prof_begin(c, &prof_slot));
    uint32_t *packed = c->d_packed + DCN_FRONT_PAD, *invmask = c->d_invmask + DCN_FRONT_PAD;
    DCN_TRY(dcn_launch_pack(d_bases, n_bases, packed, invmask, st));
    DCN_PROF_MARK(DCN_STAGE_PACK);

    dcn_plan_args pa = {};
    pa.ascii = d_bases;
    pa.offsets = d_offsets;
    pa.unit_id = d_unit_id;
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
    DCN_TRY(dcn_launch_plan(pa, st));
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
    sa.w = idx->w;
    sa.abs_threshold = params->abs_threshold;
    sa.rel_threshold = params->rel_threshold;
    sa.deplete = params->deplete;
    // decisions only: largest list length whose required hits still equal abs_threshold (dcn_required_hits is
    // monotone in the total); the scan kernel's lanes then stop at abs_threshold distinct hits (scan.hip)
    sa.early_out_max_items = 0;
    if (!d_hits && !d_total && params->abs_threshold >= 1 && params->abs_threshold <= 4 && !getenv("DCN_NO_EARLY_OUT")) {
        uint32_t lo = 0, hi = 65535; // required(lo) == abs always holds for lo = 0
        while (lo < hi) {
            uint32_t mid = (lo + hi + 1) / 2;
            if (dcn_required_hits(params->abs_threshold, params->rel_threshold, mid) == params->abs_threshold) lo = mid;
            else hi = mid - 1;
        }
        sa.early_out_max_items = lo;
        sa.early_out_pairs = getenv("DCN_NO_EARLY_OUT_PAIRS") ? 0u : 1u;
    }
    sa.keep = d_keep;
    sa.hits = d_hits;
    sa.total = d_total;
    sa.unit_state = c->d_unit_state;
    sa.g_total = g_total;
    sa.g_hitcnt = g_hitcnt;
    sa.rec_unit = c->d_rec_unit;
    sa.rec_hash = c->d_rec_hash;
    sa.rec_capacity = c->rec_capacity;
    sa.status = c->d_status;
    uint64_t tile_bound = (uint64_t)n_reads + n_bases / c->tile_windows + 1;
    if (tile_bound > c->max_tiles) tile_bound = c->max_tiles;
    DCN_TRY(dcn_launch_scan(sa, (uint32_t)tile_bound, false, st));
    DCN_PROF_MARK(DCN_STAGE_SCAN);

    dcn_distinct_args da;
    da.rec_unit = c->d_rec_unit;
    da.rec_hash = c->d_rec_hash;
    da.rec_capacity = c->rec_capacity;
    da.g_hitcnt = g_hitcnt;
    da.g_distinct = g_distinct;
    da.g_zero = g_zero;
    da.set_off = c->d_set_off;
    da.set_slots = c->d_set_slots;
    da.set_capacity = 4 * c->rec_capacity + 64;
    da.n_units = n_units;
    da.status = c->d_status;
    da.caps = c->d_caps;
    DCN_TRY(dcn_launch_distinct(da, st));
    DCN_PROF_MARK(DCN_STAGE_DISTINCT);

    dcn_finish_args fa;
    fa.n_units = n_units;
    fa.unit_first_read = d_unit_id ? c->d_unit_first_read : nullptr;
    fa.offsets = d_offsets;
    fa.unit_state = c->d_unit_state;
    fa.g_total = g_total;
    fa.g_distinct = g_distinct;
    fa.abs_threshold = params->abs_threshold;
    fa.rel_threshold = params->rel_threshold;
    fa.deplete = params->deplete;
    fa.keep = d_keep;
    fa.hits = d_hits;
    fa.total = d_total;
    fa.status_stats = c->d_status->stats;
    fa.status = c->d_status;
    DCN_TRY(dcn_launch_finish(fa, st));
    DCN_PROF_MARK(DCN_STAGE_FINISH);
    if (prof_slot >= 0) c->prof_used[prof_slot] = true;
    c->batch_pending = true;
    return DCN_OK;
}

// wait for the compute stream and surface deferred pipeline errors
int sync_and_check(dcn_ctx *c, uint64_t *needed_records) {
    DCN_HIP(hipStreamSynchronize(c->stream));
    if (needed_records) *needed_records = 0;
    if (c->profiling) DCN_TRY(prof_harvest(c));
    if (!c->batch_pending) return DCN_OK;
    c->batch_pending = false;
    DCN_HIP(hipMemcpy(c->h_status, c->d_status, sizeof(dcn_status), hipMemcpyDeviceToHost));
    if (c->h_status->rec_overflow) {
        // size the retry for the fullest shard (shards fill unevenly)
        uint64_t need = 0;
        for (uint32_t sidx = 0; sidx < DCN_REC_SHARDS; ++sidx)
            need = std::max<uint64_t>(need, c->h_status->rec_count[sidx]);
        need *= DCN_REC_SHARDS;
        if (needed_records) *needed_records = need;
        return dcn_fail(DCN_ERR_CAPACITY, "hit-record scratch overflow: need " + std::to_string(need) +
                                              " records, have " + std::to_string(c->rec_capacity) +
                                              " (dcn_ctx_reserve_records)");
    }
    return DCN_OK;
}

// A few host threads that split one large memcpy: a single core moves ~10 GB/s into the pinned staging
// buffer, which is less than the PCIe link takes out of it.  Process-wide, created on first use;
// DCN_HOST_THREADS sets the width (default: up to 8, 1 = plain memcpy).
class HostCopyPool {
  public:
    static HostCopyPool &get() {
        static HostCopyPool pool;
        return pool;
    }
    void copy(void *dst, const void *src, size_t n) {
        if (n_threads_ <= 1 || n < (4u << 20)) {
            memcpy(dst, src, n);
            return;
        }
        std::lock_guard<std::mutex> user(user_mu_); // one copy at a time
        {
            std::lock_guard<std::mutex> g(mu_);
            dst_ = (uint8_t *)dst;
            src_ = (const uint8_t *)src;
            n_ = n;
            pending_ = n_threads_ - 1;
            ++generation_;
        }
        cv_.notify_all();
        slice(0);
        std::unique_lock<std::mutex> g(mu_);
        done_cv_.wait(g, [&] { return pending_ == 0; });
    }
    ~HostCopyPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }

  private:
    HostCopyPool() {
        unsigned hw = std::thread::hardware_concurrency();
        int want = (int)std::min<unsigned>(8, hw ? hw : 1);
        if (const char *e = getenv("DCN_HOST_THREADS")) want = atoi(e);
        n_threads_ = std::max(1, std::min(want, 64));
        for (int i = 1; i < n_threads_; ++i) workers_.emplace_back([this, i] { run(i); });
    }
    void slice(int i) {
        size_t per = ((n_ / n_threads_) + 4095) & ~(size_t)4095;
        size_t lo = std::min(n_, per * i), hi = i == n_threads_ - 1 ? n_ : std::min(n_, per * (i + 1));
        if (hi > lo) memcpy(dst_ + lo, src_ + lo, hi - lo);
    }
    void run(int i) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            slice(i);
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    int n_threads_ = 1;
    std::vector<std::thread> workers_;
    std::mutex mu_, user_mu_;
    std::condition_variable cv_, done_cv_;
    uint8_t *dst_ = nullptr;
    const uint8_t *src_ = nullptr;
    size_t n_ = 0;
    int pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};

// page-locked host memory (hipHostMalloc / hipHostRegister, e.g. from dcn_host_alloc) needs no staging
bool is_pinned_host(const void *p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // plain malloc memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// copy `bytes` of host memory to device on the copy stream: directly when the source is page-locked,
// otherwise through the two pinned staging buffers (the host fills one while the other is in flight)
int staged_h2d(dcn_ctx *c, void *d_dst, const void *h_src, uint64_t bytes) {
    const uint8_t *src = (const uint8_t *)h_src;
    uint8_t *dst = (uint8_t *)d_dst;
    if (bytes == 0) return DCN_OK;
    if (is_pinned_host(h_src)) {
        DCN_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->copy_stream));
        return DCN_OK;
    }
    int which = 0;
    for (uint64_t off = 0; off < bytes; off += c->stage_bytes, which ^= 1) {
        uint64_t m = std::min<uint64_t>(c->stage_bytes, bytes - off);
        DCN_HIP(hipEventSynchronize(c->stage_free[which])); // previous copy out of this buffer finished
        HostCopyPool::get().copy(c->h_stage[which], src + off, m);
        DCN_HIP(hipMemcpyAsync(dst + off, c->h_stage[which], m, hipMemcpyHostToDevice, c->copy_stream));
        DCN_HIP(hipEventRecord(c->stage_free[which], c->copy_stream));
    }
    return DCN_OK;
}

int validate_host_batch(const dcn_ctx *c, const uint64_t *offsets, const uint32_t *unit_id, uint32_t n_reads,
                        uint32_t *n_units) {
    if (n_reads > c->max_reads) return dcn_fail(DCN_ERR_CAPACITY, "n_reads exceeds the context's max_batch_reads");
    if (offsets[0] != 0) return dcn_fail(DCN_ERR_ARG, "offsets[0] must be 0");
    for (uint32_t r = 0; r < n_reads; ++r) {
        if (offsets[r + 1] < offsets[r]) return dcn_fail(DCN_ERR_ARG, "offsets must be non-decreasing");
        if (offsets[r + 1] - offsets[r] > 0xFFFFFFF0ull) return dcn_fail(DCN_ERR_ARG, "read longer than 2^32 bases");
    }
    if (offsets[n_reads] > c->max_bases) return dcn_fail(DCN_ERR_CAPACITY, "batch exceeds the context's max_batch_bases");
    if (unit_id) {
        if (n_reads && unit_id[0] != 0) return dcn_fail(DCN_ERR_ARG, "unit_id[0] must be 0");
        for (uint32_t r = 1; r < n_reads; ++r)
            if (unit_id[r] != unit_id[r - 1] && unit_id[r] != unit_id[r - 1] + 1)
                return dcn_fail(DCN_ERR_ARG, "unit_id must stay equal or grow by one");
        *n_units = n_reads ? unit_id[n_reads - 1] + 1 : 0;
    } else {
        *n_units = n_reads;
    }
    return DCN_OK;
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
    if (count == 0 || count > (1ull << 40) || size - pos != 9 * count || check_kw(k, w) != DCN_OK ||
        dcn_device_count(&ndev) != DCN_OK || device < 0 || device >= ndev) {
        close(fd);
        return DCN_OK;
    }
    void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return DCN_OK;
    madvise(map, size, MADV_SEQUENTIAL);
    const uint8_t *src = (const uint8_t *)map + pos;

    const uint64_t CH = std::min<uint64_t>(count, 8ull << 20); // records per chunk (72 MB)
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
    int which = 0;
    for (uint64_t off = 0; off < count && rc == DCN_OK; off += CH, which ^= 1) {
        const uint64_t m = std::min<uint64_t>(CH, count - off);
        if (!hip_ok(hipEventSynchronize(ev[which]), "event wait")) break; // the copy out of this buffer is done
        HostCopyPool::get().copy(h_buf[which], src + 9 * off, 9 * m);
        if (!hip_ok(hipMemcpyAsync(d_raw[which], h_buf[which], 9 * m, hipMemcpyHostToDevice, st_), "hipMemcpyAsync")) break;
        rc = dcn_table_insert_varint9(idx, d_raw[which], m, d_new, d_flags, d_flags + 1, st_);
        if (rc == DCN_OK) hip_ok(hipEventRecord(ev[which], st_), "event record");
    }
    unsigned long long h_new = 0;
    uint32_t h_flags[2] = {0, 0};
    if (rc == DCN_OK) {
        hip_ok(hipStreamSynchronize(st_), "index load");
        hip_ok(hipMemcpy(&h_new, d_new, sizeof h_new, hipMemcpyDeviceToHost), "hipMemcpy");
        hip_ok(hipMemcpy(h_flags, d_flags, sizeof h_flags, hipMemcpyDeviceToHost), "hipMemcpy");
    }
    if (rc == DCN_OK && h_flags[1]) rc = dcn_fail(DCN_ERR_FORMAT, "Failed to deserialise minimizer hash");
    munmap(map, size);
    if (st_) hipStreamSynchronize(st_);
    for (int i = 0; i < 2; ++i) {
        if (h_buf[i]) hipHostFree(h_buf[i]);
        if (d_raw[i]) hipFree(d_raw[i]);
        if (ev[i]) hipEventDestroy(ev[i]);
    }
    if (d_new) hipFree(d_new);
    if (d_flags) hipFree(d_flags);
    if (st_) hipStreamDestroy(st_);
    if (rc != DCN_OK) {
        if (idx && idx->d_slots) hipFree(idx->d_slots);
        delete idx;
        return rc;
    }
    idx->n_keys = h_new;
    idx->has_zero = h_flags[0] != 0;
    *out = idx;
    *handled = true;
    return DCN_OK;
}

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
    int rc = DCN_OK;
    auto fail = [&](int code) {
        free_ctx(c);
        return code;
    };
    if (hipSetDevice(c->device) != hipSuccess) return fail(dcn_fail(DCN_ERR_HIP, "hipSetDevice failed"));
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->stage_free[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->stage_free[1], hipEventDisableTiming) != hipSuccess)
        return fail(dcn_fail(DCN_ERR_HIP, "stream/event creation failed"));
    uint64_t MR = max_batch_reads;
#define A(ptr, count, what)                          \
    if ((rc = dev_alloc(&c->ptr, (count), what)) != DCN_OK) return fail(rc)
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
    A(d_status, 1, "status");
#undef A
    // hit records: sized for the expected long-read density (1 minimizer per 8 windows, half of them hits),
    // grown on demand by the host API / dcn_ctx_reserve_records
    if ((rc = alloc_records(c, std::min<uint64_t>(std::max<uint64_t>(max_batch_bases / 16, 1u << 16), 1ull << 29))) != DCN_OK)
        return fail(rc);
    c->stage_bytes = std::min<uint64_t>(std::max<uint64_t>(max_batch_bases, 4096), 32ull << 20);
    for (int i = 0; i < 2; ++i)
        if (hipHostMalloc((void **)&c->h_stage[i], c->stage_bytes, hipHostMallocDefault) != hipSuccess)
            return fail(dcn_fail(DCN_ERR_NOMEM, "pinned staging allocation failed"));
    if (hipHostMalloc((void **)&c->h_status, sizeof(dcn_status), hipHostMallocDefault) != hipSuccess)
        return fail(dcn_fail(DCN_ERR_NOMEM, "pinned status allocation failed"));
    // zero padding in front of / behind the packed stream is written once; pack only touches the middle
    if (hipMemset(c->d_packed, 0, packed_words(max_batch_bases) * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(c->d_invmask, 0, mask_words(max_batch_bases) * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(c->d_status, 0, sizeof(dcn_status)) != hipSuccess)
        return fail(dcn_fail(DCN_ERR_HIP, "hipMemset failed"));
    *out = c;
    return DCN_OK;
}

extern "C" void dcn_ctx_destroy(dcn_ctx *ctx) { free_ctx(ctx); }

extern "C" void *dcn_ctx_stream(dcn_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int dcn_ctx_reserve_records(dcn_ctx *ctx, uint64_t n_records) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
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
    DCN_HIP(hipSetDevice(ctx->device));
    return enqueue_batch(ctx, d_bases, d_offsets, d_unit_id, n_reads, n_bases, n_units, params, d_keep, d_hits, d_total);
}

extern "C" int dcn_filter_batch(dcn_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, const uint32_t *unit_id,
                                uint32_t n_reads, const dcn_params *params, uint8_t *keep, uint32_t *hits,
                                uint32_t *total) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    auto t_enter = std::chrono::steady_clock::now();
    DCN_TRY(check_params(params));
    if (n_reads == 0) return DCN_OK;
    if (!offsets || !keep) return dcn_fail(DCN_ERR_ARG, "offsets/keep is NULL");
    uint32_t n_units = 0;
    DCN_TRY(validate_host_batch(ctx, offsets, unit_id, n_reads, &n_units));
    uint64_t n_bases = offsets[n_reads];
    if (n_bases > 0 && !bases) return dcn_fail(DCN_ERR_ARG, "bases is NULL");
    DCN_HIP(hipSetDevice(ctx->device));
    static const bool timing = getenv("DCN_HOST_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    auto t_stage = now();
    // stage inputs on the copy stream; the compute stream waits for the last copy
    DCN_TRY(staged_h2d(ctx, ctx->d_ascii, bases, n_bases));
    DCN_TRY(staged_h2d(ctx, ctx->d_offsets, offsets, (uint64_t)(n_reads + 1) * sizeof(uint64_t)));
    if (unit_id) DCN_TRY(staged_h2d(ctx, ctx->d_unit_id, unit_id, (uint64_t)n_reads * sizeof(uint32_t)));
    DCN_HIP(hipEventRecord(ctx->copy_done, ctx->copy_stream));
    DCN_HIP(hipStreamWaitEvent(ctx->stream, ctx->copy_done, 0));
    auto t_run = now();
    for (int attempt = 0;; ++attempt) {
        // without hit counts / totals the kernels only have to fix the decisions (early-out, see enqueue_batch)
        DCN_TRY(enqueue_batch(ctx, ctx->d_ascii, ctx->d_offsets, unit_id ? ctx->d_unit_id : nullptr, n_reads, n_bases,
                              n_units, params, ctx->d_keep, (hits || total) ? ctx->d_hits : nullptr,
                              (hits || total) ? ctx->d_total : nullptr));
        uint64_t need = 0;
        int rc = sync_and_check(ctx, &need);
        if (rc == DCN_OK) break;
        if (rc != DCN_ERR_CAPACITY || attempt >= 3) return rc;
        // grow the record scratch and run the batch again (inputs are still resident)
        uint64_t want = std::max<uint64_t>(need + need / 8 + 1024, ctx->rec_capacity * 2);
        DCN_TRY(alloc_records(ctx, std::min<uint64_t>(want, 1ull << 29)));
        // (the finish kernel skips the counters of an overflowed attempt, so nothing is double counted)
    }
    auto t_back = now();
    DCN_HIP(hipMemcpy(keep, ctx->d_keep, n_units, hipMemcpyDeviceToHost));
    if (hits) DCN_HIP(hipMemcpy(hits, ctx->d_hits, (uint64_t)n_units * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (total) DCN_HIP(hipMemcpy(total, ctx->d_total, (uint64_t)n_units * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (timing)
        fprintf(stderr, "dcn_filter_batch: validate %.2f ms, stage+enqueue copies %.2f ms, copies drain+kernels %.2f ms, "
                        "results back %.2f ms\n", ms(t_enter, t_stage), ms(t_stage, t_run), ms(t_run, t_back), ms(t_back, now()));
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
    ctx->profiling = enable != 0;
    return DCN_OK;
}

extern "C" int dcn_ctx_profile(dcn_ctx *ctx, double stage_ms[DCN_N_STAGES], uint64_t *n_batches) {
    if (!ctx || !stage_ms) return dcn_fail(DCN_ERR_ARG, "ctx/stage_ms is NULL");
    for (int j = 0; j < DCN_N_STAGES; ++j) stage_ms[j] = ctx->prof_ms[j];
    if (n_batches) *n_batches = ctx->prof_batches;
    return DCN_OK;
}

extern "C" int dcn_ctx_stats(dcn_ctx *ctx, uint64_t counters[DCN_N_STATS]) {
    if (!ctx || !counters) return dcn_fail(DCN_ERR_ARG, "ctx/counters is NULL");
    DCN_HIP(hipSetDevice(ctx->device));
    DCN_HIP(hipStreamSynchronize(ctx->stream));
    DCN_HIP(hipMemcpy(ctx->h_status, ctx->d_status, sizeof(dcn_status), hipMemcpyDeviceToHost));
    for (int i = 0; i < DCN_N_STATS; ++i) counters[i] = ctx->h_status->stats[i];
    return DCN_OK;
}

extern "C" int dcn_ctx_reset_stats(dcn_ctx *ctx) {
    if (!ctx) return dcn_fail(DCN_ERR_ARG, "ctx is NULL");
    DCN_HIP(hipSetDevice(ctx->device));
    DCN_HIP(hipStreamSynchronize(ctx->stream));
    DCN_HIP(hipMemset(ctx->d_status->stats, 0, sizeof(unsigned long long) * DCN_N_STATS));
    return DCN_OK;
}

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
    uint32_t n_units = 0;
    DCN_TRY(validate_host_batch(ctx, offsets, nullptr, n_reads, &n_units));
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
    DCN_TRY(staged_h2d(c, c->d_ascii, bases, n_bases));
    DCN_TRY(staged_h2d(c, c->d_offsets, offsets, (uint64_t)(n_reads + 1) * sizeof(uint64_t)));
    DCN_HIP(hipEventRecord(c->copy_done, c->copy_stream));
    DCN_HIP(hipStreamWaitEvent(c->stream, c->copy_done, 0));
    hipStream_t st = c->stream;
    DCN_HIP(hipMemsetAsync(c->d_status, 0, offsetof(dcn_status, stats), st));
    uint32_t *packed = c->d_packed + DCN_FRONT_PAD, *invmask = c->d_invmask + DCN_FRONT_PAD;
    DCN_TRY(dcn_launch_pack(c->d_ascii, n_bases, packed, invmask, st));
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
    pa.status = c->d_status;
    DCN_TRY(dcn_launch_plan(pa, st));
    dcn_scan_args sa;
    memset(&sa, 0, sizeof(sa));
    sa.packed = packed;
    sa.invmask = invmask;
    sa.tiles = c->d_tiles;
    sa.n_tiles = &c->d_status->n_tiles;
    sa.table = c->index->view();
    sa.k = c->index->k;
    sa.w = c->index->w;
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
            uint64_t base = tiles[t].scan_start + (tiles[t].flags & 1u);
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
    if (hash_offsets[0] != 0) return dcn_fail(DCN_ERR_ARG, "hash_offsets[0] must be 0");
    for (uint32_t u = 0; u < n_units; ++u) {
        if (hash_offsets[u + 1] < hash_offsets[u]) return dcn_fail(DCN_ERR_ARG, "hash_offsets must be non-decreasing");
        if (hash_offsets[u + 1] - hash_offsets[u] > 0xFFFFFFF0ull) return dcn_fail(DCN_ERR_ARG, "unit has more than 2^32 hashes");
    }
    uint64_t n_hashes = hash_offsets[n_units];
    if (n_hashes > 0 && !hashes) return dcn_fail(DCN_ERR_ARG, "hashes is NULL");
    dcn_ctx *c = ctx;
    DCN_HIP(hipSetDevice(c->device));
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
        DCN_HIP(hipMemsetAsync(c->d_status, 0, offsetof(dcn_status, stats), st));
        DCN_HIP(hipMemsetAsync(c->d_unit_state, 0, n_units, st));
        DCN_HIP(hipMemsetAsync(c->d_unit_scratch, 0, (uint64_t)c->max_reads * 4 * sizeof(uint32_t), st));
        uint32_t *g_total = c->d_unit_scratch, *g_hitcnt = g_total + c->max_reads,
                 *g_distinct = g_hitcnt + c->max_reads, *g_zero = g_distinct + c->max_reads;
        dcn_probe_hashes_args pa;
        pa.table = c->index->view();
        pa.hashes = d_hashes;
        pa.hash_offsets = d_hoff;
        pa.n_hashes = n_hashes;
        pa.n_units = n_units;
        pa.g_total = g_total;
        pa.g_hitcnt = g_hitcnt;
        pa.rec_unit = c->d_rec_unit;
        pa.rec_hash = c->d_rec_hash;
        pa.rec_capacity = c->rec_capacity;
        pa.status = c->d_status;
        DCN_TRY(dcn_launch_probe_hashes(pa, st));
        dcn_distinct_args da;
        da.rec_unit = c->d_rec_unit;
        da.rec_hash = c->d_rec_hash;
        da.rec_capacity = c->rec_capacity;
        da.g_hitcnt = g_hitcnt;
        da.g_distinct = g_distinct;
        da.g_zero = g_zero;
        da.set_off = c->d_set_off;
        da.set_slots = c->d_set_slots;
        da.set_capacity = 4 * c->rec_capacity + 64;
        da.n_units = n_units;
        da.status = c->d_status;
        da.caps = c->d_caps;
    DCN_TRY(dcn_launch_distinct(da, st));
        dcn_finish_args fa;
        fa.n_units = n_units;
        fa.unit_first_read = nullptr;
        fa.offsets = nullptr; // no read lengths here: the counters are untouched
        fa.unit_state = c->d_unit_state;
        fa.g_total = g_total;
        fa.g_distinct = g_distinct;
        fa.abs_threshold = params->abs_threshold;
        fa.rel_threshold = params->rel_threshold;
        fa.deplete = params->deplete;
        fa.keep = c->d_keep;
        fa.hits = c->d_hits;
        fa.total = c->d_total;
        fa.status_stats = c->d_status->stats;
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
            DCN_HIP(hipMemsetAsync(c->d_status, 0, offsetof(dcn_status, stats), st));
            DCN_HIP(hipMemsetAsync(c->d_dump_valid, 0, nb + 2, st));
            uint32_t *packed = c->d_packed + DCN_FRONT_PAD, *invmask = c->d_invmask + DCN_FRONT_PAD;
            DCN_TRY(dcn_launch_pack(c->d_ascii, nb, packed, invmask, st, /*index_side=*/true));
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
            sa.w = idx->w;
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
