/*
 * deacon_hip.h -- C ABI of the MI355X-native read-filtering core for Deacon.
 *
 * This is the drop-in boundary for ONE path of the reference (crate deacon 0.10.0): the per-read
 * "pack -> canonical minimizer scan -> k-mer hash -> index probe -> distinct-hit count -> threshold"
 * loop that `deacon filter` runs on CPU worker threads.  The reference has no FFI of its own; each
 * entry point below names the Rust item (file:line under the reference's src/) it replaces.  A Rust
 * maintainer binds them with a plain `extern "C"` block.
 *
 * Conventions
 *   - every function returns DCN_OK (0) or a negative DCN_ERR_* code and never aborts;
 *     dcn_last_error() returns a thread-local message for the last failure on this thread;
 *   - plain pointers and sizes only; "host" pointers are ordinary process memory, "device" pointers are
 *     HIP device allocations on the context's GPU;
 *   - a dcn_index is immutable after creation and may be shared by any number of contexts/threads
 *     (the reference shares its set through an Arc: src/local_filter.rs:156,631);
 *   - a dcn_ctx owns one HIP stream pair plus staging/scratch buffers and is NOT thread-safe: use one
 *     per host thread (the reference clones one FilterProcessor per worker: src/local_filter.rs:153).
 */
#ifndef DEACON_HIP_H
#define DEACON_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCN_OK 0
#define DCN_ERR_ARG (-1)      /* invalid argument (NULL, k/w out of range, k+w-1 even, ...) */
#define DCN_ERR_HIP (-2)      /* a HIP runtime call failed (message has the HIP error string) */
#define DCN_ERR_NOMEM (-3)    /* host or device allocation failed */
#define DCN_ERR_IO (-4)       /* index file could not be opened / read */
#define DCN_ERR_FORMAT (-5)   /* index file is not a format-version-2 deacon index */
#define DCN_ERR_CAPACITY (-6) /* batch larger than the context was created for, or output too small */
#define DCN_ERR_INTERNAL (-7)

typedef struct dcn_index dcn_index; /* device-resident minimizer set: replaces Arc<FxHashSet<u64>> */
typedef struct dcn_ctx dcn_ctx;     /* per-thread pipeline context */

/* Filter-time parameters: the fields of FilterProcessor that the decision depends on
 * (src/local_filter.rs:159-162; CLI defaults -a 2 -r 0.01 -p 0, src/main.rs:43-60).
 * k and w are NOT here: like the reference they come from the index header
 * (src/local_filter.rs:633-634). */
typedef struct dcn_params {
    uint64_t abs_threshold; /* -a: minimum absolute number of distinct minimizer hits */
    double rel_threshold;   /* -r: minimum hits relative to the unit's minimizer count */
    uint64_t prefix_length; /* -p: 0 = whole read, else only the first prefix_length bases */
    uint32_t deplete;       /* -d: 0 = keep matching units, 1 = keep non-matching units */
    uint32_t reserved;      /* must be 0 */
} dcn_params;

/* The six ProcessingStats counters of src/local_filter.rs:179-187, in that order. */
enum {
    DCN_STAT_TOTAL_SEQS = 0,
    DCN_STAT_FILTERED_SEQS = 1,
    DCN_STAT_TOTAL_BP = 2,
    DCN_STAT_OUTPUT_BP = 3,
    DCN_STAT_FILTERED_BP = 4,
    DCN_STAT_OUTPUT_SEQ_COUNTER = 5,
    DCN_N_STATS = 6
};

/* ---- library ------------------------------------------------------------------------------------ */
/* ABI version of this header.  MAJOR changes whenever the signature of an exported function or the layout of a struct
 * changes (a binding built against another major must refuse to run: its calls would pass the wrong arguments), MINOR
 * when entry points are added.  History: 0.x = the headers before versioning (dcn_pack_ascii took four arguments there);
 * 1.0 = dcn_pack_ascii(bases, n_bases, packed, invmask, saw_newline); 1.1 = dcn_abi_version, dcn_comm_* / dcn_stats_allreduce_rccl. */
#define DCN_ABI_MAJOR 1
#define DCN_ABI_MINOR 1
/* What the loaded library was built as: a binding asserts *major == DCN_ABI_MAJOR it was written against and
 * *minor >= the minor it needs, before its first other call (no reference counterpart: the reference is one crate). */
int dcn_abi_version(uint32_t *major, uint32_t *minor);
const char *dcn_version(void);
const char *dcn_last_error(void);
int dcn_device_count(int *count);

/* ---- parity-pinning switch ------------------------------------------------------------------------------
 * The minimizer rule itself lives in a crate the reference only calls: simd_minimizers::canonical_minimizer_positions
 * (src/filter_common.rs:261-267, src/minimizers.rs:143-148; simd-minimizers 1.3.0 in Cargo.lock:1954-1957).  Three
 * of its details could not be executed where this library was written ("Notes on A4"): the ntHash
 * rotation per base (1, or 7 as in the crate's later line), how many hash bits the window minimum compares (the top
 * 16, or all 32) and how the two strands' hashes are combined (wrapping add, or xor).  The defaults are (1, 16, 0).
 * The setting is process-wide and is CAPTURED by every index when it is created (built, loaded, merged, cloned): the
 * rule an index's keys were selected by travels with it, every context filters by its index's rule whatever the
 * setting has become since, and union / diff refuse operands created under different rules.  Set it before the first
 * index is built or loaded.  Any other setting than the default runs a generic kernel (slower, counting mode only); tests/golden/
 * dump_crate_vectors prints vectors from the real crates, and tests/test_crate_vectors.py names the setting that
 * reproduces them. */
int dcn_set_minimizer_variant(uint32_t nt_rot, uint32_t cmp_bits, uint32_t combine /* 0: fw + rc, 1: fw ^ rc */);
int dcn_get_minimizer_variant(uint32_t *nt_rot, uint32_t *cmp_bits, uint32_t *combine);

/* ---- index: replaces index::load_minimizer_hashes (src/index.rs:80-107) ----------------------------- */

/* Build the device set from `n` host u64 minimizer hashes (duplicates allowed; they are merged, as by
 * FxHashSet::insert at src/index.rs:104).  k, w as in IndexHeader (src/index.rs:17-31): 1<=k<=56
 * (the filter path asserts k<=56 at src/filter_common.rs:269), w>=1, k+w-1 odd (src/index.rs:186-194). */
int dcn_index_from_keys(const uint64_t *keys, uint64_t n, uint8_t k, uint8_t w, int device, dcn_index **out);

/* Load a deacon index file (bincode 2 "standard" varint layout written by write_minimizers,
 * src/index.rs:130-164; header validation as IndexHeader::validate, src/index.rs:34-43). */
int dcn_index_from_file(const char *path, int device, dcn_index **out);

/* Build the index from sequences held in host memory: index::build (src/index.rs:167-308) minus FASTX parsing.
 * Every sequence goes through the index-side rule (minimizers::fill_minimizer_hashes, src/minimizers.rs:125-191:
 * IUPAC codes canonicalised to ACGT before the scan, minimizers whose k-mer has a non-ACGT base in the ORIGINAL
 * bytes dropped, optional scaled-entropy floor) and the hashes are merged into one device set.
 *   bases / offsets   concatenated sequences and n_seqs+1 byte offsets, as for dcn_filter_batch
 *   entropy_threshold 0 = off (the reference's default, src/lib.rs:224)
 *   capacity_keys     pre-allocation hint (IndexConfig::capacity_millions, src/lib.rs:200); 0 = grow as needed */
int dcn_index_build(const uint8_t *bases, const uint64_t *offsets, uint32_t n_seqs, uint8_t k, uint8_t w,
                    float entropy_threshold, uint64_t capacity_keys, int device, dcn_index **out);

/* Copy the distinct keys to host memory in arbitrary order (the reference iterates its FxHashSet, src/index.rs:159);
 * *n receives the key count; DCN_ERR_CAPACITY if capacity is smaller. */
int dcn_index_keys(const dcn_index *index, uint64_t *out, uint64_t capacity, uint64_t *n);

/* Write the index in the reference's file format: write_minimizers (src/index.rs:130-164). */
int dcn_index_write_file(const dcn_index *index, const char *path);

/* Set union of n >= 1 indexes on the same device: index::union (src/index.rs:563-664).  All inputs must have the
 * same k and w (the reference refuses otherwise, :611-626). */
int dcn_index_union(const dcn_index *const *inputs, uint32_t n, dcn_index **out);

/* Set difference first \ second: index::diff (src/index.rs:421-536); k and w must match (:478-490). */
int dcn_index_diff(const dcn_index *first, const dcn_index *second, dcn_index **out);

/* Header fields and the number of DISTINCT keys (what `deacon index info` prints, src/index.rs:539-560). */
int dcn_index_header(const dcn_index *index, uint8_t *k, uint8_t *w, uint64_t *n_keys);

/* The HIP device the index lives on. */
int dcn_index_device(const dcn_index *index, int *device);

/* Bytes of device memory the index's hash table occupies (a power-of-two number of 16-byte groups with at least
 * four slots per key, eight while that stays within 34 GB and a third of the device's free memory:
 * DESIGN.md section 3; the reference's FxHashSet<u64> is ~9-18 bytes per key of host memory). */
int dcn_index_memory(const dcn_index *index, uint64_t *table_bytes);

/* Set membership for `n` host keys -> out[i] in {0,1}: FxHashSet::contains (src/filter_common.rs:144). */
int dcn_index_contains(const dcn_index *index, const uint64_t *keys, uint64_t n, uint8_t *out);

/* Same for keys already in device memory (d_keys, d_out are DEVICE pointers on the index's GPU); enqueued on
 * `stream` (a hipStream_t, NULL = default stream) without waiting. */
int dcn_index_contains_device(const dcn_index *index, const uint64_t *d_keys, uint64_t n, uint8_t *d_out,
                              void *stream);

/* Measurement: the rate (home-group reads per second) at which this table serves a stream of keys when nothing else
 * runs -- every key's home group is read (one 16-byte request), nothing is resolved or written; the best of several
 * launch shapes over `reps` repetitions.  d_keys (DEVICE pointer) is the stream to replay, e.g. a batch's minimizer
 * hashes; NULL probes n uniformly random groups instead (every request an L2 miss).  This is the ceiling the probe
 * stage of the filter kernel is measured against (bench.py: roofline.probe_ceiling_*; the counterpart of timing
 * FxHashSet::contains alone, src/filter_common.rs:144).  Blocks until done. */
int dcn_index_probe_ceiling(const dcn_index *index, const uint64_t *d_keys, uint64_t n, uint32_t reps,
                            double *probes_per_s);

/* Replica of an index on another (or the same) device, made device to device: the reference shares ONE set between its
 * workers through an Arc (src/local_filter.rs:630-631); a multi-GPU host loads or builds the index once and clones it to
 * every other device instead of repeating the host-to-device copy.  To another device the compacted KEYS cross xGMI
 * (hipMemcpyPeer of 8 bytes per key, a tenth of the sparse table) and are inserted into an empty table there; on the same
 * device the table itself is copied.  Same key set, k, w and minimizer rule either way. */
int dcn_index_clone(const dcn_index *index, int device, dcn_index **out);

void dcn_index_destroy(dcn_index *index);

/* ---- context -------------------------------------------------------------------------------------- */

/* max_batch_bases / max_batch_reads bound one call of dcn_filter_batch*; buffers are sized once here. */
int dcn_ctx_create(const dcn_index *index, uint64_t max_batch_bases, uint32_t max_batch_reads, dcn_ctx **out);
void dcn_ctx_destroy(dcn_ctx *ctx);

/* ---- the hot path -------------------------------------------------------------------------------- */

/* Filter one batch of reads held in host memory.  Replaces, for every unit of the batch,
 * FilterProcessor::should_keep_sequence (src/local_filter.rs:221-252) or ::should_keep_pair (:254-285),
 * i.e. get_minimizer_hashes_and_positions (src/filter_common.rs:211-310) + sequence_matches /
 * pair_matches (:129-198) + meets_filtering_criteria (:99-112).
 *
 *   bases    concatenated ASCII sequences (record.seq() bytes, no separators), offsets[n_reads] bytes
 *   offsets  n_reads+1 byte offsets into `bases`, offsets[0] == 0, non-decreasing
 *   unit_id  NULL: every read is its own unit.  Otherwise n_reads entries, unit_id[0]==0, each
 *            entry equal to its predecessor or predecessor+1: consecutive reads with equal id form
 *            one unit (a pair: mate 1 then mate 2 -- src/filter_common.rs:312-348)
 *   keep     out, one byte per unit: 1 = write the unit's records to the output, 0 = drop
 *   hits     out (may be NULL), distinct minimizer hits per unit
 *   total    out (may be NULL), minimizer count per unit (duplicates included) -- the
 *            (bool, usize, usize) of should_keep_*
 * With hits == NULL and total == NULL only the decisions are produced, which is all `deacon filter` consumes
 * outside --debug (src/local_filter.rs:350-371): the kernels may then stop probing a unit as soon as its
 * decision is fixed (abs_threshold distinct hits reached while the relative threshold cannot ask for more).
 * keep and the six counters are identical in both forms.
 *
 * Blocking (= dcn_filter_batch_submit + dcn_filter_batch_wait).  Inside, the batch is cut at unit boundaries into
 * chunks of ~64 Mbp (DCN_CHUNK_BASES): chunk i's kernels run while chunk i+1 crosses PCIe on a side stream and chunk i-1's results
 * travel back on a third.  What crosses the link:
 *   pageable memory   host threads pack it to 2 bits + 1 mask bit per base straight into the context's pinned
 *                     staging ring (0.375 instead of 1 byte per base on the link -- the reference packs on the host
 *                     too: src/filter_common.rs:238-258);
 *   page-locked memory (dcn_host_alloc / hipHostRegister)  the same where the host packs fast enough to beat the
 *                     link's 1 byte per base (AVX-512BW hosts), else -- and always with DCN_PINNED_ASCII_DMA=1 -- the
 *                     ASCII is DMA'd as it is and packed on the device (no host work at all).
 * A batch that needed more hit-record scratch than the context has is re-run after growing it: callers never see
 * DCN_ERR_CAPACITY for that. */
int dcn_filter_batch(dcn_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, const uint32_t *unit_id,
                     uint32_t n_reads, const dcn_params *params, uint8_t *keep, uint32_t *hits,
                     uint32_t *total);

/* Asynchronous form: the paraseq workers of the reference overlap reading, filtering and writing across threads
 * (src/local_filter.rs:696-709); here ONE caller thread keeps up to two batches in flight per context.
 * submit validates the batch, enqueues its copies and kernels and returns a ticket; wait(ticket) blocks until the
 * batch is done, delivers keep/hits/total and adds the batch's six counters to the context's.  All input and
 * output arrays must stay valid and unmodified until wait returns, and the CONTENTS of keep/hits/total are undefined
 * until then: results of early chunks are copied out while later chunks are still being prepared, and a batch that
 * has to be run again (a newline found while packing, a run of the record array that overflowed) overwrites them.
 * A third submit without a wait fails with DCN_ERR_CAPACITY.  Tickets may be waited for in any order. */
int dcn_filter_batch_submit(dcn_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, const uint32_t *unit_id,
                            uint32_t n_reads, const dcn_params *params, uint8_t *keep, uint32_t *hits,
                            uint32_t *total, uint64_t *ticket);
int dcn_filter_batch_wait(dcn_ctx *ctx, uint64_t ticket);

/* The same batch handed over already packed, as the reference holds it after PackedSeqVec::from_ascii and its mask
 * loop (src/filter_common.rs:238-258) -- for the concatenated batch instead of one read at a time:
 *   packed   2 bits per base, code (c >> 1) & 3 of the ASCII byte (A=0 C=1 T=2 G=3, non-ACGT mapped the same
 *            lossy way); base i of the batch = bits [2(i%16), +2) of packed[i/16], i.e. bits 2(i%4) of byte i/4:
 *            packed-seq's own byte order
 *   invmask  1 bit per base, bit i%32 of invmask[i/32] set iff byte i is not one of ACGTacgt
 *   offsets / unit_id / outputs as for dcn_filter_batch (offsets are BASE indices into the stream)
 * Both arrays must be allocated in whole 32-base groups: 2 * ceil(n_bases/32) and ceil(n_bases/32) words
 * (dcn_pack_ascii fills them).  Reads must not end in a newline byte, 0x0A (src/filter_common.rs:229 strips one from
 * the ASCII; a packed stream cannot show it: dcn_pack_ascii reports whether it met one).  The pack kernel is skipped;
 * 0.25 bytes per base cross the link, plus the non-zero words of invmask (the rest of it is a memset on the device). */
int dcn_filter_batch_packed(dcn_ctx *ctx, const uint32_t *packed, const uint32_t *invmask, const uint64_t *offsets,
                            const uint32_t *unit_id, uint32_t n_reads, const dcn_params *params, uint8_t *keep,
                            uint32_t *hits, uint32_t *total);
int dcn_filter_batch_packed_submit(dcn_ctx *ctx, const uint32_t *packed, const uint32_t *invmask,
                                   const uint64_t *offsets, const uint32_t *unit_id, uint32_t n_reads,
                                   const dcn_params *params, uint8_t *keep, uint32_t *hits, uint32_t *total,
                                   uint64_t *ticket);

/* Host-side packer producing exactly that layout from concatenated ASCII (AVX2 + BMI2 where the CPU has them,
 * split over the library's host threads, DCN_HOST_THREADS).  Input formatting only: nothing here hashes or
 * decides.  packed / invmask: 2 * ceil(n_bases/32) and ceil(n_bases/32) u32 words.  saw_newline (may be NULL)
 * receives 1 if some byte of the input was 0x0A, else 0: a read that ENDS in one is shortened by the ASCII entry
 * points (src/filter_common.rs:229) and cannot be by the packed ones, so a caller whose record buffers may carry
 * line ends must strip them, or send that batch through dcn_filter_batch, when the flag comes back set. */
int dcn_pack_ascii(const uint8_t *bases, uint64_t n_bases, uint32_t *packed, uint32_t *invmask, uint32_t *saw_newline);

/* Same computation on inputs already resident in device memory (all pointers are DEVICE pointers on the
 * context's GPU; d_unit_id / d_hits / d_total may be NULL).  n_bases = offsets[n_reads], n_units = number
 * of units (n_reads when d_unit_id is NULL).  Enqueues on the context's stream and returns without
 * waiting; call dcn_ctx_synchronize() before reading the outputs.  The arrays are read WHEN THE KERNELS RUN, on the
 * context's stream: a producer on another stream must be ordered before it (event or synchronize).  Nothing on the
 * host has seen d_offsets / d_unit_id, so the planning kernel checks them: a read with offsets[r] > offsets[r+1] or
 * offsets[r+1] > n_bases is planned as empty, unit ids that are not 0, then equal or +1, ending at n_units - 1 are
 * flagged, and the next dcn_ctx_synchronize() returns DCN_ERR_ARG (never tiles or read ranges that point outside
 * the batch's buffers). */
int dcn_filter_batch_device(dcn_ctx *ctx, const uint8_t *d_bases, const uint64_t *d_offsets,
                            const uint32_t *d_unit_id, uint32_t n_reads, uint64_t n_bases, uint32_t n_units,
                            const dcn_params *params, uint8_t *d_keep, uint32_t *d_hits, uint32_t *d_total);

/* Wait for everything enqueued on the context by dcn_filter_batch_device; reports deferred errors of the device
 * pipeline.  DCN_ERR_CAPACITY: the hit-record scratch overflowed in SOME batch enqueued since the previous
 * synchronize (the flag is sticky across batches); results and counters of all of them are then unspecified:
 * dcn_ctx_reserve_records(), dcn_ctx_reset_stats() and enqueue them again.  DCN_ERR_ARG: the device found the
 * d_offsets / d_unit_id of some batch since the previous synchronize inconsistent (see above); outputs and counters
 * of those batches are undefined, the context itself stays usable (reported once). */
int dcn_ctx_synchronize(dcn_ctx *ctx);

/* Grow the scratch that holds (unit, hash) hit records of units spanning several tiles (long reads). */
int dcn_ctx_reserve_records(dcn_ctx *ctx, uint64_t n_records);

/* Raw HIP stream (hipStream_t) the context enqueues on, so a caller can time or order against it. */
void *dcn_ctx_stream(dcn_ctx *ctx);

/* Page-locked host memory for batch buffers.  dcn_filter_batch copies from such memory (or any memory the
 * caller registered with hipHostRegister) straight over PCIe; pageable memory goes through the context's
 * pinned staging buffers first.  Stands for nothing in the reference (its batches never leave the host). */
int dcn_host_alloc(uint64_t bytes, void **out);
void dcn_host_free(void *p);

/* Minimizer hashes and positions of every read of a host batch: the (Vec<u64>, Vec<u32>) of
 * get_minimizer_hashes_and_positions (src/filter_common.rs:211-310), concatenated read by read.
 *   out_offsets  n_reads+1 entries: read r owns [out_offsets[r], out_offsets[r+1]) of the two arrays
 *   capacity     entries available in out_hashes / out_positions; on DCN_ERR_CAPACITY
 *                out_offsets[n_reads] holds the required size */
int dcn_minimizer_hashes_batch(dcn_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint32_t n_reads,
                               uint64_t prefix_length, uint64_t *out_offsets, uint64_t *out_hashes,
                               uint32_t *out_positions, uint64_t capacity);

/* Batch seam of the server engine: unpaired_should_keep / paired_should_keep
 * (src/remote_filter.rs:230-301) -- minimizer hashes precomputed by the client, unit u owns
 * hashes[hash_offsets[u] .. hash_offsets[u+1]) (for a pair: both mates' hashes concatenated).
 * Host pointers; hits/total may be NULL.  prefix_length of params is ignored here. */
int dcn_should_keep_hashes(dcn_ctx *ctx, const uint64_t *hashes, const uint64_t *hash_offsets, uint32_t n_units,
                           const dcn_params *params, uint8_t *keep, uint32_t *hits, uint32_t *total);

/* ---- counters: ProcessingStats (src/local_filter.rs:179-187, merged at :388-396) -------------------------- */

/* Counters accumulated on the device over every dcn_filter_batch* call since the last reset. */
int dcn_ctx_stats(dcn_ctx *ctx, uint64_t counters[DCN_N_STATS]);
int dcn_ctx_reset_stats(dcn_ctx *ctx);

/* Sum of the six counters over n_ctx contexts of THIS process (several devices, or several contexts per device):
 * the merge of the per-worker ProcessingStats at src/local_filter.rs:388-396.  In-process contexts share an address
 * space, so this is a host sum; one-process-per-GPU jobs reduce the same six words with RCCL (bench.py). */
int dcn_stats_allreduce(dcn_ctx *const *ctxs, int n_ctx, uint64_t counters[DCN_N_STATS]);

/* The same merge ACROSS PROCESSES (one process per GPU, reads sharded, index replicated): an RCCL all-reduce(sum) of the
 * six u64 words, the path's only collective -- 48 bytes, once per run.  For a host that is not Python (which reduces them
 * with torch.distributed over RCCL): rank 0 makes an id with dcn_comm_unique_id and hands its 128 bytes to the other ranks
 * by any means of its own (a file, a socket, MPI, the job launcher); every rank then calls dcn_comm_create -- a collective
 * call: it returns when all world_size ranks are in it -- and, at the end of its share of the input,
 * dcn_stats_allreduce_rccl with its contexts (summed on the host first, as dcn_stats_allreduce does).  Every rank gets the
 * job's totals.  RCCL is bound at first use (dlopen librccl.so.1; DCN_RCCL_LIB names another file): dcn_comm_available
 * says whether it can be, without touching a GPU.  The reference's counterpart is the mutex-guarded merge of its
 * worker threads' ProcessingStats (src/local_filter.rs:388-396); it has no multi-process form. */
#define DCN_COMM_ID_BYTES 128
typedef struct dcn_comm dcn_comm;
int dcn_comm_available(void);
int dcn_comm_unique_id(uint8_t id[DCN_COMM_ID_BYTES]);
int dcn_comm_create(const uint8_t id[DCN_COMM_ID_BYTES], int world_size, int rank, int device, dcn_comm **out);
int dcn_stats_allreduce_rccl(dcn_comm *comm, dcn_ctx *const *ctxs, int n_ctx, uint64_t counters[DCN_N_STATS]);
void dcn_comm_destroy(dcn_comm *comm);

/* ---- measurement ------------------------------------------------------------------------------------ */

/* Stages of one batch on the context's stream, timed with HIP events recorded on that stream. */
enum {
    DCN_STAGE_PACK = 0,     /* ASCII -> 2-bit stream + invalid mask */
    DCN_STAGE_PLAN = 1,     /* effective lengths, prefix sum, tile descriptors */
    DCN_STAGE_SCAN = 2,     /* minimizer scan + k-mer hash + index probe + in-wave distinct count (dominant) */
    DCN_STAGE_DISTINCT = 3, /* exact distinct count for units spanning several waves */
    DCN_STAGE_FINISH = 4,   /* decisions of those units + the six counters */
    DCN_N_STAGES = 5
};

/* enable 1: record events around every stage of every following batch (and clear the accumulators); 2: around the
 * scan stage only (two marker packets per batch instead of six: the form a throughput measurement can afford);
 * 0: off. */
int dcn_ctx_set_profiling(dcn_ctx *ctx, int enable);

/* Accumulated device time per stage in milliseconds and the number of batches measured, for batches that
 * have completed (call after dcn_ctx_synchronize). */
int dcn_ctx_profile(dcn_ctx *ctx, double stage_ms[DCN_N_STAGES], uint64_t *n_batches);

#ifdef __cplusplus
}
#endif
#endif /* DEACON_HIP_H */
