// dcn_plan.h -- argument blocks of the planning / distinct / finish kernels (plan.hip).
#pragma once

#include "dcn_internal.h"

struct dcn_plan_args {
    const uint8_t *ascii;
    const uint64_t *offsets;
    const uint32_t *unit_id; // may be null: unit == read
    uint32_t unit_base;      // subtracted from every unit_id entry (a chunk of a larger batch); 0 otherwise
    uint32_t n_reads, n_units;
    uint32_t k, w;
    uint64_t prefix_length;
    uint32_t tile_windows;
    uint32_t *read_tiles;       // n_reads: tiles of each read (may be null together with read_tile_first)
    uint32_t *read_tile_first;  // n_reads: first tile of each read (tile ranges of different reads never overlap)
    uint32_t *unit_first_read;  // n_units + 1 (only written when unit_id != null)
    uint32_t *unit_tile_first;  // n_units: first tile of the unit
    uint32_t *unit_tile_count;  // n_units: its tile count; 0xFFFFFFFF = tiles not contiguous (never resolved in-wave)
    uint8_t *unit_state;        // n_units, cleared here (1 = resolved by the scan kernel)
    uint32_t *unit_scratch;     // g_total | g_hitcnt | g_distinct | g_zero, scratch_stride entries each (zero between batches: finish_kernel)
    uint32_t scratch_stride;
    dcn_tile *tiles;
    uint32_t *tile_read_pos; // null, or per tile the position of its scan_start in its read (minimizer dump)
    uint32_t *tile_cursor;      // global tile counter (= &status->n_tiles, zeroed per batch)
    dcn_status *status;
    uint64_t stream_bases;      // check_offsets: every read must lie inside [0, stream_bases)
    uint32_t max_tiles;         // check_offsets: capacity of tiles[] (offsets that pass read by read can still overlap)
    uint32_t check_offsets;     // 1: a read whose offsets are decreasing or beyond the stream is planned as empty and reported
                                // (status->bad_offsets) instead of being followed outside the batch's buffers
    const uint32_t *newline_flag; // null: status->any_newline; else the word the pack kernel of this batch wrote (it may have
                                  // run ahead of the batch's own status words: api.hip, pack one batch ahead)
};

struct dcn_distinct_args {
    const dcn_tile *tiles;
    const uint32_t *n_tiles;         // device-side tile count
    const uint32_t *unit_tile_first; // n_units
    const uint32_t *unit_tile_count; // n_units; 0xFFFFFFFF: not contiguous (hit count in g_hitcnt, always a global set)
    const uint8_t *unit_state;       // 1 = finished by the scan kernel
    const uint32_t *tile_hits;       // per tile: length of the run that starts at it
    const uint64_t *rec_hash;        // runs of hit hashes, 0 = no entry
    uint32_t rec_shift;              // a run starts at slot (scan_start + carry of its first tile) >> rec_shift
    const uint32_t *pending;         // work list: status->n_pending units
    const uint32_t *g_hitcnt;        // per unit: total run length
    uint32_t *g_distinct;
    uint32_t *set_off; // n_units: first slot of a unit's global set (only for units with caps != 0)
    uint32_t *caps;    // n_units: global set size (power of two), 0 = none (counted in LDS, or no hits)
    uint2 *big;        // pass B's work items: (unit, first of 64 tiles), status->n_big of them
    uint64_t *set_slots;
    uint64_t set_capacity;
    uint32_t n_units;
    dcn_status *status;
    // decisions only (the caller takes neither hit counts nor totals, src/local_filter.rs:350-371): a unit's count may stop
    // at the hits its decision needs.  g_total is complete when this pass runs, so `required` is known per unit.
    const uint32_t *g_total; // null: count everything
    uint64_t abs_threshold;
    double rel_threshold;
};

struct dcn_finish_args {
    uint32_t n_units;
    const uint32_t *unit_first_read; // null: unit == read
    const uint64_t *offsets;         // null: no counters (hash seam)
    const uint8_t *unit_state;
    uint32_t *g_total, *g_hitcnt, *g_distinct, *g_zero; // read for undecided units, and put back to zero
    uint64_t abs_threshold;
    double rel_threshold;
    uint32_t deplete;
    uint8_t *keep;
    uint32_t *hits, *total;
    dcn_batch_report *report; // counters and the sticky overflow word of the batch this chunk belongs to
    const dcn_status *status;
};

struct dcn_probe_hashes_args {
    dcn_table_view table;
    uint64_t *hashes; // device copy of the request's hashes; misses are overwritten with 0
    const uint64_t *hash_offsets;
    uint64_t n_hashes;
    uint32_t n_units;
    // one pseudo-tile per unit, so that the distinct pass of the scan path serves this seam too
    dcn_tile *tiles;
    uint32_t *n_tiles;
    uint32_t *tile_hits, *unit_tile_first, *unit_tile_count, *pending;
    uint8_t *unit_state;
    uint32_t *g_total, *g_hitcnt, *g_distinct, *g_zero;
    dcn_status *status;
};

int dcn_launch_plan(const dcn_plan_args &a, hipStream_t stream);
int dcn_launch_distinct(const dcn_distinct_args &a, hipStream_t stream);
int dcn_launch_finish(const dcn_finish_args &a, hipStream_t stream);
int dcn_launch_probe_hashes(const dcn_probe_hashes_args &a, hipStream_t stream);
