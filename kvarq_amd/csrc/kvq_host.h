// kvarq_amd/csrc/kvq_host.h -- host-side internals of libkvarq_hip.so
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/kvarq_hip.h"
#include "kvq_device.h"

// ---- error state (thread local) ----------------------------------------------
void kvq_set_error(int code, const char *fmt, ...);
void kvq_clear_error();
int  kvq_error_code();

#define KVQ_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            kvq_set_error(KVQ_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return KVQ_ERR_DEVICE;                                                            \
        }                                                                                     \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    int ensure(size_t n);          // KVQ_OK / error code; contents are NOT preserved
    void release();
    template <class T> T *as() const { return (T *)p; }
};

// per-batch tables (chunk offsets, tile maps): pinned host memory mirrored by device
// memory, bump-allocated so that a batch can be enqueued without waiting for the previous one
struct TablePool {
    uint8_t *h = nullptr, *d = nullptr; size_t cap = 0, used = 0, h_cap = 0, d_cap = 0;
    int reserve(size_t bytes, hipStream_t stream);   // makes room (may synchronise the stream when growing)
    size_t take(size_t bytes) { const size_t at = used; used += (bytes + 255) & ~(size_t)255; return at; }
    void release();
};

struct kvq_table {
    kvq_config cfg;
    int32_t nseq = 0;
    int64_t bases = 0;
    std::vector<uint8_t> h_tab;          // concatenated bytes
    std::vector<int32_t> h_off;          // nseq + 1
    std::vector<int32_t> exhaustive;     // sequences the exhaustive kernel serves
    std::vector<int32_t> seeded;         // sequences the seed-filter kernel serves
    std::vector<uint8_t> is_seeded;
    int32_t seed_k = 0;
    int device = 0;                      // the device the table's blocks live on
    DevBuf d_tab, d_off, d_exh, d_all, d_seeded;
    struct SeedIndex *index = nullptr;   // kernels_seeded
    int64_t ctr_len, off_nseqhits, off_nseqbasehits, off_cov, off_mut;
};

// where each array of a finished scan lives inside the result buffer (device copy and pinned host copy alike)
struct KvqResultLayout {
    size_t file_pos = 0, hitseq_off = 0, seq_nr = 0, seq_pos = 0, length = 0, readlength = 0, blob = 0, total = 0;
};

struct Batch {
    const uint8_t *d_data; int64_t nbytes; int64_t fpos_base;
    std::vector<int64_t> chunk_off;
    bool redone = false;      // its seed-filter pass failed validation; an exhaustive redo batch follows
    bool is_redo = false;     // this batch is such a redo (or the redo of another batch's skipped tiles)
    bool skips_done = false;  // its skipped tiles have been scanned again
    size_t skip_at = 0;       // its list of skipped tiles in the table pool
    uint32_t tile_bytes = 0;  // bytes a tile owned when it was scanned
};

struct kvq_comm;
#define KVQ_REDO_CAP 16384u            // records that the skipped tiles of one launch may leave (beyond: the batch is redone as a whole)
// The scan kernel's survivors (round 4): a work item that passed the 16-base test -- a true hit as a rule, 0.02 per read -- is put on
// this list instead of being verified where it was found; kvq_verify_survivors, right behind the scan kernel, does the byte-exact
// part for all of them at once (a lane each).  Where it was found it cost its wave some twenty dependent trips to memory, and the
// other seven waves of the tile the wait for it: 6 % of the kernel's time for 0.003 hits per read.  A full list costs nothing but
// speed: the items that do not fit are verified in place, as before.
struct KvqSurvivor { uint32_t boff; uint16_t rl, p; uint64_t en; uint32_t kind, pad; };      // batch offset of the trimmed read, its length, read position of the seed, index entry, which index
#define KVQ_SURV_CAP (1u << 22)
#define KVQ_SURV_CHUNK 16u             // a wave takes the list's slots a chunk at a time (one atomic on the list's counter per chunk, not per survivor:
                                       // a word in memory takes ~88 atomics per microsecond, hit-dense input would queue up on it)
struct KvqSurvivors {
    unsigned int *count; KvqSurvivor *item;
    static size_t bytes() { return 256 + (size_t)KVQ_SURV_CAP * sizeof(KvqSurvivor); }
    __host__ __device__ explicit KvqSurvivors(void *p) { count = (unsigned int *)p; item = (KvqSurvivor *)((char *)p + 256); }
};
struct KvqRedo {                      // where the pieces lie inside kvq_scan::d_redo
    unsigned int *count; uint32_t *nl4, *rec_start, *read_off; int32_t *read_len;
    static size_t bytes() { return 256 + (size_t)KVQ_REDO_CAP * (16 + 4 + 4 + 4); }
    __host__ __device__ explicit KvqRedo(void *p) { char *c = (char *)p; count = (unsigned int *)c; nl4 = (uint32_t *)(c + 256); rec_start = nl4 + 4 * (size_t)KVQ_REDO_CAP; read_off = rec_start + KVQ_REDO_CAP; read_len = (int32_t *)(read_off + KVQ_REDO_CAP); }
};
int kvq_live_scans();                 // scan objects alive in this process
// the persistent scan kernels of a process run one behind the other (two at once only get in each other's way): a launch waits for
// the event the last one published, on its own stream, right in front of its scan kernel -- its table upload and kvq_expand_tiles do not wait
int kvq_chain_wait(struct kvq_scan *s, bool *behind_a_running_scan = nullptr);
bool kvq_chain_busy(const struct kvq_scan *s);     // a scan kernel of another scan object of the process is still on the device
int kvq_chain_publish(struct kvq_scan *s);
uint32_t kvq_device_cu_count();       // compute units of the current device
int kvq_comm_reduce_counters(kvq_comm *c, const unsigned long long *d_in, unsigned long long *d_out, int64_t ctr_len, unsigned long long *d_scratch, hipStream_t stream);
int kvq_comm_max_status(kvq_comm *c, unsigned long long mine, unsigned long long *d_scratch, hipStream_t stream, unsigned long long *out);

struct kvq_scan {
    const kvq_table *t = nullptr;
    kvq_comm *comm = nullptr;            // several GPUs: `finish` sums the counters of all ranks over it (kvq_dist.hip)
    hipStream_t stream = nullptr;
    bool force_exhaustive = false;
    // counters
    unsigned long long *d_ctr = nullptr; bool own_ctr = false; DevBuf d_ctr_own;
    std::vector<int64_t> h_ctr;
    // per-batch scratch
    uint32_t cus = 0;                  // compute units of the scan's device (asked once)
    bool seen_skips = false;           // a tile of this scan object has left records to the redo before (kept across resets: sizes the redo's launches)
    bool tail_pending = false; size_t tail_nb0 = 0, tail_spec = 0;      // the tail of the scan (enqueue_tail) is on the stream for these batches: kvq_scan_finish_begin
    uint32_t surv_cap = 0;             // slots of d_surv's list (KVQ_SURV_CAP, or what the environment cut it to)
    DevBuf d_surv;                     // what passed the scan kernel's 16-base test, for kvq_verify_survivors (KvqSurvivors)
    DevBuf d_redo;                     // the redo of skipped tiles (KvqRedo: count, newline quadruples, record starts, trimmed reads)
    DevBuf d_skipped, d_chunk_off, d_seg_base, d_seg_cnt, d_chunk_nrec, d_rec_base, d_nl4, d_rec_start, d_read_off, d_read_len;
    // hit arena
    DevBuf d_covdiff;                  // coverage marks (KvqParams::covdiff)
    uint32_t tile_bytes = 0;           // bytes a tile of the seed-filter kernel owns (0 = not chosen yet; kvq_choose_tile)
    uint32_t rec_bytes = 0;            // average record among the first bytes of the text (0 = unknown)
    DevBuf d_arena, d_blob, d_small;   // d_small: arena_n, batch range words, blob_n, err
    uint32_t arena_cap = 0; uint64_t blob_cap = 0;
    unsigned int *d_arena_n = nullptr, *d_range = nullptr, *d_fail = nullptr, *cur_fail = nullptr;
    unsigned long long *d_blob_n = nullptr, *d_err = nullptr, *d_err_stage = nullptr, *d_stage_ctr = nullptr;
    int path_bits = 0;
    std::vector<int64_t> cur_chunk_off;  // chunk offsets of the batch being enqueued
    size_t cur_co_at = 0;                // ... and where run_batch put them in the pool
    size_t cur_skip_at = 0, cur_first_at = 0; uint32_t cur_ntiles = 0;   // the batch's list of skipped tiles, its first-tile table
    TablePool pool;
    // staging for host batches
    // Host batches: two staging buffers.  A batch's text crosses PCIe on copy_stream as soon as it is handed over; its
    // kernels are enqueued one call later, when the batch in front of it has been settled -- the copy engine never waits
    // for kernels, the kernels never wait for the host
    DevBuf d_stage, d_stage_b;
    int run_slot = 1;                    // the buffer of the batch whose kernels are in flight (the next text goes to the other one)
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copy[2] = { nullptr, nullptr };
    bool copied_pending = false; int copied_slot = 0; Batch copied;   // the batch whose text is on its way (or there), kernels not yet enqueued
    // replay list (device batches) + bookkeeping
    std::vector<Batch> batches;
    bool host_batches = false;
    int64_t host_pending = -1;           // index of the host batch in flight (kvq_scan_host_async), -1: none
    hipEvent_t ev_chain = nullptr;       // this scan's last seed-filter launch and its kvq_verify_survivors are through (recorded behind the two, in front of kvq_validate_tiles: what the next scan of the process waits for)
    int64_t records = 0;
    int64_t parsed = 0, total = 0;
    // timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_all, ev_main, ev_free;     // ev_free: pairs of earlier scans, reused
    double ms_all = 0, ms_main = 0; int64_t main_launches = 0;
    // results: ordered and laid out on the device (kernels_results.hip), one copy into pinned host memory
    DevBuf d_sort_tmp, d_sorted, d_result, d_order, d_finish;   // d_order: bucket arrays of the ordering; d_finish: KvqFinishState
    uint32_t order_nb_max = 0;                    // the bucket arrays in d_order are laid out for this many buckets
    uint8_t *pin_res = nullptr;                   // where the result arrays start inside pin (behind the counters)
    size_t spec_bytes = 1u << 20;                 // result bytes fetched together with the counters, before their number is known (the last scan's)
    uint8_t *pin = nullptr; size_t pin_cap = 0;   // pinned landing buffer of finish: the result arrays, then the counters
    uint8_t *pin_small = nullptr; size_t pin_small_cap = 0;   // pinned landing buffer for the scan's small words and fail flags
    KvqResultLayout res;                          // where the arrays sit inside pin
    uint64_t n_hits = 0;
    bool finished = false;
    // several ranks (kvq_dist.hip)
    DevBuf d_ctr_all;                             // the counters of all ranks, summed (the rank's own stay in d_ctr: a repeated finish sums them afresh)
    DevBuf d_gather_cnt, d_gather_res;            // kvq_scan_gather_hits: counts of the ranks, the gathered arrays
    bool reduced = false, gathered = false;
};

// kernels_seeded.hip
struct SeedIndex;
SeedIndex *kvq_seed_index_build(kvq_table *t);           // nullptr when no sequence qualifies
void       kvq_seed_index_destroy(SeedIndex *ix);
// enqueue the fused seed-filter scan of one batch; returns KVQ_OK or error
int kvq_seeded_launch(kvq_scan *s, const KvqParams &P, const uint8_t *d_data, int64_t nbytes,
                      const uint32_t *d_chunk_off, int64_t nchunks, int64_t fpos_base, uint32_t max_chunk_bytes);

uint32_t kvq_choose_tile(uint32_t maxline, uint32_t rec_bytes);
uint32_t kvq_min_tile();
uint32_t kvq_tile_for_text(const uint8_t *text, size_t n, uint32_t *rec_bytes_out = nullptr);

// synth.hip
// (C ABI only)
