/*
 * include/kvarq_hip.h -- C ABI of libkvarq_hip.so, the MI355X-native engine
 * behind KvarQ's `kvarq.engine` module.
 *
 * Plain pointers and sizes only (no torch / HIP types in the signatures);
 * "device pointer" arguments are ordinary HIP device addresses passed as
 * void*.  Every entry point names the reference interface it replaces
 * (/root/reference/csrc/workhorse.c unless said otherwise).  The Python side
 * (kvarq_amd/engine.py) binds these with ctypes; INTEGRATION.md shows the stub
 * a reference maintainer would add.
 *
 * Threading contract (same as the reference): one scan at a time
 * per process for kvq_findseqs (a second concurrent call fails with
 * KVQ_ERR_RUNTIME "findseqs() already running!"); kvq_poll_stats and
 * kvq_request_stop may be called from any thread while a scan runs.
 */
#ifndef KVARQ_HIP_H
#define KVARQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVQ_MAX_READLENGTH 1024          /* MAX_READLENGTH, workhorse.c:105 */
#define KVQ_SCANBUFSIZE    (1024*1024)   /* SCANBUFSIZE,    workhorse.c:15  */
#define KVQ_MUT_CLASSES    6             /* A C G T N other                 */

/* error classes; kvarq_amd/engine.py maps them to the exceptions the
 * reference raises (workhorse.c:577-605, 794-829, 1038-1047, 1260, 1295) */
enum {
    KVQ_OK          = 0,
    KVQ_ERR_FORMAT  = 1,   /* kvarq.fastq.FastqFileFormatException */
    KVQ_ERR_IO      = 2,   /* IOError      */
    KVQ_ERR_MEMORY  = 3,   /* MemoryError  */
    KVQ_ERR_RUNTIME = 4,   /* RuntimeError */
    KVQ_ERR_TYPE    = 5,   /* TypeError    */
    KVQ_ERR_DEVICE  = 6,   /* no usable GPU / HIP failure: RuntimeError, never a CPU fallback */
    KVQ_ERR_RESCAN  = 7    /* kvq_scan_finish after host batches: the hit arena was too small and has been enlarged;
                            * kvq_scan_reset and feed the same batches again (kvq_findseqs does so itself;
                            * device batches are replayed by the library).  kvarq_amd.scan.Scanner replays them. */
};

/* ---- engine.config / engine.get_config (workhorse.c:1484-1507) -------------
 * process-global, persists across calls; defaults maxerrors 0, minoverlap 20,
 * minreadlength 10, nthreads 1, Amin '!', Azero '!' (workhorse.c:71-75).
 * nthreads = number of host reader/inflate threads; it never changes results. */
typedef struct kvq_config {
    int32_t maxerrors;
    int32_t minoverlap;
    int32_t minreadlength;
    int32_t nthreads;
    int8_t  Amin;
    int8_t  Azero;
} kvq_config;

void kvq_config_set(const kvq_config *cfg);
void kvq_config_get(kvq_config *cfg);

/* ---- error state -----------------------------------------------------------
 * code + message of the last failed call on this thread's most recent scan
 * (the reference's `exception`/`errstr` globals, workhorse.c:95-97,1454-1458) */
int32_t kvq_last_error(char *msg, size_t cap);

/* ---- target-sequence table (the `sequences` argument of findseqs,
 *      workhorse.c:1300-1338; built by kvarq/analyse.py:352-354) -------------- */
typedef struct kvq_table kvq_table;

/* uploads the sequences (arbitrary bytes, any length >= 0) and builds the
 * device-side seed index for the given config (NULL = current global config).
 * Returns NULL on error (kvq_last_error). */
kvq_table *kvq_table_create(const uint8_t *const *seqs, const int32_t *seqlens, int32_t nseq,
                            const kvq_config *cfg);
void    kvq_table_destroy(kvq_table *t);
int32_t kvq_table_nseq(const kvq_table *t);
int64_t kvq_table_bases(const kvq_table *t);       /* sum of sequence lengths */
/* 1 if the seed-filter kernel serves sequence s, 0 if the exhaustive kernel does */
int32_t kvq_table_seq_is_seeded(const kvq_table *t, int32_t s);
int32_t kvq_table_seed_k(const kvq_table *t);

/* ---- counters ----------------------------------------------------------------
 * One flat int64 array in device memory, summable across GPUs with a single
 * all-reduce; slot KVQ_CTR_LONGEST is max-reduced instead.
 *   [0] records_parsed   (add_records_parsed, workhorse.c:387-392)
 *   [1] longest read + 1 (rls_longest + 1, workhorse.c:399-400; 0 = none)
 *   [2] hits             (length of the hit list)
 *   [3] reserved
 *   [4 .. 4+1024)                 readlengths      (rls_buf, workhorse.c:394-402)
 *   [.. +S)                       nseqhits         (seqhits, workhorse.c:435)
 *   [.. +S)                       nseqbasehits     (seqbasehits, workhorse.c:434)
 *   [.. +B)                       coverage         (Coverage.coverage per sequence, kvarq/analyse.py:76)
 *   [.. +6B)                      mutations        (Coverage.mutations as A,C,G,T,N,other counts, analyse.py:77-78)
 * with S = number of sequences, B = kvq_table_bases(). */
enum { KVQ_CTR_RECORDS = 0, KVQ_CTR_LONGEST = 1, KVQ_CTR_HITS = 2, KVQ_CTR_READLENGTHS = 4 };
int64_t kvq_counters_len(const kvq_table *t);
int64_t kvq_counters_off_nseqhits(const kvq_table *t);
int64_t kvq_counters_off_nseqbasehits(const kvq_table *t);
int64_t kvq_counters_off_coverage(const kvq_table *t);
int64_t kvq_counters_off_mutations(const kvq_table *t);
/* start of sequence s inside the coverage block (prefix sum of lengths) */
int64_t kvq_table_seq_offset(const kvq_table *t, int32_t s);

/* ---- scan object: scan_filepart (workhorse.c:976-1197) on the GPU ----------- */
typedef struct kvq_scan kvq_scan;

/* d_counters: device buffer of kvq_counters_len() int64 owned by the caller
 * (e.g. a torch tensor that is all-reduced afterwards), zeroed by the caller;
 * NULL = the scan allocates and zeroes its own.
 *
 * Several scan objects may be alive at once (each has its own stream): a caller may enqueue the next job
 * (kvq_scan_reset + kvq_scan_device) on one object before it waits for kvq_scan_finish of another -- the
 * reference's counterpart is a caller that starts the next file while it post-processes the last result.
 * The persistent scan kernels of a process are chained (one at a time, in the order they were enqueued);
 * the small kernels of a finish run beside the next scan. */
kvq_scan *kvq_scan_create(const kvq_table *t, void *d_counters);
void      kvq_scan_destroy(kvq_scan *s);

/* chunk boundaries fastq_read/fastq_rewind (workhorse.c:696-718, 737-956) put
 * on an in-memory stream: writes nchunks+1 offsets (cap >= nbytes/512Ki + 4);
 * returns nchunks, or -1 when a chunk holds no record start. Host only. */
int64_t kvq_chunk_offsets(const uint8_t *data, int64_t nbytes, int64_t *offsets, int64_t cap);

/* Scan nbytes of FastQ text resident in device memory (16-byte aligned,
 * readable up to the next 16-byte boundary).  chunk_off[0..nchunks] are host
 * offsets into d_data; every chunk starts a fresh record count like one
 * fastq_read buffer.  fpos_base = offset of d_data[0] in the concatenated
 * inflated stream (file_pos of a Hit is global, workhorse.c:777).  nbytes must
 * be at most 4 GiB - 1 MiB (offsets inside a batch are 32-bit).  Launches are asynchronous; d_data must stay valid until
 * kvq_scan_finish.  Returns KVQ_OK or an error code. */
int32_t kvq_scan_device(kvq_scan *s, const void *d_data, int64_t nbytes,
                        const int64_t *chunk_off, int64_t nchunks, int64_t fpos_base);

/* same for host memory: copies to a device staging buffer first (the buffer
 * the reference fills with fread/inflate, workhorse.c:986,998) */
int32_t kvq_scan_host(kvq_scan *s, const void *h_data, int64_t nbytes,
                      const int64_t *chunk_off, int64_t nchunks, int64_t fpos_base);

/* the same without waiting for the copy: one host batch is in flight at a time, so that the
 * caller can fill its next buffer (fread / inflate) while this one is copied and scanned.
 * h_data must stay untouched until kvq_scan_host_copied(s) has returned (or the next
 * kvq_scan_host_async / kvq_scan_host_drain / kvq_scan_finish).  kvq_scan_host_drain waits for the
 * batch in flight and settles it (a batch whose record-split speculation failed validation is
 * scanned again with the exhaustive kernels while its text is still staged). */
int32_t kvq_scan_host_async(kvq_scan *s, const void *h_data, int64_t nbytes,
                            const int64_t *chunk_off, int64_t nchunks, int64_t fpos_base);
int32_t kvq_scan_host_copied(kvq_scan *s);
int32_t kvq_scan_host_drain(kvq_scan *s);

/* wait for the device, fold hits into the counters, bring hits (canonical
 * order), hit bytes and counters to the host.  On a malformed
 * record returns KVQ_ERR_FORMAT with the reference's message
 * (workhorse.c:1037-1048) for the first bad record in stream order. */
int32_t kvq_scan_finish(kvq_scan *s);
/* optional, for a caller that keeps several scan objects in flight: every batch of this scan has been fed -- enqueue what
 * kvq_scan_finish would enqueue (the ordering of the hits, the gather, the copies to the host) behind the scan's kernels
 * now and return at once; kvq_scan_finish later only waits for it.  Without it the host sits out those kernels inside
 * kvq_scan_finish before it can enqueue its next job (they run beside another job's scan, slowly).  Feeding another batch
 * afterwards is allowed (kvq_scan_finish enqueues again).  Not part of the reference's interface: engine.findseqs is one call. */
int32_t kvq_scan_finish_begin(kvq_scan *s);

/* results, valid after kvq_scan_finish until kvq_scan_destroy; arrays are
 * owned by the scan object (engine.Hit fields, workhorse.c:1579-1586) */
int64_t        kvq_scan_n_hits(const kvq_scan *s);
const int32_t *kvq_scan_hit_seq_nr(const kvq_scan *s);
const int64_t *kvq_scan_hit_file_pos(const kvq_scan *s);
const int32_t *kvq_scan_hit_seq_pos(const kvq_scan *s);
const int32_t *kvq_scan_hit_length(const kvq_scan *s);
const int32_t *kvq_scan_hit_readlength(const kvq_scan *s);
const uint8_t *kvq_scan_hitseq_blob(const kvq_scan *s);       /* workhorse.c:437-439 */
const int64_t *kvq_scan_hitseq_offsets(const kvq_scan *s);    /* n_hits + 1 */
const int64_t *kvq_scan_counters(const kvq_scan *s);          /* host copy, kvq_counters_len() */
void          *kvq_scan_device_counters(const kvq_scan *s);   /* the device array (several ranks: the sum over all of them once finish has taken it) */
void          *kvq_scan_device_counters_own(const kvq_scan *s);   /* ... this rank's own counters, whatever finish has summed */
int64_t        kvq_scan_parsed(const kvq_scan *s);            /* fastq_parsed          */
int64_t        kvq_scan_total(const kvq_scan *s);             /* fastq_size_estimated  */

/* GPU time of all scan kernels enqueued so far on this scan's stream, from HIP
 * events around the launches (valid after kvq_scan_finish); and the same for
 * the dominant (read-scanning) kernel alone plus its launch count */
double  kvq_scan_kernel_ms(const kvq_scan *s);
double  kvq_scan_main_kernel_ms(const kvq_scan *s);
/* (measurement) ms between the end of a's last main kernel and the start of b's first one; both finished, not reset since */
double  kvq_scan_gap_ms(const kvq_scan *a, const kvq_scan *b);
int64_t kvq_scan_main_kernel_launches(const kvq_scan *s);
/* forget accumulated hits/counters/timers but keep buffers (bench steps) */
int32_t kvq_scan_reset(kvq_scan *s);
/* which kernels produced the result: bit 0 = the seed-filter kernel ran, bit 1 = the
 * exhaustive kernels ran, bit 2 = the seed-filter pass of a batch was discarded (its speculated
 * record split failed validation, one read flooded a wave's queues) and the batch rescanned,
 * bit 3 = tiles of the seed-filter pass left their records alone (a record longer than a tile's
 * look-ahead, more newlines than its tables hold) and those records were scanned again */
int32_t kvq_scan_path(const kvq_scan *s);
/* 0 = let the table decide, 1 = force the exhaustive kernel for every sequence */
void    kvq_scan_force_exhaustive(kvq_scan *s, int32_t on);

/* ---- several GPUs, one process each ----------------------------------------------------------
 * The reference starts nthreads workers on one file and joins them into one result
 * (workhorse.c:1375-1447: pthread_create / pthread_join, counters under mutexes, one hit list).
 * Here every process scans its own stretch of the reads on its own GPU -- no exchange on the data
 * path -- and the join is a collective over RCCL (librccl.so, loaded on first use):
 *   kvq_comm_unique_id   rank 0 makes the 128-byte id of a communicator, the caller hands it to the
 *                        other ranks (any way it likes: a file, MPI, torch.distributed ...)
 *   kvq_comm_create      every rank, collectively, with its number and the id; after kvq_set_device
 *   kvq_scan_set_comm    kvq_scan_finish then is collective.  Behind the rank's own finish the ranks
 *                        agree on how they ended (a maximum over one status word); when all are fine the
 *                        counter arrays of all ranks are summed (one all-reduce on the scan's stream; the
 *                        slot of the longest read takes the maximum) into kvq_scan_counters and
 *                        kvq_scan_device_counters -- the rank's own counters are kept, a second finish
 *                        sums them afresh.  When ANY rank returns KVQ_ERR_RESCAN (hit arena overflow on
 *                        host batches) EVERY rank does, with no sum taken: all ranks reset, feed their
 *                        batches again and finish again, so their collectives stay in step.  Any other
 *                        failure of one rank is an error on all of them.
 *   kvq_comm_create_local  instead of kvq_comm_create: the ranks are threads of ONE process (each with a
 *                        scan of its own, on whatever device it has set) that exchange through host
 *                        memory -- no RCCL; `world_key` is any number the ranks of one communicator
 *                        share.  For hosts without RCCL and for testing the join on a single GPU.
 *   kvq_scan_gather_hits after kvq_scan_finish, collective: the hits of all ranks, in rank order --
 *                        ranks scan consecutive stretches of the stream, so that is the reference's
 *                        order -- replace the rank's own behind kvq_scan_n_hits / kvq_scan_hit_* /
 *                        kvq_scan_hitseq_*: a count exchange (all-gather), then one broadcast per
 *                        rank and array into its place
 *   kvq_comm_allreduce_counters  the same sum for a counter array the caller keeps itself
 *                        (d_scratch16: 16 bytes of device memory) */
typedef struct kvq_comm kvq_comm;
int32_t   kvq_comm_unique_id(void *id128);
kvq_comm *kvq_comm_create(int32_t nranks, int32_t rank, const void *id128);
void      kvq_comm_destroy(kvq_comm *c);
int32_t   kvq_comm_nranks(const kvq_comm *c);
int32_t   kvq_comm_rank(const kvq_comm *c);
int32_t   kvq_scan_set_comm(kvq_scan *s, kvq_comm *c);
int32_t   kvq_scan_gather_hits(kvq_scan *s, kvq_comm *c);
int32_t   kvq_comm_allreduce_counters(kvq_comm *c, void *d_counters, int64_t ctr_len, void *d_scratch16);
kvq_comm *kvq_comm_create_local(int32_t nranks, int32_t rank, uint64_t world_key);
/* where kvq_scan_gather_hits puts the ranks' arrays -- a host function of the counts alone (counts[2r] hits,
 * counts[2r+1] hit bytes of rank r).  parts: 7 x {offset in the rank's own result buffer, offset in the gathered
 * one, bytes} per rank, in the order file_pos, hitseq offsets, seq_nr, seq_pos, length, readlength, hit bytes;
 * blob_base[r]: added to rank r's hitseq offsets; totals: {hits, hit bytes, bytes of the gathered buffer, offset
 * of its hitseq-offset array}.  kvq_gather_host carries the plan out in host memory; kvq_result_layout_words:
 * the offsets of those seven arrays in a result buffer of n hits and blob_bytes hit bytes, then its size. */
int32_t   kvq_gather_plan(int32_t nranks, const uint64_t *counts, uint64_t *parts, uint64_t *blob_base, uint64_t *totals);
int32_t   kvq_gather_host(int32_t nranks, const uint64_t *counts, const uint8_t *const *rank_bufs, uint8_t *out);
void      kvq_result_layout_words(uint64_t n, uint64_t blob_bytes, uint64_t *out8);

/* ---- engine.findseqs (workhorse.c:1249-1464) -------------------------------
 * files: plain or ".gz" (by suffix, workhorse.c:582), scanned as one stream
 * with cumulative file_pos.  Blocking; uses the global config.  Returns a scan
 * object holding the results (never NULL unless out of memory); check
 * kvq_last_error() -- on error the scan object still has to be destroyed. */
kvq_scan *kvq_findseqs(const char *const *files, int32_t nfiles,
                       const uint8_t *const *seqs, const int32_t *seqlens, int32_t nseq);

/* destroys a scan returned by kvq_findseqs together with the table it built */
void kvq_findseqs_free(kvq_scan *s);

/* host-only (no GPU): the chunks the reader cuts from the files, exactly the
 * buffers fastq_read (workhorse.c:737-956) would hand to the scanning threads,
 * as (offset in the inflated stream, length) pairs; returns the number of
 * chunks or -1 (kvq_last_error).  batch_bytes <= 0 selects the default. */
int64_t kvq_host_chunk_plan(const char *const *files, int32_t nfiles, int64_t *chunk_fpos, int64_t *chunk_len,
                            int64_t cap, int64_t *parsed, int64_t *total, int64_t batch_bytes);

/* ---- engine.stats / engine.stop (workhorse.c:1205-1244, 1469-1479) ---------- */
typedef struct kvq_live_stats {
    int64_t records_parsed;
    int64_t parsed;            /* fastq_parsed         */
    int64_t total;             /* fastq_size_estimated */
    int64_t rls_longest;       /* -1 if none */
    int32_t nseq;
    int32_t running;
    int32_t sigints;
    int32_t stop_requested;
} kvq_live_stats;

/* snapshot of the running (or last) findseqs; readlengths/nseqhits/nseqbasehits
 * may be NULL; nseq_cap bounds the two per-sequence arrays */
void kvq_poll_stats(kvq_live_stats *out, int64_t *readlengths /*[1024]*/,
                    int64_t *nseqhits, int64_t *nseqbasehits, int32_t nseq_cap);
void kvq_request_stop(void);
void kvq_count_sigint(void);    /* sigint_cb, workhorse.c:133-136 */
/* signal(SIGINT, sigint_cb) of the module init (workhorse.c:1632) as an explicit, reversible call: a C handler
 * that only counts, so that SIGINTs are counted while the calling thread is inside kvq_findseqs */
int  kvq_sigint_counter_install(void);
void kvq_sigint_counter_remove(void);

/* ---- device + synthetic workload (bench / tests plumbing) -------------------- */
int32_t kvq_device_count(void);
int32_t kvq_set_device(int32_t ordinal);
void   *kvq_device_alloc(int64_t nbytes);               /* 256-byte aligned, NULL on failure */
void    kvq_device_free(void *p);
int32_t kvq_memcpy_h2d(void *d, const void *h, int64_t nbytes);
int32_t kvq_memcpy_d2h(void *h, const void *d, int64_t nbytes);
int32_t kvq_memset_d(void *d, int32_t value, int64_t nbytes);
int32_t kvq_device_synchronize(void);
/* Device and pinned blocks of destroyed scans and tables are kept for the next ones (at most 2 GiB + 512 MiB;
 * KVQ_BLOCK_CACHE=0 turns that off): this hands them back to the driver.  Nothing needs calling it. */
void    kvq_release_cached(void);

/* records first..first+n-1 of the synthetic FastQ stream of SURVEY 8(d)
 * (kvarq_amd/synth.py is the byte-identical numpy statement) written to
 * d_out (n * (2L+25) bytes); d_genome = the synthetic genome in device memory */
int32_t kvq_synth_reads_device(void *d_out, int64_t first, int64_t n, int32_t L,
                               uint64_t seed, const void *d_genome, int64_t genome_size);
/* the same on the host (single thread) */
void    kvq_synth_reads_host(uint8_t *out, int64_t first, int64_t n, int32_t L,
                             uint64_t seed, const uint8_t *genome, int64_t genome_size);
/* i.i.d. genome bases before planting (synth.genome does the planting) */
void    kvq_synth_genome_host(uint8_t *out, int64_t size, uint64_t seed);

const char *kvq_version(void);

#ifdef __cplusplus
}
#endif
#endif
