/*
 * oracle/kvarq_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the KvarQ read-scanning hot path (reference:
 * /root/reference/csrc/workhorse.c).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (kvarq_amd/, libkvarq_hip.so) never links, imports or calls it.
 *
 * Parity status: PINNED -- checked against the reference's own known-answer
 * tests (tests/test_engine.py literal expectations on its tests/fastqs files) and
 * against the reference C engine itself compiled from /root/reference into
 * oracle/_ref (see tests/test_oracle_*.py, tests/golden/).
 */
#ifndef KVARQ_ORACLE_H
#define KVARQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVO_MAX_READLENGTH 1024      /* workhorse.c:105 */
#define KVO_SCANBUFSIZE (1024*1024)  /* workhorse.c:15  */

/* error classes, mapping to the Python exceptions the reference raises */
enum {
    KVO_OK = 0,
    KVO_ERR_FORMAT = 1,   /* kvarq.fastq.FastqFileFormatException  workhorse.c:1037-1048 */
    KVO_ERR_IO = 2,       /* IOError                               workhorse.c:577,617,665,794,819 */
    KVO_ERR_MEMORY = 3,   /* MemoryError */
    KVO_ERR_RUNTIME = 4   /* RuntimeError                          workhorse.c:759,922 */
};

/* the six engine.config() globals, workhorse.c:69-76 */
typedef struct kvo_config {
    int32_t maxerrors;
    int32_t minoverlap;
    int32_t minreadlength;
    int32_t nthreads;
    int8_t  Amin;
    int8_t  Azero;
} kvo_config;

typedef struct kvo_result {
    /* hits in canonical order, struct-of-arrays; engine.Hit fields, workhorse.c:1579-1586 */
    int64_t  n_hits;
    int32_t *seq_nr;
    int64_t *file_pos;
    int32_t *seq_pos;
    int32_t *length;
    int32_t *readlength;
    uint8_t *hitseq_blob;      /* concatenated hit bytes            workhorse.c:437 */
    int64_t *hitseq_off;       /* n_hits+1 offsets into the blob */
    /* stats, workhorse.c:1205-1244 */
    int64_t  readlengths[KVO_MAX_READLENGTH];
    int64_t  rls_longest;      /* -1 if no record was parsed */
    int32_t  nseq;
    int64_t *nseqhits;
    int64_t *nseqbasehits;
    int64_t  records_parsed;
    int64_t  parsed;
    int64_t  total;
    /* error state (what the reference would raise after joining its workers) */
    int32_t  err_code;
    char     errmsg[1024];
    /* bookkeeping */
    int64_t  cap_hits, cap_blob;
} kvo_result;

/* engine.findseqs on files (plain or .gz by suffix); never returns NULL unless
 * out of memory.  workhorse.c:1249-1464 */
kvo_result *kvo_findseqs(const char *const *files, int nfiles,
                         const uint8_t *const *seqs, const int32_t *seqlens, int nseq,
                         const kvo_config *cfg);

/* same scan over one in-memory (already inflated) stream, cut into chunks the
 * way fastq_read would cut it; used by tests on synthetic data and by
 * bench.py's cpu_baseline leg.  fpos_base is added to every file_pos. */
kvo_result *kvo_scan_memory(const uint8_t *data, int64_t nbytes, int64_t fpos_base,
                            const uint8_t *const *seqs, const int32_t *seqlens, int nseq,
                            const kvo_config *cfg);

void kvo_free(kvo_result *r);

/* chunk boundaries fastq_read/fastq_rewind would produce on an in-memory
 * stream (workhorse.c:696-718, 737-956).  Writes up to cap chunk start offsets
 * followed by the end offset; returns number of chunks, or -1 if a chunk has
 * no record start (the reference's RuntimeError). */
int64_t kvo_chunk_offsets(const uint8_t *data, int64_t nbytes, int64_t *offsets, int64_t cap);

/* Coverage.apply_hit fold (kvarq/analyse.py:57-78) over all hits of r, per
 * SEQUENCE INDEX (strand folding is the caller's business):
 *   cov[off[s]+j]      += 1            for every covered base j of sequence s
 *   mut[(off[s]+j)*6+c] += 1           if the read base differs; c = A,C,G,T,N,other
 * off[] has nseq+1 entries (prefix sums of seqlens). */
void kvo_fold_coverage(const kvo_result *r, const uint8_t *const *seqs, const int32_t *seqlens,
                       const int64_t *off, int64_t *cov, int64_t *mut);

#ifdef __cplusplus
}
#endif
#endif
