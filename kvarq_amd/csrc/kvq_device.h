// kvarq_amd/csrc/kvq_device.h -- structures and wave-level helpers shared by the
// HIP kernels (gfx950 only: 64-lane wavefronts are assumed throughout).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define KVQ_WAVE 64
#define KVQ_SEG_BYTES 4096u   // one wave indexes 4 KiB of text: 4 rounds x 64 lanes x 16 B

// one hit as the kernels emit it (32 bytes).  Canonical order of the
// reference for one worker: by read, by sequence, then class A
// (i descending), class B (i descending), class C (i ascending) --
// key = class << 30 | ordinal inside the class.
struct KvqHit {
    int64_t  fpos;        // file_pos: stream offset of the first base of the trimmed read (workhorse.c:1124)
    int32_t  seq_nr;
    int32_t  seq_pos;
    int32_t  length;
    int32_t  readlength;
    uint32_t key;
    uint32_t blob_off;    // where kvq_fold_hits put the hit bytes
};

// counters layout (include/kvarq_hip.h)
#define KVQ_CTR_RECORDS_ 0
#define KVQ_CTR_LONGEST_ 1
#define KVQ_CTR_HITS_ 2
#define KVQ_CTR_RL_ 4
#define KVQ_RL_BINS 1024
// the counters the fused scan kernel adds to are staged per batch (kvq_commit_batch), in KVQ_STAGE_COPIES copies: a workgroup
// takes the copy of its blockIdx.x % KVQ_STAGE_COPIES
#define KVQ_STAGE_SLOTS (KVQ_CTR_RL_ + KVQ_RL_BINS)
#define KVQ_STAGE_COPIES 8

// parameters every scanning kernel needs
struct KvqParams {
    int32_t maxerrors, minoverlap, minreadlength;
    int32_t amin;                 // signed char value of Amin
    int32_t nseq;                 // all sequences
    const uint8_t *tab;           // concatenated sequence bytes
    const int32_t *tab_off;       // nseq + 1 prefix sums
    unsigned long long *ctr;      // counters (int64 slots)
    unsigned long long *covdiff;  // coverage as +1 / -1 marks per hit (slot tab_off[s] + s + position), summed up into ctr by kvq_cov_apply
    int64_t off_nseqhits, off_nseqbasehits, off_cov, off_mut;
    KvqHit *arena; uint32_t arena_cap;
    unsigned int *arena_n;        // hits emitted so far (keeps counting past cap)
    uint8_t *blob; unsigned long long blob_cap;
    unsigned long long *blob_n;
    unsigned long long *err;      // min over bad records of (fpos << 16 | kind << 8 | byte); ~0 = none
};

__device__ __forceinline__ int kvq_lane() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ uint64_t kvq_lanemask_lt()
{
    return (1ull << kvq_lane()) - 1ull;
}

// 0x80 in every byte of x that equals '\n' (exact per byte, no borrow artefacts)
__device__ __forceinline__ uint32_t kvq_nl_flags(uint32_t x)
{
    // low seven bits differ from 0x0A -> the add carries into bit 7; three instructions on gfx950
    // (v_bitop3_b32 spelled out: with its constants as literals the compiler falls back to two-input logic)
    const uint32_t t = __builtin_amdgcn_bitop3_b32(x, 0x0A0A0A0Au, 0x7F7F7F7Fu, 0x6c) + 0x7F7F7F7Fu;   // ((x & 0x7F..) ^ 0x0A..) + 0x7F..
    return __builtin_amdgcn_bitop3_b32(t, 0x80808080u, x, 0x04);                                      // ~t & 0x80.. & ~x
}

// 0x80 flags of the bytes of a dword at batch offset p that lie inside [a, b)
__device__ __forceinline__ uint32_t kvq_range_flags(uint32_t p, uint32_t a, uint32_t b)
{
    long long lo = (long long)a - (long long)p, hi = (long long)b - (long long)p;
    lo = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
    hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
    if (hi <= lo) return 0u;
    const uint64_t m = ((1ull << (8 * hi)) - 1ull) & ~((1ull << (8 * lo)) - 1ull);
    return (uint32_t)m & 0x80808080u;
}

// 0x80 byte flags of up to four dwords -> one bit per byte (dword 0 in bits 0..3, ...): the
// byte-wise dot product does the gathering (weights 1, 2, 4, ... on flags of 128 each)
__device__ __forceinline__ uint32_t kvq_flags16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3)
{
    uint32_t a = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false);
    a = __builtin_amdgcn_udot4(f1, 0x80402010u, a, false);
    uint32_t b = __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false);
    b = __builtin_amdgcn_udot4(f3, 0x80402010u, b, false);
    return (a >> 7) | (b << 1);
}

// inclusive prefix sum over the 64 lanes of a wave, all lanes active: six DPP adds
// (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast 15 and 31 across rows), no LDS traffic
__device__ __forceinline__ uint32_t kvq_wave_incl_scan(uint32_t v)
{
    // (row_shr: a lane without a source takes zero either way; saying so lets the compiler drop the "old" value)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// append one hit per flagged lane to the arena with a single atomic per wave.
// Must be called by all lanes of the wave (convergent).
__device__ __forceinline__ void kvq_emit(const KvqParams &P, bool hit, int64_t fpos, int seq_nr,
                                         int seq_pos, int length, int rl, uint32_t key)
{
    const uint64_t m = __ballot(hit);
    if (m == 0) return;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (kvq_lane() == leader) base = atomicAdd(P.arena_n, (unsigned int)__popcll(m));
    base = __shfl(base, leader, 64);
    if (hit) {
        const uint32_t idx = base + (uint32_t)__popcll(m & kvq_lanemask_lt());
        if (idx < P.arena_cap) {
            KvqHit h;
            h.fpos = fpos; h.seq_nr = seq_nr; h.seq_pos = seq_pos; h.length = length;
            h.readlength = rl; h.key = key; h.blob_off = 0;
            P.arena[idx] = h;
        }
    }
}

// the redo of what the fused scan leaves (kvq_runtime.hip: run_batch): reads of KVQ_LONG_READ bases or more are matched by a
// launch of their own, at most KVQ_LONG_CAP of them; a list entry whose record start is KVQ_REDO_TRIMMED has been trimmed
// (and counted) by the scan kernel already
// The small kernels of a step (validation, redo, fold, ordering, gather) run BESIDE the next step's persistent scan kernel, in the
// workgroup slots it leaves free -- on SIMDs whose other six waves are the scan's, at wave priority 1 to 3 in its passes.  At the
// default priority 0 they only issue when no scan wave can: kvq_bucket_sort took 160 us there against 16 us alone, and the step of a
// small input was the wait for them.  They are a few microseconds of work: they go first.
#define KVQ_BESIDE_SCAN() __builtin_amdgcn_s_setprio(3)

#define KVQ_LONG_READ 1024
#define KVQ_LONG_CAP 2048u
#define KVQ_REDO_TRIMMED 0xFFFFFFFFu

