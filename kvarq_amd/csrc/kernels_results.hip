// kvarq_amd/csrc/kernels_results.hip -- the hits of a finished scan, put on the device into the
// order and the arrays the accessors of include/kvarq_hip.h hand out.
//
// The scanning kernels append hits to the arena in whatever order their waves finish.  The
// reference returns them worker by worker, read by read, sequence by sequence, class A / B / C
// (csrc/workhorse.c:1398-1447 after the join; 1097-1180 inside one read), which for the stream as a
// whole is ascending (file_pos, seq_nr, class, ordinal).  Sorting and regrouping a few MB on the
// GPU takes a fraction of the time the host needs for it and leaves the host one copy to wait for.
//
// Nothing here needs a number from the host that the host would first have to fetch: the hit count,
// the bucket plan and the layout of the result arrays are worked out on the device
// (kvq_finish_plan) and read from device memory by the kernels that follow, whose grids do not
// depend on them.  So `finish` enqueues the whole tail behind the scan's kernels and waits once.
//
// This file is included by kvq_unity.hip after kvq_device.h.

#include "kvq_host.h"

#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

struct KvqHitBefore {
    __host__ __device__ bool operator()(const KvqHit &a, const KvqHit &b) const
    {
        if (a.fpos != b.fpos) return a.fpos < b.fpos;
        if (a.seq_nr != b.seq_nr) return a.seq_nr < b.seq_nr;
        return a.key < b.key;
    }
};

struct KvqHitBytes {
    __host__ __device__ long long operator()(const KvqHit &h) const { return h.length > 0 ? (long long)h.length : 0ll; }
};

__host__ __device__ static inline KvqResultLayout kvq_result_layout(uint64_t n, uint64_t blob_bytes)
{
    KvqResultLayout L; size_t at = 0;
    auto take = [&](size_t b) { const size_t r = at; at += (b + 255) & ~(size_t)255; return r; };
    L.file_pos = take((size_t)n * 8); L.hitseq_off = take(((size_t)n + 1) * 8);
    L.seq_nr = take((size_t)n * 4); L.seq_pos = take((size_t)n * 4); L.length = take((size_t)n * 4); L.readlength = take((size_t)n * 4);
    L.blob = take((size_t)blob_bytes);
    L.total = at;
    return L;
}

// ---- order by buckets (the usual case) -------------------------------------------------------
// Hits spread over the stream roughly evenly (a few per thousand reads), so a counting sort into
// about 4 n buckets of equal file_pos width leaves almost every bucket with zero to a few hits,
// which one thread puts in order.  A bucket with more than KVQ_BUCKET_MAX hits (hits crowded into a
// corner of the stream) raises a flag and the host orders the scan with the merge sort below instead.
#define KVQ_BUCKET_MAX 64u
#define KVQ_BUCKETS_MAX (1u << 22)
#define KVQ_ORDER_BLOCKS 512u      // the buckets are dealt to this many workgroups in contiguous shares (their sums are scanned by every block for itself)

// what kvq_finish_plan leaves for the kernels behind it, and (copied to pinned memory) for the host
struct KvqFinishState {
    uint32_t n;                    // hits to put in order: min(hits counted, arena capacity)
    uint32_t n_raw;                // hits counted (beyond the capacity: the scan has to be repeated with a larger arena)
    uint32_t crowded;              // a bucket held more than KVQ_BUCKET_MAX hits
    uint32_t nb, shift, bc;        // bucket = (file_pos - lo) >> shift, nb buckets; a block of the ordering kernels takes bc of them
    long long lo;
    unsigned long long blob_n, err;
    KvqResultLayout L;
};

extern "C" __global__ void __launch_bounds__(64)
kvq_finish_plan(const unsigned int *__restrict__ arena_n, uint32_t arena_cap, const unsigned long long *__restrict__ blob_n,
                unsigned long long blob_cap, const unsigned long long *__restrict__ err, long long lo, long long hi,
                uint32_t nb_max, KvqFinishState *__restrict__ st)
{
    KVQ_BESIDE_SCAN();
    if (threadIdx.x != 0) return;
    KvqFinishState s;
    s.n_raw = *arena_n; s.n = s.n_raw < arena_cap ? s.n_raw : arena_cap;
    s.blob_n = *blob_n; s.err = *err; s.crowded = 0; s.lo = lo;
    uint32_t want = 256; while ((unsigned long long)want < 4ull * s.n && want < nb_max) want <<= 1;    // about four buckets per hit: most hold none or one
    const unsigned long long span = (unsigned long long)(hi > lo ? hi - lo : 1);
    s.shift = 0; while ((span >> s.shift) >= (unsigned long long)want) s.shift++;
    s.nb = (uint32_t)(span >> s.shift) + 1u;                              // <= want
    s.bc = (s.nb + KVQ_ORDER_BLOCKS - 1u) / KVQ_ORDER_BLOCKS;
    s.L = kvq_result_layout(s.n, s.blob_n < blob_cap ? s.blob_n : blob_cap);
    *st = s;
}

__global__ void __launch_bounds__(256)
kvq_bucket_count(const KvqHit *__restrict__ arena, const KvqFinishState *__restrict__ st, uint32_t *__restrict__ cnt, uint32_t *__restrict__ share_cnt)
{
    KVQ_BESIDE_SCAN();
    const uint32_t n = st->n, shift = st->shift, bc = st->bc; const long long lo = st->lo;
    const int lane = kvq_lane();
    for (uint32_t h0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); h0 < n; h0 += gridDim.x * blockDim.x) {     // (a wave stays together)
        const uint32_t h = h0 + (uint32_t)lane;
        uint32_t share = 0xFFFFFFFFu;
        if (h < n) {
            const uint32_t b = (uint32_t)((uint64_t)(arena[h].fpos - lo) >> shift);
            atomicAdd(&cnt[b], 1u);
            share = b / bc;
        }
        // hits per share of buckets: neighbours in the arena mostly fall into one share -- one add per wave and share
        unsigned long long todo = __ballot(share != 0xFFFFFFFFu);
        while (todo) {
            const uint32_t s0 = (uint32_t)__shfl((int)share, __ffsll((long long)todo) - 1, 64);
            const unsigned long long same = __ballot(share == s0);
            if (lane == __ffsll((long long)same) - 1) atomicAdd(&share_cnt[s0], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
}

// sum of v over the block's threads in front of this one (exclusive), and the total; blockDim.x == 256
__device__ __forceinline__ unsigned long long kvq_block_excl_scan_256(unsigned long long v, unsigned long long *sh, unsigned long long &total)
{
    const uint32_t tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {
        const unsigned long long u = tid >= d ? sh[tid - d] : 0ull;
        __syncthreads();
        sh[tid] += u;
        __syncthreads();
    }
    const unsigned long long incl = sh[tid];
    total = sh[255];
    __syncthreads();
    return incl - v;
}

// counts per bucket -> where each bucket's hits start in the order (start[b], start[nb] = n): block k takes
// the buckets [k bc, (k + 1) bc); the hits in front of its share come from the share counts
__global__ void __launch_bounds__(256)
kvq_bucket_starts(const KvqFinishState *__restrict__ st, uint32_t *__restrict__ cnt_start, const uint32_t *__restrict__ share_cnt)
{
    KVQ_BESIDE_SCAN();
    __shared__ unsigned long long sh[256];
    const uint32_t nb = st->nb, bc = st->bc, k = blockIdx.x, tid = threadIdx.x;
    unsigned long long part = 0, tot;
    for (uint32_t j = tid; j < k; j += 256u) part += share_cnt[j];
    (void)kvq_block_excl_scan_256(part, sh, tot);
    uint32_t run = (uint32_t)tot;                                   // hits in front of this share
    const uint32_t b_lo = k * bc, b_hi = b_lo + bc < nb ? b_lo + bc : nb;
    for (uint32_t b0 = b_lo; b0 < b_hi; b0 += 256u) {
        const uint32_t b = b0 + tid;
        const uint32_t c = b < b_hi ? cnt_start[b] : 0u;
        const uint32_t ex = (uint32_t)kvq_block_excl_scan_256(c, sh, tot);
        if (b < b_hi) cnt_start[b] = run + ex;
        run += (uint32_t)tot;
    }
    if (k == 0 && tid == 0) cnt_start[nb] = st->n;
}

__global__ void __launch_bounds__(256)
kvq_bucket_scatter(const KvqHit *__restrict__ arena, const KvqFinishState *__restrict__ st, const uint32_t *__restrict__ start,
                   uint32_t *__restrict__ fill, uint32_t *__restrict__ idx)
{
    KVQ_BESIDE_SCAN();
    const uint32_t n = st->n, shift = st->shift; const long long lo = st->lo;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < n; h += gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)((uint64_t)(arena[h].fpos - lo) >> shift);
        idx[start[b] + atomicAdd(&fill[b], 1u)] = h;
    }
}

// one thread per bucket: its (few) hit numbers put in order.  Up to eight hits: sorting network in
// registers.  Nine to KVQ_BUCKET_MAX (a read on a locus that many templates share): the wave takes
// such buckets one at a time, lane i holds hit i and counts the hits in front of it (no dependent
// loads: a serial insertion sort of 20 hits used to set the time of the whole kernel).
// Block k takes the buckets [k bc, (k + 1) bc) and also sums up the bytes of their hits (share_len[k]): the
// gather kernel, which takes the same shares, knows from those where its hits' bytes start.
__global__ void __launch_bounds__(256)
kvq_bucket_sort(const KvqHit *__restrict__ arena, KvqFinishState *__restrict__ st, const uint32_t *__restrict__ start, uint32_t *__restrict__ idx,
                unsigned long long *__restrict__ share_len)
{
    KVQ_BESIDE_SCAN();
    const int lane = kvq_lane();
    const uint32_t nb = st->nb, bc = st->bc;
    const uint32_t b_lo = blockIdx.x * bc, b_hi = b_lo + bc < nb ? b_lo + bc : nb;
    unsigned long long bytes = 0;
    for (uint32_t b0 = b_lo; b0 < b_hi; b0 += blockDim.x) {      // (the same trip count for every lane of a wave)
        // (neighbouring buckets go to different waves: a share of some sixty buckets is then the work of four waves, not of the first
        // one alone -- on input whose reads hit a dozen templates each, the wave takes its crowded buckets one behind the other)
        const uint32_t b = b0 + ((uint32_t)lane * (blockDim.x >> 6) + (threadIdx.x >> 6));
        uint32_t s0 = 0, m = 0;
        if (b < b_hi) { s0 = start[b]; m = start[b + 1] - s0; }
        // (the bytes of the bucket's hits are summed up where the hits are fetched anyway -- side by side, not one behind the other: a
        // loop over the bucket's hits in front of everything was two dependent trips to memory per hit, and on input whose reads hit a
        // dozen templates each it was the whole time of this kernel, and of the finish)
        auto len_of = [&](uint32_t hit) -> unsigned long long { const int l = arena[hit].length; return l > 0 ? (unsigned long long)l : 0ull; };
        if (m > KVQ_BUCKET_MAX) {
            for (uint32_t i = 0; i < m; i++) bytes += len_of(idx[s0 + i]);
            st->crowded = 1u; m = 0;
        }
        if (m == 1u) bytes += len_of(idx[s0]);
        if (m >= 2u && m <= 8u) {
            // the usual bucket: the sort keys of its hits (file_pos; seq_nr, class | ordinal: 16 of a hit's 32
            // bytes) are fetched side by side and sorted in registers together with the hit numbers
            uint32_t ix[8]; long long k1[8]; unsigned long long k2[8];
#pragma unroll
            for (int i = 0; i < 8; i++) ix[i] = (uint32_t)i < m ? idx[s0 + i] : 0u;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const KvqHit *h = &arena[ix[i]];
                k1[i] = (uint32_t)i < m ? h->fpos : 0x7FFFFFFFFFFFFFFFll;                      // (empty slots sort last)
                k2[i] = (uint32_t)i < m ? ((unsigned long long)(uint32_t)h->seq_nr << 32) | h->key : ~0ull;
                bytes += (uint32_t)i < m && h->length > 0 ? (unsigned long long)h->length : 0ull;
            }
            // odd-even transposition network over the 8 slots (no dynamic register indexing)
#pragma unroll
            for (int round = 0; round < 8; round++) {
#pragma unroll
                for (int i = round & 1; i + 1 < 8; i += 2) {
                    const bool swap = k1[i + 1] < k1[i] || (k1[i + 1] == k1[i] && k2[i + 1] < k2[i]);
                    const long long a1 = swap ? k1[i + 1] : k1[i], b1 = swap ? k1[i] : k1[i + 1];
                    const unsigned long long a2 = swap ? k2[i + 1] : k2[i], b2 = swap ? k2[i] : k2[i + 1];
                    const uint32_t ai = swap ? ix[i + 1] : ix[i], bi = swap ? ix[i] : ix[i + 1];
                    k1[i] = a1; k1[i + 1] = b1; k2[i] = a2; k2[i + 1] = b2; ix[i] = ai; ix[i + 1] = bi;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) if ((uint32_t)i < m) idx[s0 + i] = ix[i];
        }
        // the wave's larger buckets, all 64 lanes on one at a time (every lane of the wave gets here)
        static_assert(KVQ_BUCKET_MAX <= 64u, "one lane per hit");
        unsigned long long todo = __ballot(m > 8u);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1; todo &= todo - 1ull;
            const uint32_t bs0 = (uint32_t)__shfl((int)s0, src, 64), bm = (uint32_t)__shfl((int)m, src, 64);
            uint32_t mine = 0; long long a1 = 0x7FFFFFFFFFFFFFFFll; unsigned long long a2 = ~0ull;
            if ((uint32_t)lane < bm) {
                mine = idx[bs0 + (uint32_t)lane];
                const KvqHit *h = &arena[mine];
                a1 = h->fpos; a2 = ((unsigned long long)(uint32_t)h->seq_nr << 32) | h->key;
                bytes += h->length > 0 ? (unsigned long long)h->length : 0ull;          // (any lane's sum will do: the block's lanes are added up below)
            }
            uint32_t rank = 0;                                       // hits that come before this lane's
            for (uint32_t j = 0; j < bm; j++) {
                const long long b1 = ((long long)__shfl((int)(a1 >> 32), (int)j, 64) << 32) | (uint32_t)__shfl((int)a1, (int)j, 64);
                const unsigned long long b2 = ((unsigned long long)(uint32_t)__shfl((int)(a2 >> 32), (int)j, 64) << 32) | (uint32_t)__shfl((int)a2, (int)j, 64);
                rank += (b1 < a1 || (b1 == a1 && (b2 < a2 || (b2 == a2 && j < (uint32_t)lane)))) ? 1u : 0u;
            }
            // (every lane has fetched its hit number before any lane stores: one wave, in step)
            if ((uint32_t)lane < bm) idx[bs0 + rank] = mine;
        }
    }
    if (bytes) atomicAdd(&share_len[blockIdx.x], bytes);
}

// off[i] = bytes of the hits in front of hit i of the order (order == nullptr: arena order), off[n] = all:
// the same single-workgroup scan, over the hit lengths
__global__ void __launch_bounds__(1024)
kvq_scan_hit_bytes(const KvqHit *__restrict__ arena, const uint32_t *__restrict__ order, const KvqFinishState *__restrict__ st, uint8_t *__restrict__ res)
{
    KVQ_BESIDE_SCAN();
    __shared__ unsigned long long part[1024];
    const uint32_t n = st->n, tid = threadIdx.x;
    long long *const off = reinterpret_cast<long long *>(res + st->L.hitseq_off);
    const uint32_t per = (n + 1023u) / 1024u, i0 = tid * per, i1 = i0 + per < n ? i0 + per : n;
    unsigned long long sum = 0;
    for (uint32_t i = i0; i < i1; i++) { const int l = arena[order ? order[i] : i].length; sum += l > 0 ? (unsigned long long)l : 0ull; }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = tid ? part[tid - 1] : 0ull;
    for (uint32_t i = i0; i < i1; i++) { off[i] = (long long)run; const int l = arena[order ? order[i] : i].length; run += l > 0 ? (unsigned long long)l : 0ull; }
    if (tid == 1023u) off[n] = (long long)part[1023];
}

// one wave per hit: the five columns and the hit bytes moved from where the fold left them to their place
// in canonical order; hit i is arena[order[i]] (order == nullptr: arena[i]) -- the merge sort's way
__global__ void __launch_bounds__(256)
kvq_gather_results(const KvqHit *__restrict__ arena, const uint32_t *__restrict__ order, const KvqFinishState *__restrict__ st,
                   const uint8_t *__restrict__ blob_in, unsigned long long blob_cap, uint8_t *__restrict__ res)
{
    KVQ_BESIDE_SCAN();
    const int lane = kvq_lane();
    const uint32_t n = st->n;
    const KvqResultLayout L = st->L;
    const uint32_t per = (blockDim.x >> 6) * gridDim.x;
    const long long *off = reinterpret_cast<const long long *>(res + L.hitseq_off);
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += per) {
        const KvqHit h = arena[order ? order[i] : i];
        const long long at = off[i];
        const int len = h.length > 0 ? h.length : 0;
        if (lane == 0) {
            reinterpret_cast<long long *>(res + L.file_pos)[i] = h.fpos;
            reinterpret_cast<int32_t *>(res + L.seq_nr)[i] = h.seq_nr;
            reinterpret_cast<int32_t *>(res + L.seq_pos)[i] = h.seq_pos;
            reinterpret_cast<int32_t *>(res + L.length)[i] = h.length;
            reinterpret_cast<int32_t *>(res + L.readlength)[i] = h.readlength;
        }
        if (h.blob_off != 0xFFFFFFFFu && (unsigned long long)at + (unsigned long long)len <= blob_cap) {      // (an overflowing blob: the scan is repeated anyway)
            const uint8_t *src = blob_in + h.blob_off;
            uint8_t *dst = res + L.blob + at;
            for (int j = lane; j < len; j += 64) dst[j] = src[j];
        }
    }
}

// Block k takes the hits of the buckets [k bc, (k + 1) bc) -- order[i_lo .. i_hi) -- and puts them into the
// result arrays: the five columns, hitseq offsets (running sum of the hit lengths, started from the share
// sums of the blocks in front), and the hit bytes moved from where the fold left them.
__global__ void __launch_bounds__(256)
kvq_gather_shares(const KvqHit *__restrict__ arena, const uint32_t *__restrict__ order, const uint32_t *__restrict__ start,
                  const unsigned long long *__restrict__ share_len, const KvqFinishState *__restrict__ st,
                  const uint8_t *__restrict__ blob_in, unsigned long long blob_cap, uint8_t *__restrict__ res)
{
    KVQ_BESIDE_SCAN();
    __shared__ unsigned long long sh[256];
    __shared__ uint32_t c_src[256], c_len[256]; __shared__ unsigned long long c_dst[256];
    const uint32_t nb = st->nb, bc = st->bc, n = st->n, k = blockIdx.x, tid = threadIdx.x;
    const int lane = kvq_lane();
    const KvqResultLayout L = st->L;
    long long *const off = reinterpret_cast<long long *>(res + L.hitseq_off);
    unsigned long long part = 0, run;
    for (uint32_t j = tid; j < k; j += 256u) part += share_len[j];
    (void)kvq_block_excl_scan_256(part, sh, run);                     // bytes in front of this share
    const uint32_t b_lo = k * bc < nb ? k * bc : nb, b_hi = b_lo + bc < nb ? b_lo + bc : nb;
    const uint32_t i_lo = start[b_lo], i_hi = start[b_hi];
    for (uint32_t i0 = i_lo; i0 < i_hi; i0 += 256u) {
        const uint32_t i = i0 + tid;
        KvqHit h; h.length = 0; h.blob_off = 0xFFFFFFFFu;
        if (i < i_hi) h = arena[order[i]];
        const uint32_t len = i < i_hi && h.length > 0 ? (uint32_t)h.length : 0u;
        unsigned long long tot;
        const unsigned long long at = run + kvq_block_excl_scan_256(len, sh, tot);
        if (i < i_hi) {
            reinterpret_cast<long long *>(res + L.file_pos)[i] = h.fpos;
            reinterpret_cast<int32_t *>(res + L.seq_nr)[i] = h.seq_nr;
            reinterpret_cast<int32_t *>(res + L.seq_pos)[i] = h.seq_pos;
            reinterpret_cast<int32_t *>(res + L.length)[i] = h.length;
            reinterpret_cast<int32_t *>(res + L.readlength)[i] = h.readlength;
            off[i] = (long long)at;
        }
        // the chunk's hit bytes, a wave per hit (an overflowing blob: the scan is repeated anyway)
        c_src[tid] = h.blob_off; c_dst[tid] = at; c_len[tid] = (h.blob_off != 0xFFFFFFFFu && at + len <= blob_cap) ? len : 0u;
        __syncthreads();
        const uint32_t cn = i_hi - i0 < 256u ? i_hi - i0 : 256u;
        for (uint32_t j = tid >> 6; j < cn; j += 4u) {
            const uint8_t *src = blob_in + c_src[j];
            uint8_t *dst = res + L.blob + c_dst[j];
            for (uint32_t q = (uint32_t)lane; q < c_len[j]; q += 64u) dst[q] = src[q];
        }
        __syncthreads();
        run += tot;
    }
    if (k == gridDim.x - 1 && tid == 0) off[n] = (long long)run;
}

// the bucket arrays cleared for the next scan
__global__ void __launch_bounds__(256)
kvq_order_clear(const KvqFinishState *__restrict__ st, uint32_t *__restrict__ start, uint32_t *__restrict__ fill,
                uint32_t *__restrict__ share_cnt, unsigned long long *__restrict__ share_len)
{
    KVQ_BESIDE_SCAN();
    const uint32_t words = st->nb + 2u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) { start[i] = 0u; fill[i] = 0u; }
    if (blockIdx.x == 0) for (uint32_t i = threadIdx.x; i < KVQ_ORDER_BLOCKS; i += blockDim.x) { share_cnt[i] = 0u; share_len[i] = 0ull; }
}

// the scratch of the bucket ordering: start[nb_max + 64] | fill[nb_max + 64] | share_cnt[KVQ_ORDER_BLOCKS] | share_len[KVQ_ORDER_BLOCKS] (all zero between scans) | idx[arena_cap]
struct KvqOrderScratch { uint32_t *start, *fill, *share_cnt; unsigned long long *share_len; uint32_t *idx; uint32_t nb_max; };
static size_t kvq_order_scratch_zero_bytes(uint32_t nb_max) { return ((size_t)2 * (nb_max + 64) + KVQ_ORDER_BLOCKS) * 4 + (size_t)KVQ_ORDER_BLOCKS * 8; }
static KvqOrderScratch kvq_order_scratch(void *base, uint32_t nb_max)
{
    KvqOrderScratch W; W.nb_max = nb_max;
    W.start = (uint32_t *)base; W.fill = W.start + nb_max + 64; W.share_cnt = W.fill + nb_max + 64;
    W.share_len = (unsigned long long *)(W.share_cnt + KVQ_ORDER_BLOCKS); W.idx = (uint32_t *)(W.share_len + KVQ_ORDER_BLOCKS);
    return W;
}

// enqueue ordering + offsets + gather, by buckets
static int kvq_order_by_buckets(hipStream_t stream, const KvqHit *arena, const uint8_t *blob_in, unsigned long long blob_cap,
                                KvqFinishState *d_st, const KvqOrderScratch &W, uint8_t *res)
{
    const KvqFinishState *cst = d_st;
    hipLaunchKernelGGL(kvq_bucket_count, dim3(256), dim3(256), 0, stream, arena, cst, W.start, W.share_cnt);
    hipLaunchKernelGGL(kvq_bucket_starts, dim3(KVQ_ORDER_BLOCKS), dim3(256), 0, stream, cst, W.start, (const uint32_t *)W.share_cnt);
    hipLaunchKernelGGL(kvq_bucket_scatter, dim3(256), dim3(256), 0, stream, arena, cst, (const uint32_t *)W.start, W.fill, W.idx);
    hipLaunchKernelGGL(kvq_bucket_sort, dim3(KVQ_ORDER_BLOCKS), dim3(256), 0, stream, arena, d_st, (const uint32_t *)W.start, W.idx, W.share_len);
    hipLaunchKernelGGL(kvq_gather_shares, dim3(KVQ_ORDER_BLOCKS), dim3(256), 0, stream, arena, (const uint32_t *)W.idx, (const uint32_t *)W.start,
                       (const unsigned long long *)W.share_len, cst, blob_in, blob_cap, res);
    hipLaunchKernelGGL(kvq_order_clear, dim3(256), dim3(256), 0, stream, cst, W.start, W.fill, W.share_cnt, W.share_len);
    KVQ_HIP(hipGetLastError());
    return KVQ_OK;
}

// the same by a comparison sort of the n hits (n known on the host: the path for crowded buckets)
static int kvq_order_by_mergesort(hipStream_t stream, const KvqHit *arena, uint32_t n, const uint8_t *blob_in, unsigned long long blob_cap,
                                  KvqFinishState *d_st, DevBuf &tmp, DevBuf &sorted, uint8_t *res)
{
    if (n == 0) return KVQ_OK;
    int rc;
    if ((rc = sorted.ensure((size_t)n * sizeof(KvqHit)))) return rc;
    KvqHit *out = sorted.as<KvqHit>();
    size_t need_sort = 0;
    KVQ_HIP(rocprim::merge_sort(nullptr, need_sort, arena, out, (size_t)n, KvqHitBefore(), stream));
    if ((rc = tmp.ensure(need_sort + 256))) return rc;
    KVQ_HIP(rocprim::merge_sort(tmp.p, need_sort, arena, out, (size_t)n, KvqHitBefore(), stream));
    hipLaunchKernelGGL(kvq_scan_hit_bytes, dim3(1), dim3(1024), 0, stream, (const KvqHit *)out, (const uint32_t *)nullptr, (const KvqFinishState *)d_st, res);
    hipLaunchKernelGGL(kvq_gather_results, dim3(1024), dim3(256), 0, stream, (const KvqHit *)out, (const uint32_t *)nullptr, (const KvqFinishState *)d_st,
                       blob_in, blob_cap, res);
    KVQ_HIP(hipGetLastError());
    return KVQ_OK;
}
