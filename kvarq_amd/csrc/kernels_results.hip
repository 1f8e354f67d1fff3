// kvarq_amd/csrc/kernels_results.hip -- the hits of a finished scan, put on the device into the
// order and the arrays the accessors of include/kvarq_hip.h hand out.
//
// The scanning kernels append hits to the arena in whatever order their waves finish.  The
// reference returns them worker by worker, read by read, sequence by sequence, class A / B / C
// (csrc/workhorse.c:1398-1447 after the join; 1097-1180 inside one read), which for the stream as a
// whole is ascending (file_pos, seq_nr, class, ordinal).  Sorting and regrouping a few MB on the
// GPU takes a fraction of the time the host needs for it and leaves the host one copy to wait for.
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

static KvqResultLayout kvq_result_layout(uint64_t n, uint64_t blob_bytes)
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
// about n buckets of equal file_pos width leaves almost every bucket with zero to a few hits,
// which one thread puts in order by insertion.  Five small kernels instead of the six passes of
// a comparison sort over the whole array.  A bucket with more than KVQ_BUCKET_MAX hits (hits
// crowded into a corner of the stream) raises a flag and the host orders the scan with the merge
// sort below instead.
#define KVQ_BUCKET_MAX 64u

struct KvqBucketPlan { int64_t lo; uint32_t shift, nb; };       // bucket = (file_pos - lo) >> shift, nb buckets

__global__ void __launch_bounds__(256)
kvq_bucket_count(const KvqHit *__restrict__ arena, uint32_t n, KvqBucketPlan B, uint32_t *__restrict__ cnt)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < n) atomicAdd(&cnt[1u + (uint32_t)((uint64_t)(arena[h].fpos - B.lo) >> B.shift)], 1u);
}

__global__ void __launch_bounds__(256)
kvq_bucket_scatter(const KvqHit *__restrict__ arena, uint32_t n, KvqBucketPlan B, const uint32_t *__restrict__ start,
                   uint32_t *__restrict__ fill, uint32_t *__restrict__ idx)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n) return;
    const uint32_t b = (uint32_t)((uint64_t)(arena[h].fpos - B.lo) >> B.shift);
    idx[start[b] + atomicAdd(&fill[b], 1u)] = h;
}

// one thread per bucket: its (few) hit numbers put in order.  Up to eight hits: sorting network in
// registers.  Nine to KVQ_BUCKET_MAX (a read on a locus that many templates share): the wave takes
// such buckets one at a time, lane i holds hit i and counts the hits in front of it (no dependent
// loads: a serial insertion sort of 20 hits used to set the time of the whole kernel).
__global__ void __launch_bounds__(256)
kvq_bucket_sort(const KvqHit *__restrict__ arena, KvqBucketPlan B, const uint32_t *__restrict__ start, uint32_t *__restrict__ idx,
                uint32_t *__restrict__ crowded)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = kvq_lane();
    uint32_t s0 = 0, m = 0;
    if (b < B.nb) { s0 = start[b]; m = start[b + 1] - s0; }
    if (m > KVQ_BUCKET_MAX) { *crowded = 1u; m = 0; }
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

struct KvqHitBytesAt {
    const KvqHit *arena;
    __host__ __device__ long long operator()(uint32_t h) const { const int l = arena[h].length; return l > 0 ? (long long)l : 0ll; }
};

// one wave per hit: the five columns, the closing offset, and the hit bytes moved from where
// kvq_fold_hits left them to their place in canonical order; hit i is arena[order[i]] (order == nullptr: arena[i])
__global__ void __launch_bounds__(256)
kvq_gather_results(const KvqHit *__restrict__ arena, const uint32_t *__restrict__ order, uint32_t n, const uint8_t *__restrict__ blob_in,
                   uint8_t *__restrict__ res, KvqResultLayout L)
{
    const int lane = kvq_lane();
    const uint32_t per = (blockDim.x >> 6) * gridDim.x;
    long long *off = reinterpret_cast<long long *>(res + L.hitseq_off);
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
            if (i + 1 == n) off[n] = at + len;
        }
        const uint8_t *src = blob_in + h.blob_off;
        uint8_t *dst = res + L.blob + at;
        for (int j = lane; j < len; j += 64) dst[j] = src[j];
    }
}

// enqueue ordering + offsets + gather for the n hits of the arena whose file positions lie in
// [lo, hi); tmp / sorted are grow-only scratch.  by_buckets: the counting sort (*d_crowded is
// raised when it gave up; the caller then calls again with by_buckets = false)
static int kvq_order_results(hipStream_t stream, const KvqHit *arena, uint32_t n, const uint8_t *blob_in,
                             DevBuf &tmp, DevBuf &sorted, uint8_t *res, const KvqResultLayout &L,
                             bool by_buckets, int64_t lo, int64_t hi, uint32_t *d_crowded)
{
    if (n == 0) return KVQ_OK;
    int rc;
    long long *off = reinterpret_cast<long long *>(res + L.hitseq_off);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 3) / 4, 8192);
    if (by_buckets) {
        KvqBucketPlan B; B.lo = lo;
        uint32_t want = 256; while (want < 4ull * n && want < (1u << 22)) want <<= 1;      // about four buckets per hit: most hold none or one
        const uint64_t span = (uint64_t)(hi > lo ? hi - lo : 1);
        B.shift = 0; while ((span >> B.shift) >= (uint64_t)want) B.shift++;
        B.nb = (uint32_t)(span >> B.shift) + 1u;                           // <= want
        // scratch: start[nb + 1] | fill[nb] | idx[n] | scan storage
        const size_t start_b = (((size_t)B.nb + 1) * 4 + 255) & ~(size_t)255, fill_b = ((size_t)B.nb * 4 + 255) & ~(size_t)255,
                     idx_b = ((size_t)n * 4 + 255) & ~(size_t)255;
        size_t need_scan_a = 0, need_scan_b = 0;
        uint32_t *null32 = nullptr;
        KVQ_HIP(rocprim::inclusive_scan(nullptr, need_scan_a, null32, null32, (size_t)B.nb, rocprim::plus<uint32_t>(), stream));
        auto lens0 = rocprim::make_transform_iterator(null32, KvqHitBytesAt{ arena });
        KVQ_HIP(rocprim::exclusive_scan(nullptr, need_scan_b, lens0, off, 0ll, (size_t)n, rocprim::plus<long long>(), stream));
        const size_t scan_b = std::max(need_scan_a, need_scan_b) + 256;
        if ((rc = tmp.ensure(start_b + fill_b + idx_b + scan_b))) return rc;
        uint8_t *base = (uint8_t *)tmp.p;
        uint32_t *start = (uint32_t *)base, *fill = (uint32_t *)(base + start_b), *idx = (uint32_t *)(base + start_b + fill_b);
        void *scan_tmp = base + start_b + fill_b + idx_b;
        KVQ_HIP(hipMemsetAsync(base, 0, start_b + fill_b, stream));
        KVQ_HIP(hipMemsetAsync(d_crowded, 0, 4, stream));
        hipLaunchKernelGGL(kvq_bucket_count, dim3((n + 255) / 256), dim3(256), 0, stream, arena, n, B, start);
        KVQ_HIP(rocprim::inclusive_scan(scan_tmp, need_scan_a, start + 1, start + 1, (size_t)B.nb, rocprim::plus<uint32_t>(), stream));
        hipLaunchKernelGGL(kvq_bucket_scatter, dim3((n + 255) / 256), dim3(256), 0, stream, arena, n, B, (const uint32_t *)start, fill, idx);
        hipLaunchKernelGGL(kvq_bucket_sort, dim3((B.nb + 255) / 256), dim3(256), 0, stream, arena, B, (const uint32_t *)start, idx, d_crowded);
        auto lens = rocprim::make_transform_iterator((const uint32_t *)idx, KvqHitBytesAt{ arena });
        KVQ_HIP(rocprim::exclusive_scan(scan_tmp, need_scan_b, lens, off, 0ll, (size_t)n, rocprim::plus<long long>(), stream));
        hipLaunchKernelGGL(kvq_gather_results, dim3(blocks), dim3(256), 0, stream, arena, (const uint32_t *)idx, n, blob_in, res, L);
        KVQ_HIP(hipGetLastError());
        return KVQ_OK;
    }
    if ((rc = sorted.ensure((size_t)n * sizeof(KvqHit)))) return rc;
    KvqHit *out = sorted.as<KvqHit>();
    size_t need_sort = 0, need_scan = 0;
    auto lens = rocprim::make_transform_iterator(out, KvqHitBytes());
    KVQ_HIP(rocprim::merge_sort(nullptr, need_sort, arena, out, (size_t)n, KvqHitBefore(), stream));
    KVQ_HIP(rocprim::exclusive_scan(nullptr, need_scan, lens, off, 0ll, (size_t)n, rocprim::plus<long long>(), stream));
    if ((rc = tmp.ensure(std::max(need_sort, need_scan) + 256))) return rc;
    KVQ_HIP(rocprim::merge_sort(tmp.p, need_sort, arena, out, (size_t)n, KvqHitBefore(), stream));
    KVQ_HIP(rocprim::exclusive_scan(tmp.p, need_scan, lens, off, 0ll, (size_t)n, rocprim::plus<long long>(), stream));
    hipLaunchKernelGGL(kvq_gather_results, dim3(blocks), dim3(256), 0, stream, (const KvqHit *)out, (const uint32_t *)nullptr, n, blob_in, res, L);
    KVQ_HIP(hipGetLastError());
    return KVQ_OK;
}
