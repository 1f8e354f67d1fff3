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

// one wave per hit: the five columns, the closing offset, and the hit bytes moved from where
// kvq_fold_hits left them to their place in canonical order
__global__ void __launch_bounds__(256)
kvq_gather_results(const KvqHit *__restrict__ sorted, uint32_t n, const uint8_t *__restrict__ blob_in,
                   uint8_t *__restrict__ res, KvqResultLayout L)
{
    const int lane = kvq_lane();
    const uint32_t per = (blockDim.x >> 6) * gridDim.x;
    long long *off = reinterpret_cast<long long *>(res + L.hitseq_off);
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += per) {
        const KvqHit h = sorted[i];
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

// enqueue sort + offsets + gather for the n hits of the arena; *tmp / *sorted are grow-only scratch
static int kvq_order_results(hipStream_t stream, const KvqHit *arena, uint32_t n, const uint8_t *blob_in,
                             DevBuf &tmp, DevBuf &sorted, uint8_t *res, const KvqResultLayout &L)
{
    if (n == 0) return KVQ_OK;
    int rc;
    if ((rc = sorted.ensure((size_t)n * sizeof(KvqHit)))) return rc;
    KvqHit *out = sorted.as<KvqHit>();
    size_t need_sort = 0, need_scan = 0;
    auto lens = rocprim::make_transform_iterator(out, KvqHitBytes());
    long long *off = reinterpret_cast<long long *>(res + L.hitseq_off);
    KVQ_HIP(rocprim::merge_sort(nullptr, need_sort, arena, out, (size_t)n, KvqHitBefore(), stream));
    KVQ_HIP(rocprim::exclusive_scan(nullptr, need_scan, lens, off, 0ll, (size_t)n, rocprim::plus<long long>(), stream));
    if ((rc = tmp.ensure(std::max(need_sort, need_scan) + 256))) return rc;
    KVQ_HIP(rocprim::merge_sort(tmp.p, need_sort, arena, out, (size_t)n, KvqHitBefore(), stream));
    KVQ_HIP(rocprim::exclusive_scan(tmp.p, need_scan, lens, off, 0ll, (size_t)n, rocprim::plus<long long>(), stream));
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)n + 3) / 4, 8192);
    hipLaunchKernelGGL(kvq_gather_results, dim3(blocks), dim3(256), 0, stream, (const KvqHit *)out, n, blob_in, res, L);
    KVQ_HIP(hipGetLastError());
    return KVQ_OK;
}
