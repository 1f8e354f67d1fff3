// kvarq_amd/csrc/kvq_dist.hip -- several GPUs, one process each: the reference joins its worker threads and
// hands back one set of counters and one list of hits (csrc/workhorse.c:1375-1447).  Reads shard without any
// exchange on the data path; what the ranks exchange at the end goes over RCCL (xGMI inside a node):
//   * kvq_scan_set_comm: `finish` then sums the counter array of all ranks on the scan's stream (one
//     all-reduce; the slot of the longest read takes the maximum) before it goes to the host;
//   * kvq_scan_gather_hits: the result arrays of all ranks, concatenated in rank order (ranks scan
//     consecutive stretches of the stream, so that is file order), on every rank.
// librccl.so is loaded when the first communicator is made: a single-GPU user of the library never touches it.
#include "kvq_host.h"

#include <dlfcn.h>
#include <mutex>

// the few RCCL entry points used, by their C signatures (rccl/rccl.h)
typedef struct ncclComm *kvq_nccl_comm;
typedef struct { char internal[128]; } kvq_nccl_id;
struct KvqRccl {
    void *lib = nullptr;
    int (*GetUniqueId)(kvq_nccl_id *) = nullptr;
    int (*CommInitRank)(kvq_nccl_comm *, int, kvq_nccl_id, int) = nullptr;
    int (*CommDestroy)(kvq_nccl_comm) = nullptr;
    int (*CommCount)(const kvq_nccl_comm, int *) = nullptr;
    int (*CommUserRank)(const kvq_nccl_comm, int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, kvq_nccl_comm, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, kvq_nccl_comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, kvq_nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
enum { KVQ_NCCL_SUM = 0, KVQ_NCCL_MAX = 2, KVQ_NCCL_UINT8 = 1, KVQ_NCCL_UINT64 = 5 };      // ncclRedOp_t / ncclDataType_t

static KvqRccl g_rccl;
static std::mutex g_rccl_lock;

static int rccl_load()
{
    std::lock_guard<std::mutex> l(g_rccl_lock);
    if (g_rccl.lib) return KVQ_OK;
    const char *names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void *h = nullptr;
    // an RCCL that the process has loaded already (a host that runs torch.distributed has its own copy) is the one to
    // use: one instance per process; only otherwise is one looked for
    for (const char *n : { "librccl.so.1", "librccl.so" }) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h) for (const char *n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { kvq_set_error(KVQ_ERR_RUNTIME, "cannot load librccl.so: %s", dlerror()); return KVQ_ERR_RUNTIME; }
    KvqRccl r; r.lib = h;
#define KVQ_SYM(field, name) do { *(void **)(&r.field) = dlsym(h, name); if (!r.field) { kvq_set_error(KVQ_ERR_RUNTIME, "librccl.so lacks %s", name); dlclose(h); return KVQ_ERR_RUNTIME; } } while (0)
    KVQ_SYM(GetUniqueId, "ncclGetUniqueId"); KVQ_SYM(CommInitRank, "ncclCommInitRank"); KVQ_SYM(CommDestroy, "ncclCommDestroy");
    KVQ_SYM(CommCount, "ncclCommCount"); KVQ_SYM(CommUserRank, "ncclCommUserRank");
    KVQ_SYM(AllReduce, "ncclAllReduce"); KVQ_SYM(AllGather, "ncclAllGather"); KVQ_SYM(Broadcast, "ncclBroadcast");
    KVQ_SYM(GroupStart, "ncclGroupStart"); KVQ_SYM(GroupEnd, "ncclGroupEnd"); KVQ_SYM(GetErrorString, "ncclGetErrorString");
#undef KVQ_SYM
    g_rccl = r;
    return KVQ_OK;
}

#define KVQ_NCCL(call)                                                                          \
    do {                                                                                        \
        const int e_ = (call);                                                                  \
        if (e_ != 0) {                                                                          \
            kvq_set_error(KVQ_ERR_RUNTIME, "%s failed: %s", #call, g_rccl.GetErrorString(e_));  \
            return KVQ_ERR_RUNTIME;                                                             \
        }                                                                                       \
    } while (0)

struct kvq_comm { kvq_nccl_comm c = nullptr; int nranks = 1, rank = 0; };

extern "C" int32_t kvq_comm_unique_id(void *id128)
{
    kvq_clear_error();
    int rc = rccl_load(); if (rc) return rc;
    kvq_nccl_id id;
    KVQ_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return KVQ_OK;
}

extern "C" kvq_comm *kvq_comm_create(int32_t nranks, int32_t rank, const void *id128)
{
    kvq_clear_error();
    if (rccl_load()) return nullptr;
    kvq_nccl_id id; memcpy(&id, id128, sizeof(id));
    kvq_comm *c = new kvq_comm();
    const int e = g_rccl.CommInitRank(&c->c, nranks, id, rank);
    if (e != 0) { kvq_set_error(KVQ_ERR_RUNTIME, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(e)); delete c; return nullptr; }
    c->nranks = nranks; c->rank = rank;
    return c;
}

extern "C" void kvq_comm_destroy(kvq_comm *c)
{
    if (!c) return;
    if (c->c) (void)g_rccl.CommDestroy(c->c);
    delete c;
}
extern "C" int32_t kvq_comm_nranks(const kvq_comm *c) { return c ? c->nranks : 1; }
extern "C" int32_t kvq_comm_rank(const kvq_comm *c) { return c ? c->rank : 0; }

// the counter array of all ranks summed in place on `stream`; slot KVQ_CTR_LONGEST takes the maximum
// (d_scratch: two 8-byte words of device memory)
int kvq_comm_reduce_counters(kvq_comm *c, unsigned long long *d_ctr, int64_t ctr_len, unsigned long long *d_scratch, hipStream_t stream)
{
    if (!c) return KVQ_OK;            // (a communicator of one rank goes through RCCL like any other)
    KVQ_HIP(hipMemcpyAsync(d_scratch, d_ctr + KVQ_CTR_LONGEST_, 8, hipMemcpyDeviceToDevice, stream));
    KVQ_NCCL(g_rccl.GroupStart());
    KVQ_NCCL(g_rccl.AllReduce(d_ctr, d_ctr, (size_t)ctr_len, KVQ_NCCL_UINT64, KVQ_NCCL_SUM, c->c, stream));
    KVQ_NCCL(g_rccl.AllReduce(d_scratch, d_scratch + 1, 1, KVQ_NCCL_UINT64, KVQ_NCCL_MAX, c->c, stream));
    KVQ_NCCL(g_rccl.GroupEnd());
    KVQ_HIP(hipMemcpyAsync(d_ctr + KVQ_CTR_LONGEST_, d_scratch + 1, 8, hipMemcpyDeviceToDevice, stream));
    return KVQ_OK;
}

extern "C" int32_t kvq_scan_set_comm(kvq_scan *s, kvq_comm *c) { s->comm = c; return KVQ_OK; }

// a free-standing form for callers that hold the counters of several scans in one device array of their own
extern "C" int32_t kvq_comm_allreduce_counters(kvq_comm *c, void *d_counters, int64_t ctr_len, void *d_scratch16)
{
    kvq_clear_error();
    int rc = kvq_comm_reduce_counters(c, (unsigned long long *)d_counters, ctr_len, (unsigned long long *)d_scratch16, nullptr);
    if (rc) return rc;
    KVQ_HIP(hipStreamSynchronize(nullptr));
    return KVQ_OK;
}

// After kvq_scan_finish on every rank: the hits of all ranks, in rank order, take the place of this
// rank's own in the scan's result arrays (kvq_scan_n_hits, kvq_scan_hit_*, kvq_scan_hitseq_*).  Counts first
// (one all-gather of two words per rank), then every array as one broadcast per rank into its place.
extern "C" int32_t kvq_scan_gather_hits(kvq_scan *s, kvq_comm *c)
{
    kvq_clear_error();
    if (!s->finished) { kvq_set_error(KVQ_ERR_RUNTIME, "kvq_scan_gather_hits before kvq_scan_finish"); return KVQ_ERR_RUNTIME; }
    if (!c) return KVQ_OK;
    const int N = c->nranks;
    int rc;
    DevBuf d_cnt;
    if ((rc = d_cnt.ensure((size_t)(N + 1) * 16))) return rc;
    unsigned long long mine[2] = { s->n_hits, (unsigned long long)(s->n_hits ? reinterpret_cast<const long long *>(s->pin_res + s->res.hitseq_off)[s->n_hits] : 0) };
    std::vector<unsigned long long> all((size_t)N * 2);
    unsigned long long *d_mine = d_cnt.as<unsigned long long>(), *d_all = d_mine + 2;
    KVQ_HIP(hipMemcpyAsync(d_mine, mine, 16, hipMemcpyHostToDevice, s->stream));
    KVQ_NCCL(g_rccl.AllGather(d_mine, d_all, 2, KVQ_NCCL_UINT64, c->c, s->stream));
    KVQ_HIP(hipMemcpyAsync(all.data(), d_all, (size_t)N * 16, hipMemcpyDeviceToHost, s->stream));
    KVQ_HIP(hipStreamSynchronize(s->stream));
    std::vector<uint64_t> h0((size_t)N + 1, 0), b0((size_t)N + 1, 0);
    for (int r = 0; r < N; r++) { h0[r + 1] = h0[r] + all[2 * r]; b0[r + 1] = b0[r] + all[2 * r + 1]; }
    const uint64_t n = h0[N], blob = b0[N];
    if (n > 0xFFFFFFF0ull) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results"); d_cnt.release(); return KVQ_ERR_MEMORY; }
    const KvqResultLayout L = kvq_result_layout(n, blob), M = s->res;
    DevBuf d_all_res;
    if ((rc = d_all_res.ensure(L.total + 256))) { d_cnt.release(); return rc; }
    uint8_t *dst = d_all_res.as<uint8_t>(); const uint8_t *src = s->d_result.as<uint8_t>();
    KVQ_NCCL(g_rccl.GroupStart());
    for (int r = 0; r < N; r++) {
        const size_t k = (size_t)all[2 * r], kb = (size_t)all[2 * r + 1];
        struct { size_t from, to, bytes; } parts[7] = {
            { M.file_pos, L.file_pos + h0[r] * 8, k * 8 }, { M.hitseq_off, L.hitseq_off + h0[r] * 8, k * 8 },
            { M.seq_nr, L.seq_nr + h0[r] * 4, k * 4 }, { M.seq_pos, L.seq_pos + h0[r] * 4, k * 4 },
            { M.length, L.length + h0[r] * 4, k * 4 }, { M.readlength, L.readlength + h0[r] * 4, k * 4 },
            { M.blob, L.blob + b0[r], kb } };
        for (auto &p : parts)
            if (p.bytes) KVQ_NCCL(g_rccl.Broadcast(src + p.from, dst + p.to, p.bytes, KVQ_NCCL_UINT8, r, c->c, s->stream));
    }
    KVQ_NCCL(g_rccl.GroupEnd());
    // to the host (behind the counters, as `finish` lays the landing buffer out); hitseq offsets: rank r's start at b0[r]
    const size_t ctr_b = (size_t)(s->pin_res - s->pin);
    if (ctr_b + L.total > s->pin_cap) {
        const size_t want = (ctr_b + L.total) * 5 / 4 + (1 << 20);
        uint8_t *np = nullptr;
        if (hipHostMalloc((void **)&np, want, hipHostMallocDefault) != hipSuccess) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results"); return KVQ_ERR_MEMORY; }
        memcpy(np, s->pin, ctr_b);
        (void)hipHostFree(s->pin);
        s->pin = np; s->pin_cap = want; s->pin_res = np + ctr_b;
    }
    KVQ_HIP(hipMemcpyAsync(s->pin_res, dst, L.total, hipMemcpyDeviceToHost, s->stream));
    KVQ_HIP(hipStreamSynchronize(s->stream));
    long long *off = reinterpret_cast<long long *>(s->pin_res + L.hitseq_off);
    for (int r = 0; r < N; r++)
        for (uint64_t i = h0[r]; i < h0[r + 1]; i++) off[i] += (long long)b0[r];
    off[n] = (long long)blob;
    s->res = L; s->n_hits = n;
    d_cnt.release(); d_all_res.release();
    return KVQ_OK;
}
