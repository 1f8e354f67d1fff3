// kvarq_amd/csrc/kvq_dist.hip -- several GPUs, one process each: the reference joins its worker threads and
// hands back one set of counters and one list of hits (csrc/workhorse.c:1375-1447).  Reads shard without any
// exchange on the data path; what the ranks exchange at the end goes over RCCL (xGMI inside a node):
//   * kvq_scan_set_comm: `finish` becomes collective.  The ranks first agree on how their own scans ended (one
//     max over a status word): only when every rank is fine are the counter arrays summed (one all-reduce into a
//     buffer of its own, so that a repeated finish does not sum sums; the slot of the longest read takes the
//     maximum); when a rank has to feed its host batches again (KVQ_ERR_RESCAN) EVERY rank is told so and the
//     sum is not taken, so that all ranks go round together; any other failure surfaces on all of them.
//   * kvq_scan_gather_hits: the result arrays of all ranks, concatenated in rank order (ranks scan
//     consecutive stretches of the stream, so that is file order), on every rank.  Where each rank's arrays go is
//     worked out by kvq_gather_plan, a host function of the counts alone (tests/test_coverage_and_dist.py runs it
//     for 2, 3 and 8 ranks on the CPU against the oracle's single scan).
// A communicator is either RCCL (librccl.so, loaded when the first one is made: a single-GPU user of the library
// never touches it) or the in-process loopback kvq_comm_create_local makes -- N threads of one process, each with
// a scan of its own on whatever GPU it has set, exchanging through host memory: the very same join code with more
// than one rank on a box that has one GPU (RCCL refuses two ranks on one device).
#include "kvq_host.h"

#include <dlfcn.h>
#include <rccl/rccl.h>            // (types and enumerators only: nothing of it is linked)
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string.h>

// ---- RCCL, by dlopen ---------------------------------------------------------------------------------------
struct KvqRccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static_assert(sizeof(ncclUniqueId) == 128, "include/kvarq_hip.h promises a 128-byte id");

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
    KVQ_SYM(AllReduce, "ncclAllReduce"); KVQ_SYM(AllGather, "ncclAllGather"); KVQ_SYM(Broadcast, "ncclBroadcast");
    KVQ_SYM(GroupStart, "ncclGroupStart"); KVQ_SYM(GroupEnd, "ncclGroupEnd"); KVQ_SYM(GetErrorString, "ncclGetErrorString");
#undef KVQ_SYM
    g_rccl = r;
    return KVQ_OK;
}

#define KVQ_NCCL(call)                                                                          \
    do {                                                                                        \
        const ncclResult_t e_ = (call);                                                         \
        if (e_ != ncclSuccess) {                                                                \
            kvq_set_error(KVQ_ERR_RUNTIME, "%s failed: %s", #call, g_rccl.GetErrorString(e_));  \
            return KVQ_ERR_RUNTIME;                                                             \
        }                                                                                       \
    } while (0)

// ---- the loopback: N threads of one process --------------------------------------------------------------
struct KvqLocalWorld {
    int nranks = 0;
    std::mutex m; std::condition_variable cv;
    int waiting = 0; uint64_t generation = 0;
    std::vector<std::vector<uint8_t>> stage;          // what each rank has put up for the others
    void barrier()
    {
        std::unique_lock<std::mutex> l(m);
        const uint64_t g = generation;
        if (++waiting == nranks) { waiting = 0; generation++; cv.notify_all(); }
        else cv.wait(l, [&] { return generation != g; });
    }
};
static std::mutex g_worlds_lock;
static std::map<uint64_t, std::weak_ptr<KvqLocalWorld>> g_worlds;

struct KvqBcastPart { const void *src; void *dst; size_t bytes; int root; };

struct kvq_comm {
    int nranks = 1, rank = 0;
    ncclComm_t c = nullptr;                           // RCCL ...
    std::shared_ptr<KvqLocalWorld> w;                 // ... or the loopback
    std::vector<KvqBcastPart> parts;                  // the broadcasts of an open group

    int all_reduce_u64(const unsigned long long *d_in, unsigned long long *d_out, size_t n, bool take_max, hipStream_t stream)
    {
        if (!w) { KVQ_NCCL(g_rccl.AllReduce(d_in, d_out, n, ncclUint64, take_max ? ncclMax : ncclSum, c, stream)); return KVQ_OK; }
        KVQ_HIP(hipStreamSynchronize(stream));
        std::vector<uint8_t> &mine = w->stage[(size_t)rank];
        mine.resize(n * 8);
        KVQ_HIP(hipMemcpy(mine.data(), d_in, n * 8, hipMemcpyDeviceToHost));
        w->barrier();
        std::vector<unsigned long long> acc(n, 0);
        for (int r = 0; r < nranks; r++) {
            const unsigned long long *v = reinterpret_cast<const unsigned long long *>(w->stage[(size_t)r].data());
            for (size_t i = 0; i < n; i++) acc[i] = take_max ? std::max(acc[i], v[i]) : acc[i] + v[i];
        }
        w->barrier();                                 // (everyone has read every stage: they may be written again)
        KVQ_HIP(hipMemcpy(d_out, acc.data(), n * 8, hipMemcpyHostToDevice));
        return KVQ_OK;
    }
    int all_gather_u64(const unsigned long long *d_in, unsigned long long *d_out, size_t n_per_rank, hipStream_t stream)
    {
        if (!w) { KVQ_NCCL(g_rccl.AllGather(d_in, d_out, n_per_rank, ncclUint64, c, stream)); return KVQ_OK; }
        KVQ_HIP(hipStreamSynchronize(stream));
        std::vector<uint8_t> &mine = w->stage[(size_t)rank];
        mine.resize(n_per_rank * 8);
        KVQ_HIP(hipMemcpy(mine.data(), d_in, n_per_rank * 8, hipMemcpyDeviceToHost));
        w->barrier();
        std::vector<uint8_t> all((size_t)nranks * n_per_rank * 8);
        for (int r = 0; r < nranks; r++) memcpy(all.data() + (size_t)r * n_per_rank * 8, w->stage[(size_t)r].data(), n_per_rank * 8);
        w->barrier();
        KVQ_HIP(hipMemcpy(d_out, all.data(), all.size(), hipMemcpyHostToDevice));
        return KVQ_OK;
    }
    // broadcasts are collected and sent as one group (every rank lists the same parts in the same order)
    void bcast(const void *d_src, void *d_dst, size_t bytes, int root) { if (bytes) parts.push_back({ d_src, d_dst, bytes, root }); }
    int bcast_flush(hipStream_t stream)
    {
        std::vector<KvqBcastPart> todo; todo.swap(parts);
        if (!w) {
            KVQ_NCCL(g_rccl.GroupStart());
            ncclResult_t bad = ncclSuccess;
            for (const KvqBcastPart &p : todo) { const ncclResult_t e = g_rccl.Broadcast(p.src, p.dst, p.bytes, ncclUint8, p.root, c, stream); if (e != ncclSuccess && bad == ncclSuccess) bad = e; }
            const ncclResult_t e2 = g_rccl.GroupEnd();            // (the group is closed whatever a broadcast said)
            if (bad != ncclSuccess || e2 != ncclSuccess) { kvq_set_error(KVQ_ERR_RUNTIME, "ncclBroadcast failed: %s", g_rccl.GetErrorString(bad != ncclSuccess ? bad : e2)); return KVQ_ERR_RUNTIME; }
            return KVQ_OK;
        }
        KVQ_HIP(hipStreamSynchronize(stream));
        std::vector<uint8_t> &mine = w->stage[(size_t)rank];
        size_t own = 0; for (const KvqBcastPart &p : todo) if (p.root == rank) own += p.bytes;
        mine.resize(own);
        { size_t at = 0; for (const KvqBcastPart &p : todo) if (p.root == rank) { KVQ_HIP(hipMemcpy(mine.data() + at, p.src, p.bytes, hipMemcpyDeviceToHost)); at += p.bytes; } }
        w->barrier();
        std::vector<size_t> at((size_t)nranks, 0);
        for (const KvqBcastPart &p : todo) {
            KVQ_HIP(hipMemcpy(p.dst, w->stage[(size_t)p.root].data() + at[(size_t)p.root], p.bytes, hipMemcpyHostToDevice));
            at[(size_t)p.root] += p.bytes;
        }
        w->barrier();
        return KVQ_OK;
    }
};

extern "C" int32_t kvq_comm_unique_id(void *id128)
{
    kvq_clear_error();
    int rc = rccl_load(); if (rc) return rc;
    ncclUniqueId id;
    KVQ_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return KVQ_OK;
}

extern "C" kvq_comm *kvq_comm_create(int32_t nranks, int32_t rank, const void *id128)
{
    kvq_clear_error();
    if (rccl_load()) return nullptr;
    ncclUniqueId id; memcpy(&id, id128, sizeof(id));
    kvq_comm *c = new kvq_comm();
    const ncclResult_t e = g_rccl.CommInitRank(&c->c, nranks, id, rank);
    if (e != ncclSuccess) { kvq_set_error(KVQ_ERR_RUNTIME, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(e)); delete c; return nullptr; }
    c->nranks = nranks; c->rank = rank;
    return c;
}

extern "C" kvq_comm *kvq_comm_create_local(int32_t nranks, int32_t rank, uint64_t world_key)
{
    kvq_clear_error();
    if (nranks < 1 || rank < 0 || rank >= nranks) { kvq_set_error(KVQ_ERR_RUNTIME, "bad rank %d of %d", (int)rank, (int)nranks); return nullptr; }
    std::shared_ptr<KvqLocalWorld> w;
    {
        std::lock_guard<std::mutex> l(g_worlds_lock);
        w = g_worlds[world_key].lock();
        if (!w) { w = std::make_shared<KvqLocalWorld>(); w->nranks = nranks; w->stage.resize((size_t)nranks); g_worlds[world_key] = w; }
    }
    if (w->nranks != nranks) { kvq_set_error(KVQ_ERR_RUNTIME, "the ranks of world %llu disagree about their number", (unsigned long long)world_key); return nullptr; }
    kvq_comm *c = new kvq_comm();
    c->nranks = nranks; c->rank = rank; c->w = w;
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

// the counter arrays of all ranks summed on `stream`: d_in -> d_out (they may be the same array); slot
// KVQ_CTR_LONGEST takes the maximum (d_scratch: two 8-byte words of device memory)
int kvq_comm_reduce_counters(kvq_comm *c, const unsigned long long *d_in, unsigned long long *d_out, int64_t ctr_len, unsigned long long *d_scratch, hipStream_t stream)
{
    if (!c) return KVQ_OK;            // (a communicator of one rank goes through the transport like any other)
    int rc;
    KVQ_HIP(hipMemcpyAsync(d_scratch, d_in + KVQ_CTR_LONGEST_, 8, hipMemcpyDeviceToDevice, stream));
    if ((rc = c->all_reduce_u64(d_in, d_out, (size_t)ctr_len, false, stream))) return rc;
    if ((rc = c->all_reduce_u64(d_scratch, d_scratch + 1, 1, true, stream))) return rc;
    KVQ_HIP(hipMemcpyAsync(d_out + KVQ_CTR_LONGEST_, d_scratch + 1, 8, hipMemcpyDeviceToDevice, stream));
    return KVQ_OK;
}

// the largest of the ranks' status words, on every rank (d_scratch: two words of device memory)
int kvq_comm_max_status(kvq_comm *c, unsigned long long mine, unsigned long long *d_scratch, hipStream_t stream, unsigned long long *out)
{
    int rc;
    KVQ_HIP(hipMemcpyAsync(d_scratch, &mine, 8, hipMemcpyHostToDevice, stream));
    KVQ_HIP(hipStreamSynchronize(stream));                       // (`mine` lives on this stack frame)
    if ((rc = c->all_reduce_u64(d_scratch, d_scratch + 1, 1, true, stream))) return rc;
    KVQ_HIP(hipMemcpyAsync(out, d_scratch + 1, 8, hipMemcpyDeviceToHost, stream));
    KVQ_HIP(hipStreamSynchronize(stream));
    return KVQ_OK;
}

// (what the collective finish needs on the device is allocated HERE, where a failure is this rank's own affair: a rank that
// could not allocate inside `finish` would leave the others waiting in a collective it never enters)
extern "C" int32_t kvq_scan_set_comm(kvq_scan *s, kvq_comm *c)
{
    kvq_clear_error();
    if (c) {
        int rc;
        if ((rc = s->d_finish.ensure(sizeof(KvqFinishState) + 256))) return rc;
        if ((rc = s->d_ctr_all.ensure((size_t)s->t->ctr_len * 8))) return rc;
    }
    s->comm = c;
    return KVQ_OK;
}

// a free-standing form for callers that hold the counters of several scans in one device array of their own
extern "C" int32_t kvq_comm_allreduce_counters(kvq_comm *c, void *d_counters, int64_t ctr_len, void *d_scratch16)
{
    kvq_clear_error();
    int rc = kvq_comm_reduce_counters(c, (const unsigned long long *)d_counters, (unsigned long long *)d_counters, ctr_len, (unsigned long long *)d_scratch16, nullptr);
    if (rc) return rc;
    KVQ_HIP(hipStreamSynchronize(nullptr));
    return KVQ_OK;
}

// ---- where the ranks' result arrays go -------------------------------------------------------------------
// counts[2 r] = hits of rank r, counts[2 r + 1] = its hit bytes.  Out, per rank, seven parts in the order
// file_pos, hitseq offsets, seq_nr, seq_pos, length, readlength, hit bytes: {offset in the rank's own result
// buffer, offset in the gathered one, bytes} = parts[(7 r + i) * 3 ...]; totals = {hits, hit bytes, bytes of the
// gathered buffer, offset of its hitseq-offset array}; blob_base[r] = what is added to rank r's hitseq offsets.
extern "C" int32_t kvq_gather_plan(int32_t nranks, const uint64_t *counts, uint64_t *parts, uint64_t *blob_base, uint64_t *totals)
{
    std::vector<uint64_t> h0((size_t)nranks + 1, 0), b0((size_t)nranks + 1, 0);
    for (int r = 0; r < nranks; r++) { h0[(size_t)r + 1] = h0[(size_t)r] + counts[2 * r]; b0[(size_t)r + 1] = b0[(size_t)r] + counts[2 * r + 1]; }
    const uint64_t n = h0[(size_t)nranks], blob = b0[(size_t)nranks];
    if (n > 0xFFFFFFF0ull) return KVQ_ERR_MEMORY;
    const KvqResultLayout L = kvq_result_layout(n, blob);
    for (int r = 0; r < nranks; r++) {
        const uint64_t k = counts[2 * r], kb = counts[2 * r + 1];
        const KvqResultLayout M = kvq_result_layout(k, kb);
        const uint64_t row[7][3] = {
            { M.file_pos, L.file_pos + h0[(size_t)r] * 8, k * 8 }, { M.hitseq_off, L.hitseq_off + h0[(size_t)r] * 8, k * 8 },
            { M.seq_nr, L.seq_nr + h0[(size_t)r] * 4, k * 4 }, { M.seq_pos, L.seq_pos + h0[(size_t)r] * 4, k * 4 },
            { M.length, L.length + h0[(size_t)r] * 4, k * 4 }, { M.readlength, L.readlength + h0[(size_t)r] * 4, k * 4 },
            { M.blob, L.blob + b0[(size_t)r], kb } };
        memcpy(parts + (size_t)r * 21, row, sizeof(row));
        blob_base[r] = b0[(size_t)r];
    }
    totals[0] = n; totals[1] = blob; totals[2] = L.total; totals[3] = L.hitseq_off;
    return KVQ_OK;
}

// the same plan carried out in host memory (tests: the plan without a GPU): rank_bufs[r] = rank r's result
// buffer as kvq_result_layout lays it out, out = the gathered buffer (totals[2] bytes)
extern "C" int32_t kvq_gather_host(int32_t nranks, const uint64_t *counts, const uint8_t *const *rank_bufs, uint8_t *out)
{
    std::vector<uint64_t> parts((size_t)nranks * 21), base((size_t)nranks); uint64_t tot[4];
    const int rc = kvq_gather_plan(nranks, counts, parts.data(), base.data(), tot);
    if (rc) return rc;
    for (int r = 0; r < nranks; r++)
        for (int i = 0; i < 7; i++) {
            const uint64_t *p = &parts[((size_t)r * 7 + (size_t)i) * 3];
            if (p[2]) memcpy(out + p[1], rank_bufs[r] + p[0], (size_t)p[2]);
        }
    long long *off = reinterpret_cast<long long *>(out + tot[3]);
    uint64_t at = 0;
    for (int r = 0; r < nranks; r++) { for (uint64_t i = 0; i < counts[2 * r]; i++) off[at + i] += (long long)base[(size_t)r]; at += counts[2 * r]; }
    off[tot[0]] = (long long)tot[1];
    return KVQ_OK;
}
extern "C" void kvq_result_layout_words(uint64_t n, uint64_t blob_bytes, uint64_t *out8)
{
    const KvqResultLayout L = kvq_result_layout(n, blob_bytes);
    const uint64_t v[8] = { L.file_pos, L.hitseq_off, L.seq_nr, L.seq_pos, L.length, L.readlength, L.blob, L.total };
    memcpy(out8, v, sizeof(v));
}

// After kvq_scan_finish on every rank: the hits of all ranks, in rank order, take the place of this
// rank's own in the scan's result arrays (kvq_scan_n_hits, kvq_scan_hit_*, kvq_scan_hitseq_*).  Counts first
// (one all-gather of two words per rank), then every array as one broadcast per rank into its place.  The
// rank's own arrays stay where they are on the device: a second call finds the gathered ones in place.
extern "C" int32_t kvq_scan_gather_hits(kvq_scan *s, kvq_comm *c)
{
    kvq_clear_error();
    if (!s->finished) { kvq_set_error(KVQ_ERR_RUNTIME, "kvq_scan_gather_hits before kvq_scan_finish"); return KVQ_ERR_RUNTIME; }
    if (!c || s->gathered) return KVQ_OK;
    const int N = c->nranks;
    int rc;
    if ((rc = s->d_gather_cnt.ensure((size_t)(N + 1) * 16))) return rc;
    unsigned long long mine[2] = { s->n_hits, (unsigned long long)(s->n_hits ? reinterpret_cast<const long long *>(s->pin_res + s->res.hitseq_off)[s->n_hits] : 0) };
    std::vector<uint64_t> all((size_t)N * 2);
    unsigned long long *d_mine = s->d_gather_cnt.as<unsigned long long>(), *d_all = d_mine + 2;
    KVQ_HIP(hipMemcpyAsync(d_mine, mine, 16, hipMemcpyHostToDevice, s->stream));
    KVQ_HIP(hipStreamSynchronize(s->stream));
    if ((rc = c->all_gather_u64(d_mine, d_all, 2, s->stream))) return rc;
    KVQ_HIP(hipMemcpyAsync(all.data(), d_all, (size_t)N * 16, hipMemcpyDeviceToHost, s->stream));
    KVQ_HIP(hipStreamSynchronize(s->stream));
    std::vector<uint64_t> parts((size_t)N * 21), base((size_t)N); uint64_t tot[4];
    if (kvq_gather_plan(N, all.data(), parts.data(), base.data(), tot)) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results"); return KVQ_ERR_MEMORY; }
    const uint64_t n = tot[0], blob = tot[1];
    const KvqResultLayout L = kvq_result_layout(n, blob);
    if ((rc = s->d_gather_res.ensure(L.total + 256))) return rc;
    uint8_t *dst = s->d_gather_res.as<uint8_t>(); const uint8_t *src = s->d_result.as<uint8_t>();
    for (int r = 0; r < N; r++)
        for (int i = 0; i < 7; i++) {
            const uint64_t *p = &parts[((size_t)r * 7 + (size_t)i) * 3];
            c->bcast(src + p[0], dst + p[1], (size_t)p[2], r);            // (p[0] is the sender's own layout: the same function of its counts on every rank)
        }
    if ((rc = c->bcast_flush(s->stream))) return rc;
    // to the host (behind the counters, as `finish` lays the landing buffer out); hitseq offsets: rank r's start at base[r]
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
    uint64_t at = 0;
    for (int r = 0; r < N; r++) { for (uint64_t i = 0; i < all[2 * (size_t)r]; i++) off[at + i] += (long long)base[(size_t)r]; at += all[2 * (size_t)r]; }
    off[n] = (long long)blob;
    s->res = L; s->n_hits = n; s->gathered = true;
    return KVQ_OK;
}
