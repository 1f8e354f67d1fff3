// kvarq_amd/csrc/kvq_runtime.hip -- host runtime of libkvarq_hip.so: config and
// error state, the target table, the scan object (batches -> kernels -> hits),
// results.  The file reader / engine.findseqs driver lives in kvq_findseqs.hip.
#include "kvq_host.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <stdarg.h>
#include <string.h>
#include <chrono>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const bool g_timing = getenv("KVQ_TIMING") != nullptr;

// internal: arena too small or speculation failed; the caller must rescan
#define KVQ_NEED_RESCAN (-2)

// ---------------------------------------------------------------------------
// error state and config
// ---------------------------------------------------------------------------

static thread_local int  tl_err_code = 0;
static thread_local char tl_err_msg[1024] = "";

void kvq_set_error(int code, const char *fmt, ...)
{
    tl_err_code = code;
    va_list ap; va_start(ap, fmt);
    vsnprintf(tl_err_msg, sizeof(tl_err_msg), fmt, ap);
    va_end(ap);
}
void kvq_clear_error() { tl_err_code = 0; tl_err_msg[0] = 0; }
int  kvq_error_code() { return tl_err_code; }

extern "C" int32_t kvq_last_error(char *msg, size_t cap)
{
    if (msg && cap) { strncpy(msg, tl_err_msg, cap - 1); msg[cap - 1] = 0; }
    return tl_err_code;
}

static std::mutex g_cfg_lock;
static kvq_config g_cfg = { 0, 20, 10, 1, '!', '!' };      // workhorse.c:71-75

extern "C" void kvq_config_set(const kvq_config *cfg) { std::lock_guard<std::mutex> l(g_cfg_lock); g_cfg = *cfg; }
extern "C" void kvq_config_get(kvq_config *cfg) { std::lock_guard<std::mutex> l(g_cfg_lock); *cfg = g_cfg; }
extern "C" const char *kvq_version(void) { return "kvarq_hip 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------
// device plumbing
// ---------------------------------------------------------------------------

// Blocks a destroyed scan or table gives back are kept for the next one (per device, up to a bound): a scan's dozen
// hipMalloc / hipFree calls and its pinned buffers cost about 15 ms a scan, more than the kernels of a 3 GB file.
// A block in the cache is idle: whoever releases one has waited for the work that used it (kvq_scan_destroy,
// kvq_table_destroy); a buffer that grows while its scan is running goes back to the driver instead (hipFree waits).
namespace {
struct BlockCache {
    struct E { void *p; size_t cap; int dev; };
    std::mutex m; std::vector<E> v; size_t bytes = 0; const size_t max_bytes; const bool host;
    BlockCache(size_t mb, bool h) : max_bytes(mb), host(h) {}
    void *take(size_t n, size_t *cap)
    {
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> l(m);
        size_t best = v.size();
        for (size_t i = 0; i < v.size(); i++)
            if (v[i].dev == dev && v[i].cap >= n && v[i].cap <= 2 * n + (1u << 20) && (best == v.size() || v[i].cap < v[best].cap)) best = i;
        if (best == v.size()) return nullptr;
        void *p = v[best].p; *cap = v[best].cap; bytes -= v[best].cap;
        v.erase(v.begin() + (long)best);
        return p;
    }
    bool put(void *p, size_t cap)
    {
        // the device that OWNS the block, not the one that is current in the calling thread: a scan may be destroyed from
        // another thread than the one that made it (Python's garbage collector; ranks as threads closed by their parent),
        // and a block filed under the wrong device would later be handed to a scan there
        int dev = 0; (void)hipGetDevice(&dev);
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device; else (void)hipGetLastError();
        std::lock_guard<std::mutex> l(m);
        if (cap > max_bytes / 2 || v.size() >= 256) return false;
        while (bytes + cap > max_bytes && !v.empty()) {          // the oldest blocks make room
            if (host) (void)hipHostFree(v[0].p); else (void)hipFree(v[0].p);
            bytes -= v[0].cap; v.erase(v.begin());
        }
        v.push_back({ p, cap, dev }); bytes += cap;
        return true;
    }
    size_t held() { std::lock_guard<std::mutex> l(m); return bytes; }
    void drop()
    {
        std::lock_guard<std::mutex> l(m);
        for (auto &e : v) { if (host) (void)hipHostFree(e.p); else (void)hipFree(e.p); }
        v.clear(); bytes = 0;
    }
};
static bool cache_on() { static const bool on = !(getenv("KVQ_BLOCK_CACHE") && getenv("KVQ_BLOCK_CACHE")[0] == '0'); return on; }
static BlockCache g_dev_blocks((size_t)2 << 30, false), g_pin_blocks((size_t)512 << 20, true);
}

// give the cached blocks back to the driver (a host that wants the memory; nothing needs calling this)
void kvq_drop_kept_scan();      // kvq_findseqs.hip
extern "C" void kvq_release_cached(void) { kvq_drop_kept_scan(); g_dev_blocks.drop(); g_pin_blocks.drop(); }

int DevBuf::ensure(size_t n)
{
    if (n <= cap && p) return KVQ_OK;
    size_t want = n + n / 4 + 256;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (cache_on()) { size_t c = 0; if (void *q = g_dev_blocks.take(n + 256, &c)) { p = q; cap = c; return KVQ_OK; } }
    hipError_t e = hipMalloc(&p, want);
    if (e == hipErrorOutOfMemory && g_dev_blocks.held()) { g_dev_blocks.drop(); e = hipMalloc(&p, want); }
    if (e != hipSuccess) {
        p = nullptr;
        kvq_set_error(e == hipErrorOutOfMemory ? KVQ_ERR_MEMORY : KVQ_ERR_DEVICE, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return tl_err_code;
    }
    cap = want;
    return KVQ_OK;
}
void DevBuf::release() { if (p && !(cache_on() && g_dev_blocks.put(p, cap))) (void)hipFree(p); p = nullptr; cap = 0; }

// pinned host memory through the same kind of cache (*cap: what the block holds, at least n)
static void *pinned_take(size_t n, size_t *cap)
{
    if (cache_on()) if (void *q = g_pin_blocks.take(n, cap)) return q;
    void *p = nullptr;
    if (hipHostMalloc(&p, n, hipHostMallocDefault) != hipSuccess) return nullptr;
    *cap = n;
    return p;
}
static void pinned_give(void *p, size_t cap) { if (p && !(cache_on() && g_pin_blocks.put(p, cap))) (void)hipHostFree(p); }

int TablePool::reserve(size_t bytes, hipStream_t stream)
{
    bytes = (bytes + 4095) & ~(size_t)4095;
    if (used + bytes <= cap) return KVQ_OK;
    // grow: earlier batches may still read the old blocks, so wait for them first; what they have left there
    // (tile reports, lists of skipped tiles: found again by their offsets) moves over
    if (stream) KVQ_HIP(hipStreamSynchronize(stream));
    const size_t want = std::max<size_t>(2 * cap, std::max<size_t>(used + bytes, (size_t)8 << 20));
    size_t hcap = 0; uint8_t *nh = (uint8_t *)pinned_take(want, &hcap);
    DevBuf nd;
    if (!nh || nd.ensure(want) != KVQ_OK) {
        if (nh) pinned_give(nh, hcap);
        kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate %zu bytes of batch tables", want);
        return KVQ_ERR_MEMORY;
    }
    if (used) { memcpy(nh, h, used); KVQ_HIP(hipMemcpy(nd.p, d, used, hipMemcpyDeviceToDevice)); }
    const size_t keep = used;
    release();
    h = nh; h_cap = hcap; d = (uint8_t *)nd.p; d_cap = nd.cap; cap = want; used = keep;
    return KVQ_OK;
}
void TablePool::release()
{
    if (h) pinned_give(h, h_cap);
    if (d) { DevBuf b; b.p = d; b.cap = d_cap; b.release(); }
    h = d = nullptr; cap = used = 0; h_cap = d_cap = 0;
}

extern "C" int32_t kvq_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int32_t kvq_set_device(int32_t o) { KVQ_HIP(hipSetDevice(o)); return KVQ_OK; }
extern "C" void *kvq_device_alloc(int64_t n)
{
    void *p = nullptr;
    if (hipMalloc(&p, (size_t)n + 256) != hipSuccess) { kvq_set_error(KVQ_ERR_MEMORY, "hipMalloc(%lld) failed", (long long)n); return nullptr; }
    return p;
}
extern "C" void kvq_device_free(void *p) { if (p) (void)hipFree(p); }
extern "C" int32_t kvq_memcpy_h2d(void *d, const void *h, int64_t n) { KVQ_HIP(hipMemcpy(d, h, (size_t)n, hipMemcpyHostToDevice)); return KVQ_OK; }
extern "C" int32_t kvq_memcpy_d2h(void *h, const void *d, int64_t n) { KVQ_HIP(hipMemcpy(h, d, (size_t)n, hipMemcpyDeviceToHost)); return KVQ_OK; }
extern "C" int32_t kvq_memset_d(void *d, int32_t v, int64_t n) { KVQ_HIP(hipMemset(d, v, (size_t)n)); return KVQ_OK; }
extern "C" int32_t kvq_device_synchronize(void) { KVQ_HIP(hipDeviceSynchronize()); return KVQ_OK; }

// ---------------------------------------------------------------------------
// table
// ---------------------------------------------------------------------------

extern "C" kvq_table *kvq_table_create(const uint8_t *const *seqs, const int32_t *seqlens, int32_t nseq, const kvq_config *cfg)
{
    kvq_clear_error();
    if (kvq_device_count() <= 0) { kvq_set_error(KVQ_ERR_DEVICE, "no HIP device available: libkvarq_hip has no CPU path"); return nullptr; }
    kvq_table *t = new kvq_table();
    if (cfg) t->cfg = *cfg; else kvq_config_get(&t->cfg);
    t->nseq = nseq;
    t->h_off.resize((size_t)nseq + 1);
    int64_t tot = 0;
    for (int i = 0; i < nseq; i++) {
        if (seqlens[i] < 0) { kvq_set_error(KVQ_ERR_TYPE, "seqlist must be list of strings"); delete t; return nullptr; }
        t->h_off[i] = (int32_t)tot; tot += seqlens[i];
        if (tot > 0x7FFFFFFF) { kvq_set_error(KVQ_ERR_MEMORY, "sequence table too large"); delete t; return nullptr; }
    }
    t->h_off[nseq] = (int32_t)tot;
    t->bases = tot;
    t->h_tab.resize((size_t)tot + 64, 0);
    for (int i = 0; i < nseq; i++) if (seqlens[i]) memcpy(&t->h_tab[t->h_off[i]], seqs[i], (size_t)seqlens[i]);
    // counters layout (include/kvarq_hip.h)
    t->off_nseqhits = KVQ_CTR_RL_ + KVQ_RL_BINS;
    t->off_nseqbasehits = t->off_nseqhits + nseq;
    t->off_cov = t->off_nseqbasehits + nseq;
    t->off_mut = t->off_cov + tot;
    t->ctr_len = t->off_mut + 6 * tot;

    (void)hipGetDevice(&t->device);
    t->is_seeded.assign((size_t)nseq, 0);
    t->index = kvq_seed_index_build(t);      // fills seeded / is_seeded / seed_k
    if (kvq_error_code()) { kvq_table_destroy(t); return nullptr; }
    for (int i = 0; i < nseq; i++) if (!t->is_seeded[i]) t->exhaustive.push_back(i);

    std::vector<int32_t> all((size_t)nseq);
    for (int i = 0; i < nseq; i++) all[i] = i;
    bool ok = t->d_tab.ensure(t->h_tab.size()) == KVQ_OK && t->d_off.ensure((size_t)(nseq + 1) * 4) == KVQ_OK &&
              t->d_exh.ensure((size_t)(nseq + 1) * 4) == KVQ_OK && t->d_all.ensure((size_t)(nseq + 1) * 4) == KVQ_OK &&
              t->d_seeded.ensure((size_t)(nseq + 1) * 4) == KVQ_OK;
    if (ok) {
        ok = hipMemcpy(t->d_tab.p, t->h_tab.data(), t->h_tab.size(), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(t->d_off.p, t->h_off.data(), (size_t)(nseq + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
             (t->exhaustive.empty() || hipMemcpy(t->d_exh.p, t->exhaustive.data(), t->exhaustive.size() * 4, hipMemcpyHostToDevice) == hipSuccess) &&
             (t->seeded.empty() || hipMemcpy(t->d_seeded.p, t->seeded.data(), t->seeded.size() * 4, hipMemcpyHostToDevice) == hipSuccess) &&
             (nseq == 0 || hipMemcpy(t->d_all.p, all.data(), (size_t)nseq * 4, hipMemcpyHostToDevice) == hipSuccess);
        if (!ok) kvq_set_error(KVQ_ERR_DEVICE, "uploading the sequence table failed");
    }
    if (!ok) { kvq_table_destroy(t); return nullptr; }
    return t;
}

extern "C" void kvq_table_destroy(kvq_table *t)
{
    if (!t) return;
    {
        // (its blocks go to the cache, not through hipFree: nothing may still be reading them -- on the device the table lives on,
        // which need not be the calling thread's current one)
        int cur = 0; (void)hipGetDevice(&cur);
        if (t->device != cur) (void)hipSetDevice(t->device);
        (void)hipDeviceSynchronize();
        if (t->device != cur) (void)hipSetDevice(cur);
    }
    if (t->index) kvq_seed_index_destroy(t->index);
    t->d_tab.release(); t->d_off.release(); t->d_exh.release(); t->d_all.release(); t->d_seeded.release();
    delete t;
}
extern "C" int32_t kvq_table_nseq(const kvq_table *t) { return t->nseq; }
extern "C" int64_t kvq_table_bases(const kvq_table *t) { return t->bases; }
extern "C" int32_t kvq_table_seq_is_seeded(const kvq_table *t, int32_t s) { return (s >= 0 && s < t->nseq) ? t->is_seeded[s] : 0; }
extern "C" int32_t kvq_table_seed_k(const kvq_table *t) { return t->seed_k; }
extern "C" int64_t kvq_counters_len(const kvq_table *t) { return t->ctr_len; }
extern "C" int64_t kvq_counters_off_nseqhits(const kvq_table *t) { return t->off_nseqhits; }
extern "C" int64_t kvq_counters_off_nseqbasehits(const kvq_table *t) { return t->off_nseqbasehits; }
extern "C" int64_t kvq_counters_off_coverage(const kvq_table *t) { return t->off_cov; }
extern "C" int64_t kvq_counters_off_mutations(const kvq_table *t) { return t->off_mut; }
extern "C" int64_t kvq_table_seq_offset(const kvq_table *t, int32_t s) { return (s >= 0 && s <= t->nseq) ? t->h_off[s] : -1; }

// ---------------------------------------------------------------------------
// scan object
// ---------------------------------------------------------------------------

#define KVQ_MAX_BATCHES 65536
// d_small layout (bytes): [0] arena_n u32, [8] blob_n u64, [16] err u64, [24] err of the batch in flight u64,
// [64 ..) range words u32 x (KVQ_MAX_BATCHES+1), then per-batch "speculation failed" flags u32 x KVQ_MAX_BATCHES,
// then the staging counters of the batch in flight (records, longest, read-length histogram)
static const size_t SMALL_RANGE = 64, SMALL_FAIL = SMALL_RANGE + 4 * (KVQ_MAX_BATCHES + 1),
                    SMALL_STAGE = (SMALL_FAIL + 4 * KVQ_MAX_BATCHES + 255) & ~(size_t)255,
                    SMALL_BYTES = SMALL_STAGE + 8 * KVQ_STAGE_SLOTS * KVQ_STAGE_COPIES;

// after the kernels of one batch: merge the seed-filter kernel's staged counters and error
// into the scan's, or -- when its speculated record split failed validation -- forget
// everything the batch appended to the hit arena (the host rescans it exhaustively)
extern "C" __global__ void __launch_bounds__(256)
kvq_commit_batch(unsigned long long *stage, unsigned long long *ctr, unsigned long long *err_stage, unsigned long long *err,
                 const unsigned int *fail, unsigned int *arena_n, unsigned int *range)
{
    KVQ_BESIDE_SCAN();
    const bool bad = (*fail & 1u) != 0u;          // (the bits above count skipped tiles: kvq_validate_tiles)
    // (the scan kernel's workgroups add to one of KVQ_STAGE_COPIES copies of the staged counters -- a thousand atomics on
    // one word take 12 us at the end of every launch, an eighth of them a fraction of that: the copies are put together here)
    for (int i = threadIdx.x; i < KVQ_STAGE_SLOTS; i += blockDim.x) {
        unsigned long long v = 0;
        for (int c = 0; c < KVQ_STAGE_COPIES; c++) {
            const unsigned long long w = stage[(size_t)c * KVQ_STAGE_SLOTS + i];
            v = i == KVQ_CTR_LONGEST_ ? (v > w ? v : w) : v + w;
            stage[(size_t)c * KVQ_STAGE_SLOTS + i] = 0;
        }
        if (v && !bad) { if (i == KVQ_CTR_LONGEST_) atomicMax(&ctr[i], v); else atomicAdd(&ctr[i], v); }
    }
    if (threadIdx.x == 0) {
        if (bad) *arena_n = range[0];
        else if (*err_stage != ~0ull) atomicMin(err, *err_stage);
        *err_stage = ~0ull;
        range[1] = *arena_n;                       // the batch's hits end here
    }
}

// the scan's device state back to "nothing scanned" in one launch: the small words (err and the
// staged err to all ones), counters, coverage marks (all 8-byte words)
extern "C" __global__ void __launch_bounds__(256)
kvq_reset_state(unsigned long long *small, size_t small_words, unsigned long long *ctr, size_t ctr_words,
                unsigned long long *cov, size_t cov_words)
{
    KVQ_BESIDE_SCAN();
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = i0; i < small_words; i += step) small[i] = (i == 2 || i == 3) ? ~0ull : 0ull;      // bytes 16..31: err, staged err
    for (size_t i = i0; i < ctr_words; i += step) ctr[i] = 0ull;
    for (size_t i = i0; i < cov_words; i += step) cov[i] = 0ull;
}

// the scan's eight small words, its per-batch "speculation failed" flags and what kvq_finish_plan and the
// ordering left (hit count, layout, "crowded") -> pinned host memory
extern "C" __global__ void __launch_bounds__(256)
kvq_publish_small(const unsigned int *__restrict__ small, const unsigned int *__restrict__ fail, unsigned int nbatches,
                  const unsigned int *__restrict__ state, unsigned int state_words,
                  unsigned int *host_small, unsigned int *host_fail, unsigned int *host_state)
{
    KVQ_BESIDE_SCAN();
    if (threadIdx.x < 8) host_small[threadIdx.x] = small[threadIdx.x];
    for (unsigned int i = threadIdx.x; i < nbatches; i += blockDim.x) host_fail[i] = fail[i];
    for (unsigned int i = threadIdx.x; i < state_words; i += blockDim.x) host_state[i] = state[i];
    __threadfence_system();
}

static int reset_device_state(kvq_scan *s)
{
    static_assert(SMALL_BYTES % 8 == 0, "kvq_reset_state writes 8-byte words");
    hipLaunchKernelGGL(kvq_reset_state, dim3(256), dim3(256), 0, s->stream, (unsigned long long *)s->d_small.p, SMALL_BYTES / 8,
                       s->d_ctr, (size_t)s->t->ctr_len, s->d_covdiff.as<unsigned long long>(), (size_t)s->t->bases + (size_t)s->t->nseq + 1);
    KVQ_HIP(hipGetLastError());
    return KVQ_OK;
}

static int ensure_arena(kvq_scan *s, uint64_t hits, uint64_t blob)
{
    if (hits > 0xFFFFFFF0ull) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results"); return KVQ_ERR_MEMORY; }
    if (hits > s->arena_cap) {
        int rc = s->d_arena.ensure((size_t)hits * sizeof(KvqHit)); if (rc) return rc;
        s->arena_cap = (uint32_t)hits;
    }
    if (blob > s->blob_cap) {
        if (blob > 0xFFFFFFF0ull) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results"); return KVQ_ERR_MEMORY; }
        int rc = s->d_blob.ensure((size_t)blob); if (rc) return rc;
        s->blob_cap = blob;
    }
    return KVQ_OK;
}

static std::atomic<int> g_live_scans{0};
int kvq_live_scans() { return g_live_scans.load(); }
uint32_t kvq_device_cu_count()
{
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess || pr.multiProcessorCount <= 0) return 256u;
    return (uint32_t)pr.multiProcessorCount;
}

extern "C" kvq_scan *kvq_scan_create(const kvq_table *t, void *d_counters)
{
    kvq_clear_error();
    kvq_scan *s = new kvq_scan();
    g_live_scans++;
    s->t = t;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        kvq_set_error(KVQ_ERR_DEVICE, "hipStreamCreate failed"); delete s; return nullptr;
    }
    if (d_counters) { s->d_ctr = (unsigned long long *)d_counters; s->own_ctr = false; }
    else {
        if (s->d_ctr_own.ensure((size_t)t->ctr_len * 8) != KVQ_OK) { kvq_scan_destroy(s); return nullptr; }
        s->d_ctr = s->d_ctr_own.as<unsigned long long>(); s->own_ctr = true;
    }
    if (s->d_small.ensure(SMALL_BYTES) != KVQ_OK) { kvq_scan_destroy(s); return nullptr; }
    if (s->d_covdiff.ensure(((size_t)t->bases + (size_t)t->nseq + 1) * 8) != KVQ_OK) { kvq_scan_destroy(s); return nullptr; }
    s->d_arena_n = (unsigned int *)s->d_small.p;
    s->d_blob_n = (unsigned long long *)((char *)s->d_small.p + 8);
    s->d_err = (unsigned long long *)((char *)s->d_small.p + 16);
    s->d_err_stage = (unsigned long long *)((char *)s->d_small.p + 24);
    s->d_range = (unsigned int *)((char *)s->d_small.p + SMALL_RANGE);
    s->d_fail = (unsigned int *)((char *)s->d_small.p + SMALL_FAIL);
    s->d_stage_ctr = (unsigned long long *)((char *)s->d_small.p + SMALL_STAGE);
    if (ensure_arena(s, 1u << 20, 64ull << 20) != KVQ_OK) { kvq_scan_destroy(s); return nullptr; }
    if (s->d_surv.ensure(KvqSurvivors::bytes()) != KVQ_OK) { kvq_scan_destroy(s); return nullptr; }
    {
        // header: slots handed out, the list's size (KVQ_SURV_CAP=<slots> shrinks it for the tests: a full list costs speed, never results)
        unsigned int hdr[64] = { 0 };
        hdr[1] = KVQ_SURV_CAP;
        if (const char *e = getenv("KVQ_SURV_CAP")) { const long v = atol(e); if (v >= 0 && v < (long)KVQ_SURV_CAP) hdr[1] = (unsigned int)v; }
        s->surv_cap = hdr[1];
        if (hipMemcpyAsync(s->d_surv.p, hdr, 256, hipMemcpyHostToDevice, s->stream) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) { kvq_scan_destroy(s); return nullptr; }
    }
    if (s->d_redo.ensure(KvqRedo::bytes()) != KVQ_OK) { kvq_scan_destroy(s); return nullptr; }
    if (hipMemsetAsync(s->d_redo.p, 0, 256, s->stream) != hipSuccess) { kvq_scan_destroy(s); return nullptr; }      // (the block comes from the cache as it was left: the redo's two counts start at zero)
    s->pin_cap = (size_t)t->ctr_len * 8 + (4u << 20);
    {
        const size_t small_b = 64 + 4 * (size_t)KVQ_MAX_BATCHES + 512;
        size_t c = 0;
        s->pin_small = (uint8_t *)pinned_take(small_b, &c); s->pin_small_cap = c;
        s->pin = (uint8_t *)pinned_take(s->pin_cap, &c);
        if (s->pin) s->pin_cap = c;
        if (!s->pin_small || !s->pin) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results"); kvq_scan_destroy(s); return nullptr; }
        memset(s->pin_small, 0, small_b);
    }
    memset(s->pin, 0, s->pin_cap < (1u << 20) ? s->pin_cap : (1u << 20));
    s->res = kvq_result_layout(0, 0);
    s->pin_res = s->pin + ((((size_t)t->ctr_len * 8) + 255) & ~(size_t)255);
    if (reset_device_state(s) != KVQ_OK) { kvq_scan_destroy(s); return nullptr; }
    s->h_ctr.assign((size_t)t->ctr_len, 0);
    return s;
}

// The scan kernel is persistent and takes every wave slot and all of the LDS of every CU: two of them at once
// (scan objects on different streams, e.g. a caller that enqueues the next job while it collects the last one)
// only get in each other's way.  So the main kernels of a process form a chain: each waits for the one enqueued
// before it, whatever stream that was on.  Everything else of a scan (tables, validation, fold, ordering, copies)
// is left free to run beside the next scan's kernel.
static std::mutex g_chain_lock;
static hipEvent_t g_chain_done = nullptr;         // recorded behind the main kernel enqueued last
static const kvq_scan *g_chain_owner = nullptr;   // (the event is its: forgotten when that scan goes away)

// is a scan kernel of ANOTHER scan object of this process still on the device (enqueued or running)?
bool kvq_chain_busy(const kvq_scan *s)
{
    std::lock_guard<std::mutex> l(g_chain_lock);
    if (!g_chain_done || g_chain_owner == s) return false;
    const bool busy = hipEventQuery(g_chain_done) == hipErrorNotReady;
    (void)hipGetLastError();
    return busy;
}
int kvq_chain_wait(kvq_scan *s, bool *behind_a_running_scan)
{
    std::lock_guard<std::mutex> l(g_chain_lock);
    if (behind_a_running_scan) *behind_a_running_scan = false;
    if (g_chain_done && g_chain_owner != s) {
        // (is the scan in front still on the device?  Then the caller keeps several jobs in flight, and the one after this will
        // be enqueued behind this one in the same way: kvq_seeded_launch lets it start without waiting for this scan's survivors)
        if (behind_a_running_scan) { *behind_a_running_scan = hipEventQuery(g_chain_done) == hipErrorNotReady; (void)hipGetLastError(); }
        KVQ_HIP(hipStreamWaitEvent(s->stream, g_chain_done, 0));
    }
    return KVQ_OK;
}
int kvq_chain_publish(kvq_scan *s)
{
    if (!s->ev_chain) KVQ_HIP(hipEventCreateWithFlags(&s->ev_chain, hipEventDisableTiming));
    KVQ_HIP(hipEventRecord(s->ev_chain, s->stream));
    std::lock_guard<std::mutex> l(g_chain_lock);
    g_chain_done = s->ev_chain; g_chain_owner = s;
    return KVQ_OK;
}
static void chain_forget(const kvq_scan *s)
{
    std::lock_guard<std::mutex> l(g_chain_lock);
    if (g_chain_owner == s) { g_chain_done = nullptr; g_chain_owner = nullptr; }
}

// timing events are kept for the next scan of the same handle (creating a pair costs several microseconds)
static void drop_events(kvq_scan *s, bool destroy = false)
{
    for (auto *v : { &s->ev_all, &s->ev_main }) { s->ev_free.insert(s->ev_free.end(), v->begin(), v->end()); v->clear(); }
    if (destroy) {
        chain_forget(s);
        for (auto &e : s->ev_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        s->ev_free.clear();
    }
}

extern "C" void kvq_scan_destroy(kvq_scan *s)
{
    if (!s) return;
    g_live_scans--;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    drop_events(s, true);
    s->d_ctr_own.release();
    if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    if (s->pin) pinned_give(s->pin, s->pin_cap);
    if (s->pin_small) pinned_give(s->pin_small, s->pin_small_cap);
    for (int i = 0; i < 2; i++) if (s->ev_copy[i]) (void)hipEventDestroy(s->ev_copy[i]);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
    if (s->ev_chain) (void)hipEventDestroy(s->ev_chain);
    DevBuf *bufs[] = { &s->d_surv, &s->d_redo, &s->d_ctr_all, &s->d_gather_cnt, &s->d_gather_res, &s->d_sort_tmp, &s->d_sorted, &s->d_result, &s->d_order, &s->d_finish, &s->d_covdiff, &s->d_skipped, &s->d_chunk_off, &s->d_seg_base, &s->d_seg_cnt, &s->d_chunk_nrec, &s->d_rec_base, &s->d_nl4,
                       &s->d_rec_start, &s->d_read_off, &s->d_read_len, &s->d_arena, &s->d_blob, &s->d_small, &s->d_stage, &s->d_stage_b };
    for (DevBuf *b : bufs) b->release();
    s->pool.release();
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" int32_t kvq_scan_path(const kvq_scan *s) { return s->path_bits; }
extern "C" void kvq_scan_force_exhaustive(kvq_scan *s, int32_t on) { s->force_exhaustive = on != 0; }

extern "C" int32_t kvq_scan_reset(kvq_scan *s)
{
    kvq_clear_error();
    const double tr0 = now_ms();
    KVQ_HIP(hipStreamSynchronize(s->stream));
    drop_events(s);
    if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    s->batches.clear(); s->host_batches = false; s->host_pending = -1; s->copied_pending = false; s->records = 0; s->parsed = 0; s->total = 0;
    s->ms_all = s->ms_main = 0; s->main_launches = 0; s->finished = false; s->reduced = false; s->gathered = false; s->path_bits = 0; s->n_hits = 0;
    s->tail_pending = false;
    s->pool.used = 0;
    const int rr = reset_device_state(s);
    if (g_timing) fprintf(stderr, "reset host %.3f ms\n", now_ms() - tr0);
    return rr;
}

static KvqParams make_params(const kvq_scan *s)
{
    const kvq_table *t = s->t;
    KvqParams P;
    P.maxerrors = t->cfg.maxerrors; P.minoverlap = t->cfg.minoverlap; P.minreadlength = t->cfg.minreadlength;
    P.amin = (int32_t)t->cfg.Amin; P.nseq = t->nseq;
    P.tab = t->d_tab.as<uint8_t>(); P.tab_off = t->d_off.as<int32_t>();
    P.ctr = s->d_ctr; P.covdiff = s->d_covdiff.as<unsigned long long>();
    P.off_nseqhits = t->off_nseqhits; P.off_nseqbasehits = t->off_nseqbasehits; P.off_cov = t->off_cov; P.off_mut = t->off_mut;
    P.arena = s->d_arena.as<KvqHit>(); P.arena_cap = s->arena_cap; P.arena_n = s->d_arena_n;
    P.blob = s->d_blob.as<uint8_t>(); P.blob_cap = s->blob_cap; P.blob_n = s->d_blob_n;
    P.err = s->d_err;
    return P;
}

static int new_event_pair(kvq_scan *s, std::vector<std::pair<hipEvent_t, hipEvent_t>> &v)
{
    if (!s->ev_free.empty()) { v.push_back(s->ev_free.back()); s->ev_free.pop_back(); return KVQ_OK; }
    hipEvent_t a, b;
    KVQ_HIP(hipEventCreate(&a)); KVQ_HIP(hipEventCreate(&b));
    v.emplace_back(a, b);
    return KVQ_OK;
}

// enqueue every kernel of one batch.  The exhaustive path needs one host
// round trip (records per chunk) to size its record arrays.
static int run_batch(kvq_scan *s, const uint8_t *d_data, int64_t nbytes, const int64_t *chunk_off, int64_t nchunks,
                     int64_t fpos_base, size_t batch_no, bool exhaustive_only)
{
    const kvq_table *t = s->t;
    const double tb0 = now_ms();
    if (nbytes <= 0 || nchunks <= 0) return KVQ_OK;
    if (nbytes > 0xFFF00000ll) { kvq_set_error(KVQ_ERR_RUNTIME, "batch of %lld bytes is too large (< 4 GiB - 1 MiB)", (long long)nbytes); return KVQ_ERR_RUNTIME; }
    if (batch_no >= KVQ_MAX_BATCHES) { kvq_set_error(KVQ_ERR_RUNTIME, "too many batches in one scan"); return KVQ_ERR_RUNTIME; }
    if (((uintptr_t)d_data & 15u) != 0) { kvq_set_error(KVQ_ERR_RUNTIME, "device buffer must be 16-byte aligned"); return KVQ_ERR_RUNTIME; }

    KvqParams P = make_params(s);
    const bool use_seeded = t->index && !t->seeded.empty() && !s->force_exhaustive && !exhaustive_only;
    s->cur_chunk_off.assign(chunk_off, chunk_off + nchunks + 1);
    const std::vector<int32_t> *exh = &t->exhaustive;
    const int32_t *d_exh = t->d_exh.as<int32_t>();
    std::vector<int32_t> all;
    if (!use_seeded) { d_exh = t->d_all.as<int32_t>(); }
    const int32_t n_exh = use_seeded ? (int32_t)exh->size() : t->nseq;

    // chunk table: written into the pinned half of the pool, copied to its device half (async)
    int rc;
    // (room for everything this batch puts into the pool -- chunk offsets here; first tiles, parameter
    // block, tile table and tile reports in kvq_seeded_launch, whose tiles own at least kvq_min_tile() bytes --
    // is made in one go: the pool must not move between the two)
    const size_t tiles_bound = (size_t)(nbytes / kvq_min_tile()) + (size_t)nchunks + 2;
    if ((rc = s->pool.reserve(((size_t)nchunks + 1) * 8 + tiles_bound * 24 + 65536, s->stream))) return rc;
    const size_t co_at = s->pool.take(((size_t)nchunks + 1) * 4);
    s->cur_co_at = co_at;
    uint32_t *co = reinterpret_cast<uint32_t *>(s->pool.h + co_at);
    const uint32_t *d_co = reinterpret_cast<const uint32_t *>(s->pool.d + co_at);
    std::vector<uint32_t> sb((size_t)nchunks + 1);
    uint32_t maxseg = 0, maxchunk = 0; uint64_t segs = 0;
    for (int64_t c = 0; c <= nchunks; c++) {
        if (chunk_off[c] < 0 || chunk_off[c] > nbytes || (c && chunk_off[c] < chunk_off[c - 1])) {
            kvq_set_error(KVQ_ERR_RUNTIME, "bad chunk offsets"); return KVQ_ERR_RUNTIME;
        }
        co[c] = (uint32_t)chunk_off[c];
    }
    for (int64_t c = 0; c < nchunks; c++) {
        const uint32_t a = co[c], b = co[c + 1];
        const uint32_t n = b > a ? (uint32_t)(((uint64_t)b - (a & ~15u) + KVQ_SEG_BYTES - 1) / KVQ_SEG_BYTES) : 0u;
        sb[c] = (uint32_t)segs; segs += n;
        maxseg = std::max(maxseg, n); maxchunk = std::max(maxchunk, b - a);
    }
    sb[nchunks] = (uint32_t)segs;
    // (the seed-filter launch copies the chunk offsets together with its own tables: one transfer)
    if (!use_seeded) KVQ_HIP(hipMemcpyAsync(s->pool.d + co_at, co, ((size_t)nchunks + 1) * 4, hipMemcpyHostToDevice, s->stream));

    if ((rc = new_event_pair(s, s->ev_all))) return rc;
    KVQ_HIP(hipEventRecord(s->ev_all.back().first, s->stream));

    bool hist_done = false;
    if (use_seeded) {
        if ((rc = new_event_pair(s, s->ev_main))) return rc;
        KvqParams PS = P;                          // counters and error of this batch are staged until it is validated (the pair of events is recorded right around the scan kernel: kvq_seeded_launch)
        PS.ctr = s->d_stage_ctr; PS.err = s->d_err_stage;
        s->cur_fail = s->d_fail + batch_no;
        if ((rc = kvq_seeded_launch(s, PS, d_data, nbytes, d_co, nchunks, fpos_base, maxchunk))) return rc;
        s->batches[batch_no].skip_at = s->cur_skip_at; s->batches[batch_no].tile_bytes = s->tile_bytes;
        s->main_launches++; s->path_bits |= 1;
        hist_done = true;
    }

    if (!hist_done || n_exh > 0) {
        if (n_exh > 0) s->path_bits |= 2;
        if ((rc = s->d_seg_base.ensure(sb.size() * 4))) return rc;
        if ((rc = s->d_seg_cnt.ensure((size_t)(segs + 1) * 4))) return rc;
        if ((rc = s->d_chunk_nrec.ensure((size_t)(nchunks + 1) * 4))) return rc;
        if ((rc = s->d_rec_base.ensure((size_t)(nchunks + 1) * 4))) return rc;
        KVQ_HIP(hipMemcpyAsync(s->d_seg_base.p, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, s->stream));
        const uint32_t gx = std::max(1u, std::min(65u, (maxseg + 3) / 4));
        for (int64_t c0 = 0; c0 < nchunks; c0 += 32768) {
            const uint32_t ny = (uint32_t)std::min<int64_t>(32768, nchunks - c0);
            hipLaunchKernelGGL(kvq_count_lines, dim3(gx, ny), dim3(256), 0, s->stream, d_data,
                               d_co + c0, s->d_seg_base.as<uint32_t>() + c0, s->d_seg_cnt.as<uint32_t>());
        }
        hipLaunchKernelGGL(kvq_scan_segments, dim3((uint32_t)((nchunks + 3) / 4)), dim3(256), 0, s->stream, (uint32_t)nchunks,
                           s->d_seg_base.as<uint32_t>(), s->d_seg_cnt.as<uint32_t>(), s->d_chunk_nrec.as<uint32_t>());
        std::vector<uint32_t> nrec((size_t)nchunks), rbase((size_t)nchunks + 1);
        KVQ_HIP(hipMemcpyAsync(nrec.data(), s->d_chunk_nrec.p, (size_t)nchunks * 4, hipMemcpyDeviceToHost, s->stream));
        KVQ_HIP(hipStreamSynchronize(s->stream));
        uint64_t R = 0;
        for (int64_t c = 0; c < nchunks; c++) { rbase[c] = (uint32_t)R; R += nrec[c]; }
        rbase[nchunks] = (uint32_t)R;
        if (R > 0) {
            if ((rc = s->d_nl4.ensure((size_t)R * 16))) return rc;
            if ((rc = s->d_rec_start.ensure((size_t)R * 4))) return rc;
            if ((rc = s->d_read_off.ensure((size_t)R * 4))) return rc;
            if ((rc = s->d_read_len.ensure((size_t)R * 4))) return rc;
            KVQ_HIP(hipMemcpyAsync(s->d_rec_base.p, rbase.data(), rbase.size() * 4, hipMemcpyHostToDevice, s->stream));
            KVQ_HIP(hipStreamSynchronize(s->stream));
            for (int64_t c0 = 0; c0 < nchunks; c0 += 32768) {
                const uint32_t ny = (uint32_t)std::min<int64_t>(32768, nchunks - c0);
                hipLaunchKernelGGL(kvq_index_records, dim3(gx, ny), dim3(256), 0, s->stream, d_data,
                                   d_co + c0, s->d_seg_base.as<uint32_t>() + c0, s->d_seg_cnt.as<uint32_t>(),
                                   s->d_chunk_nrec.as<uint32_t>() + c0, s->d_rec_base.as<uint32_t>() + c0,
                                   s->d_nl4.as<uint32_t>(), s->d_rec_start.as<uint32_t>());
            }
            const uint32_t per_block = 4 * 16;       // KVQ_TRIM_RPW records per wave
            hipLaunchKernelGGL(kvq_trim_records, dim3((uint32_t)((R + per_block - 1) / per_block)), dim3(256), 0, s->stream, P, d_data,
                               fpos_base, (uint32_t)R, KvqDevCount{ nullptr, 0, 0, nullptr }, s->d_nl4.as<uint32_t>(), s->d_rec_start.as<uint32_t>(),
                               s->d_read_off.as<uint32_t>(), s->d_read_len.as<int32_t>(), hist_done ? 0 : 1, 16u, (unsigned int *)nullptr, 0u);
            if (n_exh > 0) {
                const bool main_here = !use_seeded;
                if (main_here) { if ((rc = new_event_pair(s, s->ev_main))) return rc; KVQ_HIP(hipEventRecord(s->ev_main.back().first, s->stream)); }
                hipLaunchKernelGGL(kvq_match_all, dim3((uint32_t)((R + 3) / 4)), dim3(256), 0, s->stream, P, d_data, fpos_base, (uint32_t)R, KvqDevCount{ nullptr, 0, 0, nullptr },
                                   s->d_read_off.as<uint32_t>(), s->d_read_len.as<int32_t>(), d_exh, n_exh);
                if (main_here) { KVQ_HIP(hipEventRecord(s->ev_main.back().second, s->stream)); s->main_launches++; }
            }
        }
    }
    // The records of tiles that the fused scan skipped (a record longer than the tile's look-ahead, more newlines than a
    // tile's tables hold) go through the exhaustive kernels for the seeded sequences -- found again from the exact newline
    // counts (kvq_collect_skipped walks them from what kvq_validate_tiles wrote for each such tile), trimmed, matched --
    // right here, behind every seed-filter launch, WITHOUT the host looking: the kernels are launched with fixed grids, read
    // the number of tiles and of records from device memory and return at once when there are none (the usual case: three
    // empty launches).  A batch that failed validation is left alone (it is redone as a whole), and so is one whose
    // skipped tiles hold more records than KVQ_REDO_CAP (kvq_dev_count raises its fail bit).  Their hits lie in the
    // batch's own range of the arena, closed below.
    if (use_seeded && s->cur_ntiles) {         // (a batch of empty chunks has scanned no tile: nothing was skipped, and the redo's counts are the LAST launch's)
        const KvqRedo Rd(s->d_redo.p);
        unsigned int *const failw = s->d_fail + batch_no;
        const KvqSkippedTile *tiles = reinterpret_cast<const KvqSkippedTile *>(s->pool.d + s->cur_skip_at);
        const KvqDevCount ntile{ failw, 8, KVQ_SKIP_CAP, failw }, nrec{ Rd.count, 0, KVQ_REDO_CAP - KVQ_LONG_CAP, failw }, nlong{ Rd.count + 1, 0, 0xFFFFFFFFu, failw };
        hipLaunchKernelGGL(kvq_collect_skipped, dim3(16), dim3(256), 0, s->stream, d_data, tiles, 0u, ntile, Rd.nl4, Rd.rec_start, Rd.count, KVQ_REDO_CAP - KVQ_LONG_CAP);
        // (few records, some of them very long: a wave per record for the trim; the matcher shares a record's sequences and
        // alignments out over many waves)
        hipLaunchKernelGGL(kvq_trim_records, dim3(32), dim3(256), 0, s->stream, P, d_data, fpos_base, 0u, nrec, Rd.nl4, Rd.rec_start, Rd.read_off, Rd.read_len, 1, 1u, Rd.count + 1, KVQ_REDO_CAP - 1u);
        // The matcher, twice: the ordinary reads a wave each (the sequences of a read shared out over a few workgroups), the
        // long ones -- a handful of reads of thousands of bases, which the trim has put on a list of their own -- spread out
        // over sequences and alignments.  Grids of a fixed, modest size (an empty launch of sixteen thousand workgroups costs
        // 50 us, one of a few hundred next to nothing; the kernels stride over what there is): small as long as this scan
        // object has never had a skipped tile.
        static const char *mg = getenv("KVQ_MGRID");           // (experiments: workgroups of the long reads' launch)
        const uint32_t lgrid = mg && atoi(mg) > 0 ? (uint32_t)atoi(mg) : 1536u;
        const dim3 ogrid = s->seen_skips ? dim3(128, (uint32_t)std::min<size_t>(s->t->seeded.size(), 4), 1) : dim3(16, (uint32_t)std::min<size_t>(s->t->seeded.size(), 4), 1);
        if (!s->t->seeded.empty()) {
            hipLaunchKernelGGL(kvq_match_all, ogrid, dim3(256), 0, s->stream, P, d_data, fpos_base, 0u, nrec,
                               Rd.read_off, Rd.read_len, s->t->d_seeded.as<int32_t>(), (int32_t)s->t->seeded.size());
            hipLaunchKernelGGL(kvq_match_long, dim3(s->seen_skips ? lgrid : 256u), dim3(256), 0, s->stream, P, d_data, fpos_base, nlong,
                               Rd.read_off, Rd.read_len, s->t->d_seeded.as<int32_t>(), (int32_t)s->t->seeded.size(), KVQ_REDO_CAP - 1u);
        }
    }
    // hits of this batch = arena[range[batch_no], range[batch_no + 1]) (kvq_commit_batch closes the range)
    if (use_seeded)
        hipLaunchKernelGGL(kvq_commit_batch, dim3(1), dim3(256), 0, s->stream, s->d_stage_ctr, s->d_ctr, s->d_err_stage, s->d_err,
                           (const unsigned int *)(s->d_fail + batch_no), s->d_arena_n, s->d_range + batch_no);
    else
        KVQ_HIP(hipMemcpyAsync(s->d_range + batch_no + 1, s->d_arena_n, 4, hipMemcpyDeviceToDevice, s->stream));
    hipLaunchKernelGGL(kvq_fold_batch, dim3(512), dim3(256), 0, s->stream, P, d_data, fpos_base,
                       (const unsigned int *)(s->d_range + batch_no), (const unsigned int *)(s->d_range + batch_no + 1));
    KVQ_HIP(hipEventRecord(s->ev_all.back().second, s->stream));
    KVQ_HIP(hipGetLastError());
    if (g_timing) fprintf(stderr, "run_batch host %.3f ms\n", now_ms() - tb0);
    return KVQ_OK;
}

// what run_batch would reject, found out before the batch is put on the scan's list (a listed batch must
// close its range of hits: one that was refused would leave a hole that the batches behind it fall into)
static int check_batch(const void *data, int64_t nbytes, const int64_t *chunk_off, int64_t nchunks, size_t batch_no, bool device)
{
    if (nbytes > 0xFFF00000ll) { kvq_set_error(KVQ_ERR_RUNTIME, "batch of %lld bytes is too large (< 4 GiB - 1 MiB)", (long long)nbytes); return KVQ_ERR_RUNTIME; }
    if (batch_no >= KVQ_MAX_BATCHES) { kvq_set_error(KVQ_ERR_RUNTIME, "too many batches in one scan"); return KVQ_ERR_RUNTIME; }
    if (device && ((uintptr_t)data & 15u) != 0) { kvq_set_error(KVQ_ERR_RUNTIME, "device buffer must be 16-byte aligned"); return KVQ_ERR_RUNTIME; }
    for (int64_t c = 0; c <= nchunks; c++)
        if (chunk_off[c] < 0 || chunk_off[c] > nbytes || (c && chunk_off[c] < chunk_off[c - 1])) {
            kvq_set_error(KVQ_ERR_RUNTIME, "bad chunk offsets"); return KVQ_ERR_RUNTIME;
        }
    return KVQ_OK;
}

extern "C" int32_t kvq_scan_device(kvq_scan *s, const void *d_data, int64_t nbytes, const int64_t *chunk_off, int64_t nchunks, int64_t fpos_base)
{
    kvq_clear_error();
    if (nbytes <= 0 || nchunks <= 0) return KVQ_OK;                  // nothing to scan: not a batch
    int rc = check_batch(d_data, nbytes, chunk_off, nchunks, s->batches.size(), true); if (rc) return rc;
    Batch b; b.d_data = (const uint8_t *)d_data; b.nbytes = nbytes; b.fpos_base = fpos_base;
    b.chunk_off.assign(chunk_off, chunk_off + nchunks + 1);
    s->batches.push_back(b);
    s->parsed += nbytes; s->total += nbytes;
    return run_batch(s, (const uint8_t *)d_data, nbytes, chunk_off, nchunks, fpos_base, s->batches.size() - 1, false);
}

// Host batches.  kvq_scan_host_async(k) sends batch k's text across PCIe at once (copy stream, the staging buffer that is
// free) and enqueues the kernels of batch k - 1, whose text has arrived meanwhile, behind the settled batch k - 2: copies
// follow each other without a gap, kernels run beside them, and the host is back reading the next batch while both go on.
// Settling a batch = waiting for its kernels and looking at its fail word: when its seed-filter pass failed validation it is
// scanned again, exhaustively, while its text is still in its staging buffer.
static DevBuf &stage_of(kvq_scan *s, int slot) { return slot == 0 ? s->d_stage : s->d_stage_b; }

static int settle_in_flight(kvq_scan *s)
{
    KVQ_HIP(hipStreamSynchronize(s->stream));
    if (s->host_pending < 0) return KVQ_OK;
    const size_t b = (size_t)s->host_pending;
    s->host_pending = -1;
    if (!(s->path_bits & 1)) return KVQ_OK;
    const unsigned int fail = *reinterpret_cast<const unsigned int *>(s->pin_small + 40);     // copied behind the batch
    if (!fail) return KVQ_OK;
    if (!(fail & 1u)) { s->path_bits |= 8 | 2; s->seen_skips = true; return KVQ_OK; }      // only some tiles were skipped: their records have been through the exhaustive kernels behind the scan (run_batch)
    s->batches[b].redone = true;
    s->tile_bytes = kvq_choose_tile(1u << 20, 0); s->rec_bytes = 0;  // (a record may have outgrown the look-ahead: back to the full one)
    Batch again = s->batches[b]; again.is_redo = true;
    s->batches.push_back(again);
    s->path_bits |= 4;
    int rc = run_batch(s, stage_of(s, s->run_slot).as<uint8_t>(), again.nbytes, again.chunk_off.data(), (int64_t)again.chunk_off.size() - 1,
                       again.fpos_base, s->batches.size() - 1, true);
    if (rc) return rc;
    KVQ_HIP(hipStreamSynchronize(s->stream));
    return KVQ_OK;
}

// the kernels of the batch whose text has been sent (the batch in flight has been settled: the table pool is free)
static int launch_copied(kvq_scan *s)
{
    if (!s->copied_pending) return KVQ_OK;
    s->copied_pending = false;
    const int slot = s->copied_slot;
    DevBuf &stage = stage_of(s, slot);
    s->pool.used = 0;
    *reinterpret_cast<unsigned int *>(s->pin_small + 40) = 0;    // "speculation failed" of the batch about to be enqueued
    KVQ_HIP(hipStreamWaitEvent(s->stream, s->ev_copy[slot], 0));
    s->batches.push_back(s->copied);
    const Batch &b = s->batches.back();
    s->run_slot = slot;
    int rc = run_batch(s, stage.as<uint8_t>(), b.nbytes, b.chunk_off.data(), (int64_t)b.chunk_off.size() - 1, b.fpos_base, s->batches.size() - 1, false);
    if (rc) return rc;
    if (s->path_bits & 1)
        KVQ_HIP(hipMemcpyAsync(s->pin_small + 40, s->d_fail + (s->batches.size() - 1), 4, hipMemcpyDeviceToHost, s->stream));
    s->host_pending = (int64_t)s->batches.size() - 1;
    return KVQ_OK;
}

// everything handed over so far is scanned and settled
extern "C" int32_t kvq_scan_host_drain(kvq_scan *s)
{
    int rc;
    if ((rc = settle_in_flight(s))) return rc;
    if ((rc = launch_copied(s))) return rc;
    return settle_in_flight(s);
}

// hand over one host batch and return; h_data must stay untouched until kvq_scan_host_copied(s) (or the next
// kvq_scan_host_async / kvq_scan_host_drain / kvq_scan_finish) has returned
extern "C" int32_t kvq_scan_host_async(kvq_scan *s, const void *h_data, int64_t nbytes, const int64_t *chunk_off, int64_t nchunks, int64_t fpos_base)
{
    kvq_clear_error();
    if (nbytes <= 0 || nchunks <= 0) return KVQ_OK;
    int rc;
    if ((rc = check_batch(h_data, nbytes, chunk_off, nchunks, s->batches.size() + (s->copied_pending ? 1u : 0u), false))) return rc;
    if ((rc = settle_in_flight(s))) return rc;                     // the batch whose kernels ran while the caller read this one
    if ((rc = launch_copied(s))) return rc;                        // the batch handed over last call: its text has arrived meanwhile
    const int slot = s->run_slot == 0 ? 1 : 0;                     // (the buffer of the batch settled just now, or one never used)
    DevBuf &stage = stage_of(s, slot);
    if ((rc = stage.ensure((size_t)nbytes + 64))) return rc;
    if (!s->copy_stream) KVQ_HIP(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) if (!s->ev_copy[i]) KVQ_HIP(hipEventCreateWithFlags(&s->ev_copy[i], hipEventDisableTiming));
    if (s->tile_bytes == 0)                                      // size the seed-filter tiles from the head of the text
        s->tile_bytes = kvq_tile_for_text((const uint8_t *)h_data, (size_t)std::min<int64_t>(nbytes, 128 << 10), &s->rec_bytes);
    KVQ_HIP(hipMemcpyAsync(stage.p, h_data, (size_t)nbytes, hipMemcpyHostToDevice, s->copy_stream));
    KVQ_HIP(hipEventRecord(s->ev_copy[slot], s->copy_stream));
    s->copied = Batch(); s->copied.d_data = nullptr; s->copied.nbytes = nbytes; s->copied.fpos_base = fpos_base;
    s->copied.chunk_off.assign(chunk_off, chunk_off + nchunks + 1);
    s->copied_pending = true; s->copied_slot = slot;
    s->host_batches = true;
    s->parsed += nbytes; s->total += nbytes;
    // (a caller that alternates two host buffers writes next into the one of the call before: that text has left it)
    if (s->host_pending >= 0) KVQ_HIP(hipEventSynchronize(s->ev_copy[s->run_slot]));
    return KVQ_OK;
}

// wait until the text of the last kvq_scan_host_async batch has left the host buffer
extern "C" int32_t kvq_scan_host_copied(kvq_scan *s)
{
    if (s->copied_pending && s->ev_copy[s->copied_slot]) KVQ_HIP(hipEventSynchronize(s->ev_copy[s->copied_slot]));
    return KVQ_OK;
}

// the blocking form: h_data may be reused when the call returns
extern "C" int32_t kvq_scan_host(kvq_scan *s, const void *h_data, int64_t nbytes, const int64_t *chunk_off, int64_t nchunks, int64_t fpos_base)
{
    const int rc = kvq_scan_host_async(s, h_data, nbytes, chunk_off, nchunks, fpos_base);
    return rc ? rc : kvq_scan_host_copied(s);
}

// The tail of a scan -- coverage marks -> counters, the plan of the ordering, the ordering itself, the gather into the result arrays,
// the words the host needs, the copies -- enqueued behind the scan's kernels: nothing in it needs a number the host would first have
// to fetch.  finish_once waits for it ONCE; kvq_scan_finish_begin enqueues it ahead of time (a job whose batches are all fed), so that
// a caller with several jobs in flight finds it done when it comes to kvq_scan_finish -- otherwise the host sits out the ordering
// kernels of every small job before it enqueues the next one, and those kernels run beside another job's scan at a tenth of their speed.
static int enqueue_tail(kvq_scan *s)
{
    int rc;
    const kvq_table *t = s->t;
    const size_t ctr_b = ((size_t)t->ctr_len * 8 + 255) & ~(size_t)255;
    unsigned char *small = s->pin_small; unsigned int *fail = reinterpret_cast<unsigned int *>(s->pin_small + 64);
    static const bool no_buckets = getenv("KVQ_ORDER") && !strcmp(getenv("KVQ_ORDER"), "mergesort");
    uint32_t nb_max = 256; while ((uint64_t)nb_max < 4ull * s->arena_cap && nb_max < KVQ_BUCKETS_MAX) nb_max <<= 1;
    const size_t order_b = kvq_order_scratch_zero_bytes(nb_max) + (size_t)s->arena_cap * 4;
    if (order_b > s->d_order.cap || nb_max != s->order_nb_max) {
        if ((rc = s->d_order.ensure(order_b))) return rc;
        KVQ_HIP(hipMemsetAsync(s->d_order.p, 0, kvq_order_scratch_zero_bytes(nb_max), s->stream));    // (kvq_order_clear leaves them zero again)
        s->order_nb_max = nb_max;
    }
    const KvqOrderScratch W = kvq_order_scratch(s->d_order.p, nb_max);
    if ((rc = s->d_finish.ensure(sizeof(KvqFinishState) + 256))) return rc;
    KvqFinishState *d_st = s->d_finish.as<KvqFinishState>();
    const size_t res_cap = kvq_result_layout(s->arena_cap, s->blob_cap).total + 256;
    if ((rc = s->d_result.ensure(res_cap))) return rc;
    // file positions of this scan lie in [lo, hi)
    int64_t lo = 0, hi = 1;
    for (size_t b = 0; b < s->batches.size(); b++) {
        const int64_t a = s->batches[b].fpos_base, e = a + s->batches[b].nbytes;
        if (b == 0 || a < lo) lo = a;
        if (b == 0 || e > hi) hi = e;
    }
    const size_t nb0 = s->batches.size();
    hipLaunchKernelGGL(kvq_finish_plan, dim3(1), dim3(64), 0, s->stream, (const unsigned int *)s->d_arena_n, s->arena_cap,
                       (const unsigned long long *)s->d_blob_n, (unsigned long long)s->blob_cap, (const unsigned long long *)s->d_err,
                       (long long)lo, (long long)hi, nb_max, d_st);
    if (t->nseq > 0)
        hipLaunchKernelGGL(kvq_cov_apply, dim3((uint32_t)((t->nseq + 3) / 4)), dim3(256), 0, s->stream, make_params(s));
    if (!no_buckets &&
        (rc = kvq_order_by_buckets(s->stream, s->d_arena.as<KvqHit>(), s->d_blob.as<uint8_t>(), s->blob_cap, d_st, W, s->d_result.as<uint8_t>()))) return rc;
    hipLaunchKernelGGL(kvq_publish_small, dim3(1), dim3(256), 0, s->stream, (const unsigned int *)s->d_small.p,
                       (const unsigned int *)s->d_fail, (unsigned int)nb0, (const unsigned int *)d_st, (unsigned int)(sizeof(KvqFinishState) / 4),
                       (unsigned int *)small, fail, (unsigned int *)(s->pin_small + 64 + 4 * (size_t)KVQ_MAX_BATCHES));
    KVQ_HIP(hipGetLastError());
    // results: as many bytes as the last scan of this handle had (a guess: what is missing is fetched by finish_once)
    size_t spec = std::min(s->spec_bytes, res_cap);
    if (ctr_b + spec > s->pin_cap) spec = s->pin_cap > ctr_b ? s->pin_cap - ctr_b : 0;
    KVQ_HIP(hipMemcpyAsync(s->pin, s->d_ctr, (size_t)t->ctr_len * 8, hipMemcpyDeviceToHost, s->stream));
    if (spec) KVQ_HIP(hipMemcpyAsync(s->pin + ctr_b, s->d_result.p, spec, hipMemcpyDeviceToHost, s->stream));
    s->tail_nb0 = nb0; s->tail_spec = spec; s->tail_pending = true;
    return KVQ_OK;
}

extern "C" int32_t kvq_scan_finish_begin(kvq_scan *s)
{
    kvq_clear_error();
    if (!s || s->finished) return KVQ_OK;
    // (a host batch in flight is settled first -- that waits for it; a scan of device batches is not waited for at all)
    if (s->host_pending >= 0 || s->copied_pending) { const int rc0 = kvq_scan_host_drain(s); if (rc0) return rc0; }
    return enqueue_tail(s);
}

static int finish_once(kvq_scan *s)
{
    const double t0 = now_ms();
    { const int rc0 = kvq_scan_host_drain(s); if (rc0) return rc0; }          // settle the host batch in flight
    int rc;
    const kvq_table *t = s->t;
    const size_t ctr_b = ((size_t)t->ctr_len * 8 + 255) & ~(size_t)255;
    unsigned int *fail = reinterpret_cast<unsigned int *>(s->pin_small + 64);
    const KvqFinishState *h_st = reinterpret_cast<const KvqFinishState *>(s->pin_small + 64 + 4 * (size_t)KVQ_MAX_BATCHES);
    static const bool no_buckets = getenv("KVQ_ORDER") && !strcmp(getenv("KVQ_ORDER"), "mergesort");

    for (int round = 0; round < 3; round++) {
        // (the tail may have been enqueued ahead of time, by kvq_scan_finish_begin, for exactly the batches there are)
        if (!(s->tail_pending && s->tail_nb0 == s->batches.size()) && (rc = enqueue_tail(s))) return rc;
        s->tail_pending = false;
        const size_t nb0 = s->tail_nb0;
        size_t spec = s->tail_spec;
        KvqFinishState *d_st = s->d_finish.as<KvqFinishState>();
        KVQ_HIP(hipStreamSynchronize(s->stream));
        const double t1 = now_ms();

        // batches whose seed-filter pass failed validation (a tile's speculated record split disagreed with the
        // newline count, one read flooded a wave's queues) were rolled back on the device: scan those again with
        // the exhaustive kernels (device batches only; host batches were redone on the spot) and finish again
        if (nb0 && (s->path_bits & 1)) {
            bool any = false;
            for (size_t b = 0; b < nb0; b++) {
                if (!fail[b] || s->batches[b].redone || s->batches[b].is_redo || !s->batches[b].d_data) continue;
                if (!(fail[b] & 1u)) { s->path_bits |= 8 | 2; s->seen_skips = true; continue; }      // only some tiles were skipped: their records went through the exhaustive kernels behind the scan (run_batch)
                s->batches[b].redone = true;
                s->tile_bytes = kvq_choose_tile(1u << 20, 0); s->rec_bytes = 0;   // (a record may have outgrown the look-ahead: back to the full one)
                Batch again = s->batches[b]; again.is_redo = true;
                s->batches.push_back(again);
                s->path_bits |= 4; any = true;
                rc = run_batch(s, again.d_data, again.nbytes, again.chunk_off.data(), (int64_t)again.chunk_off.size() - 1,
                               again.fpos_base, s->batches.size() - 1, true);
                if (rc) return rc;
            }
            if (any) continue;
        }
        const KvqFinishState st = *h_st;
        const uint32_t n_hits = st.n_raw; const unsigned long long blob_n = st.blob_n, err = st.err;
        if (err != ~0ull) {
            // first malformed record in stream order (workhorse.c:1037-1048)
            const long fpos = (long)(err >> 16); const int kind = (int)((err >> 8) & 0xFF); const int ch = (int)(err & 0xFF);
            if (kind == 0) kvq_set_error(KVQ_ERR_FORMAT, "record must start with '@' (and not '%c') fpos=%ld", ch, fpos);
            else kvq_set_error(KVQ_ERR_FORMAT, "3rd line of record must start with '+' fpos=%ld", fpos);
            return KVQ_ERR_FORMAT;
        }
        if (n_hits > s->arena_cap || blob_n > s->blob_cap) {
            // grow to what this scan needs and ask for a rescan
            const uint64_t want_hits = std::max<uint64_t>(n_hits + n_hits / 8 + 1024, s->arena_cap);
            // blob_n undercounts when the arena overflowed (dropped hits were never folded): scale it
            uint64_t want_blob = blob_n;
            if (n_hits > s->arena_cap && s->arena_cap) want_blob = (uint64_t)((double)blob_n * ((double)n_hits / s->arena_cap) * 1.25) + (1 << 20);
            want_blob = std::max<uint64_t>(want_blob + want_blob / 8, s->blob_cap);
            rc = ensure_arena(s, want_hits, want_blob); if (rc) return rc;
            return KVQ_NEED_RESCAN;
        }
        const KvqResultLayout L = st.L;
        if (ctr_b + L.total > s->pin_cap) {
            // (a larger landing buffer: the counters, already there, move over)
            const size_t want = (ctr_b + L.total) * 5 / 4 + (1 << 20);
            size_t got = 0; uint8_t *np = (uint8_t *)pinned_take(want, &got);
            if (!np) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results"); return KVQ_ERR_MEMORY; }
            memcpy(np, s->pin, ctr_b);
            pinned_give(s->pin, s->pin_cap);
            s->pin = np; s->pin_cap = got; spec = 0;
        }
        bool refetch = L.total > spec;
        if (n_hits && (no_buckets || st.crowded)) {
            // crowded buckets (or the comparison sort asked for): order the hits with the merge sort instead
            if ((rc = kvq_order_by_mergesort(s->stream, s->d_arena.as<KvqHit>(), n_hits, s->d_blob.as<uint8_t>(), s->blob_cap, d_st,
                                             s->d_sort_tmp, s->d_sorted, s->d_result.as<uint8_t>()))) return rc;
            refetch = true;
        }
        if (n_hits && refetch) {
            KVQ_HIP(hipMemcpyAsync(s->pin + ctr_b, s->d_result.p, L.total, hipMemcpyDeviceToHost, s->stream));
            KVQ_HIP(hipStreamSynchronize(s->stream));
        }
        memcpy(s->h_ctr.data(), s->pin, (size_t)t->ctr_len * 8);
        s->pin_res = s->pin + ctr_b;
        if (!n_hits) memset(s->pin_res + L.hitseq_off, 0, 8);
        s->res = L; s->n_hits = n_hits;
        s->spec_bytes = L.total + L.total / 8 + 65536;

        s->ms_all = s->ms_main = 0;
        for (auto &e : s->ev_all) { float ms = 0; if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) s->ms_all += ms; }
        for (auto &e : s->ev_main) { float ms = 0; if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) s->ms_main += ms; }
        (void)hipGetLastError();                 // (a pair that was never recorded is not an error of the scan)
        s->finished = true;
        if (g_timing) {
            unsigned int rc2[2] = { 0, 0 };
            if (s->d_redo.p) (void)hipMemcpy(rc2, s->d_redo.p, 8, hipMemcpyDeviceToHost);
            fprintf(stderr, "finish: enqueue + wait %.3f  rest %.3f ms (%u hits; the last launch's skipped tiles left %u records, %u of them long)\n", t1 - t0, now_ms() - t1, n_hits, rc2[0], rc2[1]);
        }
        return KVQ_OK;
    }
    kvq_set_error(KVQ_ERR_RUNTIME, "a redone batch failed validation again");
    return KVQ_ERR_RUNTIME;
}

// returns KVQ_OK, an error code, or KVQ_NEED_RESCAN when host batches must be fed again
int kvq_scan_finish_internal(kvq_scan *s)
{
    for (int attempt = 0; attempt < 4; attempt++) {
        int rc = finish_once(s);
        if (rc != KVQ_NEED_RESCAN) return rc;
        if (s->host_batches) return KVQ_NEED_RESCAN;
        // device batches are still resident: replay them into the larger arena
        std::vector<Batch> again; again.swap(s->batches);
        drop_events(s); s->main_launches = 0; s->path_bits = 0; s->tail_pending = false;
        if ((rc = reset_device_state(s))) return rc;
        s->pool.used = 0;
        for (size_t b = 0; b < again.size(); b++) {
            if (again[b].is_redo) continue;            // the exhaustive redo of a failed batch: its original is replayed and judged afresh
            Batch nb = again[b]; nb.redone = false; nb.skips_done = false;
            s->batches.push_back(nb);
            rc = run_batch(s, nb.d_data, nb.nbytes, nb.chunk_off.data(), (int64_t)nb.chunk_off.size() - 1, nb.fpos_base, s->batches.size() - 1, false);
            if (rc) return rc;
        }
    }
    kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results");
    return KVQ_ERR_MEMORY;
}

// several GPUs (kvq_scan_set_comm): `finish` is collective.  First the ranks agree on how their own scans
// ended -- the maximum of 0 (fine), 1 (this rank has to be fed its host batches again) and 2 (failed): only when
// every rank is fine are the counters of all ranks summed (into an array of their own: the rank's counters stay
// what they are, so that finishing twice does not sum sums); when some rank has to go round again every rank
// returns KVQ_ERR_RESCAN and goes round with it, with no sum taken (the collectives of the ranks stay in step);
// a failure anywhere is an error everywhere.
static int finish_over_ranks(kvq_scan *s, int rc_own)
{
    int rc;
    // (d_finish and d_ctr_all were made by kvq_scan_set_comm: nothing in front of the status exchange can fail on this rank alone)
    if ((rc = s->d_finish.ensure(sizeof(KvqFinishState) + 256))) return rc;
    unsigned long long *scratch = (unsigned long long *)((char *)s->d_finish.p + ((sizeof(KvqFinishState) + 15) & ~(size_t)15));
    unsigned long long worst = 0;
    const unsigned long long mine = rc_own == KVQ_OK ? 0ull : rc_own == KVQ_ERR_RESCAN ? 1ull : 2ull;
    if ((rc = kvq_comm_max_status(s->comm, mine, scratch, s->stream, &worst))) return rc;
    if (worst == 2) {
        if (rc_own != KVQ_OK && rc_own != KVQ_ERR_RESCAN) return rc_own;             // (its own message stands)
        kvq_set_error(KVQ_ERR_RUNTIME, "the scan of another rank has failed");
        return KVQ_ERR_RUNTIME;
    }
    if (worst == 1) {
        if (rc_own != KVQ_ERR_RESCAN) kvq_set_error(KVQ_ERR_RESCAN, "the hit arena of another rank overflowed on host batches: every rank resets its scan and feeds its batches again");
        s->finished = false;
        return KVQ_ERR_RESCAN;
    }
    if ((rc = s->d_ctr_all.ensure((size_t)s->t->ctr_len * 8))) return rc;
    if ((rc = kvq_comm_reduce_counters(s->comm, s->d_ctr, s->d_ctr_all.as<unsigned long long>(), s->t->ctr_len, scratch, s->stream))) return rc;
    KVQ_HIP(hipMemcpyAsync(s->pin, s->d_ctr_all.p, (size_t)s->t->ctr_len * 8, hipMemcpyDeviceToHost, s->stream));
    KVQ_HIP(hipStreamSynchronize(s->stream));
    memcpy(s->h_ctr.data(), s->pin, (size_t)s->t->ctr_len * 8);
    s->reduced = true;
    return KVQ_OK;
}

extern "C" int32_t kvq_scan_finish(kvq_scan *s)
{
    kvq_clear_error();
    int rc = kvq_scan_finish_internal(s);
    if (rc == KVQ_NEED_RESCAN) { kvq_set_error(KVQ_ERR_RESCAN, "hit arena overflow on host batches: the arena has been enlarged, reset the scan and feed the batches again"); rc = KVQ_ERR_RESCAN; }
    if (s->comm) {
        const int saved = kvq_error_code(); char msg[1024]; kvq_last_error(msg, sizeof(msg));
        const int rc2 = finish_over_ranks(s, rc);
        if (rc && rc2 == rc) kvq_set_error(saved, "%s", msg);
        rc = rc2;
    }
    return rc;
}

extern "C" int64_t kvq_scan_n_hits(const kvq_scan *s) { return (int64_t)s->n_hits; }
extern "C" const int32_t *kvq_scan_hit_seq_nr(const kvq_scan *s) { return reinterpret_cast<const int32_t *>(s->pin_res + s->res.seq_nr); }
extern "C" const int64_t *kvq_scan_hit_file_pos(const kvq_scan *s) { return reinterpret_cast<const int64_t *>(s->pin_res + s->res.file_pos); }
extern "C" const int32_t *kvq_scan_hit_seq_pos(const kvq_scan *s) { return reinterpret_cast<const int32_t *>(s->pin_res + s->res.seq_pos); }
extern "C" const int32_t *kvq_scan_hit_length(const kvq_scan *s) { return reinterpret_cast<const int32_t *>(s->pin_res + s->res.length); }
extern "C" const int32_t *kvq_scan_hit_readlength(const kvq_scan *s) { return reinterpret_cast<const int32_t *>(s->pin_res + s->res.readlength); }
extern "C" const uint8_t *kvq_scan_hitseq_blob(const kvq_scan *s) { return s->pin_res + s->res.blob; }
extern "C" const int64_t *kvq_scan_hitseq_offsets(const kvq_scan *s) { return reinterpret_cast<const int64_t *>(s->pin_res + s->res.hitseq_off); }
extern "C" const int64_t *kvq_scan_counters(const kvq_scan *s) { return s->h_ctr.data(); }
extern "C" void *kvq_scan_device_counters(const kvq_scan *s) { return s->reduced ? s->d_ctr_all.p : (void *)s->d_ctr; }
extern "C" void *kvq_scan_device_counters_own(const kvq_scan *s) { return s->d_ctr; }
extern "C" int64_t kvq_scan_parsed(const kvq_scan *s) { return s->parsed; }
extern "C" int64_t kvq_scan_total(const kvq_scan *s) { return s->total; }
extern "C" double kvq_scan_kernel_ms(const kvq_scan *s) { return s->ms_all; }
extern "C" double kvq_scan_main_kernel_ms(const kvq_scan *s) { return s->ms_main; }
// (measurement) ms from the end of a's last main kernel to the start of b's first one (both finished, neither reset since); < 0: unknown
extern "C" double kvq_scan_gap_ms(const kvq_scan *a, const kvq_scan *b)
{
    if (!a || !b || a->ev_main.empty() || b->ev_main.empty()) return -1.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, a->ev_main.back().second, b->ev_main.front().first) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
    return (double)ms;
}
extern "C" int64_t kvq_scan_main_kernel_launches(const kvq_scan *s) { return s->main_launches; }

// ---------------------------------------------------------------------------
// fastq_rewind / fastq_read chunk cuts on an in-memory stream
// ---------------------------------------------------------------------------

// length of the trailing partial record of buf[0..n): going backwards, the
// first line start '@' met after a line start '+' (workhorse.c:696-718)
int64_t kvq_tail_record(const uint8_t *buf, int64_t n)
{
    bool plus_seen = false;
    for (int64_t k = n - 1; k >= 2; k--) {                 // i = n - k runs 1 .. n-2 (706)
        const uint8_t prev = buf[k - 1];
        if (prev != '\n' && prev != '\r') continue;
        if (buf[k] == '+') plus_seen = true;
        else if (buf[k] == '@' && plus_seen) return n - k;
    }
    return -1;
}

extern "C" int64_t kvq_chunk_offsets(const uint8_t *data, int64_t nbytes, int64_t *offsets, int64_t cap)
{
    int64_t n = 0, cs = 0, fill = 0;
    for (;;) {
        const int64_t want = KVQ_SCANBUFSIZE - (fill - cs);
        const int64_t have = nbytes - fill;
        if (have >= want) {                                // buffer filled, source not dry: cut (916-943)
            const int64_t end = fill + want;
            const int64_t keep = kvq_tail_record(data + cs, end - cs);
            if (keep < 0) return -1;
            if (n < cap) offsets[n] = cs;
            n++;
            cs = end - keep; fill = end;
        } else {                                           // short read: eof, no cut (901-910)
            if (nbytes > cs) { if (n < cap) offsets[n] = cs; n++; }
            break;
        }
    }
    if (n < cap + 1) offsets[n] = nbytes;
    return n;
}
