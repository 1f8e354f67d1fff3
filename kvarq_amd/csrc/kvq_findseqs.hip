// kvarq_amd/csrc/kvq_findseqs.hip -- engine.findseqs (workhorse.c:1249-1464) on
// top of the scan object: a reader that produces the concatenated inflated
// stream of the input files (plain or gzip, workhorse.c:559-629), cut into the
// chunks fastq_read would hand out (workhorse.c:737-956), fed batch by batch
// through pinned host buffers to the GPU; live stats and cooperative stop
// (workhorse.c:1205-1244, 1469-1479).
#include "kvq_host.h"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <signal.h>
#include <string.h>
#include <zlib.h>
#include <thread>
#include <sys/mman.h>
#include <unistd.h>

int64_t kvq_tail_record(const uint8_t *buf, int64_t n);
int kvq_scan_finish_internal(kvq_scan *s);

// ---------------------------------------------------------------------------
// live state shared with engine.stats()/engine.stop()
// ---------------------------------------------------------------------------

static std::atomic<int> g_running{0}, g_stop{0}, g_sigints{0};
static struct { kvq_table *t = nullptr; kvq_scan *s = nullptr; int dev = -1; } g_kept;       // the last kvq_findseqs call's table and scan object (kvq_findseqs_free)
static std::mutex g_kept_lock;
void kvq_drop_kept_scan()
{
    std::lock_guard<std::mutex> l(g_kept_lock);
    if (g_kept.s) { kvq_table *kt = g_kept.t; kvq_scan_destroy(g_kept.s); kvq_table_destroy(kt); g_kept.t = nullptr; g_kept.s = nullptr; }
}
static std::mutex g_live_lock;
static struct {
    int64_t records = 0, parsed = 0, total = 0, longest = -1;
    int32_t nseq = 0;
    std::vector<int64_t> readlengths = std::vector<int64_t>(KVQ_MAX_READLENGTH, 0), nseqhits, nseqbasehits;
} g_live;

extern "C" void kvq_request_stop(void) { g_stop++; }
extern "C" void kvq_count_sigint(void) { g_sigints++; }

// The reference installs its counting handler with signal() when the module is imported (workhorse.c:133-136,
// 1632): a C handler, so that a SIGINT is counted while the main thread sits inside findseqs (a Python-level
// handler only runs once the interpreter gets control back).  Here installing it is an explicit call, and it
// can be undone; a lock-free atomic increment is all the handler does.
static struct sigaction g_sigint_before;
static bool g_sigint_installed = false;
static void on_sigint(int) { g_sigints++; }
static_assert(std::atomic<int>::is_always_lock_free, "the SIGINT handler only touches a lock-free counter");

extern "C" int kvq_sigint_counter_install(void)
{
    if (g_sigint_installed) return KVQ_OK;
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_sigint; sigemptyset(&sa.sa_mask); sa.sa_flags = SA_RESTART;
    if (sigaction(SIGINT, &sa, &g_sigint_before) != 0) { kvq_set_error(KVQ_ERR_RUNTIME, "cannot install the SIGINT handler"); return KVQ_ERR_RUNTIME; }
    g_sigint_installed = true;
    return KVQ_OK;
}

extern "C" void kvq_sigint_counter_remove(void)
{
    if (!g_sigint_installed) return;
    (void)sigaction(SIGINT, &g_sigint_before, nullptr);
    g_sigint_installed = false;
}

extern "C" void kvq_poll_stats(kvq_live_stats *out, int64_t *readlengths, int64_t *nseqhits, int64_t *nseqbasehits, int32_t nseq_cap)
{
    std::lock_guard<std::mutex> l(g_live_lock);
    if (out) {
        out->records_parsed = g_live.records; out->parsed = g_live.parsed; out->total = g_live.total;
        out->rls_longest = g_live.longest; out->nseq = g_live.nseq; out->running = g_running.load();
        out->sigints = g_sigints.load(); out->stop_requested = g_stop.load();
    }
    if (readlengths) memcpy(readlengths, g_live.readlengths.data(), KVQ_MAX_READLENGTH * sizeof(int64_t));
    const int32_t n = std::min<int32_t>(nseq_cap, g_live.nseq);
    if (nseqhits && n > 0) memcpy(nseqhits, g_live.nseqhits.data(), (size_t)n * sizeof(int64_t));
    if (nseqbasehits && n > 0) memcpy(nseqbasehits, g_live.nseqbasehits.data(), (size_t)n * sizeof(int64_t));
}

static void live_reset(int32_t nseq, int64_t total)
{
    std::lock_guard<std::mutex> l(g_live_lock);
    g_live.records = 0; g_live.parsed = 0; g_live.total = total; g_live.longest = -1; g_live.nseq = nseq;
    std::fill(g_live.readlengths.begin(), g_live.readlengths.end(), 0);
    g_live.nseqhits.assign((size_t)nseq, 0); g_live.nseqbasehits.assign((size_t)nseq, 0);
}

static void live_from_counters(const kvq_table *t, const int64_t *ctr, int64_t parsed, int64_t total)
{
    std::lock_guard<std::mutex> l(g_live_lock);
    g_live.records = ctr[KVQ_CTR_RECORDS]; g_live.longest = ctr[KVQ_CTR_LONGEST] - 1;
    g_live.parsed = parsed; g_live.total = total;
    memcpy(g_live.readlengths.data(), ctr + KVQ_CTR_READLENGTHS, KVQ_MAX_READLENGTH * sizeof(int64_t));
    if (t->nseq) {
        memcpy(g_live.nseqhits.data(), ctr + t->off_nseqhits, (size_t)t->nseq * sizeof(int64_t));
        memcpy(g_live.nseqbasehits.data(), ctr + t->off_nseqbasehits, (size_t)t->nseq * sizeof(int64_t));
    }
}

// ---------------------------------------------------------------------------
// the inflated stream of a list of files
// ---------------------------------------------------------------------------

// Serial inflate of one .gz file, member after member (workhorse.c:482-541, 559-629, 790-884).  It owns its
// FILE and knows nothing of the stream around it, so that it can run ahead of the stream in a thread of its
// own (GzAhead): an error is kept -- code, message, position within the file's inflated bytes -- for the
// stream's thread to raise.
struct GzSerial {
    FILE *fd = nullptr; z_stream zs; bool zs_live = false; uint8_t *inbuf = nullptr;
    int64_t remaining = 0;      // compressed bytes of the file not yet read
    int64_t consumed = 0;       // compressed bytes of the file behind the read position (what ftell() says)
    int64_t produced = 0;       // inflated bytes handed out
    int err = 0; char msg[256]; int64_t err_at = -1;          // err_at >= 0: the message ends " fpos=<stream offset of the file + err_at>"

    ~GzSerial() { close(); free(inbuf); }
    void close()
    {
        if (zs_live) { inflateEnd(&zs); zs_live = false; }
        if (fd) { fclose(fd); fd = nullptr; }
    }
    int fail(int code, const char *fmt, const char *a = "", const char *b = "")
    {
        err = code; snprintf(msg, sizeof(msg), fmt, a, b); return code;
    }
    int getc_counted() { const int c = fgetc(fd); if (c != EOF) consumed++; return c; }

    // workhorse.c:482-541
    const char *skip_gz_header(int dist)
    {
        int state = 0, y = 0, c;
        for (c = getc_counted(); state != 2 && y <= dist && c != EOF; c = getc_counted()) {
            if (c == 0x1F && state == 0) state = 1;
            else if (c == 0x8B && state == 1) state = 2;
            else { state = 0; y++; }
        }
        if (state != 2) return "magic bytes not found";
        if (c != 8) return "expected method==DEFLATED";
        const int flags = getc_counted();
        if (flags & (0x02 | 0x20 | 0xC0)) return "unsupported flags (CONTINUATION or ENCRYPTED or RESERVED)";
        for (int i = 0; i < 6; i++) (void)getc_counted();
        if (flags & 0x04) { int n = getc_counted(); n |= getc_counted() << 8; while (n-- > 0) (void)getc_counted(); }
        if (flags & 0x08) { do c = getc_counted(); while (c > 0); }
        if (flags & 0x10) { do c = getc_counted(); while (c > 0); }
        return nullptr;
    }

    // takes over `f`, positioned at file offset `at` of `file_size`: at == 0 is the beginning of the file (the header
    // must start right there, workhorse.c:613-621), anything else a later member (header within 10 bytes; *no_member
    // when there is none: the stream ends, workhorse.c:851-853)
    int start(FILE *f, int64_t file_size, int64_t at, bool *no_member)
    {
        fd = f; consumed = at; produced = 0; err = 0; err_at = -1;
        if (no_member) *no_member = false;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -MAX_WBITS) != Z_OK) return fail(KVQ_ERR_RUNTIME, "cannot mz_inflateInit()");
        zs_live = true;
        if (!inbuf) inbuf = (uint8_t *)malloc(KVQ_SCANBUFSIZE);
        if (!inbuf) return fail(KVQ_ERR_MEMORY, "cannot allocate inbuf");
        fseek(fd, (long)at, SEEK_SET);
        remaining = file_size - at;
        const char *m = skip_gz_header(at == 0 ? 0 : 10);
        if (m) {
            if (at == 0) return fail(KVQ_ERR_IO, "no valid gzip header found at beginning of file : %s", m);
            *no_member = true; return KVQ_OK;
        }
        remaining -= consumed - at;
        return KVQ_OK;
    }

    // up to cap inflated bytes; *eof when the file is exhausted; -1 (and err/msg/err_at) on failure
    int64_t read(uint8_t *dst, int64_t cap, bool *eof)
    {
        *eof = false;
        zs.next_out = dst; zs.avail_out = (uInt)cap;
        bool done = false;
        while (zs.avail_out > 0 && !done) {
            if (zs.avail_in == 0) {
                if (remaining <= 0) { done = true; break; }
                const int64_t m = std::min<int64_t>(KVQ_SCANBUFSIZE, remaining);
                if ((int64_t)fread(inbuf, 1, (size_t)m, fd) != m) {
                    fail(KVQ_ERR_IO, "could not read enough bytes from .fastq.gz%s%s", ferror(fd) ? " : I/O error" : "", feof(fd) ? " : premature EOF" : "");
                    return -1;
                }
                consumed += m; remaining -= m;
                zs.next_in = inbuf; zs.avail_in = (uInt)m;
            }
            const int st = inflate(&zs, Z_SYNC_FLUSH);
            if (st != Z_OK && st != Z_STREAM_END && st != Z_BUF_ERROR) {
                err = KVQ_ERR_IO; snprintf(msg, sizeof(msg), "error while inflating compressed data : status=%d", st);
                err_at = produced + (cap - zs.avail_out);
                return -1;
            }
            if (st == Z_STREAM_END) {
                // another gzip member follows when more than a trailer is left (842-866)
                if (remaining + (int64_t)zs.avail_in > 10) {
                    fseek(fd, -(long)zs.avail_in, SEEK_CUR);
                    consumed -= zs.avail_in; remaining += zs.avail_in; zs.avail_in = 0;
                    const int64_t before = consumed;
                    const char *m = skip_gz_header(10);
                    if (m) { remaining = 0; done = true; }
                    else {
                        remaining -= consumed - before;
                        uint8_t *no = zs.next_out; const uInt ao = zs.avail_out;
                        inflateEnd(&zs); memset(&zs, 0, sizeof(zs)); inflateInit2(&zs, -MAX_WBITS);
                        zs.next_out = no; zs.avail_out = ao;
                    }
                } else done = true;
            } else if (st == Z_BUF_ERROR && zs.avail_in == 0 && remaining <= 0) done = true;
        }
        const int64_t n = cap - zs.avail_out;
        if (zs.avail_out > 0) *eof = true;
        produced += n;
        return n;
    }
};

// A GzSerial in a thread of its own, inflating into a queue of blocks for the stream to pick up.  The stream
// is strictly one file after the other (the file positions of hits count inflated bytes of every file before,
// workhorse.c:641-686), but nothing says the files must be INFLATED one after the other: the reader of the
// second file of a pair starts together with the first file's and runs up to `budget` inflated bytes ahead.
struct GzAhead {
    struct Block { uint8_t *p; int64_t n, used, consumed_after; bool eof, failed; };
    static const int64_t BLOCK = 4 << 20;
    GzSerial z; int open_err = 0;
    std::thread th; std::mutex m; std::condition_variable cv; std::deque<Block> q;
    int64_t queued = 0, budget = 0; bool quit = false;
    std::vector<uint8_t *> spare;                 // blocks handed back by the consumer, written again without page faults

    // a fresh block costs a page fault per 4 KiB written, a third of the inflate time itself: ask for huge pages
    static uint8_t *new_block()
    {
        void *p = nullptr;
        if (posix_memalign(&p, 2 << 20, (size_t)BLOCK)) return nullptr;
        (void)madvise(p, (size_t)BLOCK, MADV_HUGEPAGE);
        return (uint8_t *)p;
    }

    GzAhead(const char *name, int64_t budget_bytes) : budget(budget_bytes)
    {
        FILE *f = fopen(name, "rb");
        if (!f) { open_err = z.fail(KVQ_ERR_IO, "cannot open file"); return; }
        fseek(f, 0, SEEK_END); const int64_t size = ftell(f);
        open_err = z.start(f, size, 0, nullptr);
        if (!open_err) th = std::thread([this] { run(); });
    }
    ~GzAhead()
    {
        { std::lock_guard<std::mutex> l(m); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
        for (auto &b : q) free(b.p);
        for (auto p : spare) free(p);
    }
    void run()
    {
        for (;;) {
            uint8_t *mem = nullptr;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return quit || queued < budget; });
                if (quit) return;
                if (!spare.empty()) { mem = spare.back(); spare.pop_back(); }
            }
            Block b = { mem ? mem : new_block(), 0, 0, 0, false, false };
            if (!b.p) { z.fail(KVQ_ERR_MEMORY, "cannot allocate memory for scanning"); b.failed = true; }
            else {
                const int64_t n = z.read(b.p, BLOCK, &b.eof);
                if (n < 0) b.failed = true; else b.n = n;
            }
            b.consumed_after = z.consumed;
            const bool last = b.eof || b.failed;
            {
                std::lock_guard<std::mutex> l(m);
                q.push_back(b); queued += b.n;
            }
            cv.notify_all();
            if (last) return;
        }
    }
    // the consumer's side of GzSerial::read; *consumed follows the compressed bytes behind what has been handed out.
    // Whatever is queued is copied out by up to `nthreads` threads at once: behind the first file of a pair the whole
    // second file may be waiting, and one memcpy stream would then be what the scan waits for.
    int64_t read(uint8_t *dst, int64_t cap, bool *eof, int64_t *consumed, int nthreads)
    {
        struct Span { const uint8_t *from; uint8_t *to; int64_t n; };
        *eof = false;
        int64_t n = 0;
        while (n < cap && !*eof) {
            std::vector<Span> spans;
            size_t whole = 0;                             // blocks used up by this round
            bool failed = false;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return !q.empty(); });
                for (auto &b : q) {                       // (only this thread pops, and a deque keeps its elements in place when the reader pushes)
                    const int64_t k = std::min(cap - n, b.n - b.used);
                    if (k > 0) spans.push_back({ b.p + b.used, dst + n, k });
                    n += k; b.used += k;
                    if (b.used < b.n) break;              // cap reached inside the block
                    *consumed = b.consumed_after;
                    if (b.failed) { failed = true; break; }
                    whole++;
                    if (b.eof) { *eof = true; break; }
                    if (n == cap) break;
                }
            }
            int64_t bytes = 0;
            for (auto &sp : spans) bytes += sp.n;
            const int nt = bytes < (8 << 20) ? 1 : std::max(1, std::min<int>(std::min<int>(nthreads, 8), (int)spans.size()));
            auto copy = [&](int t) { for (size_t i = (size_t)t; i < spans.size(); i += (size_t)nt) memcpy(spans[i].to, spans[i].from, (size_t)spans[i].n); };
            std::vector<std::thread> helpers;
            for (int t = 1; t < nt; t++) helpers.emplace_back(copy, t);
            copy(0);
            for (auto &h : helpers) h.join();
            if (failed) return -1;
            {
                std::lock_guard<std::mutex> l(m);
                for (; whole > 0; whole--) {
                    if (spare.size() < 4) spare.push_back(q.front().p); else free(q.front().p);
                    queued -= q.front().n; q.pop_front();
                }
            }
            cv.notify_all();
        }
        return n;
    }
};

class StreamSource {
public:
    ~StreamSource() { close_file(); }

    // workhorse.c:641-686: sizes of all files first, then the first file is opened
    int open(const char *const *files, int nfiles)
    {
        for (int i = 0; i < nfiles; i++) files_.push_back(files[i]);
        for (auto &f : files_) {
            FILE *fd = fopen(f.c_str(), "rb");
            if (!fd) { kvq_set_error(KVQ_ERR_IO, "cannot open file '%s' for getting filesize", f.c_str()); return KVQ_ERR_IO; }
            fseek(fd, 0, SEEK_END); size_ += ftell(fd); fclose(fd);
        }
        total_ = size_;
        return KVQ_OK;
    }
    bool has_next_file() const { return next_ < files_.size(); }

    // workhorse.c:559-629
    int open_next()
    {
        close_file();
        const std::string &name = files_[next_++];
        fd_ = fopen(name.c_str(), "rb");
        if (!fd_) { kvq_set_error(KVQ_ERR_IO, "cannot open file"); return KVQ_ERR_IO; }
        consumed_ = 0; file_done_ = false; opened_ = true; file_fpos0_ = fpos_; serial_fpos0_ = fpos_;
        fseek(fd_, 0, SEEK_END); file_size_ = ftell(fd_); fseek(fd_, 0, SEEK_SET);
        gz_ = name.size() >= 3 && name.compare(name.size() - 3, 3, ".gz") == 0;     // by suffix (582)
        bgzf_ = false;
        if (gz_) {
            // a file of BGZF blocks (bgzip: gzip members of at most 64 KiB that carry their own size)
            // is inflated by `nthreads` workers, block by block; anything else by the serial path below,
            // which also takes over should a later member not be a BGZF block
            BgzfBlock first;
            const char *sw = getenv("KVQ_BGZF");                              // KVQ_BGZF=0: serial reader only (diagnostic)
            if (!(sw && sw[0] == '0') && bgzf_peek(0, &first)) { bgzf_ = true; boff_ = 0; total_ *= 3; start_next_ahead(); return KVQ_OK; }
            if (ahead_next_ && ahead_next_for_ == next_ - 1) ahead_ = std::move(ahead_next_);     // its reader has been running since the file before was opened
            else if (ahead_budget() > 0) ahead_.reset(new GzAhead(name.c_str(), ahead_budget()));
            if (ahead_) {
                if (ahead_->open_err) return raise(ahead_->z);
            } else {
                int rc = z_.start(fd_, file_size_, 0, nullptr);
                fd_ = nullptr;                                                        // (z_ owns the FILE now)
                if (rc) return raise(z_);
                consumed_ = z_.consumed;
            }
            total_ *= 3;                                                              // "random guess" (625)
        }
        start_next_ahead();
        return KVQ_OK;
    }

    // up to cap bytes of the current file's inflated stream; *eof when the file is exhausted
    int64_t read(uint8_t *dst, int64_t cap, bool *eof)
    {
        *eof = false;
        if (file_done_) { *eof = true; return 0; }
        int64_t n = 0;
        if (!gz_) {
            // plain file: `nthreads` readers pread() disjoint slices straight into the pinned buffer
            // (the reference's workers share one fread under a mutex, workhorse.c:746,890)
            const int fdn = fileno(fd_);
            const int64_t left = file_size_ - consumed_;
            n = left < cap ? (left < 0 ? 0 : left) : cap;
            kvq_config cfg; kvq_config_get(&cfg);
            int nt = cfg.nthreads < 1 ? 1 : (cfg.nthreads > 32 ? 32 : cfg.nthreads);
            if (n < (4 << 20)) nt = 1;
            std::atomic<int> bad{0};
            auto slice = [&](int t) {
                int64_t a = n * t / nt, b = n * (t + 1) / nt;
                while (a < b) {
                    const ssize_t got = pread(fdn, dst + a, (size_t)(b - a), (off_t)(consumed_ + a));
                    if (got <= 0) { bad = 1; return; }
                    a += got;
                }
            };
            if (nt == 1) slice(0);
            else {
                std::vector<std::thread> th;
                for (int t = 1; t < nt; t++) th.emplace_back(slice, t);
                slice(0);
                for (auto &x : th) x.join();
            }
            if (bad.load()) { kvq_set_error(KVQ_ERR_IO, "error while reading from file in fastq_read"); return -1; }
            if (n < cap) { *eof = true; file_done_ = true; }
            consumed_ += n;
        } else {
            if (bgzf_) {
                const int64_t got = read_bgzf(dst, cap, eof);
                if (got != -2) return got;            // -2: the next member is no BGZF block -> serial path from here on
            }
            bool end = false;
            if (ahead_) {
                kvq_config cfg; kvq_config_get(&cfg);
                n = ahead_->read(dst, cap, &end, &consumed_, cfg.nthreads);
                if (n < 0) { raise(ahead_->z); return -1; }
            } else {
                n = z_.read(dst, cap, &end);
                if (n < 0) { raise(z_); return -1; }
                consumed_ = z_.consumed;
            }
            if (end) { *eof = true; file_done_ = true; }
            // running estimate of the inflated size, float arithmetic as in 883-884
            if (ftell0_ + consumed_ > 0)
                total_ = (int64_t)(size_t)((float)size_ * (fpos_ + n) / (ftell0_ + consumed_));
        }
        fpos_ += n;
        return n;
    }

    int64_t fpos() const { return fpos_; }
    int64_t total() const { return total_; }

private:
    // ---- BGZF (SAM/BAM specification, section 4.1): gzip member with FEXTRA subfield 'B','C',2,0,BSIZE ----
    struct BgzfBlock { int64_t off; uint32_t size, hdr, isize; };       // file offset, block bytes, header bytes, inflated bytes

    // is there a well-formed BGZF block at file offset `off`?
    bool bgzf_peek(int64_t off, BgzfBlock *b)
    {
        uint8_t h[12];
        const int fdn = fileno(fd_);
        if (off + 28 > file_size_ || pread(fdn, h, 12, (off_t)off) != 12) return false;
        if (h[0] != 0x1F || h[1] != 0x8B || h[2] != 8 || h[3] != 4) return false;             // exactly FEXTRA, as bgzip writes
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (xlen < 6 || xlen > 4096) return false;
        uint8_t x[4096];
        if (pread(fdn, x, xlen, (off_t)(off + 12)) != (ssize_t)xlen) return false;
        uint32_t bsize = 0;
        for (uint32_t i = 0; i + 4 <= xlen; ) {
            const uint32_t slen = x[i + 2] | (x[i + 3] << 8);
            if (x[i] == 'B' && x[i + 1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = (x[i + 4] | (x[i + 5] << 8)) + 1u;
            i += 4 + slen;
        }
        const uint32_t hdr = 12 + xlen;
        if (bsize < hdr + 8 || off + bsize > file_size_) return false;
        uint8_t t[4];
        if (pread(fdn, t, 4, (off_t)(off + bsize - 4)) != 4) return false;
        b->off = off; b->size = bsize; b->hdr = hdr;
        b->isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        return b->isize <= 65536;
    }

    // inflate as many whole BGZF blocks as fit into cap bytes, nthreads workers; -2 = hand over to the serial path
    int64_t read_bgzf(uint8_t *dst, int64_t cap, bool *eof)
    {
        std::vector<BgzfBlock> blocks;
        int64_t out = 0;
        bool handover = false;
        while (true) {
            if (file_size_ - boff_ <= 10) {                                               // at most a trailer is left (workhorse.c:842)
                consumed_ += file_size_ - boff_; boff_ = file_size_;                      // (the serial reader has read those bytes too)
                *eof = true; file_done_ = true; break;
            }
            BgzfBlock b;
            if (!bgzf_peek(boff_, &b)) { handover = true; break; }
            if (out + b.isize > cap) break;
            blocks.push_back(b); out += b.isize; boff_ += b.size;
        }
        if (handover && blocks.empty()) {
            // the serial reader continues at this member: position the file, skip its header as open_next does
            bgzf_ = false;
            bool no_member = false;
            z_.consumed = consumed_;
            serial_fpos0_ = fpos_;                                                 // (the serial reader counts what IT produces: its error positions are relative to here)
            const int rc = z_.start(fd_, file_size_, boff_, &no_member);
            fd_ = nullptr;
            if (rc) { raise(z_); return -1; }
            // (GzSerial counts file offsets; the stream's count of this file also holds what the block reader skipped)
            consumed_ += z_.consumed - boff_; z_.consumed = consumed_;
            if (no_member) { *eof = true; file_done_ = true; return 0; }          // as behind any member: no further header, the stream ends (851-853)
            return -2;
        }
        // read the compressed bytes of the whole run once, then inflate block by block in parallel
        if (!blocks.empty()) {
            const int64_t c0 = blocks.front().off, c1 = blocks.back().off + blocks.back().size;
            cbuf_.resize((size_t)(c1 - c0));
            const int fdn = fileno(fd_);
            for (int64_t a = 0; a < c1 - c0; ) {
                const ssize_t got = pread(fdn, cbuf_.data() + a, (size_t)(c1 - c0 - a), (off_t)(c0 + a));
                if (got <= 0) { kvq_set_error(KVQ_ERR_IO, "could not read enough bytes from .fastq.gz : I/O error"); return -1; }
                a += got;
            }
            std::vector<int64_t> at(blocks.size());
            int64_t o = 0;
            for (size_t i = 0; i < blocks.size(); i++) { at[i] = o; o += blocks[i].isize; }
            kvq_config cfg; kvq_config_get(&cfg);
            int nt = cfg.nthreads < 1 ? 1 : (cfg.nthreads > 32 ? 32 : cfg.nthreads);
            if ((size_t)nt > blocks.size()) nt = (int)blocks.size();
            std::atomic<int> bad{0};
            auto work = [&](int t) {
                z_stream z; memset(&z, 0, sizeof(z));
                if (inflateInit2(&z, -MAX_WBITS) != Z_OK) { bad = 1; return; }
                for (size_t i = blocks.size() * t / nt; i < blocks.size() * (t + 1) / nt; i++) {
                    const BgzfBlock &b = blocks[i];
                    z.next_in = cbuf_.data() + (b.off - c0) + b.hdr; z.avail_in = b.size - b.hdr - 8;
                    z.next_out = dst + at[i]; z.avail_out = b.isize;
                    const int st = inflate(&z, Z_FINISH);
                    if (st != Z_STREAM_END || z.avail_out != 0) { bad = (st == Z_STREAM_END || st == Z_OK || st == Z_BUF_ERROR) ? 2 : 3; break; }
                    inflateReset(&z);
                }
                inflateEnd(&z);
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; t++) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
            if (bad.load()) {
                kvq_set_error(KVQ_ERR_IO, "error while inflating compressed data : status=%d fpos=%ld", bad.load() == 3 ? Z_DATA_ERROR : Z_BUF_ERROR, (long)fpos_);
                return -1;
            }
            consumed_ += c1 - c0;
        }
        if (ftell0_ + consumed_ > 0)
            total_ = (int64_t)(size_t)((float)size_ * (fpos_ + out) / (ftell0_ + consumed_));      // as in the serial path (883-884)
        fpos_ += out;
        return out;
    }

    // an error met by a gzip reader, possibly in its own thread, raised in this one
    int raise(const GzSerial &z)
    {
        if (z.err_at >= 0) kvq_set_error(z.err, "%s fpos=%ld", z.msg, (long)((&z == &z_ ? serial_fpos0_ : file_fpos0_) + z.err_at));
        else kvq_set_error(z.err, "%s", z.msg);
        return z.err;
    }

    // inflated bytes a reader may run ahead of the stream, per reader (the current file's and the next file's run at once):
    // KVQ_GZ_AHEAD_MB, else 256 MiB -- the scan takes 64 MiB at a time, a few batches of look-ahead keep it fed -- and never
    // more than an eighth of the free memory; none with nthreads == 1 (one worker was asked for).  Worked out once per walk.
    int64_t ahead_budget()
    {
        if (ahead_budget_ >= 0) return ahead_budget_;
        kvq_config cfg; kvq_config_get(&cfg);
        if (cfg.nthreads <= 1) return ahead_budget_ = 0;
        if (const char *e = getenv("KVQ_GZ_AHEAD_MB")) return ahead_budget_ = (int64_t)atol(e) << 20;
        const int64_t avail = (int64_t)sysconf(_SC_AVPHYS_PAGES) * sysconf(_SC_PAGESIZE);
        return ahead_budget_ = std::max<int64_t>(64ll << 20, std::min<int64_t>(avail / 8, 256ll << 20));
    }
    int64_t ahead_budget_ = -1;

    // the file after the one just opened: when it is a plain .gz (not BGZF -- those are inflated block-parallel when
    // their turn comes), its reader starts now
    void start_next_ahead()
    {
        if (next_ >= files_.size() || ahead_next_ || ahead_budget() <= 0) return;
        const std::string &name = files_[next_];
        if (!(name.size() >= 3 && name.compare(name.size() - 3, 3, ".gz") == 0)) return;
        const char *sw = getenv("KVQ_BGZF");
        if (!(sw && sw[0] == '0')) {
            FILE *keep = fd_; const int64_t keep_size = file_size_;
            FILE *f = fopen(name.c_str(), "rb");
            if (!f) return;                                     // (open_next reports it when the file's turn comes)
            fseek(f, 0, SEEK_END); file_size_ = ftell(f); fd_ = f;
            BgzfBlock first;
            const bool is_bgzf = bgzf_peek(0, &first);
            fclose(f); fd_ = keep; file_size_ = keep_size;
            if (is_bgzf) return;
        }
        ahead_next_.reset(new GzAhead(name.c_str(), ahead_budget()));
        ahead_next_for_ = next_;
    }

    void close_file()
    {
        if (opened_) { ftell0_ += consumed_; opened_ = false; }
        if (fd_) { fclose(fd_); fd_ = nullptr; }
        z_.close(); ahead_.reset();
    }

    std::vector<std::string> files_; size_t next_ = 0;
    FILE *fd_ = nullptr; bool gz_ = false, file_done_ = true;
    bool opened_ = false; int64_t file_fpos0_ = 0;                              // a file is open / the stream offset it began at
    int64_t serial_fpos0_ = 0;                                                  // ... / the stream offset at which z_ started producing (behind a BGZF run: later than the file)
    GzSerial z_;                                                                // serial .gz reader in this thread ...
    std::unique_ptr<GzAhead> ahead_, ahead_next_; size_t ahead_next_for_ = 0;   // ... or in its own; the next file's, already running
    bool bgzf_ = false; int64_t boff_ = 0; std::vector<uint8_t> cbuf_;       // BGZF: next block's file offset, compressed run
    int64_t size_ = 0, ftell0_ = 0, consumed_ = 0, fpos_ = 0, total_ = 0, file_size_ = 0;
};

// ---------------------------------------------------------------------------
// driver
// ---------------------------------------------------------------------------

// new stream bytes per batch (KVQ_BATCH_BYTES_MB: 16..512, read once; the two pinned buffers are of this size)
static const int64_t BATCH_BYTES = [] { const char *e = getenv("KVQ_BATCH_BYTES_MB"); const long v = e ? atol(e) : 0; return (int64_t)(v >= 16 && v <= 512 ? v : 64) << 20; }();

// Walk the files once: Sink::batch(data, nbytes, chunk offsets, nchunks, fpos,
// parsed, total) is called for every run of whole chunks, in stream order.
// pin2 (optional): a second buffer of the same size; the walk then alternates between the two
// after every batch, so that the sink may still be reading the batch it was handed last (the
// sink must be done with a batch when it is handed the next one)
template <class Sink>
static int stream_batches(Sink &sink, const char *const *files, int nfiles, uint8_t *pin, int64_t pin_cap,
                          int64_t *parsed, int64_t *total, uint8_t *pin2 = nullptr)
{
    StreamSource src;
    int rc = src.open(files, nfiles);
    if (rc) return rc;
    sink.begin(src.total());
    double t_read = 0, t_cut = 0, t_sink = 0, t_carry = 0; int64_t nbatch = 0;       // (KVQ_TIMING=1: where the host's time goes)
    struct Report { double &a, &b, &c, &d; int64_t &n; ~Report() { if (g_timing) fprintf(stderr, "stream_batches: %lld batches; read %.1f  cut %.1f  sink (wait for the last batch + enqueue) %.1f  carry %.1f ms\n", (long long)n, a, b, c, d); } } report{ t_read, t_cut, t_sink, t_carry, nbatch };

    while (src.has_next_file() && !g_stop.load()) {
        if ((rc = src.open_next())) return rc;
        // chunker state of this file, offsets relative to pin[0]
        int64_t have = 0;              // bytes of the file's stream sitting in pin
        int64_t pin_fpos = src.fpos(); // stream offset of pin[0]
        int64_t cs = 0, fill = 0;      // current chunk start / how far the reference has read (== cs + leftover)
        bool eof = false;
        while (!g_stop.load()) {
            // top up
            const double tr0 = now_ms();
            while (!eof && have < pin_cap) {
                const int64_t n = src.read(pin + have, pin_cap - have, &eof);
                if (n < 0) return kvq_error_code();
                have += n;
                if (n == 0 && !eof) break;
            }
            const double tr1 = now_ms(); t_read += tr1 - tr0;
            // cut chunks the way fastq_read does (workhorse.c:737-956)
            std::vector<int64_t> off;
            bool file_finished = false;
            for (;;) {
                const int64_t want = KVQ_SCANBUFSIZE - (fill - cs);
                if (have - fill >= want) {
                    const int64_t end = fill + want;
                    const int64_t keep = kvq_tail_record(pin + cs, end - cs);
                    if (keep < 0) {
                        kvq_set_error(KVQ_ERR_RUNTIME, "could find beginning of record; read %ld bytes up to %ld", (long)want, (long)(pin_fpos + end));
                        return KVQ_ERR_RUNTIME;
                    }
                    off.push_back(cs);
                    cs = end - keep; fill = end;
                } else if (eof) {
                    if (have > cs) off.push_back(cs);
                    cs = fill = have; file_finished = true;
                    break;
                } else break;        // need more data
            }
            const int64_t batch_begin = off.empty() ? cs : off[0];
            const int64_t batch_end = cs;
            const double tr2 = now_ms(); t_cut += tr2 - tr1;
            if (!off.empty()) {
                off.push_back(batch_end);
                for (auto &o : off) o -= batch_begin;
                rc = sink.batch(pin + batch_begin, batch_end - batch_begin, off.data(), (int64_t)off.size() - 1,
                                pin_fpos + batch_begin, src.fpos(), src.total());
                if (rc) return rc;
                nbatch++;
            }
            const double tr3 = now_ms(); t_sink += tr3 - tr2;
            if (file_finished) {
                if (pin2 && !off.empty()) std::swap(pin, pin2);       // the next file starts in the other buffer
                break;
            }
            // carry the unfinished chunk to the front of the (other) buffer
            const int64_t carry = have - cs;
            if (carry >= pin_cap) { kvq_set_error(KVQ_ERR_RUNTIME, "buf_size < fastq->buf_size !"); return KVQ_ERR_RUNTIME; }
            if (pin2 && !off.empty()) { memcpy(pin2, pin + cs, (size_t)carry); std::swap(pin, pin2); }
            else memmove(pin, pin + cs, (size_t)carry);
            pin_fpos += cs; fill -= cs; have = carry; cs = 0;
            t_carry += now_ms() - tr3;
        }
    }
    *parsed = src.fpos(); *total = src.total();
    return KVQ_OK;
}

// the GPU sink: one kvq_scan_host per batch, live stats after each
struct ScanSink {
    kvq_scan *s = nullptr;
    // The live statistics (engine.stats() from another thread while findseqs runs: workhorse.c:1205-1244 reads the workers' counters
    // under their lock) are ONE consistent snapshot: the head of the counter array copied on the scan's own stream, behind the kernels
    // of the batches enqueued so far and in front of the next one's, into pinned memory -- published by the next call together with
    // the byte count that belongs to it.  (Round 3 copied on the null stream, beside the non-blocking stream's kernels: records, read
    // lengths and hits per sequence of different batches in one answer.)
    int64_t *snap = nullptr; size_t snap_cap = 0, head = 0; hipEvent_t snap_ev = nullptr;
    bool snap_pending = false; int64_t snap_parsed = 0, enq_parsed = 0, last_parsed = 0;
    ~ScanSink() { if (snap_ev) (void)hipEventDestroy(snap_ev); if (snap) pinned_give(snap, snap_cap); }
    void begin(int64_t total)
    {
        live_reset(s->t->nseq, total);
        // (only the head of the counter array -- scalars, read lengths, hits per sequence: 12 KB for the MTBC table -- not the
        // coverage and mutation counters behind it, 0.8 MB)
        head = std::min((size_t)std::max<int64_t>(s->t->off_nseqbasehits + s->t->nseq, KVQ_CTR_READLENGTHS + KVQ_MAX_READLENGTH), (size_t)s->t->ctr_len);
        if (!snap) snap = (int64_t *)pinned_take(head * 8, &snap_cap);
        if (!snap_ev) (void)hipEventCreateWithFlags(&snap_ev, hipEventDisableTiming);
        snap_pending = false; enq_parsed = last_parsed = 0;
    }
    int batch(const uint8_t *data, int64_t nbytes, const int64_t *off, int64_t nchunks, int64_t fpos, int64_t parsed, int64_t total)
    {
        // publish the snapshot taken behind the batches that were enqueued two calls ago (it landed while this batch was read)
        if (snap_pending) {
            if (hipEventSynchronize(snap_ev) != hipSuccess) { kvq_set_error(KVQ_ERR_DEVICE, "device failure during scan"); return KVQ_ERR_DEVICE; }
            live_from_counters(s->t, snap, snap_parsed, total);            // (it reads the head only)
            snap_pending = false;
        }
        // hand this batch over (its text sets out at once, the kernels of the batch before it are enqueued: kvq_scan_host_async) and
        // return, so that the reader fills the other host buffer while this one crosses PCIe
        const int rc = kvq_scan_host_async(s, data, nbytes, off, nchunks, fpos);
        if (rc) return rc;
        // the stream now holds the kernels of every batch up to the one handed over in the call before this: snapshot behind them
        if (snap && snap_ev && last_parsed > 0) {
            if (hipMemcpyAsync(snap, s->d_ctr, head * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess && hipEventRecord(snap_ev, s->stream) == hipSuccess) {
                snap_pending = true; snap_parsed = last_parsed;
            } else (void)hipGetLastError();
        }
        last_parsed = parsed;
        return KVQ_OK;
    }
};

// one pass over the files with the current arena; KVQ_NEED_RESCAN asks for another
static int findseqs_pass(kvq_scan *s, const char *const *files, int nfiles, uint8_t *pin, uint8_t *pin2, int64_t pin_cap)
{
    ScanSink sink; sink.s = s;
    int64_t parsed = 0, total = 0;
    const double tp0 = now_ms();
    int rc = stream_batches(sink, files, nfiles, pin, pin_cap, &parsed, &total, pin2);                 // two host buffers
    if (g_timing) fprintf(stderr, "findseqs pass: stream %.1f ms\n", now_ms() - tp0);
    if (rc) return rc;
    s->parsed = parsed; s->total = total;
    {
        std::lock_guard<std::mutex> l(g_live_lock);
        g_live.parsed = parsed; g_live.total = total;
    }
    rc = kvq_scan_finish_internal(s);
    if (rc == KVQ_OK) live_from_counters(s->t, s->h_ctr.data(), parsed, total);     // stats() after the scan == the scan's stats
    return rc;
}

// host-only view of the same walk (no GPU): the chunks fastq_read would hand
// out, as (stream offset, length) pairs -- what the CPU tests compare with the oracle
struct PlanSink {
    int64_t *fpos, *len; int64_t cap, n = 0;
    void begin(int64_t) {}
    int batch(const uint8_t *, int64_t, const int64_t *off, int64_t nchunks, int64_t base, int64_t, int64_t)
    {
        for (int64_t c = 0; c < nchunks; c++, n++)
            if (n < cap) { fpos[n] = base + off[c]; len[n] = off[c + 1] - off[c]; }
        return KVQ_OK;
    }
};

extern "C" int64_t kvq_host_chunk_plan(const char *const *files, int32_t nfiles, int64_t *chunk_fpos, int64_t *chunk_len,
                                       int64_t cap, int64_t *parsed, int64_t *total, int64_t batch_bytes)
{
    kvq_clear_error();
    const int64_t pin_cap = (batch_bytes > 0 ? batch_bytes : BATCH_BYTES) + 2 * KVQ_SCANBUFSIZE;
    uint8_t *buf = (uint8_t *)malloc((size_t)pin_cap);
    if (!buf) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for scanning"); return -1; }
    PlanSink sink; sink.fpos = chunk_fpos; sink.len = chunk_len; sink.cap = cap;
    const int rc = stream_batches(sink, files, nfiles, buf, pin_cap, parsed, total);
    free(buf);
    return rc ? -1 : sink.n;
}

extern "C" kvq_scan *kvq_findseqs(const char *const *files, int32_t nfiles,
                                  const uint8_t *const *seqs, const int32_t *seqlens, int32_t nseq)
{
    kvq_clear_error();
    int expected = 0;
    if (!g_running.compare_exchange_strong(expected, 1)) {            // workhorse.c:1258-1263
        kvq_set_error(KVQ_ERR_RUNTIME, "findseqs() already running!");
        return nullptr;
    }
    g_stop = 0; g_sigints = 0;                                         // workhorse.c:1264-1265
    const double tf0 = now_ms();
    // the table and the scan object of the last call are kept (kvq_findseqs_free): a caller that scans file after file with
    // the same sequences and settings -- the usual case -- does not build the seed index and a dozen device buffers again
    kvq_table *t = nullptr; kvq_scan *s = nullptr;
    {
        kvq_config cfg; kvq_config_get(&cfg);
        std::lock_guard<std::mutex> l(g_kept_lock);
        int dev_now = 0; (void)hipGetDevice(&dev_now);
        if (g_kept.s && g_kept.dev == dev_now && nseq == g_kept.t->nseq && memcmp(&cfg, &g_kept.t->cfg, sizeof(cfg)) == 0) {
            bool same = true;
            for (int32_t i = 0; i < nseq && same; i++) {
                const int32_t len = g_kept.t->h_off[i + 1] - g_kept.t->h_off[i];
                same = seqlens[i] == len && memcmp(seqs[i], g_kept.t->h_tab.data() + g_kept.t->h_off[i], (size_t)len) == 0;
            }
            if (same && kvq_scan_reset(g_kept.s) == KVQ_OK) {
                t = g_kept.t; s = g_kept.s; g_kept.t = nullptr; g_kept.s = nullptr;
                s->tile_bytes = 0; s->rec_bytes = 0;                  // (another file: the tiles are sized from its own head)
            }
        }
        if (!s && g_kept.s) { kvq_table *kt = g_kept.t; kvq_scan_destroy(g_kept.s); kvq_table_destroy(kt); g_kept.t = nullptr; g_kept.s = nullptr; }
        kvq_clear_error();
    }
    if (!s) {
        t = kvq_table_create(seqs, seqlens, nseq, nullptr);
        s = t ? kvq_scan_create(t, nullptr) : nullptr;
    }
    const double tf1 = now_ms();
    // the two pinned host buffers outlive the call: pinning and unpinning 130 MB costs more than
    // streaming a 1 GB file through them (only one findseqs runs at a time, g_running)
    static uint8_t *g_pin = nullptr;
    const int64_t pin_cap = BATCH_BYTES + 2 * KVQ_SCANBUFSIZE;
    if (s && !g_pin && hipHostMalloc((void **)&g_pin, (size_t)pin_cap * 2, hipHostMallocDefault) != hipSuccess) {
        kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for scanning"); g_pin = nullptr;
    }
    uint8_t *const pin = g_pin;
    // KVQ_BATCH_MB=<2..64>: smaller batches (diagnostic: many batches out of a small file); the buffers keep their size
    int64_t use_cap = pin_cap;
    if (const char *mb = getenv("KVQ_BATCH_MB")) {
        const long v = atol(mb);
        if (v >= 2 && (v << 20) < BATCH_BYTES) use_cap = ((int64_t)v << 20) + 2 * KVQ_SCANBUFSIZE;
    }
    const double tf2 = now_ms();
    if (s && pin) {
        for (int attempt = 0; attempt < 4; attempt++) {
            const int rc = findseqs_pass(s, files, nfiles, pin, pin + pin_cap, use_cap);
            if (rc != KVQ_NEED_RESCAN) break;
            // the hit arena was too small (it has been enlarged): scan again from the start
            if (kvq_scan_reset(s)) break;
            if (attempt == 3) kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for results");
        }
    }
    const double tf3 = now_ms();
    if (s && s->stream) (void)hipStreamSynchronize(s->stream);        // nothing may still be reading the host buffers
    if (g_timing) fprintf(stderr, "findseqs: table+scan %.1f  pinned alloc %.1f  passes %.1f  free %.1f ms\n", tf1 - tf0, tf2 - tf1, tf3 - tf2, now_ms() - tf3);
    g_running = 0;
    if (s) s->t = t;          // the scan owns its table: destroyed with it (kvq_findseqs_free)
    else if (t) kvq_table_destroy(t);
    return s;
}

// destroy a scan returned by kvq_findseqs together with the table it created
extern "C" void kvq_findseqs_free(kvq_scan *s)
{
    if (!s) return;
    kvq_table *t = const_cast<kvq_table *>(s->t);
    {
        // kept for the next call with the same sequences and settings (one pair; KVQ_KEEP_SCAN=0: never)
        static const bool keep = !(getenv("KVQ_KEEP_SCAN") && getenv("KVQ_KEEP_SCAN")[0] == '0');
        std::lock_guard<std::mutex> l(g_kept_lock);
        if (keep && t && !g_kept.s && s->finished) { g_kept.t = t; g_kept.s = s; (void)hipGetDevice(&g_kept.dev); return; }
    }
    kvq_scan_destroy(s);
    kvq_table_destroy(t);
}
