// kvarq_amd/csrc/kvq_findseqs.hip -- engine.findseqs (workhorse.c:1249-1464) on
// top of the scan object: a reader that produces the concatenated inflated
// stream of the input files (plain or gzip, workhorse.c:559-629), cut into the
// chunks fastq_read would hand out (workhorse.c:737-956), fed batch by batch
// through pinned host buffers to the GPU; live stats and cooperative stop
// (workhorse.c:1205-1244, 1469-1479).
#include "kvq_host.h"

#include <atomic>
#include <mutex>
#include <string.h>
#include <zlib.h>
#include <thread>
#include <unistd.h>

int64_t kvq_tail_record(const uint8_t *buf, int64_t n);
int kvq_scan_finish_internal(kvq_scan *s);

// ---------------------------------------------------------------------------
// live state shared with engine.stats()/engine.stop()
// ---------------------------------------------------------------------------

static std::atomic<int> g_running{0}, g_stop{0}, g_sigints{0};
static std::mutex g_live_lock;
static struct {
    int64_t records = 0, parsed = 0, total = 0, longest = -1;
    int32_t nseq = 0;
    std::vector<int64_t> readlengths = std::vector<int64_t>(KVQ_MAX_READLENGTH, 0), nseqhits, nseqbasehits;
} g_live;

extern "C" void kvq_request_stop(void) { g_stop++; }
extern "C" void kvq_count_sigint(void) { g_sigints++; }

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

class StreamSource {
public:
    ~StreamSource() { close_file(); free(inbuf_); }

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
        consumed_ = 0; file_done_ = false;
        fseek(fd_, 0, SEEK_END); file_size_ = ftell(fd_); fseek(fd_, 0, SEEK_SET);
        gz_ = name.size() >= 3 && name.compare(name.size() - 3, 3, ".gz") == 0;     // by suffix (582)
        bgzf_ = false;
        if (gz_) {
            // a file of BGZF blocks (bgzip: gzip members of at most 64 KiB that carry their own size)
            // is inflated by `nthreads` workers, block by block; anything else by the serial path below,
            // which also takes over should a later member not be a BGZF block
            BgzfBlock first;
            const char *sw = getenv("KVQ_BGZF");                              // KVQ_BGZF=0: serial reader only (diagnostic)
            if (!(sw && sw[0] == '0') && bgzf_peek(0, &first)) { bgzf_ = true; boff_ = 0; total_ *= 3; return KVQ_OK; }
            memset(&zs_, 0, sizeof(zs_));
            if (inflateInit2(&zs_, -MAX_WBITS) != Z_OK) { kvq_set_error(KVQ_ERR_RUNTIME, "cannot mz_inflateInit()"); return KVQ_ERR_RUNTIME; }
            zs_live_ = true;
            if (!inbuf_) inbuf_ = (uint8_t *)malloc(KVQ_SCANBUFSIZE);
            if (!inbuf_) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate inbuf"); return KVQ_ERR_MEMORY; }
            fseek(fd_, 0, SEEK_END); remaining_ = ftell(fd_); fseek(fd_, 0, SEEK_SET);
            const char *msg = skip_gz_header(0);
            if (msg) { kvq_set_error(KVQ_ERR_IO, "no valid gzip header found at beginning of file : %s", msg); return KVQ_ERR_IO; }
            remaining_ -= consumed_;
            total_ *= 3;                                                              // "random guess" (625)
        }
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
            zs_.next_out = dst; zs_.avail_out = (uInt)cap;
            bool done = false;
            while (zs_.avail_out > 0 && !done) {
                if (zs_.avail_in == 0) {
                    if (remaining_ <= 0) { done = true; break; }
                    const int64_t m = std::min<int64_t>(KVQ_SCANBUFSIZE, remaining_);
                    if ((int64_t)fread(inbuf_, 1, (size_t)m, fd_) != m) {
                        kvq_set_error(KVQ_ERR_IO, "could not read enough bytes from .fastq.gz%s%s", ferror(fd_) ? " : I/O error" : "", feof(fd_) ? " : premature EOF" : "");
                        return -1;
                    }
                    consumed_ += m; remaining_ -= m;
                    zs_.next_in = inbuf_; zs_.avail_in = (uInt)m;
                }
                const int st = inflate(&zs_, Z_SYNC_FLUSH);
                if (st != Z_OK && st != Z_STREAM_END && st != Z_BUF_ERROR) {
                    kvq_set_error(KVQ_ERR_IO, "error while inflating compressed data : status=%d fpos=%ld", st, (long)(fpos_ + (cap - zs_.avail_out)));
                    return -1;
                }
                if (st == Z_STREAM_END) {
                    // another gzip member follows when more than a trailer is left (842-866)
                    if (remaining_ + (int64_t)zs_.avail_in > 10) {
                        fseek(fd_, -(long)zs_.avail_in, SEEK_CUR);
                        consumed_ -= zs_.avail_in; remaining_ += zs_.avail_in; zs_.avail_in = 0;
                        const int64_t before = consumed_;
                        const char *msg = skip_gz_header(10);
                        if (msg) { remaining_ = 0; done = true; }
                        else {
                            remaining_ -= consumed_ - before;
                            uint8_t *no = zs_.next_out; const uInt ao = zs_.avail_out;
                            inflateEnd(&zs_); memset(&zs_, 0, sizeof(zs_)); inflateInit2(&zs_, -MAX_WBITS);
                            zs_.next_out = no; zs_.avail_out = ao;
                        }
                    } else done = true;
                } else if (st == Z_BUF_ERROR && zs_.avail_in == 0 && remaining_ <= 0) done = true;
            }
            n = cap - zs_.avail_out;
            if (zs_.avail_out > 0) { *eof = true; file_done_ = true; }
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
            memset(&zs_, 0, sizeof(zs_));
            if (inflateInit2(&zs_, -MAX_WBITS) != Z_OK) { kvq_set_error(KVQ_ERR_RUNTIME, "cannot mz_inflateInit()"); return -1; }
            zs_live_ = true;
            if (!inbuf_) inbuf_ = (uint8_t *)malloc(KVQ_SCANBUFSIZE);
            if (!inbuf_) { kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate inbuf"); return -1; }
            fseek(fd_, (long)boff_, SEEK_SET);
            remaining_ = file_size_ - boff_;
            const int64_t before = consumed_;
            const char *msg = skip_gz_header(10);
            if (msg) { *eof = true; file_done_ = true; return 0; }          // as behind any member: no further header, the stream ends (851-853)
            remaining_ -= consumed_ - before;
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

    int getc_counted() { const int c = fgetc(fd_); if (c != EOF) consumed_++; return c; }

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

    void close_file()
    {
        if (fd_) { ftell0_ += consumed_; fclose(fd_); fd_ = nullptr; }
        if (zs_live_) { inflateEnd(&zs_); zs_live_ = false; }
    }

    std::vector<std::string> files_; size_t next_ = 0;
    FILE *fd_ = nullptr; bool gz_ = false, file_done_ = true;
    z_stream zs_; bool zs_live_ = false; uint8_t *inbuf_ = nullptr; int64_t remaining_ = 0;
    bool bgzf_ = false; int64_t boff_ = 0; std::vector<uint8_t> cbuf_;       // BGZF: next block's file offset, compressed run
    int64_t size_ = 0, ftell0_ = 0, consumed_ = 0, fpos_ = 0, total_ = 0, file_size_ = 0;
};

// ---------------------------------------------------------------------------
// driver
// ---------------------------------------------------------------------------

static const int64_t BATCH_BYTES = 64ll << 20;     // new stream bytes per batch

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

    while (src.has_next_file() && !g_stop.load()) {
        if ((rc = src.open_next())) return rc;
        // chunker state of this file, offsets relative to pin[0]
        int64_t have = 0;              // bytes of the file's stream sitting in pin
        int64_t pin_fpos = src.fpos(); // stream offset of pin[0]
        int64_t cs = 0, fill = 0;      // current chunk start / how far the reference has read (== cs + leftover)
        bool eof = false;
        while (!g_stop.load()) {
            // top up
            while (!eof && have < pin_cap) {
                const int64_t n = src.read(pin + have, pin_cap - have, &eof);
                if (n < 0) return kvq_error_code();
                have += n;
                if (n == 0 && !eof) break;
            }
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
            if (!off.empty()) {
                off.push_back(batch_end);
                for (auto &o : off) o -= batch_begin;
                rc = sink.batch(pin + batch_begin, batch_end - batch_begin, off.data(), (int64_t)off.size() - 1,
                                pin_fpos + batch_begin, src.fpos(), src.total());
                if (rc) return rc;
            }
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
        }
    }
    *parsed = src.fpos(); *total = src.total();
    return KVQ_OK;
}

// the GPU sink: one kvq_scan_host per batch, live stats after each
struct ScanSink {
    kvq_scan *s; std::vector<int64_t> ctr_live; bool have_live = false; int64_t live_parsed = 0;
    void begin(int64_t total) { live_reset(s->t->nseq, total); ctr_live.resize((size_t)s->t->ctr_len); }
    int batch(const uint8_t *data, int64_t nbytes, const int64_t *off, int64_t nchunks, int64_t fpos, int64_t parsed, int64_t total)
    {
        // the batch handed over last has been read from its host buffer and scanned by now or soon:
        // wait for it, publish its counters (engine.stats() may be polling), then enqueue this one and
        // return, so that the reader fills the other host buffer while this one is copied and scanned
        int rc = kvq_scan_host_drain(s);
        if (rc) return rc;
        if (have_live) {
            if (hipMemcpy(ctr_live.data(), s->d_ctr, (size_t)s->t->ctr_len * 8, hipMemcpyDeviceToHost) != hipSuccess) {
                kvq_set_error(KVQ_ERR_DEVICE, "device failure during scan"); return KVQ_ERR_DEVICE;
            }
            live_from_counters(s->t, ctr_live.data(), live_parsed, total);
        }
        have_live = true; live_parsed = parsed;
        return kvq_scan_host_async(s, data, nbytes, off, nchunks, fpos);
    }
};

// one pass over the files with the current arena; KVQ_NEED_RESCAN asks for another
static int findseqs_pass(kvq_scan *s, const char *const *files, int nfiles, uint8_t *pin, int64_t pin_cap)
{
    ScanSink sink; sink.s = s;
    int64_t parsed = 0, total = 0;
    const double tp0 = now_ms();
    int rc = stream_batches(sink, files, nfiles, pin, pin_cap, &parsed, &total, pin + pin_cap);     // two host buffers
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
    kvq_table *t = kvq_table_create(seqs, seqlens, nseq, nullptr);
    kvq_scan *s = t ? kvq_scan_create(t, nullptr) : nullptr;
    const double tf1 = now_ms();
    // the two pinned host buffers outlive the call: pinning and unpinning 130 MB costs more than
    // streaming a 1 GB file through them (only one findseqs runs at a time, g_running)
    static uint8_t *g_pin = nullptr;
    const int64_t pin_cap = BATCH_BYTES + 2 * KVQ_SCANBUFSIZE;
    if (s && !g_pin && hipHostMalloc((void **)&g_pin, (size_t)pin_cap * 2, hipHostMallocDefault) != hipSuccess) {
        kvq_set_error(KVQ_ERR_MEMORY, "cannot allocate memory for scanning"); g_pin = nullptr;
    }
    uint8_t *const pin = g_pin;
    const double tf2 = now_ms();
    if (s && pin) {
        for (int attempt = 0; attempt < 4; attempt++) {
            const int rc = findseqs_pass(s, files, nfiles, pin, pin_cap);
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
    kvq_scan_destroy(s);
    kvq_table_destroy(t);
}
