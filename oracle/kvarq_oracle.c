/*
 * oracle/kvarq_oracle.c -- TEST INFRASTRUCTURE ONLY (see kvarq_oracle.h).
 *
 * Plain-C restatement of the KvarQ engine's scan path.  Every function names
 * the reference lines (/root/reference/csrc/workhorse.c unless said
 * otherwise) whose behaviour it restates.  It is written from the behaviour,
 * not from the text: one byte-source + chunker, one record splitter, one
 * trimmer, and a matcher expressed over alignment diagonals.
 *
 * Build: make -C oracle   (gcc, zlib, pthreads)
 */
#include "kvarq_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ */
/* growing result buffers                                               */
/* ------------------------------------------------------------------ */

typedef struct piece {              /* hits of one chunk, in scan order */
    int64_t   chunk_no;
    int64_t   n, cap;
    int32_t  *seq_nr; int64_t *file_pos; int32_t *seq_pos, *length, *readlength;
    uint8_t  *blob; int64_t blob_n, blob_cap;
} piece;

static int piece_push(piece *p, int seqi, int64_t fpos, int spos, int len, int rl, const uint8_t *bytes)
{
    if (p->n == p->cap) {
        int64_t c = p->cap ? p->cap * 2 : 64;
        p->seq_nr = realloc(p->seq_nr, c * sizeof(int32_t));
        p->file_pos = realloc(p->file_pos, c * sizeof(int64_t));
        p->seq_pos = realloc(p->seq_pos, c * sizeof(int32_t));
        p->length = realloc(p->length, c * sizeof(int32_t));
        p->readlength = realloc(p->readlength, c * sizeof(int32_t));
        if (!p->seq_nr || !p->file_pos || !p->seq_pos || !p->length || !p->readlength) return -1;
        p->cap = c;
    }
    if (p->blob_n + len > p->blob_cap) {
        int64_t c = p->blob_cap ? p->blob_cap * 2 : 4096;
        while (c < p->blob_n + len) c *= 2;
        p->blob = realloc(p->blob, c);
        if (!p->blob) return -1;
        p->blob_cap = c;
    }
    p->seq_nr[p->n] = seqi; p->file_pos[p->n] = fpos; p->seq_pos[p->n] = spos;
    p->length[p->n] = len; p->readlength[p->n] = rl;
    if (len > 0) memcpy(p->blob + p->blob_n, bytes, (size_t)len);
    p->blob_n += len;
    p->n++;
    return 0;
}

static void piece_free(piece *p)
{
    free(p->seq_nr); free(p->file_pos); free(p->seq_pos); free(p->length); free(p->readlength); free(p->blob);
    memset(p, 0, sizeof(*p));
}

/* ------------------------------------------------------------------ */
/* the matcher: one trimmed read against one sequence                   */
/* ------------------------------------------------------------------ */

/* Hamming distance of a[0..n) and b[0..n), saturating just above the budget;
 * the reference's inner loops (1118-1122, 1132-1136, 1149-1153, 1165-1169)
 * stop early once e exceeds maxerrors, which does not change the verdict. */
static int within_budget(const uint8_t *a, const uint8_t *b, int n, int budget)
{
    int e = 0;
    for (int j = 0; j < n; j++) {
        if (a[j] != b[j] && ++e > budget) return 0;
    }
    return e <= budget;    /* budget < 0 never matches (e=0 > maxerrors) */
}

/*
 * All hits of read[0..rl) on sequence seq[0..seql), emitted in the reference's
 * order: class A (1112-1127), class B (1130-1141), class C (1144-1174).
 */
static int match_read(piece *out, const kvo_config *cfg, int seqi,
                      const uint8_t *read, int rl, int64_t read_fpos,
                      const uint8_t *seq, int seql)
{
    const int mo = cfg->minoverlap, me = cfg->maxerrors;
    int i;

    if (rl > mo && seql > mo) {
        /* A: tail of the read over the head of the sequence; overlap rl-i
         * grows from minoverlap while it stays <= seql-1 (1116) */
        for (i = rl - mo; i > 0 && rl - i <= seql - 1; i--)
            if (within_budget(read + i, seq, rl - i, me))
                if (piece_push(out, seqi, read_fpos, -i, rl - i, rl, read + i)) return -1;
        /* B: head of the read over the tail of the sequence; overlap seql-i
         * grows from minoverlap while it stays <= rl (1130, non-strict) */
        for (i = seql - mo; i > 0 && seql - i <= rl; i--)
            if (within_budget(seq + i, read, seql - i, me))
                if (piece_push(out, seqi, read_fpos, i, seql - i, rl, read)) return -1;
    }
    if (rl > seql) {
        /* C1: whole sequence inside the read (1147) */
        for (i = 0; i <= rl - seql; i++)
            if (within_budget(read + i, seq, seql, me))
                if (piece_push(out, seqi, read_fpos, -i, seql, rl, read + i)) return -1;
    } else {
        /* C2: whole read inside the sequence (1163) */
        for (i = 0; i <= seql - rl; i++)
            if (within_budget(seq + i, read, rl, me))
                if (piece_push(out, seqi, read_fpos, i, rl, rl, read)) return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* one chunk: record split, sanity checks, quality trim, matching        */
/* ------------------------------------------------------------------ */

typedef struct chunk_stats {
    int64_t records;                       /* 1050, 1187 */
    int64_t rls[KVO_MAX_READLENGTH];       /* 394-402 */
    int64_t longest;
    int     err_code; char errmsg[256];
} chunk_stats;

/* longest run of score bytes >= Amin on the 4th line (1055-1068): the scan
 * covers the line including its terminating '\n'; a run is only recorded
 * when a byte below Amin closes it; the first of equally long runs wins.
 * chars are compared as signed char, like the reference's plain `char`. */
static int trim_quality(const uint8_t *score, int8_t amin, int *start)
{
    int best = 0, beststart = 0;
    const uint8_t *p = score, *run = score;
    for (;;) {
        int8_t c = (int8_t)*p;
        if (c >= amin) {
            if (!run) run = p;
        } else if (run) {
            if ((int)(p - run) > best) { best = (int)(p - run); beststart = (int)(run - score); }
            run = NULL;
        }
        if (*p == '\n') break;
        p++;
    }
    *start = beststart;   /* with best==0 the reference leaves this undefined (1070) */
    return best;
}

static int scan_chunk(const uint8_t *buf, int64_t n, int64_t fpos,
                      const uint8_t *const *seqs, const int32_t *seqlens, int nseq,
                      const kvo_config *cfg, piece *out, chunk_stats *st)
{
    int64_t at = 0;
    while (at < n) {
        /* 1018-1034: a record is the next four '\n'; fewer => partial, dropped */
        int64_t nl[4]; int lines = 0; int64_t p = at;
        while (lines < 4 && p < n) { if (buf[p] == '\n') nl[lines++] = p; p++; }
        if (lines < 4) break;
        const int64_t rstart = at, sread = nl[0] + 1, plus = nl[1] + 1, sscore = nl[2] + 1;
        if (buf[rstart] != '@') {                                    /* 1037-1042 */
            st->err_code = KVO_ERR_FORMAT;
            snprintf(st->errmsg, sizeof(st->errmsg),
                     "record must start with '@' (and not '%c') fpos=%ld", buf[rstart], (long)(fpos + rstart));
            return 1;
        }
        if (buf[plus] != '+') {                                      /* 1043-1048 */
            st->err_code = KVO_ERR_FORMAT;
            snprintf(st->errmsg, sizeof(st->errmsg),
                     "3rd line of record must start with '+' fpos=%ld", (long)(fpos + plus));
            return 1;
        }
        st->records++;
        at = p;

        int off = 0;
        const int rl = trim_quality(buf + sscore, cfg->Amin, &off);
        if (rl >= 0 && rl < KVO_MAX_READLENGTH) st->rls[rl]++;       /* 394-402, before the length gate */
        if (rl > st->longest) st->longest = rl;
        if (rl < cfg->minreadlength) continue;                       /* 1100 */

        const uint8_t *read = buf + sread + off;                     /* 1070 */
        const int64_t read_fpos = fpos + sread + off;
        for (int s = 0; s < nseq; s++)
            if (match_read(out, cfg, s, read, rl, read_fpos, seqs[s], seqlens[s])) {
                st->err_code = KVO_ERR_MEMORY;
                snprintf(st->errmsg, sizeof(st->errmsg), "cannot allocate memory for results");
                return 1;
            }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* fastq_rewind (696-718)                                               */
/* ------------------------------------------------------------------ */

/* length of the trailing partial record: walk back from the end, first past
 * a line that starts with '+', then to the next line start that is '@'.
 * A line start is a byte preceded by '\n' or '\r'.  -1 if none is found. */
static int64_t tail_record_length(const uint8_t *buf, int64_t n)
{
    int seen_plus = 0;
    for (int64_t i = 1; i + 1 < n; i++) {
        const uint8_t c = buf[n - i], before = buf[n - i - 1];
        const int linestart = (before == '\n' || before == '\r');
        if (c == '+' && linestart) seen_plus = 1;
        else if (seen_plus && c == '@' && linestart) return i;
    }
    return -1;
}

int64_t kvo_chunk_offsets(const uint8_t *data, int64_t nbytes, int64_t *offsets, int64_t cap)
{
    /* what fastq_read (737-956) hands out for a plain stream of nbytes: fill
     * the 1 MiB buffer behind the leftover; if the source did not run dry,
     * cut the trailing partial record and carry it over. */
    int64_t nch = 0, start = 0, filled_to = 0;   /* [start, filled_to) = leftover */
    while (1) {
        int64_t want = KVO_SCANBUFSIZE - (filled_to - start);
        int64_t got = nbytes - filled_to < want ? nbytes - filled_to : want;
        int hit_eof = (nbytes - filled_to) < want;   /* short fread => feof (901) */
        int64_t end = filled_to + got;
        if (got == 0) {                               /* 905-910 */
            if (end > start) { if (nch < cap) offsets[nch] = start; nch++; }
            start = end;
            break;
        }
        int64_t keep = 0;
        if (!hit_eof) {
            keep = tail_record_length(data + start, end - start);
            if (keep < 0) return -1;
        }
        if (nch < cap) offsets[nch] = start;
        nch++;
        start = end - keep; filled_to = end;
        if (hit_eof) { start = end; break; }
    }
    if (nch < cap) offsets[nch] = nbytes;
    (void)start;
    return nch;
}

/* ------------------------------------------------------------------ */
/* byte source over a list of files, plain or gzip (559-629, 482-541)    */
/* ------------------------------------------------------------------ */

typedef struct source {
    const char *const *files; int nfiles, next;
    FILE *fd; int compressed;
    z_stream zs; int zs_live; uint8_t *inbuf; int64_t remaining;
    int64_t size, ftell0, consumed;          /* consumed = ftell(fd) of the reference */
    int64_t fpos;                            /* inflated bytes handed out so far (45) */
    int eof;
    uint8_t *left; int64_t left_n;
    int64_t parsed, total;                   /* fastq_parsed, fastq_size_estimated */
    int err_code; char errmsg[1024];
} source;

static int gz_getc(source *s) { int c = fgetc(s->fd); if (c != EOF) s->consumed++; return c; }

/* 482-541: find the magic within `dist` stray bytes, check method/flags, skip
 * the optional fields.  Flag bit 1 is treated as "continuation" like the
 * reference does. */
static const char *gz_skip_header(source *s, int dist)
{
    int state = 0, c, y = 0;
    for (c = gz_getc(s); state != 2 && y <= dist && c != EOF; c = gz_getc(s)) {
        if (c == 0x1F && state == 0) state = 1;
        else if (c == 0x8B && state == 1) state = 2;
        else { state = 0; y++; }
    }
    if (state != 2) return "magic bytes not found";
    if (c != 8) return "expected method==DEFLATED";
    int flags = gz_getc(s);
    if (flags & (0x02 | 0x20 | 0xC0)) return "unsupported flags (CONTINUATION or ENCRYPTED or RESERVED)";
    for (int i = 0; i < 6; i++) (void)gz_getc(s);
    if (flags & 0x04) { int n = gz_getc(s); n |= gz_getc(s) << 8; while (n-- > 0) (void)gz_getc(s); }
    if (flags & 0x08) { do c = gz_getc(s); while (c > 0); }
    if (flags & 0x10) { do c = gz_getc(s); while (c > 0); }
    return NULL;
}

static int source_open_next(source *s)
{
    if (s->fd) { s->ftell0 += s->consumed; fclose(s->fd); s->fd = NULL; }
    if (s->zs_live) { inflateEnd(&s->zs); s->zs_live = 0; }
    const char *fname = s->files[s->next++];
    s->consumed = 0;
    s->fd = fopen(fname, "rb");
    if (!s->fd) { s->err_code = KVO_ERR_IO; snprintf(s->errmsg, sizeof(s->errmsg), "cannot open file"); return -1; }
    size_t L = strlen(fname);
    s->compressed = 0;
    if (L >= 3 && strcmp(fname + L - 3, ".gz") == 0) {               /* 582: by suffix */
        s->compressed = 1;
        memset(&s->zs, 0, sizeof(s->zs));
        if (inflateInit2(&s->zs, -MAX_WBITS) != Z_OK) {
            s->err_code = KVO_ERR_RUNTIME; snprintf(s->errmsg, sizeof(s->errmsg), "cannot mz_inflateInit()"); return -1;
        }
        s->zs_live = 1;
        if (!s->inbuf) s->inbuf = malloc(KVO_SCANBUFSIZE);
        if (!s->inbuf) { s->err_code = KVO_ERR_MEMORY; snprintf(s->errmsg, sizeof(s->errmsg), "cannot allocate inbuf"); return -1; }
        fseek(s->fd, 0, SEEK_END); s->remaining = ftell(s->fd); fseek(s->fd, 0, SEEK_SET);
        const char *msg = gz_skip_header(s, 0);
        if (msg) {
            s->err_code = KVO_ERR_IO;
            snprintf(s->errmsg, sizeof(s->errmsg), "no valid gzip header found at beginning of file : %s", msg);
            return -1;
        }
        s->remaining -= s->consumed;
        s->total *= 3;                                               /* 625: "random guess" */
    }
    return 0;
}

static int source_open(source *s, const char *const *files, int nfiles)
{
    memset(s, 0, sizeof(*s));
    s->files = files; s->nfiles = nfiles;
    for (int i = 0; i < nfiles; i++) {                               /* 659-673 */
        FILE *fd = fopen(files[i], "rb");
        if (!fd) {
            s->err_code = KVO_ERR_IO;
            snprintf(s->errmsg, sizeof(s->errmsg), "cannot open file '%s' for getting filesize", files[i]);
            return -1;
        }
        fseek(fd, 0, SEEK_END); s->size += ftell(fd); fclose(fd);
    }
    s->parsed = 0; s->total = s->size;
    if (nfiles == 0) { s->eof = 1; return 0; }
    return source_open_next(s);
}

static void source_close(source *s)
{
    if (s->fd) fclose(s->fd);
    if (s->zs_live) inflateEnd(&s->zs);
    free(s->inbuf); free(s->left);
}

/* fastq_read (737-956): returns bytes placed in buf (0 = end), -1 on error;
 * *fposp = offset of buf[0] in the concatenated inflated stream. */
static int64_t source_read(source *s, uint8_t *buf, int64_t cap, int64_t *fposp)
{
    if (!s->fd) return 0;
    if (s->eof) {                                                    /* 749-753 */
        if (s->next < s->nfiles) { if (source_open_next(s)) return -1; }
    }
    int64_t left = 0;
    if (s->left_n > 0) {                                             /* 756-774 */
        if (s->left_n > cap) { s->err_code = KVO_ERR_RUNTIME; snprintf(s->errmsg, sizeof(s->errmsg), "buf_size < fastq->buf_size !"); return -1; }
        left = s->left_n; memcpy(buf, s->left, (size_t)left); free(s->left); s->left = NULL; s->left_n = 0;
    }
    *fposp = s->fpos - left;
    int64_t n = 0;
    s->eof = 0;

    if (s->compressed) {                                             /* 781-885 */
        s->zs.next_out = buf + left; s->zs.avail_out = (uInt)(cap - left);
        int stream_done = 0;
        while (s->zs.avail_out > 0 && !stream_done) {
            if (s->zs.avail_in == 0) {
                if (s->remaining <= 0) { stream_done = 1; break; }
                int64_t m = s->remaining < KVO_SCANBUFSIZE ? s->remaining : KVO_SCANBUFSIZE;
                if ((int64_t)fread(s->inbuf, 1, (size_t)m, s->fd) != m) {
                    s->err_code = KVO_ERR_IO; snprintf(s->errmsg, sizeof(s->errmsg), "could not read enough bytes from .fastq.gz"); return -1;
                }
                s->consumed += m; s->remaining -= m;
                s->zs.next_in = s->inbuf; s->zs.avail_in = (uInt)m;
            }
            uInt before = s->zs.avail_out;
            int status = inflate(&s->zs, Z_SYNC_FLUSH);
            n += before - s->zs.avail_out;
            if (status != Z_OK && status != Z_STREAM_END && status != Z_BUF_ERROR) {
                s->err_code = KVO_ERR_IO;
                snprintf(s->errmsg, sizeof(s->errmsg), "error while inflating compressed data : status=%d fpos=%ld", status, (long)s->fpos);
                return -1;
            }
            if (status == Z_STREAM_END) {
                /* 842-866: another member follows if more than the trailer remains */
                if (s->remaining + (int64_t)s->zs.avail_in > 10) {
                    fseek(s->fd, -((long)s->zs.avail_in), SEEK_CUR);
                    s->consumed -= s->zs.avail_in; s->remaining += s->zs.avail_in;
                    int64_t pos = s->consumed;
                    const char *msg = gz_skip_header(s, 10);
                    if (msg) { s->remaining = 0; s->zs.avail_in = 0; stream_done = 1; }
                    else {
                        s->remaining -= s->consumed - pos;
                        inflateEnd(&s->zs); memset(&s->zs, 0, sizeof(s->zs)); inflateInit2(&s->zs, -MAX_WBITS);
                        s->zs.next_out = buf + left + n; s->zs.avail_out = (uInt)(cap - left - n);
                        s->zs.avail_in = 0;
                    }
                } else stream_done = 1;
            }
        }
        if (s->zs.avail_out > 0) s->eof = 1;                         /* 879-880 */
        /* 883-884: float arithmetic, as the reference computes it */
        if (s->ftell0 + s->consumed > 0)
            s->total = (int64_t)(size_t)((float)s->size * (s->fpos + n) / (s->ftell0 + s->consumed));
    } else {                                                         /* 886-903 */
        n = (int64_t)fread(buf + left, 1, (size_t)(cap - left), s->fd);
        if (ferror(s->fd)) { s->err_code = KVO_ERR_IO; snprintf(s->errmsg, sizeof(s->errmsg), "error while reading from file in fastq_read"); return -1; }
        if (feof(s->fd)) s->eof = 1;
    }
    if (n == 0) return left;                                         /* 905-910 */
    s->fpos += n; s->parsed += n;
    int64_t keep = 0;
    if (!s->eof) {                                                   /* 916-943 */
        keep = tail_record_length(buf, left + n);
        if (keep < 0) {
            s->err_code = KVO_ERR_RUNTIME;
            snprintf(s->errmsg, sizeof(s->errmsg), "could find beginning of record; read %ld bytes up to %ld", (long)n, (long)s->consumed);
            return -1;
        }
        if (keep > 0) {
            s->left = malloc((size_t)keep);
            if (!s->left) { s->err_code = KVO_ERR_MEMORY; snprintf(s->errmsg, sizeof(s->errmsg), "cannot allocate new fastq->buf"); return -1; }
            memcpy(s->left, buf + left + n - keep, (size_t)keep); s->left_n = keep;
        }
    }
    return left + n - keep;
}

/* ------------------------------------------------------------------ */
/* driver: N workers pulling chunks (976-1197, 1375-1406)               */
/* ------------------------------------------------------------------ */

typedef struct job {
    /* chunk producer: either a file source or an in-memory stream */
    source *src;
    const uint8_t *mem; int64_t mem_n, mem_base; const int64_t *mem_off; int64_t mem_nch, mem_next;
    pthread_mutex_t lock;
    int64_t chunk_counter;
    /* inputs */
    const uint8_t *const *seqs; const int32_t *seqlens; int nseq; const kvo_config *cfg;
    /* outputs */
    piece *pieces; int64_t npieces, cappieces;
    int64_t records, rls[KVO_MAX_READLENGTH], longest;
    int err_code; int64_t err_chunk; char errmsg[1024];
    int stop;
} job;

static void *worker(void *arg)
{
    job *j = (job *)arg;
    uint8_t *own = j->src ? malloc(KVO_SCANBUFSIZE) : NULL;
    chunk_stats *st = malloc(sizeof(chunk_stats));
    for (;;) {
        const uint8_t *buf; int64_t n, fpos, no;
        pthread_mutex_lock(&j->lock);
        if (j->stop) { pthread_mutex_unlock(&j->lock); break; }
        if (j->src) {
            n = source_read(j->src, own, KVO_SCANBUFSIZE, &fpos);
            if (n < 0) {
                if (!j->err_code) { j->err_code = j->src->err_code; j->err_chunk = j->chunk_counter; memcpy(j->errmsg, j->src->errmsg, sizeof(j->errmsg)); }
                j->stop = 1; pthread_mutex_unlock(&j->lock); break;
            }
            buf = own;
        } else {
            if (j->mem_next >= j->mem_nch) n = 0;
            else { int64_t a = j->mem_off[j->mem_next], b = j->mem_off[j->mem_next + 1]; buf = j->mem + a; n = b - a; fpos = j->mem_base + a; j->mem_next++; }
        }
        if (n <= 0) { pthread_mutex_unlock(&j->lock); break; }
        no = j->chunk_counter++;
        pthread_mutex_unlock(&j->lock);

        memset(st, 0, sizeof(*st)); st->longest = -1;
        piece pc; memset(&pc, 0, sizeof(pc)); pc.chunk_no = no;
        int bad = scan_chunk(buf, n, fpos, j->seqs, j->seqlens, j->nseq, j->cfg, &pc, st);

        pthread_mutex_lock(&j->lock);
        if (bad) {
            /* keep the error of the earliest chunk: what one worker would hit first */
            if (!j->err_code || no < j->err_chunk) { j->err_code = st->err_code; j->err_chunk = no; snprintf(j->errmsg, sizeof(j->errmsg), "%s", st->errmsg); }
            j->stop = 1;
        }
        j->records += st->records;                                    /* 1187 */
        for (int k = 0; k < KVO_MAX_READLENGTH; k++) j->rls[k] += st->rls[k];
        if (st->longest > j->longest) j->longest = st->longest;
        if (j->npieces == j->cappieces) {
            j->cappieces = j->cappieces ? j->cappieces * 2 : 64;
            j->pieces = realloc(j->pieces, (size_t)j->cappieces * sizeof(piece));
        }
        j->pieces[j->npieces++] = pc;
        pthread_mutex_unlock(&j->lock);
    }
    free(own); free(st);
    return NULL;
}

static int by_chunk(const void *a, const void *b)
{
    int64_t x = ((const piece *)a)->chunk_no, y = ((const piece *)b)->chunk_no;
    return x < y ? -1 : x > y;
}

static kvo_result *run_job(job *j)
{
    kvo_result *r = calloc(1, sizeof(kvo_result));
    if (!r) return NULL;
    r->nseq = j->nseq;
    r->nseqhits = calloc((size_t)(j->nseq > 0 ? j->nseq : 1), sizeof(int64_t));
    r->nseqbasehits = calloc((size_t)(j->nseq > 0 ? j->nseq : 1), sizeof(int64_t));
    r->rls_longest = -1;
    j->longest = -1;
    pthread_mutex_init(&j->lock, NULL);

    int nt = j->cfg->nthreads > 0 ? j->cfg->nthreads : 1;
    pthread_t *th = malloc(sizeof(pthread_t) * (size_t)nt);
    int started = 0;
    for (int t = 0; t < nt; t++) { if (pthread_create(&th[t], NULL, worker, j) == 0) started++; else break; }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    if (started == 0) { r->err_code = KVO_ERR_RUNTIME; snprintf(r->errmsg, sizeof(r->errmsg), "pthread_create failed"); }

    /* canonical order = chunk order, then scan order inside the chunk */
    qsort(j->pieces, (size_t)j->npieces, sizeof(piece), by_chunk);
    int64_t nh = 0, nb = 0;
    for (int64_t k = 0; k < j->npieces; k++) { nh += j->pieces[k].n; nb += j->pieces[k].blob_n; }
    r->n_hits = nh; r->cap_hits = nh; r->cap_blob = nb;
    r->seq_nr = malloc((size_t)(nh + 1) * sizeof(int32_t)); r->file_pos = malloc((size_t)(nh + 1) * sizeof(int64_t));
    r->seq_pos = malloc((size_t)(nh + 1) * sizeof(int32_t)); r->length = malloc((size_t)(nh + 1) * sizeof(int32_t));
    r->readlength = malloc((size_t)(nh + 1) * sizeof(int32_t));
    r->hitseq_blob = malloc((size_t)(nb + 1)); r->hitseq_off = malloc((size_t)(nh + 1) * sizeof(int64_t));
    int64_t h = 0, b = 0;
    for (int64_t k = 0; k < j->npieces; k++) {
        piece *p = &j->pieces[k];
        int64_t pb = 0;
        for (int64_t q = 0; q < p->n; q++, h++) {
            r->seq_nr[h] = p->seq_nr[q]; r->file_pos[h] = p->file_pos[q]; r->seq_pos[h] = p->seq_pos[q];
            r->length[h] = p->length[q]; r->readlength[h] = p->readlength[q];
            r->hitseq_off[h] = b;
            memcpy(r->hitseq_blob + b, p->blob + pb, (size_t)p->length[q]);
            b += p->length[q]; pb += p->length[q];
            r->nseqhits[p->seq_nr[q]]++;                              /* 434-435 */
            r->nseqbasehits[p->seq_nr[q]] += p->length[q];
        }
        piece_free(p);
    }
    r->hitseq_off[nh] = b;
    free(j->pieces);
    memcpy(r->readlengths, j->rls, sizeof(r->readlengths));
    r->rls_longest = j->longest;
    r->records_parsed = j->records;
    if (j->err_code && !r->err_code) { r->err_code = j->err_code; memcpy(r->errmsg, j->errmsg, sizeof(r->errmsg)); }
    pthread_mutex_destroy(&j->lock);
    return r;
}

kvo_result *kvo_findseqs(const char *const *files, int nfiles,
                         const uint8_t *const *seqs, const int32_t *seqlens, int nseq,
                         const kvo_config *cfg)
{
    source src;
    job j; memset(&j, 0, sizeof(j));
    j.seqs = seqs; j.seqlens = seqlens; j.nseq = nseq; j.cfg = cfg;
    if (source_open(&src, files, nfiles)) {
        kvo_result *r = calloc(1, sizeof(kvo_result));
        if (!r) return NULL;
        r->rls_longest = -1; r->nseq = nseq;
        r->err_code = src.err_code; memcpy(r->errmsg, src.errmsg, sizeof(r->errmsg));
        source_close(&src);
        return r;
    }
    j.src = &src;
    kvo_result *r = run_job(&j);
    if (r) { r->parsed = src.parsed; r->total = src.total; }
    source_close(&src);
    return r;
}

kvo_result *kvo_scan_memory(const uint8_t *data, int64_t nbytes, int64_t fpos_base,
                            const uint8_t *const *seqs, const int32_t *seqlens, int nseq,
                            const kvo_config *cfg)
{
    int64_t cap = nbytes / (KVO_SCANBUFSIZE / 2) + 4;
    int64_t *off = malloc((size_t)(cap + 1) * sizeof(int64_t));
    if (!off) return NULL;
    int64_t nch = kvo_chunk_offsets(data, nbytes, off, cap);
    job j; memset(&j, 0, sizeof(j));
    j.seqs = seqs; j.seqlens = seqlens; j.nseq = nseq; j.cfg = cfg;
    j.mem = data; j.mem_n = nbytes; j.mem_base = fpos_base; j.mem_off = off; j.mem_nch = nch < 0 ? 0 : nch;
    kvo_result *r = run_job(&j);
    if (r) {
        r->parsed = nbytes; r->total = nbytes;
        if (nch < 0 && !r->err_code) { r->err_code = KVO_ERR_RUNTIME; snprintf(r->errmsg, sizeof(r->errmsg), "could find beginning of record"); }
    }
    free(off);
    return r;
}

void kvo_free(kvo_result *r)
{
    if (!r) return;
    free(r->seq_nr); free(r->file_pos); free(r->seq_pos); free(r->length); free(r->readlength);
    free(r->hitseq_blob); free(r->hitseq_off); free(r->nseqhits); free(r->nseqbasehits);
    free(r);
}

/* ------------------------------------------------------------------ */
/* Coverage.apply_hit (kvarq/analyse.py:57-78)                          */
/* ------------------------------------------------------------------ */

static int base_class(uint8_t c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4; default: return 5; }
}

void kvo_fold_coverage(const kvo_result *r, const uint8_t *const *seqs, const int32_t *seqlens,
                       const int64_t *off, int64_t *cov, int64_t *mut)
{
    (void)seqlens;
    for (int64_t h = 0; h < r->n_hits; h++) {
        const int s = r->seq_nr[h];
        const int start = r->seq_pos[h] > 0 ? r->seq_pos[h] : 0;     /* analyse.py:70 */
        const uint8_t *hs = r->hitseq_blob + r->hitseq_off[h];
        for (int i = 0; i < r->length[h]; i++) {
            const int64_t at = off[s] + start + i;
            cov[at]++;
            if (hs[i] != seqs[s][start + i]) mut[at * 6 + base_class(hs[i])]++;
        }
    }
}
