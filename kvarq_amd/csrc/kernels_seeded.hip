// kvarq_amd/csrc/kernels_seeded.hip -- what the fused seed-filter scan (kvq_scan_bp, kernels_bp.hip) stands on: the seed
// index of a sequence table, the tile geometry and the tile reports, small device helpers, and the kernels around the
// scan kernel (kvq_expand_tiles in front of it, kvq_validate_tiles behind it).  Round 1's text-in-LDS kernel, which
// this file was written for, is gone (KVQ_KERNEL=v1 no longer exists).
//
// Seeding (pigeonhole, K = 8): an accepted alignment of length L >= (e+1)*K
// with <= e mismatches has an exact K-mer block.  Alignments that start at the
// read head (classes B, C "read in sequence") or end at the read tail (class A)
// are found from 2(e+1) fixed read blocks in the index of ALL sequence
// positions; a sequence contained in the read (class C "sequence in read") is
// found from its e+1 ANCHOR blocks at any read position.  A diagonal found by
// several seeds is emitted by the first live seed only (canonical order).
#include "kvq_host.h"

#include <algorithm>
#include <string.h>

#ifdef KVQ_NO_PRIO
#define KVQ_SETPRIO(n) ((void)0)
#else
#define KVQ_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
#define KVQ_K_MAX 8                // the longest seed (16 code bits; an 8 KiB bitmap of the anchors in LDS); the shortest is 5
#define ST_TILE 36640u             // bytes a tile owns at the full look-ahead: 458 scan blocks of 80 bytes (tile + look-ahead = 510 blocks)
#define ST_TILE_MIN 30960u         // the least a tile owns (kvq_choose_tile cuts a tile to a whole number of lane groups)
#define ST_OV 4160u                // look-ahead for the tile's last record (52 blocks)
#define ST_PRE 80u                 // one (zeroed) block in front of the tile: buf[ST_PRE] = first owned byte
#define ST_THREADS 512
#define ST_WAVES (ST_THREADS / 64)
#define ST_ROUNDS ((ST_TILE + ST_OV) / 16u / ST_THREADS + 1)    // 16-byte loads per thread
#define ST_NLCAP 2048
#define ST_RCAP 512
#define ST_QCAP 896             // candidates (read, position): ST_QW per wave and stretch
#define ST_Q2CAP 2048           // candidate x index entry pairs: ST_Q2W per wave and stretch
#define ST_QW (ST_QCAP / ST_WAVES)
#define ST_Q2W (ST_Q2CAP / ST_WAVES)
#define ST_BLK 80u              // bytes one thread scans for newlines (LDS conflict-free stride)
#define ST_BUF (ST_PRE + ST_TILE + ST_OV)
#define ST_HIST_TILES 100u      // x ST_RCAP reads per tile < 65536: the 16-bit histogram bins cannot run over

// tile report word for kvq_validate_tiles
#define TR_NONE 0xFFu              // no record starts in this tile
#define TR_FLAG_FALLBACK 0x2000000u   // the tile could not do its work (one read flooded a wave's queues): the batch is scanned again, exhaustively
#define TR_FLAG_SKIPPED 0x4000000u    // the tile has left records alone (the last ones, when a record is longer than the look-ahead; all
                                      // of them, when it holds more newlines or records than its tables): those are scanned again (kvq_collect_skipped)
#define TR_FLAG_PARTIAL 0x8000000u    // ... but it did scan the records in front of those (so its speculated first record counts, and is checked)
#define KVQ_SKIP_CAP 1024u            // skipped tiles a batch may have before it is scanned again as a whole

// One index entry (16 bytes), the same for both indexes: which sequence position carries the 8-mer, and the 32 sequence bases
// around it as 2-bit codes -- whatever 16-base window of the diagonal the kernel picks for its first test lies inside them, so
// a work item is dismissed without a third dependent load (round 3: code -> CSR start -> entry -> 2-bit table).
//   en  = position in sequence (12) | sequence number (20) | table offset of the sequence (20) | its length (12)
//   ctx = codes of sequence bases [pos - 16, pos + 16), base i of them in bits 2i, 2i + 1 (zero outside the sequence)
struct SeedEntry { uint64_t en, ctx; };
struct SeedTables {
    const uint32_t *bm1;                      // two bitmaps of 8 KiB, one bit per 8-mer code: anchor blocks, then anywhere in a sequence
    const uint32_t *start;                    // CSR starts into ent[]: slot code for the anchor index, 65536 + code for the index of all positions (131073 entries)
    const SeedEntry *ent;                     // the entries of both indexes, the anchors' first
    const uint32_t *tab2;                     // the sequence table as 2-bit codes, 16 bases per word (kernels_bp.hip)
    int32_t stride;                           // read positions 0, stride, 2*stride, ... are looked up for anchors (2, 4 or 8)
    int32_t pitch;                            // a sequence's anchor blocks start at offsets pitch j + sft, sft < stride
};

struct SeedIndex {
    int stride = 2;           // anchor blocks sit at sequence offsets K j + 0 .. K j + stride - 1
    int k = 8;                // seed length
    int pitch = 8;            // distance of a sequence's anchor blocks: K rounded up to a multiple of the stride
    bool dense = false;       // candidates by the dozen per read are to be expected: the kernels that drain their queues as they fill
    DevBuf d_bm1, d_start, d_ent, d_tab2;
    SeedTables dev;
};

__host__ __device__ static inline uint32_t code2_of(uint8_t c) { return (c >> 1) & 3u; }

// 16-bit seed code of 8 bases: base t in bits 2t..2t+1 (rolls along a read with one shift)
static uint32_t host_code(const uint8_t *p, int K)
{
    uint32_t c = 0;
    for (int i = 0; i < K; i++) c |= code2_of(p[i]) << (2 * i);
    return c;
}

// The seed length for a configuration: every accepted alignment is at least min(minoverlap, minreadlength) long (class A/B
// overlaps are >= minoverlap, class C lengths are min(readlength, sequence length), and a seeded sequence is longer than
// that), so with maxerrors + 1 disjoint blocks of K <= that / (maxerrors + 1) bases one block is free of errors.  8 where it
// fits (test_engine.py:208-224 sweeps maxerrors 0..3 at minoverlap 25: K = 8, 8, 8, 6; test_analyser.py:55-58 has
// minoverlap 10, maxerrors 1: K = 5); below 5 the bitmaps say nothing and the exhaustive kernels serve the table.
// KVQ_K = 5..8 caps it (tests).  0: no seed filter.
int kvq_seed_k(const kvq_config &cfg)
{
    const int e = cfg.maxerrors;
    if (e < 0 || e > 6) return 0;
    if (cfg.Amin <= 13) return 0;                // the kernel relies on '\n' and '\r' closing every quality run (1058)
    int K = std::min(cfg.minoverlap, cfg.minreadlength) / (e + 1);
    if (K > KVQ_K_MAX) K = KVQ_K_MAX;
    if (const char *kv = getenv("KVQ_K")) { const int w = atoi(kv); if (w >= 5 && w < K) K = w; }
    return K >= 5 ? K : 0;
}

SeedIndex *kvq_seed_index_build(kvq_table *t)
{
    const kvq_config &cfg = t->cfg;
    t->seed_k = 0;
    const int e = cfg.maxerrors;
    const int K = kvq_seed_k(cfg);
    if (!K) return nullptr;
    // The anchor blocks of a sequence: e + 1 blocks of K bases, `pitch` apart, in `stride` shifted copies (offsets pitch j + sft, sft < stride).
    // A sequence that lies in a read at offset d is met through the copy with sft = -d mod stride, at read positions that are multiples of the
    // stride -- for that, all blocks of a copy must sit in ONE residue class, so the pitch is K rounded up to a multiple of the stride (8 for
    // K = 8; a pitch of K = 5 with stride 4 would need block 1 of another copy, which overlaps block 0: one error could break both).  The
    // copies must fit the sequence: length >= pitch e + K + stride - 1.
    auto pitch_of = [&](int stride) { return (K + stride - 1) / stride * stride; };
    auto span_of = [&](int stride) { return pitch_of(stride) * e + K + stride - 1; };
    int minlen = 1 << 30;
    for (int s = 0; s < t->nseq; s++) {
        const int len = t->h_off[s + 1] - t->h_off[s];
        bool ok = len >= span_of(2) && len <= 4095 && t->h_off[s] < (1 << 20) && s < (1 << 20);   // (stride 2 at least)
        for (int i = 0; ok && i < len; i++) {
            const uint8_t c = t->h_tab[t->h_off[s] + i];
            ok = (c == 'A' || c == 'C' || c == 'G' || c == 'T');
        }
        if (ok) { t->seeded.push_back(s); t->is_seeded[s] = 1; minlen = std::min(minlen, len); }
    }
    if (t->seeded.empty()) return nullptr;
    t->seed_k = K;
    // the wider the stride the fewer lookups per read (and the more index entries per sequence -- the expected number of
    // candidates stays the same): the widest one whose copies every seeded sequence can hold
    int stride = minlen >= span_of(8) ? 8 : minlen >= span_of(4) ? 4 : 2;
    if (const char *sv = getenv("KVQ_STRIDE")) { const int w = atoi(sv); if ((w == 2 || w == 4 || w == 8) && w <= stride) stride = w; }

    const int pitch = pitch_of(stride);
    struct Ent { uint32_t key; SeedEntry e; };                  // key = code, + 65536 for the index of all positions
    std::vector<Ent> v;
    for (int s : t->seeded) {
        const uint8_t *q = &t->h_tab[t->h_off[s]];
        const int len = t->h_off[s + 1] - t->h_off[s];
        const uint64_t hi = ((uint64_t)(uint32_t)t->h_off[s] << 32) | ((uint64_t)(uint32_t)len << 52) | ((uint64_t)(uint32_t)s << 12);
        auto entry = [&](int pos, uint32_t all) -> Ent {
            uint64_t ctx = 0;
            for (int i = 0; i < 32; i++) { const int sp = pos - 16 + i; if (sp >= 0 && sp < len) ctx |= (uint64_t)code2_of(q[sp]) << (2 * i); }
            return Ent{ host_code(q + pos, K) + (all << (2 * K)), SeedEntry{ hi | (uint64_t)pos, ctx } };
        };
        for (int j = 0; j <= e; j++)
            for (int sft = 0; sft < stride; sft++) v.push_back(entry(j * pitch + sft, 0u));
        for (int p = 0; p + K <= len; p++) v.push_back(entry(p, 1u));
    }
    if (v.size() >= (1u << 22)) {
        // (a work item of the scan kernel names its entry with 22 bits: a table this large goes to the exhaustive kernels as a whole)
        t->seeded.clear(); std::fill(t->is_seeded.begin(), t->is_seeded.end(), 0); t->seed_k = 0;
        return nullptr;
    }
    SeedIndex *ix = new SeedIndex();
    ix->stride = stride; ix->k = K; ix->pitch = pitch;
    const uint32_t NC = 1u << (2 * K);                          // codes
    std::vector<uint8_t> bm1(2 * NC / 8 < 64 ? 64 : 2 * NC / 8, 0);          // the anchors' bitmap, then the one of all positions
    std::stable_sort(v.begin(), v.end(), [](const Ent &a, const Ent &b) { return a.key < b.key; });
    std::vector<uint32_t> start(2 * (size_t)NC + 2, 0); std::vector<SeedEntry> ent(v.size() + 1, SeedEntry{ 0, 0 });
    for (size_t i = 0; i < v.size(); i++) {
        bm1[v[i].key >> 3] |= (uint8_t)(1u << (v[i].key & 7));                  // (bit 2K of the key = the second bitmap)
        start[v[i].key + 1]++; ent[i] = v[i].e;
    }
    {
        // candidates a 150-base read of random bases would give: its lookups against the anchors' bitmap, its 2 (e + 1) fixed blocks
        // against the one of all positions.  More than the wave's queue holds for its sixteen reads: the draining kernels.
        size_t occ_anc = 0, occ_all = 0;
        for (uint32_t c = 0; c < NC; c++) { occ_anc += start[c + 1] != 0; occ_all += start[NC + c + 1] != 0; }
        const double expect = (double)occ_anc / NC * (150.0 / stride) + (double)occ_all / NC * 2 * (e + 1);
        ix->dense = expect > 5.0;
    }
    for (uint32_t c = 0; c < 2 * NC; c++) start[c + 1] += start[c];
    start[2 * (size_t)NC + 1] = start[2 * (size_t)NC];
    if (ix->d_start.ensure(start.size() * 4) != KVQ_OK || ix->d_ent.ensure(ent.size() * sizeof(SeedEntry)) != KVQ_OK ||
        hipMemcpy(ix->d_start.p, start.data(), start.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ix->d_ent.p, ent.data(), ent.size() * sizeof(SeedEntry), hipMemcpyHostToDevice) != hipSuccess ||
        ix->d_bm1.ensure(bm1.size()) != KVQ_OK || hipMemcpy(ix->d_bm1.p, bm1.data(), bm1.size(), hipMemcpyHostToDevice) != hipSuccess) {
        if (!kvq_error_code()) kvq_set_error(KVQ_ERR_DEVICE, "uploading the seed index failed");
        kvq_seed_index_destroy(ix);
        return nullptr;
    }
    ix->dev.bm1 = ix->d_bm1.as<uint32_t>();
    {
        // the table as 2-bit codes (byte >> 1) & 3, base i in bits 2(i & 15) of word i >> 4 (two words of slack)
        const size_t nb = t->h_tab.size();
        std::vector<uint32_t> t2(nb / 16 + 3, 0);
        for (size_t i = 0; i < nb; i++) t2[i >> 4] |= code2_of(t->h_tab[i]) << (2 * (i & 15));
        if (ix->d_tab2.ensure(t2.size() * 4) != KVQ_OK || hipMemcpy(ix->d_tab2.p, t2.data(), t2.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            if (!kvq_error_code()) kvq_set_error(KVQ_ERR_DEVICE, "uploading the seed index failed");
            kvq_seed_index_destroy(ix);
            return nullptr;
        }
        ix->dev.tab2 = ix->d_tab2.as<uint32_t>();
    }
    ix->dev.start = ix->d_start.as<uint32_t>(); ix->dev.ent = ix->d_ent.as<SeedEntry>();
    ix->dev.stride = stride; ix->dev.pitch = pitch;
    return ix;
}

void kvq_seed_index_destroy(SeedIndex *ix)
{
    if (!ix) return;
    DevBuf *b[] = { &ix->d_tab2, &ix->d_bm1, &ix->d_start, &ix->d_ent };
    for (DevBuf *x : b) x->release();
    delete ix;
}

// ---------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------

// A byte of LDS by its absolute address.  The scan kernel's only LDS object starts at LDS address 0 (checked at
// kernel start); spelling the address out saves the "add the base (0)" instruction the compiler otherwise emits
// for every computed index.
__device__ __forceinline__ uint32_t lds_byte_at(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>((uintptr_t)addr);
}
// values that are the same in every lane (LDS reads at uniform addresses, wave
// numbers) must be moved to scalar registers by hand: the compiler cannot know
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// bytes of the sequence table: a pointer read from memory is "generic" to the compiler, which then
// uses flat loads that also tie up the LDS counter; these say "global memory"
typedef const __attribute__((address_space(1))) uint8_t *GlbBytes;
__device__ __forceinline__ uint32_t glb_u32(GlbBytes p)                         // any alignment
{
    typedef uint32_t __attribute__((aligned(1))) u32_any;
    return *reinterpret_cast<const __attribute__((address_space(1))) u32_any *>(p);
}

// 0x80 in every byte of x that is a good score ((signed) c >= Amin, Amin in 14..127); addk = (0x80 - Amin) * 0x01010101
__device__ __forceinline__ uint32_t good_flags(uint32_t x, uint32_t addk)
{
    return __builtin_amdgcn_bitop3_b32((x & 0x7F7F7F7Fu) + addk, 0x80808080u, x, 0x40);               // t & 0x80.. & ~x
}

// product of two numbers below 2^24 at full rate (v_mul_lo_u32 takes four issue slots)
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// longest run of ones in the low n (1..64) bits of m, first one if several: length, start
__device__ __forceinline__ void longest_run64(uint64_t m, int n, int &len, int &start)
{
    const uint64_t r1 = m & (m << 1), r2 = r1 & (r1 << 2), r3 = r2 & (r2 << 4), r4 = r3 & (r3 << 8), r5 = r4 & (r4 << 16);
    // E = end positions of runs of length >= L; grow L by binary descent
    uint64_t E = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    int L = 0; uint64_t T;
    T = E & r5;        if (T) { E = T; L = 32; }
    T = E & (r4 << L); if (T) { E = T; L += 16; }
    T = E & (r3 << L); if (T) { E = T; L += 8; }
    T = E & (r2 << L); if (T) { E = T; L += 4; }
    T = E & (r1 << L); if (T) { E = T; L += 2; }
    T = E & (m << L);  if (T) { E = T; L += 1; }
    len = L;
    start = L ? (__ffsll((long long)E) - 1) - L + 1 : 0;
    if (m == ~0ull) { len = 64; start = 0; }                 // the descent tops out at 63
}

// longest-run summary of a stretch of score bytes; merge is associative (left, right)
struct Seg { int len, pre, suf, best, bstart, beg; };

__device__ __forceinline__ Seg seg_merge(const Seg &A, const Seg &B)
{
    Seg R;
    R.beg = A.beg; R.len = A.len + B.len;
    R.pre = (A.pre == A.len) ? A.len + B.pre : A.pre;
    R.suf = (B.suf == B.len) ? B.len + A.suf : B.suf;
    R.best = A.best; R.bstart = A.bstart;                       // the first of equally long runs wins (1062)
    const int cross = A.suf + B.pre;
    if (cross > R.best) { R.best = cross; R.bstart = A.beg + A.len - A.suf; }
    if (B.best > R.best) { R.best = B.best; R.bstart = B.bstart; }
    return R;
}

// number of differing bytes of two dwords
__device__ __forceinline__ int diff_bytes(uint32_t x, uint32_t y)
{
    const uint32_t v = x ^ y;
    return __popc(__builtin_amdgcn_bitop3_b32((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu, 0x80808080u, v, 0xC8));  // (t | v) & 0x80..
}

// kvq_emit with the arena words read from the device copy of the parameters, by a wave that has a hit
__device__ __forceinline__ void emit_cold(const KvqParams *__restrict__ P, bool hit, int64_t fpos, int seq_nr,
                                          int seq_pos, int length, int rl, uint32_t key)
{
    const uint64_t m = __ballot(hit);
    if (m == 0) return;
    KvqHit *const arena = P->arena; const uint32_t cap = P->arena_cap; unsigned int *const arena_n = P->arena_n;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (kvq_lane() == leader) base = atomicAdd(arena_n, (unsigned int)__popcll(m));
    base = __shfl(base, leader, 64);
    if (hit) {
        const uint32_t idx = base + (uint32_t)__popcll(m & kvq_lanemask_lt());
        if (idx < cap) {
            KvqHit h;
            h.fpos = fpos; h.seq_nr = seq_nr; h.seq_pos = seq_pos; h.length = length;
            h.readlength = rl; h.key = key; h.blob_off = 0;
            arena[idx] = h;
        }
    }
}

// a thread that has no block of text in a tile passes an offset beyond any tile to the buffer load and gets zeros
#define ST_NO_BLOCK 0x7FFFFF00u
// one thread per chunk: replay the tile reports against the exact newline
// count; any tile whose speculated first record is not the one the count gives
// sets *spec_fail (the host then rescans with the exhaustive kernels)
__host__ __device__ static inline bool kvq_tile_report_bad(uint32_t rep, bool first_tile, uint32_t seen, uint32_t total)
{
    const uint32_t n_owned = rep & 0xFFFFu, jn = (rep >> 16) & 0xFFu;
    if (rep & TR_FLAG_FALLBACK) return true;
    if ((rep & TR_FLAG_SKIPPED) && !(rep & TR_FLAG_PARTIAL)) return false;      // (it scanned nothing: its records are found again from the exact newline count)
    // the first record this tile owns starts behind its newline number `want` (tile 0: the chunk
    // start itself) -- if the chunk goes on for at least two more lines behind that newline: the
    // kernel wants to see the record's '+' line (P2), and a chunk that ends there has no record left
    // (the chunk's last tile often owns nothing but the final newline of the last record)
    const uint32_t want = first_tile ? 0u : 4u - (seen & 3u);
    const bool expect = want <= n_owned && (first_tile || want + 2u <= total - seen);
    return expect ? jn != want : jn != TR_NONE;
}

extern "C" __global__ void __launch_bounds__(256)
kvq_validate_tiles(uint32_t nchunks, const uint32_t *__restrict__ tile_first, const uint32_t *__restrict__ tile_report,
                   unsigned int *__restrict__ spec_fail, KvqSkippedTile *__restrict__ skip_list,
                   const uint32_t *__restrict__ chunk_off, uint32_t tile_bytes)
{
    KVQ_BESIDE_SCAN();
    // *spec_fail: bit 0 = the batch failed; from bit 8 on = number of skipped tiles.  What kvq_collect_skipped needs to
    // walk the records such a tile left -- its chunk, what it owns, the newlines of the chunk in front of it, or where the
    // first record it left begins -- goes to skip_list: the redo is enqueued behind this kernel without the host looking
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t total = 0;                        // newlines of the chunk
    for (uint32_t g = tile_first[c]; g < tile_first[c + 1]; g++) total += tile_report[g] & 0xFFFFu;
    uint32_t seen = 0;                         // newlines of the chunk in front of the tile
    bool bad = false;
    for (uint32_t g = tile_first[c]; g < tile_first[c + 1]; g++) {
        const uint32_t rep = tile_report[g];
        if (kvq_tile_report_bad(rep, g == tile_first[c], seen, total)) bad = true;
        if ((rep & TR_FLAG_SKIPPED) && skip_list) {
            const unsigned int k = atomicAdd(spec_fail, 0x100u) >> 8;
            if (k < KVQ_SKIP_CAP) {
                // tile geometry as kvq_seeded_launch made it
                const uint32_t a = chunk_off[c], e = chunk_off[c + 1], tn = g - tile_first[c];
                const uint32_t g0 = (a & ~15u) + tn * tile_bytes;
                KvqSkippedTile T;
                T.a = a; T.b = e; T.own_begin = tn == 0 ? a : g0; T.own_end = (uint64_t)g0 + tile_bytes < e ? g0 + tile_bytes : e;
                T.seen = seen; T.first = tn == 0 ? 1u : 0u;
                // (a tile that has scanned the records in front of z - 1: the walk begins there, at a record's first byte)
                const uint32_t z = tile_report[tile_first[nchunks] + g];
                if (z) { T.a = T.own_begin = z - 1u; T.seen = 0; T.first = 1u; }
                skip_list[k] = T;
            } else bad = true;
        }
        seen += rep & 0xFFFFu;
    }
    if (bad) atomicOr(spec_fail, 1u);
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------

// one thread per chunk: per-tile table {chunk begin, chunk end, tile number inside the chunk, chunk}
extern "C" __global__ void __launch_bounds__(256)
kvq_expand_tiles(uint32_t nchunks, const uint32_t *__restrict__ chunk_off, const uint32_t *__restrict__ tile_first, uint4 *__restrict__ tile_tab, unsigned int *__restrict__ redo_count,
                 unsigned int *__restrict__ surv_count)
{
    KVQ_BESIDE_SCAN();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && redo_count) { redo_count[0] = 0; redo_count[1] = 0; }      // (records that skipped tiles leave, and the long ones among them, are counted afresh for this launch)
    if (c == 0 && surv_count) surv_count[0] = 0;                             // (and so are the scan kernel's survivors)
    if (c >= nchunks) return;
    const uint32_t a = chunk_off[c], b = chunk_off[c + 1], g0 = tile_first[c];
    for (uint32_t g = g0; g < tile_first[c + 1]; g++) tile_tab[g] = make_uint4(a, b, g - g0, c);
}

// Bytes a tile owns (a multiple of ST_BLK, ST_TILE_MIN .. ST_TILE + ST_OV - 1040): the LDS buffer holds
// ST_TILE + ST_OV bytes, and what a tile does not own it reads a second time as the look-ahead for
// its last record.  The look-ahead must cover the longest record; `maxline` is the longest line
// (with its newline) among the first bytes of the scan, a record has four lines.  A later record
// that outgrows the look-ahead sends its batch to the exhaustive kernels (and the scan back to the
// full look-ahead), so this is a matter of speed only.
// `rec_bytes` (0 = unknown) is the average record among those first bytes: the lane group of a read is
// the widest power of two G with G * records <= 512, so a tile with a few records more than a power of
// two p runs at half the group width (145 records of 125 bp: two lanes per read, 57 % of the lanes
// busy, 63-byte slices); when owning p - 1 records costs less than 22 % of the tile, the tile is cut
// to that (measured: 125 bp + 13 %, 120 bp + 6 %, 110 bp + 3 %; 105 bp, where it costs 25 %, - 2 %).
uint32_t kvq_choose_tile(uint32_t maxline, uint32_t rec_bytes)
{
    if (const char *e = getenv("KVQ_TILE")) { const int v = atoi(e); if (v >= 4000 && v <= (int)(ST_TILE + ST_OV - 1040u)) return (uint32_t)v / ST_BLK * ST_BLK; }
    if (maxline == 0) return ST_TILE;
    uint32_t ov = (4u * (maxline + 2u) + 160u + ST_BLK - 1u) / ST_BLK * ST_BLK;
    ov = std::max<uint32_t>(1040u, std::min<uint32_t>(ov, ST_OV));
    uint32_t tile = ST_TILE + ST_OV - ov;
    if (rec_bytes >= 40u) {
        const uint32_t n_full = tile / rec_bytes + 1u;                     // records a full tile can own
        uint32_t p = 1; while (2u * p <= n_full) p *= 2u;
        if (p >= 16u && p <= (uint32_t)ST_THREADS && n_full > p) {
            const uint32_t cut = (p - 1u) * rec_bytes / ST_BLK * ST_BLK;
            if (cut >= ST_TILE_MIN && (uint64_t)cut * 100u >= (uint64_t)tile * 78u) tile = cut;
        }
    }
    return tile;
}
uint32_t kvq_min_tile()
{
    if (getenv("KVQ_TILE")) return std::min<uint32_t>(ST_TILE_MIN, kvq_choose_tile(1u << 20, 0));
    return ST_TILE_MIN;
}

// the longest line (newline included) among the first bytes of a text (an unfinished last line counts)
// and the average record (four lines) among them, 0 when there are fewer than four records
void kvq_probe_text(const uint8_t *text, size_t n, uint32_t &maxline, uint32_t &rec_bytes)
{
    uint32_t best = 0; size_t start = 0, lines = 0;
    for (size_t i = 0; i < n; i++)
        if (text[i] == '\n') { best = std::max<uint32_t>(best, (uint32_t)(i + 1 - start)); start = i + 1; lines++; }
    maxline = std::max<uint32_t>(best, (uint32_t)(n - start));
    rec_bytes = lines >= 16 ? (uint32_t)(start * 4 / lines) : 0u;           // (start = bytes in whole lines)
}
uint32_t kvq_tile_for_text(const uint8_t *text, size_t n, uint32_t *rec_bytes_out)
{
    uint32_t maxline, rec_bytes;
    kvq_probe_text(text, n, maxline, rec_bytes);
    if (rec_bytes_out) *rec_bytes_out = rec_bytes;
    return kvq_choose_tile(maxline, rec_bytes);
}

