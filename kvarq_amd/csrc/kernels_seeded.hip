// kvarq_amd/csrc/kernels_seeded.hip -- the fused seed-filter scan: one pass over
// the FastQ text does record split, quality trim, read-length histogram and
// matching (workhorse.c:1010-1175) for every sequence that qualifies for seeding.
//
// Persistent workgroups (512 threads, two per CU) walk tiles of up to 39760 bytes of the input,
// handed out by a counter (kvq_choose_tile sizes them, wave priorities rise through a tile):
//   P0  every thread fetches the 80 contiguous bytes it scans (buffer loads, issued one tile
//       ahead), writes them to LDS and takes the newline flags from the registers
//   P1  workgroup prefix sum -> sorted newline offsets in LDS
//   P2  first record of the tile: exact for the first tile of a chunk, otherwise
//       speculated from the text ("@" line followed two lines later by a "+"
//       line) and verified after the kernel by kvq_validate_tiles against the
//       exact count of newlines (four '\n' = one record, workhorse.c:1018-1034);
//       any disagreement makes the host rescan with the exhaustive kernels
//   P3  G lanes per read (G = 4 at 150 bp), a wave owns its 64/G reads from here on:
//       '@'/'+' checks (1037-1048), longest run of scores >= Amin from a SWAR bitmask of
//       good bytes (1055-1068), LDS histogram (394-402); the read's 8-mers at every
//       stride-th position are looked up in an LDS bitmap of anchor blocks, its head and
//       tail blocks in the bitmap of all sequence 8-mers; candidates (read, position) go
//       to the wave's own LDS queue
//   P4  the wave verifies its candidates, one (candidate, index entry) pair per lane:
//       (sequence, diagonal) -> byte-exact mismatch count under the class A/B/C rules
//       (1112-1174) -> hits appended to the global arena (one atomic per wave)
//   (one barrier at the end of the tile; DESIGN.md section 4.1 has the details and the measurements)
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
#define SK 8                       // seed length
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

struct SeedTables {
    const uint32_t *bm1;                      // two bitmaps of 8 KiB, one bit per 8-mer code: anchor blocks, then anywhere in a sequence
    const uint32_t *start_anc, *start_all;    // CSR starts, 65537 entries
    // entry: position in sequence (12) | sequence number (20) | table offset of the sequence (20) | its length (12)
    const uint64_t *ent_anc, *ent_all;
    const uint32_t *tab2;                     // the sequence table as 2-bit codes, 16 bases per word (kernels_bp.hip)
    // the same tables once more in ONE allocation, addressed as words from one base (kernels_pool.hip: one pointer
    // in scalar registers instead of six): start_anc at 0, start_all, tab2, then the 64-bit entries (even offsets)
    const uint32_t *blob; uint32_t off_start_all, off_tab2, off_ent_anc, off_ent_all;
    int32_t stride;                           // read positions 0, stride, 2*stride, ... are looked up for anchors (2, 4 or 8)
};

struct SeedIndex {
    int variant = 2;          // 2 = kvq_scan_bp (kernels_bp.hip, the default), 0 = kvq_scan_pool (KVQ_KERNEL=pool), 1 = kvq_scan_seeded (KVQ_KERNEL=v1)
    int stride = 2;           // anchor blocks sit at sequence offsets 8j + 0 .. 8j + stride - 1
    DevBuf d_bm1, d_start_anc, d_start_all, d_ent_anc, d_ent_all, d_tab2, d_blob;
    SeedTables dev;
};

__host__ __device__ static inline uint32_t code2_of(uint8_t c) { return (c >> 1) & 3u; }

// 16-bit seed code of 8 bases: base t in bits 2t..2t+1 (rolls along a read with one shift)
static uint32_t host_code8(const uint8_t *p)
{
    uint32_t c = 0;
    for (int i = 0; i < SK; i++) c |= code2_of(p[i]) << (2 * i);
    return c;
}

SeedIndex *kvq_seed_index_build(kvq_table *t)
{
    const kvq_config &cfg = t->cfg;
    t->seed_k = 0;
    const int e = cfg.maxerrors;
    if (e < 0 || e > 6) return nullptr;
    const int need = (e + 1) * SK;
    const char *kv = getenv("KVQ_KERNEL");
    const int variant = (kv && !strcmp(kv, "v1")) ? 1 : (kv && !strcmp(kv, "pool")) ? 0 : 2;          // which kernel walks the text; the index is the same
    // every accepted alignment must be at least `need` long: class A/B overlaps
    // are >= minoverlap, class C lengths are min(readlength, sequence length)
    if (cfg.minoverlap < need || cfg.minreadlength < need) return nullptr;
    if (cfg.Amin <= 13) return nullptr;          // the kernel relies on '\n' and '\r' closing every quality run (1058)
    int minlen = 1 << 30;
    for (int s = 0; s < t->nseq; s++) {
        const int len = t->h_off[s + 1] - t->h_off[s];
        bool ok = len >= need + 1 && len <= 4095 && t->h_off[s] < (1 << 20) && s < (1 << 20);   // the shifted anchor sets need stride - 1 more bases
        for (int i = 0; ok && i < len; i++) {
            const uint8_t c = t->h_tab[t->h_off[s] + i];
            ok = (c == 'A' || c == 'C' || c == 'G' || c == 'T');
        }
        if (ok) { t->seeded.push_back(s); t->is_seeded[s] = 1; minlen = std::min(minlen, len); }
    }
    if (t->seeded.empty()) return nullptr;
    t->seed_k = SK;
    // a sequence contained in a read at offset d is met through the anchor block at sequence
    // offset 8j + sft with sft = -d mod stride, at a read position that is a multiple of the
    // stride: the wider the stride the fewer lookups per read (and the more index entries per
    // sequence -- the expected number of candidates stays the same).  Every seeded sequence
    // must be able to hold its shifted blocks: length >= 8(e+1) + stride - 1.
    int stride = minlen >= need + 7 ? 8 : minlen >= need + 3 ? 4 : 2;
    if (const char *sv = getenv("KVQ_STRIDE")) { const int w = atoi(sv); if ((w == 2 || w == 4 || w == 8) && w <= stride) stride = w; }

    std::vector<std::pair<uint32_t, uint64_t>> anc, all;       // (code, entry)
    for (int s : t->seeded) {
        const uint8_t *q = &t->h_tab[t->h_off[s]];
        const int len = t->h_off[s + 1] - t->h_off[s];
        const uint64_t hi = ((uint64_t)(uint32_t)t->h_off[s] << 32) | ((uint64_t)(uint32_t)len << 52) | ((uint64_t)(uint32_t)s << 12);
        for (int j = 0; j <= e; j++)
            for (int sft = 0; sft < stride; sft++) anc.emplace_back(host_code8(q + j * SK + sft), hi | (uint64_t)(j * SK + sft));
        for (int p = 0; p + SK <= len; p++) all.emplace_back(host_code8(q + p), hi | (uint64_t)p);
    }
    SeedIndex *ix = new SeedIndex();
    ix->variant = variant; ix->stride = stride;
    std::vector<uint8_t> bm1(16384, 0);
    std::vector<uint32_t> h_start[2]; std::vector<uint64_t> h_ent[2]; std::vector<uint32_t> h_tab2;
    auto upload = [&](std::vector<std::pair<uint32_t, uint64_t>> &v, int bit, DevBuf &st, DevBuf &en) -> bool {
        std::sort(v.begin(), v.end());
        std::vector<uint32_t> &start = h_start[bit]; std::vector<uint64_t> &ent = h_ent[bit];
        start.assign(65537, 0); ent.assign(v.size() + 1, 0);
        for (size_t i = 0; i < v.size(); i++) {
            bm1[(size_t)bit * 8192 + (v[i].first >> 3)] |= (uint8_t)(1u << (v[i].first & 7));
            start[v[i].first + 1]++; ent[i] = v[i].second;
        }
        for (int c = 0; c < 65536; c++) start[c + 1] += start[c];
        return st.ensure(65537 * 4) == KVQ_OK && en.ensure(ent.size() * 8) == KVQ_OK &&
               hipMemcpy(st.p, start.data(), 65537 * 4, hipMemcpyHostToDevice) == hipSuccess &&
               hipMemcpy(en.p, ent.data(), ent.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!upload(anc, 0, ix->d_start_anc, ix->d_ent_anc) || !upload(all, 1, ix->d_start_all, ix->d_ent_all) ||
        ix->d_bm1.ensure(16384) != KVQ_OK || hipMemcpy(ix->d_bm1.p, bm1.data(), 16384, hipMemcpyHostToDevice) != hipSuccess) {
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
        h_tab2.swap(t2);
    }
    {
        std::vector<uint32_t> blob;
        auto put_words = [&](const uint32_t *w, size_t n) { const size_t at = blob.size(); blob.insert(blob.end(), w, w + n); while (blob.size() & 1) blob.push_back(0); return (uint32_t)at; };
        put_words(h_start[0].data(), h_start[0].size());
        ix->dev.off_start_all = put_words(h_start[1].data(), h_start[1].size());
        ix->dev.off_tab2 = put_words(h_tab2.data(), h_tab2.size());
        ix->dev.off_ent_anc = put_words(reinterpret_cast<const uint32_t *>(h_ent[0].data()), h_ent[0].size() * 2);
        ix->dev.off_ent_all = put_words(reinterpret_cast<const uint32_t *>(h_ent[1].data()), h_ent[1].size() * 2);
        blob.push_back(0); blob.push_back(0);
        if (ix->d_blob.ensure(blob.size() * 4) != KVQ_OK || hipMemcpy(ix->d_blob.p, blob.data(), blob.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            if (!kvq_error_code()) kvq_set_error(KVQ_ERR_DEVICE, "uploading the seed index failed");
            kvq_seed_index_destroy(ix);
            return nullptr;
        }
        ix->dev.blob = ix->d_blob.as<uint32_t>();
    }
    ix->dev.start_anc = ix->d_start_anc.as<uint32_t>(); ix->dev.start_all = ix->d_start_all.as<uint32_t>();
    ix->dev.ent_anc = ix->d_ent_anc.as<uint64_t>(); ix->dev.ent_all = ix->d_ent_all.as<uint64_t>();
    ix->dev.stride = stride;
    return ix;
}

void kvq_seed_index_destroy(SeedIndex *ix)
{
    if (!ix) return;
    DevBuf *b[] = { &ix->d_blob, &ix->d_tab2, &ix->d_bm1, &ix->d_start_anc, &ix->d_start_all, &ix->d_ent_anc, &ix->d_ent_all };
    for (DevBuf *x : b) x->release();
    delete ix;
}

// ---------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------

struct SeededLds {
    uint8_t  buf[ST_BUF + ST_BLK];       // buf[ST_PRE] = first byte the tile owns; one block of slack behind the look-ahead
    uint16_t nl[ST_NLCAP];               // offsets into buf of every '\n', ascending
    uint8_t  bmA[8192], bmL[8192];       // one bit per 8-mer code: an anchor block of some sequence / anywhere in some sequence
    uint32_t hist[KVQ_RL_BINS / 2];      // read-length histogram, two 16-bit bins per word, flushed every ST_HIST_TILES tiles
    uint2    q1[ST_QCAP];                // candidate: x = rec | pos << 16, y = offset of its 8-mer in buf | kind << 16
    uint32_t q2[ST_Q2CAP];               // work item: candidate << 22 | index entry
    uint32_t rinfo[ST_RCAP];             // read offset in buf | rl << 16
    __attribute__((aligned(16))) uint32_t wtot[ST_WAVES];   // newlines per wave
    uint32_t longest_p1, records, fallback, n_owned, next_tile;
};

// A byte of LDS by its absolute address.  The kernel's only LDS object is the dynamic block, which
// starts at LDS address 0 (checked at kernel start); spelling the address out saves the
// "add the base (0)" instruction the compiler otherwise emits for every computed index.
#define KVQ_LDS_BMA ((uint32_t)offsetof(SeededLds, bmA))
__device__ __forceinline__ uint32_t lds_byte_at(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>((uintptr_t)addr);
}
// a dword of buf[] at byte offset off (a multiple of 4) / at any offset (buf is the first member: LDS address = off)
static_assert(offsetof(SeededLds, buf) == 0, "buf[] first");
__device__ __forceinline__ uint32_t buf_u32(uint32_t off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)off);
}
__device__ __forceinline__ uint32_t buf_u32_any(uint32_t off)
{
    typedef uint32_t __attribute__((aligned(1))) u32_any;
    return *reinterpret_cast<const __attribute__((address_space(3))) u32_any *>((uintptr_t)off);
}

static_assert(sizeof(SeededLds) <= 80 * 1024, "two workgroups per CU: at most 80 KB of LDS each");

// values that are the same in every lane (LDS reads at uniform addresses, wave
// numbers) must be moved to scalar registers by hand: the compiler cannot know
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct TileGeo {
    uint32_t a, b;          // chunk [a, b) in batch offsets
    uint32_t t;             // tile number inside the chunk
    uint32_t g0;            // batch offset of buf[ST_PRE]
    uint32_t own_begin;     // first byte whose newlines this tile counts
    uint32_t own_end;       // batch offset where ownership ends
    uint32_t load_lo, load_hi;   // loaded byte range [load_lo, load_hi) lands at buf[ST_PRE ...]
};

__device__ __forceinline__ TileGeo tile_geo(uint32_t g, const uint4 *tiles, uint32_t tile_bytes)
{
    // the per-tile table kvq_expand_tiles wrote: {chunk begin, chunk end, tile number, chunk}
    TileGeo J;
    const uint4 q = tiles[g];
    J.a = q.x; J.b = q.y; J.t = q.z;
    // a tile owns tile_bytes and looks ST_TILE + ST_OV - tile_bytes ahead (the LDS buffer holds both)
    J.g0 = (J.a & ~15u) + J.t * tile_bytes;
    J.own_end = J.g0 + tile_bytes < J.b ? J.g0 + tile_bytes : J.b;
    J.own_begin = J.t == 0 ? J.a : J.g0;
    J.load_lo = J.g0;
    J.load_hi = J.g0 + ST_TILE + ST_OV < J.b ? J.g0 + ST_TILE + ST_OV : J.b;
    return J;
}

// eight text bytes as two dwords -> twice their 16-bit seed code (the byte-wise dot product
// gathers the 2-bit base codes: bits 1..2 of every byte, weights 1, 4, 16, 64)
__device__ __forceinline__ uint32_t code8x2_of(uint32_t lo, uint32_t hi)
{
    return __builtin_amdgcn_udot4(lo & 0x06060606u, 0x40100401u, 0u, false) +
           (__builtin_amdgcn_udot4(hi & 0x06060606u, 0x40100401u, 0u, false) << 8);
}
// ... of the 8 bytes at buf[off ...] (any alignment)
__device__ __forceinline__ uint32_t lds_code8x2(const SeededLds &S, uint32_t off)
{
    const uint32_t w = off & ~3u, sh = (off & 3u) * 8u;
    const uint32_t d0 = buf_u32(w), d1 = buf_u32(w + 4u),
                   d2 = buf_u32(w + 8u);
    return code8x2_of(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh));
}
__device__ __forceinline__ uint32_t lds_code8(const SeededLds &S, uint32_t off) { return lds_code8x2(S, off) >> 1; }
// bytes of the sequence table: a pointer read from memory is "generic" to the compiler, which then
// uses flat loads that also tie up the LDS counter; these say "global memory"
typedef const __attribute__((address_space(1))) uint8_t *GlbBytes;
__device__ __forceinline__ uint32_t glb_u32(GlbBytes p)                         // any alignment
{
    typedef uint32_t __attribute__((aligned(1))) u32_any;
    return *reinterpret_cast<const __attribute__((address_space(1))) u32_any *>(p);
}
__device__ __forceinline__ uint32_t glb_code8(GlbBytes x) { return code8x2_of(glb_u32(x), glb_u32(x + 4)) >> 1; }
__device__ __forceinline__ uint32_t glb_code8(const uint8_t *x)                 // (planes kernel: generic pointers)
{
    uint32_t lo, hi;
    __builtin_memcpy(&lo, x, 4); __builtin_memcpy(&hi, x + 4, 4);
    return code8x2_of(lo, hi) >> 1;
}

// 4 score bytes -> 4 bits, bit set = byte is good ((signed) c >= Amin, Amin in 14..127);
// addk = (0x80 - Amin) * 0x01010101
__device__ __forceinline__ uint32_t good4(uint32_t x, uint32_t addk)
{
    const uint32_t gf = ((x & 0x7F7F7F7Fu) + addk) & ~x & 0x80808080u;    // 0x80 per good byte
    return __builtin_amdgcn_udot4(gf, 0x08040201u, 0u, false) >> 7;       // bits 7,15,23,31 -> 0..3
}
// the same as 0x80 flags per byte
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

// is the seed (read block at rp, sequence block at sq) live: equal 2-bit codes
__device__ __forceinline__ bool seed_live(const SeededLds &S, uint32_t roff, int rl, int rp, GlbBytes seq, int seql, int sq)
{
    if (rp < 0 || rp + SK > rl || sq < 0 || sq + SK > seql) return false;
    return lds_code8(S, roff + (uint32_t)rp) == glb_code8(seq + sq);
}

// number of differing bytes of two dwords
__device__ __forceinline__ int diff_bytes(uint32_t x, uint32_t y)
{
    const uint32_t v = x ^ y;
    return __popc(__builtin_amdgcn_bitop3_b32((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu, 0x80808080u, v, 0xC8));  // (t | v) & 0x80..
}

// One work item = one (candidate, index entry) pair = one diagonal of one read
// against one sequence: byte-exact check under the reference's loop bounds and
// emission of its hits.  Must be called by every lane of the wave.
// the scanning loop keeps only these of KvqParams in scalar registers; the rest is read from
// the device copy where it is needed (hits, errors, the final flush)
struct HotParams {
    const KvqParams *cold;
    GlbBytes tab;
    int maxerrors, minoverlap, minreadlength, amin;
};

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

__device__ __forceinline__ void verify_item(const HotParams &P, const SeededLds &S, bool active, uint32_t rec, int p,
                                            uint32_t kind, uint64_t en, int64_t tile_fpos, int stride)
{
    bool hitAB = false, hitC = false;
    int s = 0, rl = 0, lenAB = 0, lenC = 0, sposAB = 0, sposC = 0; uint32_t keyAB = 0, keyC = 0;
    int64_t fpos = 0;
    if (active) {
        const uint32_t ri = S.rinfo[rec];
        const uint32_t roff = ri & 0xFFFFu; rl = (int)(ri >> 16);
        fpos = tile_fpos + (int64_t)roff - (int64_t)ST_PRE;
        const int q = (int)(en & 4095u);
        s = (int)((en >> 12) & 0xFFFFFu);
        const GlbBytes seq = P.tab + (uint32_t)((en >> 32) & 0xFFFFFu);
        const int seql = (int)(en >> 52);
        const int mo = P.minoverlap, me = P.maxerrors;
        const int d = q - p;                             // sequence index = read index + d
        const int a = d < 0 ? -d : 0;
        const int L = (rl < seql - d ? rl : seql - d) - a;
        // most false candidates die here, on the first 16 bytes of the diagonal (one round trip),
        // before anything else is worked out for them
        int mism = 0, j = 0;
        const uint32_t x = roff + (uint32_t)a; const GlbBytes y = seq + a + d;
        bool alive = L > 0;
        if (alive && L >= 16) {
            uint32_t rw[4], sw[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { rw[t] = buf_u32_any(x + 4u * (uint32_t)t); sw[t] = glb_u32(y + 4 * t); }
#pragma unroll
            for (int t = 0; t < 4; t++) mism += diff_bytes(rw[t], sw[t]);
            j = 16;
            alive = mism <= me;
        }
        // which reference loops visit this diagonal
        bool canAB = false, canC = false;
        const bool guard = rl > mo && seql > mo;
        if (alive) {
            if (d < 0) {
                const int i = -d;
                if (i <= rl - seql) { canC = true; lenC = seql; sposC = -i; keyC = (2u << 30) | (uint32_t)i; }              // 1147
                else if (guard && i <= rl - mo) { canAB = true; lenAB = rl - i; sposAB = -i; keyAB = (0u << 30) | (uint32_t)(rl - mo - i); }   // 1116
            } else if (d == 0) {
                canC = true; lenC = rl > seql ? seql : rl; sposC = 0; keyC = 2u << 30;                                         // 1147 / 1163
            } else {
                const int i = d;
                if (guard && i <= seql - mo && i >= seql - rl) { canAB = true; lenAB = seql - i; sposAB = i; keyAB = (1u << 30) | (uint32_t)(seql - mo - i); }   // 1130
                if (rl <= seql && i <= seql - rl) { canC = true; lenC = rl; sposC = i; keyC = (2u << 30) | (uint32_t)i; }       // 1163
            }
        }
        if (canAB || canC) {
            for (; j + 4 <= L && mism <= me; j += 4) {
                uint32_t rw, sw;
                rw = buf_u32_any(x + (uint32_t)j); sw = glb_u32(y + j);
                mism += diff_bytes(rw, sw);
            }
            for (; j < L && mism <= me; j++) mism += (S.buf[x + j] != y[j]);
            if (mism <= me) {
                // canonical discoverer: no live seed earlier in the order
                // [ALL-index read blocks by position] then [ANCHOR blocks by number]
                bool earlier = false;
                for (int jj = 0; jj <= me && !earlier; jj++) {
                    const int ph = jj * SK, pt = rl - (jj + 1) * SK;
                    if (ph + SK <= rl && (kind == 0u || ph < p)) earlier = seed_live(S, roff, rl, ph, seq, seql, ph + d);
                    if (!earlier && pt >= 0 && (kind == 0u || pt < p)) earlier = seed_live(S, roff, rl, pt, seq, seql, pt + d);
                }
                if (kind == 0u) {
                    // anchors sit at sequence offsets 8j + sft (sft < stride) and are looked up at read
                    // positions that are multiples of the stride only
                    for (int jj = 0; jj <= me && !earlier; jj++)
                        for (int sft = 0; sft < stride && !earlier; sft++) {
                            const int o = jj * SK + sft;
                            if (o < q && ((o - d) & (stride - 1)) == 0) earlier = seed_live(S, roff, rl, o - d, seq, seql, o);
                        }
                }
                if (!earlier) { hitAB = canAB; hitC = canC; }
            }
        }
    }
    emit_cold(P.cold, hitAB, fpos, s, sposAB, lenAB, rl, keyAB);
    emit_cold(P.cold, hitC, fpos, s, sposC, lenC, rl, keyC);
}

// The tile's text is fetched with raw buffer loads whose range check replaces the branch on "behind
// the end of the chunk" (no exec masking, so the loads of a tile go out back to back; beyond the
// descriptor's range a load returns zeros).  The batch buffer is padded to a multiple of 16.
// the five vectors of a thread's scan block, for the tile whose text is [lo, hi) of the batch (lo a
// multiple of 16): the buffer descriptor starts at the tile, so the per-thread offset `vo` is the
// same for every tile (no address arithmetic on the vector unit) and the 16 r come as immediates;
// a thread that has no block (thread 0) passes an offset beyond any tile and gets zeros
#define ST_NO_BLOCK 0x7FFFFF00u
__device__ __forceinline__ void tile_load80(const uint8_t *data, uint32_t lo, uint32_t hi, uint32_t vo, uint4 (&pre)[ST_ROUNDS])
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(data + lo), 0, (int)(((hi + 15u) & ~15u) - lo), 0x00020000);
#pragma unroll
    for (int r = 0; r < (int)ST_ROUNDS; r++) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo + 16u * (uint32_t)r), 0, 0);
        pre[r] = make_uint4(v.x, v.y, v.z, v.w);
    }
}


// the workgroup's read-length histogram -> global counters (all threads; the caller puts barriers around it)
__device__ __forceinline__ void flush_hist(SeededLds &S, unsigned long long *ctr, int tid)
{
    for (int i = tid; i < KVQ_RL_BINS / 2; i += ST_THREADS) {
        const uint32_t w = S.hist[i];
        if (w & 0xFFFFu) atomicAdd(&ctr[KVQ_CTR_RL_ + 2 * i], (unsigned long long)(w & 0xFFFFu));
        if (w >> 16) atomicAdd(&ctr[KVQ_CTR_RL_ + 2 * i + 1], (unsigned long long)(w >> 16));
        S.hist[i] = 0;
    }
}

// SS = lookup stride of the anchor blocks (SeedTables::stride); STAMPS = diagnostic build that
// sums wave 0's cycles per phase (KVQ_DBG=16, tools/phase_stamps.py)
template <int SS, bool STAMPS>
__global__ void __launch_bounds__(ST_THREADS, 4)
kvq_scan_seeded(const KvqParams *__restrict__ Pg, SeedTables X, const uint8_t *__restrict__ data, int64_t fpos_base,
                const uint4 *__restrict__ tiles, uint32_t ntiles, uint32_t *__restrict__ tile_report, uint32_t dbg,
                uint32_t tile_bytes, unsigned int *__restrict__ tile_ctr)
{
    __shared__ __align__(16) SeededLds S;
    uint8_t *const lds_raw = reinterpret_cast<uint8_t *>(&S);
    int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = rfl((uint32_t)tid >> 6);
    if ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t *)lds_raw) != 0u) __builtin_trap();   // lds_byte_at
    // Wave priorities (s_setprio; building with -DKVQ_NO_PRIO leaves them out).  A wave's priority rises as it gets
    // on with its tile: 0 in the front end (fetch, flags, newline list, P2 -- the part that is made of
    // barriers anyway), 1 in the trim, 2 in the seed filter, 3 while it verifies (a chain of dependent
    // table loads: the sooner they are issued the better they hide), back to 0 for the tile-end barrier.
    // "Nearest to the end of its tile goes first" is worth 5 % over equal priorities.  The issue arbiter
    // also prefers the older wave of a SIMD, so waves 4..7 (the second wave of the workgroup on each
    // SIMD) used to reach the tile-end barrier ~2 k cycles behind waves 0..3, which then waited there:
    // they run the trim one step higher.
    const bool younger = wave >= 4u;
    HotParams P;
    P.cold = Pg; P.tab = (GlbBytes)Pg->tab; P.maxerrors = Pg->maxerrors; P.minoverlap = Pg->minoverlap;
    P.minreadlength = Pg->minreadlength; P.amin = Pg->amin;
    unsigned long long stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stamp_t = 0, wave_p34 = 0;
#define STAMP(i) do { if constexpr (STAMPS) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } } while (0)

    for (int i = tid; i < 4096; i += ST_THREADS) reinterpret_cast<uint32_t *>(S.bmA)[i] = X.bm1[i];     // bmA and bmL are adjacent
    for (int i = tid; i < KVQ_RL_BINS / 2; i += ST_THREADS) S.hist[i] = 0;
    if (tid == 0) { S.longest_p1 = 0; S.records = 0; S.fallback = 0; }
    if (tid < (int)(ST_PRE / 4)) reinterpret_cast<uint32_t *>(S.buf)[tid] = 0;      // the block in front of the tile never holds text

    // the tile's text travels HBM -> registers (one tile ahead) -> LDS.  Every thread fetches the
    // ST_BLK contiguous bytes it will scan for newlines (five 16-byte vectors; thread 0 stands for
    // the empty block in front of the tile), so the scan works on registers and needs no barrier
    // behind the LDS fill
    static_assert(ST_ROUNDS * 16u == ST_BLK, "one scan block per thread");
    static_assert(ST_WAVES == 8, "the wave totals are read as two uint4 each");
    uint4 pre[ST_ROUNDS];
    const uint32_t blk = (uint32_t)tid * ST_BLK;                           // the block's place in buf
    const uint32_t toff = blk - ST_PRE;                                    // ... and in the tile's text (tid >= 1)
    const uint32_t vo = tid ? toff : ST_NO_BLOCK;                          // the block's place in any tile's text
    if (blockIdx.x < ntiles) {
        const TileGeo J = tile_geo(blockIdx.x, tiles, tile_bytes);
        tile_load80(data, J.load_lo, J.load_hi, vo, pre);
    }
    // Tiles are handed out by a counter (*tile_ctr starts at gridDim.x: tiles 0 .. gridDim.x - 1 are the
    // workgroups' first ones): the two workgroups of a CU do not run at the same pace (the older one wins
    // the issue arbitration), and with a fixed share the slower ones would finish the launch alone.
    // A workgroup always knows its next tile (gn, whose text it fetches one tile ahead); thread 0
    // draws the one after that early in the tile and posts it in LDS before the tile-end barrier.
    if (tid == 0) S.next_tile = atomicAdd(tile_ctr, 1u);
    __syncthreads();

    uint32_t tiles_done = 0;
    uint32_t gn = rfl(S.next_tile);
    for (uint32_t g = blockIdx.x; g < ntiles; ) {
        const TileGeo J = tile_geo(g, tiles, tile_bytes);
        // the thread number is made opaque once per tile: the lane masks derived from it ("thread 0",
        // "lane 63", ...) are then worked out where they are used (one compare) instead of being kept
        // in scalar registers across the whole loop, which the kernel has run out of
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        if constexpr (STAMPS) stamp_t = __builtin_amdgcn_s_memtime();
        // every vector load has to be back here anyway; saying so on all paths keeps the
        // compiler from waiting for the next tile's loads in the middle of this tile
        __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0)

        // ---- P0 + P1a: registers -> LDS and newline flags of the thread's block; the next tile's loads go out ----
        // bytes in front of the chunk start and behind the loaded text are zeroed here (at most
        // two vectors per tile), so that nothing below needs masks.  (80-byte stride: the five
        // ds_write_b128 of a wave are bank-conflict free; ownership ends on a block boundary, so a
        // block is owned entirely or not at all)
        const uint32_t own_end_l = J.own_end - J.g0 + ST_PRE;              // ownership ends here
        const uint32_t end_l = J.load_hi - J.g0 + ST_PRE;                  // end of the loaded text
        uint32_t fl[ST_BLK / 4]; uint32_t cnt = 0;
        // only the block with the chunk's first byte (when that is not 16-byte aligned) and the block with
        // its last byte need masks; a wave without such a block stores and flags its vectors as they
        // came (thread 0 and the threads behind the text hold zeros, which land in the empty block
        // in front of the tile and in the slack behind it)
        const uint32_t blk_lo = J.g0 + toff;
        const bool edge = tid && (blk_lo < J.own_begin || (blk_lo < J.load_hi && J.load_hi < blk_lo + ST_BLK && (J.load_hi & 15u)));
        if (!__any(edge)) {
#pragma unroll
            for (int r = 0; r < (int)ST_ROUNDS; r++) {
                const uint4 v = pre[r];
                *reinterpret_cast<uint4 *>(&S.buf[blk + 16u * r]) = v;
                fl[4 * r + 0] = kvq_nl_flags(v.x); fl[4 * r + 1] = kvq_nl_flags(v.y);
                fl[4 * r + 2] = kvq_nl_flags(v.z); fl[4 * r + 3] = kvq_nl_flags(v.w);
            }
        } else
#pragma unroll
        for (int r = 0; r < (int)ST_ROUNDS; r++) {
            const uint32_t off = toff + 16u * r;                                  // relative to g0
            const uint32_t gp = J.g0 + off;
            uint4 v = pre[r];
            if (tid && gp < J.load_hi) {
                if (gp < J.own_begin || gp + 16u > J.load_hi) {
                    uint32_t x[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const uint32_t keep = kvq_range_flags(gp + 4u * d, J.own_begin, J.load_hi);   // 0x80 per byte inside
                        x[d] &= (keep >> 7) * 0xFFu;
                    }
                    v = make_uint4(x[0], x[1], x[2], x[3]);
                }
                *reinterpret_cast<uint4 *>(&S.buf[ST_PRE + off]) = v;
            } else v = make_uint4(0, 0, 0, 0);
            fl[4 * r + 0] = kvq_nl_flags(v.x); fl[4 * r + 1] = kvq_nl_flags(v.y);
            fl[4 * r + 2] = kvq_nl_flags(v.z); fl[4 * r + 3] = kvq_nl_flags(v.w);
        }
        if (gn < ntiles) {
            const TileGeo N = tile_geo(gn, tiles, tile_bytes);
            tile_load80(data, N.load_lo, N.load_hi, vo, pre);
        }
        // the block's 80 flag bits in three words (all that crosses the barrier)
        static_assert(ST_BLK == 80u, "five vectors per block");
        uint32_t m0 = kvq_flags16(fl[0], fl[1], fl[2], fl[3]) | (kvq_flags16(fl[4], fl[5], fl[6], fl[7]) << 16);
        uint32_t m1 = kvq_flags16(fl[8], fl[9], fl[10], fl[11]) | (kvq_flags16(fl[12], fl[13], fl[14], fl[15]) << 16);
        uint32_t m2 = kvq_flags16(fl[16], fl[17], fl[18], fl[19]);
        cnt = (uint32_t)(__popc(m0) + __popc(m1) + __popc(m2));
        const uint32_t incl = kvq_wave_incl_scan(cnt);
        if (lane == 63) S.wtot[wave] = incl;
        STAMP(0);
        __syncthreads();
        STAMP(1);
        uint32_t n_all = 0;
        {
            // lanes 0..7 hold the eight wave totals; three DPP adds make their running sums, two lane
            // reads pick this wave's and the last one (no per-wave masks to keep in scalar registers)
            uint32_t run = S.wtot[lane & 7];
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x111, 0xf, 0xf, false);
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x112, 0xf, 0xf, false);
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x114, 0xf, 0xf, false);
            n_all = (uint32_t)__builtin_amdgcn_readlane((int)run, 7);
            const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)run, (int)wave);      // ... including this wave
            uint32_t n = upto - (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) + incl - cnt;
            // newlines the tile owns = those up to the end of its last owned block (own_end_l is a block
            // boundary or the end of the text): the thread of that block knows the count
            if (blk < own_end_l && blk + ST_BLK >= own_end_l) S.n_owned = n + cnt;
            if (__any(cnt != 0u)) {
                // one loop over the block's (few) set bits
                while (__any((m0 | m1 | m2) != 0u)) {
                    const bool in0 = m0 != 0u, in1 = m1 != 0u;
                    const uint32_t w = in0 ? m0 : in1 ? m1 : m2;
                    if (w) {
                        const uint32_t pos = blk + (in0 ? 0u : in1 ? 32u : 64u) + (uint32_t)(__ffs((int)w) - 1);
                        if (n < ST_NLCAP) S.nl[n] = (uint16_t)pos;
                        n++;
                        const uint32_t w1 = w & (w - 1u);
                        if (in0) m0 = w1; else if (in1) m1 = w1; else m2 = w1;
                    }
                }
            }
        }
        __syncthreads();
        STAMP(2);
        // the tile after next: drawn here, behind the last barrier before the long barrier-free stretch
        // (a barrier waits for outstanding atomics), wanted at the end of this tile
        // (by the last wave, which holds the fewest reads: the compiler waits for the answer on the spot)
        uint32_t drawn = 0;
        if (tid == ST_THREADS - 64) drawn = atomicAdd(tile_ctr, 1u);

        // ---- P2 (every wave, redundantly): which records does this tile own? ----
        uint32_t nrec = 0, jn = TR_NONE;
        {
            const uint32_t n_nl = n_all < ST_NLCAP ? n_all : ST_NLCAP;
            const uint32_t n_owned = rfl(S.n_owned);
            uint32_t fallback = n_all > ST_NLCAP ? 1u : 0u;
            // a record belongs to the tile that owns the '\n' in front of it (the chunk's
            // first record to tile 0), also when its first byte is the next tile's first
            if (J.t == 0) jn = 0;                                            // chunk start: exact
            else {
                // lane m (m >= 1): does the line behind the tile's m-th newline start with '@'
                // and the line two further on with '+'?
                const uint32_t m = (uint32_t)lane;
                bool ok = false;
                if (m >= 1 && m <= 8 && m <= n_owned && m + 2 <= n_nl) {
                    const uint32_t ls0 = (uint32_t)S.nl[m - 1] + 1u;
                    const uint32_t ls2 = (uint32_t)S.nl[m + 1] + 1u;
                    ok = ls0 < end_l && ls2 < end_l && S.buf[ls0] == '@' && S.buf[ls2] == '+';
                }
                const uint64_t mk = __ballot(ok);
                if (mk) jn = (uint32_t)(__ffsll((long long)mk) - 1);
            }
            // records owned by the tile start behind newline jn + 4k (k >= 0) while that newline is owned
            if (jn != TR_NONE) {
                if (jn <= n_owned) nrec = (n_owned - jn) / 4u + 1u;
                // keep only records whose four newlines were loaded; a missing one means
                // chunk end (partial record, dropped: 1033) or a record longer than the look-ahead
                if (nrec > 0 && jn + 4u * nrec > n_nl) {
                    const uint32_t fit = n_nl >= jn ? (n_nl - jn) / 4u : 0u;
                    if (J.load_hi < J.b || n_all > ST_NLCAP) fallback = 1u;
                    nrec = fit;
                }
                if (nrec > ST_RCAP) { nrec = ST_RCAP; fallback = 1u; }
            }
            if (tid == 0) {
                tile_report[g] = (n_owned & 0xFFFFu) | ((jn & 0xFFu) << 16) | (fallback ? TR_FLAG_FALLBACK : 0u);
                S.records += nrec;
            }
        }

        if (dbg & 32u) nrec = 0;                                     // diagnostic: front end only
        STAMP(3);
        if (younger) KVQ_SETPRIO(2); else KVQ_SETPRIO(1);
        unsigned long long wave_t3 = 0;
        if constexpr (STAMPS) wave_t3 = __builtin_amdgcn_s_memtime();
        // ---- P3 / P4 passes: reads -> candidates, then candidates -> hits ----
        // G lanes share one read (G = 4 for 150 bp reads): each lane scans a contiguous
        // slice of the score line / of the bases serially, so that one wave instruction
        // advances 64/G reads and hardly anything runs on the scalar unit
        const int64_t tile_fpos = fpos_base + (int64_t)J.g0;
        // the widest group (a power of two, at most 64 lanes) that still gives every read of the
        // tile its own lanes in one pass: G * nrec <= 512, i.e. lg = 9 - ceil(log2(nrec))
        static_assert(ST_THREADS == 512, "lg = 9 - ceil(log2(nrec))");
        const int lg_ = 9 - (nrec > 1u ? 32 - __builtin_clz(nrec - 1u) : 0);
        const uint32_t lg = lg_ < 0 ? 0u : lg_ > 6 ? 6u : (uint32_t)lg_;
        const uint32_t G = 1u << lg, RP = ST_THREADS >> lg;
        const uint32_t gl = (uint32_t)tid & (G - 1u), gr = (uint32_t)tid >> lg;
        for (uint32_t pass0 = 0; pass0 < nrec; pass0 += RP) {
            const uint32_t k = pass0 + gr;
            const bool have = k < nrec;
            uint32_t roff = 0; int rl = 0;
            if (have) {
                const uint32_t m = jn + 4u * k;
                const uint32_t rstart = m == 0 ? ST_PRE + (J.a - (J.a & ~15u)) : (uint32_t)S.nl[m - 1] + 1u;
                const uint32_t n0 = S.nl[m], n1 = S.nl[m + 1], n2 = S.nl[m + 2], n3 = S.nl[m + 3];
                const uint32_t sread = n0 + 1u, plus = n1 + 1u, sscore = n2 + 1u;
                if (gl == 0) {
                    const uint32_t c0 = S.buf[rstart], cp = S.buf[plus];
                    if (c0 != '@') atomicMin(Pg->err, ((unsigned long long)(tile_fpos + rstart - ST_PRE) << 16) | (0ull << 8) | c0);
                    else if (cp != '+') atomicMin(Pg->err, ((unsigned long long)(tile_fpos + plus - ST_PRE) << 16) | (1ull << 8) | cp);
                }
                // quality trim (1055-1068): this lane's slice of the score line -> bitmask of good
                // bytes (SWAR, a dword at a time) -> longest run by shifts, no per-byte branching
                const int Q = (int)(n3 - sscore);                   // the closing '\n' is implied
                const int per = (Q + (int)G - 1) >> lg;
                Seg sg; sg.beg = (int)mul_u24(gl, (uint32_t)per); if (sg.beg > Q) sg.beg = Q;    // (24-bit multiply: full rate)
                int s1 = sg.beg + per; if (s1 > Q) s1 = Q;
                sg.len = 0; sg.pre = 0; sg.suf = 0; sg.best = 0; sg.bstart = sg.beg;
                {
                    const uint32_t addk = (uint32_t)(0x80 - P.amin) * 0x01010101u;
                    // one round of at most 64 score bytes; `more`: the slice may need more than the first twelve dwords
                    auto round = [&](int c0, bool more) -> Seg {
                        const int n = s1 - c0 < 64 ? s1 - c0 : 64;
                        const uint32_t abs0 = sscore + (uint32_t)c0, abs1 = abs0 + (uint32_t)n;
                        uint32_t w = abs0 & ~3u;
                        const int lead = (int)(abs0 - w);                  // bytes of the first dword in front of the slice
                        // dwords w, w+4, ... cover the slice (bytes behind the slice are masked off below,
                        // the buffer has slack behind its end)
                        uint64_t m = 0;
                        int sh = -lead;
                        {
                            // the first 48 bytes in one go (twelve loads travel together; a 150 bp read's
                            // slice needs no more), whatever is left in rounds of 16
                            uint32_t q[12], g16[3];
#pragma unroll
                            for (int t = 0; t < 12; t++) q[t] = buf_u32(w + 4u * t);
#pragma unroll
                            for (int u = 0; u < 3; u++)
                                g16[u] = kvq_flags16(good_flags(q[4 * u], addk), good_flags(q[4 * u + 1], addk),
                                                     good_flags(q[4 * u + 2], addk), good_flags(q[4 * u + 3], addk));
                            // 48 flag bits side by side, then one shift drops the bytes in front of the slice
                            m = (((uint64_t)g16[2] << 32) | (uint64_t)(g16[0] | (g16[1] << 16))) >> lead;
                            w += 48u; sh += 48;
                        }
                        if (more)
                        for (; w < abs1; w += 16u, sh += 16) {
                            uint32_t q[4];
#pragma unroll
                            for (int t = 0; t < 4; t++) q[t] = buf_u32(w + 4u * t);
                            const uint32_t g16 = kvq_flags16(good_flags(q[0], addk), good_flags(q[1], addk), good_flags(q[2], addk), good_flags(q[3], addk));
                            m |= sh >= 0 ? ((uint64_t)g16 << sh) : ((uint64_t)g16 >> (-sh));
                        }
                        const uint64_t nmask = (!more || n < 64) ? (1ull << n) - 1ull : ~0ull;      // (one round of twelve dwords: n <= 45)
                        m &= nmask;
                        Seg sub; sub.beg = c0; sub.len = n;
                        int bl, bs;
                        uint64_t zz = ~m & nmask;                          // the bad bytes of the slice
                        if (!__any(__popcll(zz) > 4) && !(dbg & 4u)) {
                            // the usual case, few bad bytes in any lane's slice: walk them (runs = the gaps between them)
                            // (branch-free: a lane that has run out of bad bytes sees "one at n", which closes its last run)
                            int prev = 0, first = n, p; bl = 0; bs = 0;
                            do {
                                const uint32_t plo = (uint32_t)(__ffs((int)(uint32_t)zz) - 1), ph = (uint32_t)(__ffs((int)(uint32_t)(zz >> 32)) - 1);
                                const uint32_t phi = ph > 0xFFFFFFDFu ? 0xFFFFFFFFu : ph + 32u;
                                uint32_t pm = plo < phi ? plo : phi; if (pm > (uint32_t)n) pm = (uint32_t)n;
                                p = (int)pm;
                                zz &= zz - 1ull;
                                first = first < p ? first : p;
                                const int gap = p - prev;
                                if (gap > bl) { bl = gap; bs = prev; }
                                prev = p < n ? p + 1 : prev;
                            } while (__any(p < n));
                            sub.pre = first; sub.suf = n - prev;
                        } else {
                            const uint64_t inv = ~m;
                            sub.pre = inv ? __ffsll((long long)inv) - 1 : 64; if (sub.pre > n) sub.pre = n;
                            const uint64_t top = ~(m << (64 - n));         // leading ones of the n-bit mask = trailing run
                            sub.suf = top ? __clzll((long long)top) : 64; if (sub.suf > n) sub.suf = n;
                            longest_run64(m, n, bl, bs);
                        }
                        sub.best = bl; sub.bstart = c0 + bs;
                        return sub;
                    };
                    if (!__any(per > 45)) {
                        // the usual case: every lane's slice fits the first twelve dwords (45 bytes at any alignment)
                        sg = round(sg.beg, false);
                    } else {
                        for (int c0 = sg.beg; c0 < s1; c0 += 64) {         // one round per 64 bytes of the slice
                            const Seg sub = round(c0, true);
                            sg = (c0 == sg.beg) ? sub : seg_merge(sg, sub);
                        }
                    }
                }
                // ordered tree merge over the G lanes of the read
                if (G == 4u) {
                    // the common group width: neighbours inside a quad, by DPP (no LDS traffic)
#define KVQ_QUAD(v, ctl) __builtin_amdgcn_update_dpp(0, (v), (ctl), 0xf, 0xf, true)       // (every lane has a source: no "old" value to set up)
                    {
                        Seg B;                                                       // lane ^ 1: quad_perm [1,0,3,2]
                        B.len = KVQ_QUAD(sg.len, 0xB1); B.pre = KVQ_QUAD(sg.pre, 0xB1); B.suf = KVQ_QUAD(sg.suf, 0xB1);
                        B.best = KVQ_QUAD(sg.best, 0xB1); B.bstart = KVQ_QUAD(sg.bstart, 0xB1); B.beg = 0;
                        sg = seg_merge(sg, B);            // (every lane merges -- only what lanes 0 and 2 of the quad make of it is used)
                    }
                    {
                        Seg B;                                                       // lane ^ 2: quad_perm [2,3,0,1]
                        B.len = KVQ_QUAD(sg.len, 0x4E); B.pre = KVQ_QUAD(sg.pre, 0x4E); B.suf = KVQ_QUAD(sg.suf, 0x4E);
                        B.best = KVQ_QUAD(sg.best, 0x4E); B.bstart = KVQ_QUAD(sg.bstart, 0x4E); B.beg = 0;
                        sg = seg_merge(sg, B);            // (... lane 0)
                    }
                    rl = KVQ_QUAD(sg.best, 0x00);                                   // lane 0 of the quad: quad_perm [0,0,0,0]
                    roff = sread + (uint32_t)KVQ_QUAD(sg.bstart, 0x00);              // 1070
#undef KVQ_QUAD
                } else {
                    for (uint32_t d = 1; d < G; d <<= 1) {
                        Seg B;
                        B.len = __shfl_xor(sg.len, (int)d, 64); B.pre = __shfl_xor(sg.pre, (int)d, 64); B.suf = __shfl_xor(sg.suf, (int)d, 64);
                        B.best = __shfl_xor(sg.best, (int)d, 64); B.bstart = __shfl_xor(sg.bstart, (int)d, 64); B.beg = 0;
                        if ((gl & d) == 0) sg = seg_merge(sg, B);
                    }
                    rl = __shfl(sg.best, lane & ~(int)(G - 1u), 64);
                    roff = sread + (uint32_t)__shfl(sg.bstart, lane & ~(int)(G - 1u), 64);             // 1070
                }
                if (gl == 0) {
                    if (rl < KVQ_RL_BINS) atomicAdd(&S.hist[rl >> 1], 1u << (16 * (rl & 1)));          // 394-402
                    atomicMax(&S.longest_p1, (uint32_t)(rl + 1));
                    S.rinfo[k] = roff | ((uint32_t)rl << 16);
                }
            }
            STAMP(4);
            KVQ_SETPRIO(2);
            // filter + verify, wave by wave: a wave's reads (64/G of them), its candidates and its work
            // items are its own (queue segments wave * ST_QW / wave * ST_Q2W, counts in scalar
            // registers), so nothing between here and the end of the tile waits for another wave.
            // Normally one stretch; when a queue overflows (dense tables, hit-rich reads) the
            // stretch is redone in halves
            const uint32_t rpw = 64u >> lg, grw = (uint32_t)lane >> lg;        // reads per wave, this lane's read within the wave
            const uint32_t wfirst = pass0 + wave * rpw;                        // the wave's first read of this pass
            const uint32_t npass = wfirst < nrec ? (nrec - wfirst < rpw ? nrec - wfirst : rpw) : 0u;
            uint2 *const q1 = S.q1 + wave * ST_QW; uint32_t *const q2 = S.q2 + wave * ST_Q2W;
            uint32_t sub = 0, step = rpw;
            while (sub < npass) {
                const bool mine = have && rl >= P.minreadlength && !(dbg & 2u) && grw >= sub && grw - sub < step;       // 1100
                uint32_t qn = 0;                                                // candidates queued by this wave (uniform)
                // seed filter: the read's 8-mers at positions 0, SS, 2 SS, ... are looked up in the LDS
                // bitmap of anchor blocks (the G lanes of a read share the positions); its e+1 head
                // blocks (positions 8j) and tail blocks (rl - 8(j+1)) in the bitmap of all sequence 8-mers
                int e0 = 0, e1 = 0;
                if (mine) {
                    const int NPe = (rl - SK) / SS + 1;
                    const int per = (NPe + (int)G - 1) >> lg;
                    e0 = (int)mul_u24(gl, (uint32_t)per); if (e0 > NPe) e0 = NPe;
                    e1 = e0 + per; if (e1 > NPe) e1 = NPe;
                }
                const int me_ = P.maxerrors;
                // is the 8-mer at read position pp anywhere in a sequence?
                auto fixed_block = [&](int pp, bool ok) -> bool {
                    const uint32_t code = lds_code8x2(S, roff + (uint32_t)(ok ? pp : 0)) >> 1;
                    return ok && ((S.bmL[code >> 3] >> (code & 7u)) & 1u);
                };
                // lane j of the group takes head block j and tail block j (a tail block that is also a
                // head block counts as head block only); their candidates join the first round's push
                auto head_ok = [&](int jj) { return mine && jj <= me_ && (jj + 1) * SK <= rl; };
                auto tail_ok = [&](int jj) { const int pp = rl - (jj + 1) * SK; return mine && jj <= me_ && pp >= 0 && !((pp % SK) == 0 && pp <= me_ * SK); };
                const bool hhit = fixed_block((int)gl * SK, head_ok((int)gl));
                const bool thit = fixed_block(rl - ((int)gl + 1) * SK, tail_ok((int)gl));
                constexpr int NR = SS == 8 ? 6 : 48 / SS;                        // lookups per lane and round
                uint32_t pkb = 0;
                if constexpr (SS != 8) {
                    // the G lanes of a read pack its bases to 2 bits each, 16 per dword, into the record's own
                    // score line (dead after the trim); the overlapping 8-mers then come out of the packed
                    // words with one v_alignbit each
                    uint32_t sscore_l = 0; int nw = 0;
                    if (mine) {
                        sscore_l = (uint32_t)S.nl[jn + 4u * k + 2u] + 1u;         // score line of this record (all lanes of the group agree)
                        nw = (rl + 15) >> 4;
                    }
                    pkb = (sscore_l + 3u) & ~3u;
                    // (three words per lane and round: their fifteen loads travel together)
                    for (int i0 = (int)gl; __any(i0 < nw); i0 += 3 * (int)G) {
                        uint32_t d[3][5]; const uint32_t sh8 = (roff & 3u) * 8u;
#pragma unroll
                        for (int u = 0; u < 3; u++) {
                            const int i = i0 + u * (int)G;
                            const uint32_t w = (roff + 16u * (uint32_t)(i < nw ? i : 0)) & ~3u;
#pragma unroll
                            for (int t = 0; t < 5; t++) d[u][t] = buf_u32(w + 4u * t);
                        }
#pragma unroll
                        for (int u = 0; u < 3; u++) {
                            const int i = i0 + u * (int)G;
                            const uint32_t c01 = code8x2_of(__builtin_amdgcn_alignbit(d[u][1], d[u][0], sh8), __builtin_amdgcn_alignbit(d[u][2], d[u][1], sh8));
                            const uint32_t c23 = code8x2_of(__builtin_amdgcn_alignbit(d[u][3], d[u][2], sh8), __builtin_amdgcn_alignbit(d[u][4], d[u][3], sh8));
                            if (i < nw) *reinterpret_cast<uint32_t *>(&S.buf[pkb + 4u * (uint32_t)i]) = (c01 >> 1) | (c23 << 15);
                        }
                    }
                }
                bool first = true;
                for (int ee = e0; __any(ee < e1); ee += NR, first = false) {    // one round unless a slice exceeds NR lookups
                    const bool act = ee < e1;
                    uint32_t hA = 0;                                             // bit j: lookup ee + j met an anchor code
                    if constexpr (SS == 8) {
                        // blocks do not overlap: codes straight from the text, 48 bytes (13 dwords) per round
                        const uint32_t src = roff + 8u * (uint32_t)(act ? ee : 0), w = src & ~3u, sh8 = (src & 3u) * 8u;
                        uint32_t D[2 * NR + 1];
#pragma unroll
                        for (int t = 0; t < 2 * NR + 1; t++) D[t] = buf_u32(w + 4u * t);
                        uint32_t c2[NR], bb[NR];
#pragma unroll
                        for (int j = 0; j < NR; j++) {
                            c2[j] = code8x2_of(__builtin_amdgcn_alignbit(D[2 * j + 1], D[2 * j], sh8),
                                               __builtin_amdgcn_alignbit(D[2 * j + 2], D[2 * j + 1], sh8));
                            bb[j] = lds_byte_at(KVQ_LDS_BMA + (c2[j] >> 4));
                        }
                        asm volatile("" ::: "memory");                          // all bitmap bytes are on their way before the first is looked at
#pragma unroll
                        for (int j = 0; j < NR; j++) hA |= __builtin_amdgcn_ubfe(bb[j], (c2[j] >> 1) & 7u, 1u) << j;
                    } else {
                        const uint32_t bit = act ? 2u * SS * (uint32_t)ee : 0u;  // packed stream: 2 bits per base
                        const uint32_t wi = bit >> 5, bo = bit & 31u;
                        uint32_t W[5];
#pragma unroll
                        for (int t = 0; t < 5; t++) W[t] = buf_u32(pkb + 4u * (wi + (uint32_t)t));
                        const uint32_t R0 = __builtin_amdgcn_alignbit(W[1], W[0], bo), R1 = __builtin_amdgcn_alignbit(W[2], W[1], bo),
                                       R2 = __builtin_amdgcn_alignbit(W[3], W[2], bo), R3 = __builtin_amdgcn_alignbit(W[4], W[3], bo);
                        // the same stream half a word on: a 16-bit code that starts in the upper half of R[i] lies whole in H[i]
                        const uint32_t H0 = __builtin_amdgcn_alignbit(R1, R0, 16), H1 = __builtin_amdgcn_alignbit(R2, R1, 16),
                                       H2 = __builtin_amdgcn_alignbit(R3, R2, 16);
                        // four instructions and a byte from LDS per lookup: byte index and bit index straight out of
                        // the word that holds the code (v_bfe), bit, pile up; twelve bitmap bytes travel together
                        // (left alone the compiler waits for each)
                        constexpr int NB = 12;
#pragma unroll
                        for (int j0 = 0; j0 < NR; j0 += NB) {
                            uint32_t bi[NB], bb[NB];
#pragma unroll
                            for (int u = 0; u < NB; u++) {
                                const int b = 2 * SS * (j0 + u), wj = b >> 5, o = b & 31;
                                const uint32_t word = o <= 16 ? (wj == 0 ? R0 : wj == 1 ? R1 : R2) : (wj == 0 ? H0 : wj == 1 ? H1 : H2);
                                const uint32_t off = (uint32_t)(o <= 16 ? o : o - 16);                 // the code is bits off .. off + 15 of word
                                bi[u] = __builtin_amdgcn_ubfe(word, off, 3u);
                                bb[u] = lds_byte_at(KVQ_LDS_BMA + __builtin_amdgcn_ubfe(word, off + 3u, 13u));
                            }
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int u = 0; u < NB; u++) hA |= __builtin_amdgcn_ubfe(bb[u], bi[u], 1u) << (j0 + u);
                        }
                    }
                    const int nv = act ? (e1 - ee < NR ? e1 - ee : NR) : 0;
                    hA &= (1u << nv) - 1u;                                      // nv <= 24
                    const bool hh = first && hhit, th = first && thit;
                    // one queue reservation per wave and round, then every lane writes its own candidates
                    // (read | position << 16, kind << 16; P4a adds the code)
                    const uint32_t c = (uint32_t)__popc(hA) + (hh ? 1u : 0u) + (th ? 1u : 0u);
                    const uint32_t inc = kvq_wave_incl_scan(c);
                    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    if (tot) {
                        uint32_t idx = qn + inc - c;
                        while (hA) {
                            const int j = __ffs((int)hA) - 1; hA &= hA - 1u;
                            if (idx < ST_QW) q1[idx] = make_uint2(k | ((uint32_t)(SS * (ee + j)) << 16), roff + (uint32_t)(SS * (ee + j)));   // beyond the cap: dropped, the stretch is redone in halves
                            idx++;
                        }
                        if (hh) { if (idx < ST_QW) q1[idx] = make_uint2(k | ((gl * SK) << 16), (1u << 16) | (roff + gl * SK)); idx++; }
                        if (th && idx < ST_QW) { const uint32_t pt = (uint32_t)(rl - ((int)gl + 1) * SK); q1[idx] = make_uint2(k | (pt << 16), (1u << 16) | (roff + pt)); }
                        qn += tot;
                    }
                }
                // groups narrower than e+1 lanes: the remaining head and tail blocks, one push round each
                for (int t = (int)G; t <= me_; t += (int)G) {
                    const int jj = t + (int)gl;
#pragma unroll
                    for (int side = 0; side < 2; side++) {
                        const int pp = side ? rl - (jj + 1) * SK : jj * SK;
                        const bool hit = fixed_block(pp, side ? tail_ok(jj) : head_ok(jj));
                        const uint64_t mm = __ballot(hit);
                        if (mm) {
                            const uint32_t idx = qn + (uint32_t)__popcll(mm & kvq_lanemask_lt());
                            if (hit && idx < ST_QW) q1[idx] = make_uint2(k | ((uint32_t)pp << 16), (1u << 16) | (roff + (uint32_t)pp));
                            qn += (uint32_t)__popcll(mm);
                        }
                    }
                }
                STAMP(5);
                KVQ_SETPRIO(3);

                // ---- P4a: one candidate per lane: index range -> (candidate, entry) work items ----
                const bool over1 = qn > ST_QW;                            // candidates were dropped
                const uint32_t qn_ok = (over1 || (dbg & 1u)) ? 0u : qn;
                uint32_t q2n = 0;                                         // work items queued by this wave (uniform)
                for (uint32_t q0 = 0; q0 < qn_ok; q0 += 64u) {
                    const uint32_t qi = q0 + lane;
                    uint32_t e0 = 0, ne = 0;
                    if (qi < qn_ok) {
                        const uint2 cd = q1[qi];
                        const uint32_t code = lds_code8(S, cd.y & 0xFFFFu);            // (the 8-mer's place came with the candidate: one LDS round trip less)
                        const uint32_t *st = (cd.y >> 16) ? X.start_all : X.start_anc;
                        e0 = st[code]; ne = st[code + 1u] - e0;
                    }
                    const uint32_t inc = kvq_wave_incl_scan(ne);
                    const uint32_t base = q2n + inc - ne;
                    for (uint32_t j = 0; j < ne; j++)
                        if (base + j < ST_Q2W) q2[base + j] = (qi << 22) | (e0 + j);
                    q2n += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                }
                const bool over = over1 || q2n > ST_Q2W;
                if (over && step > 1u) { step >>= 1; continue; }
                if (over && lane == 0) S.fallback = 1u;                   // one read floods the queues: the batch goes to the exhaustive kernels

                // ---- P4b: one work item per lane ----
                {
                    const uint32_t q2n_ok = over ? 0u : q2n;
                    for (uint32_t i0 = 0; i0 < q2n_ok; i0 += 64u) {
                        const uint32_t ii = i0 + lane;
                        const bool active = ii < q2n_ok;
                        uint32_t rec = 0, kind = 0; int p = 0; uint64_t en = 0;
                        if (active) {
                            const uint32_t it = q2[ii];
                            const uint2 cd = q1[it >> 22];
                            rec = cd.x & 0xFFFFu; p = (int)(cd.x >> 16); kind = cd.y >> 16;
                            en = (kind ? X.ent_all : X.ent_anc)[it & 0x3FFFFFu];
                        }
                        verify_item(P, S, active, rec, p, kind, en, tile_fpos, SS);
                    }
                }
                KVQ_SETPRIO(2);
                sub += step;
            }
            KVQ_SETPRIO(0);
            STAMP(6);
        }
        if constexpr (STAMPS) wave_p34 += __builtin_amdgcn_s_memtime() - wave_t3;
        // everyone is done with the tile's text before the next tile's fill
        if (tid == ST_THREADS - 64) S.next_tile = drawn;
        __syncthreads();
        STAMP(7);
        if (tid == 0 && S.fallback) { atomicOr(&tile_report[g], TR_FLAG_FALLBACK); S.fallback = 0; }
        g = gn; gn = rfl(S.next_tile);
        if (++tiles_done == ST_HIST_TILES) {                      // (uniform: every thread counts the same tiles)
            flush_hist(S, Pg->ctr, tid);
            tiles_done = 0;
            __syncthreads();
        }
    }

    unsigned long long *const ctr = Pg->ctr;
    if constexpr (STAMPS) {
        if (tid == 0) for (int i = 0; i < 8; i++) atomicAdd(&ctr[KVQ_CTR_RL_ + 900 + i], stamp_acc[i]);
        // every wave: cycles from the end of P2 to the end of its own P4 (bins 908 + wave)
        if (lane == 0) atomicAdd(&ctr[KVQ_CTR_RL_ + 908 + wave], wave_p34);
    }
    // ---- flush per-workgroup counters ----
    flush_hist(S, ctr, tid);
    if (tid == 0) {
        if (S.longest_p1) atomicMax(&ctr[KVQ_CTR_LONGEST_], (unsigned long long)S.longest_p1);
        if (S.records) atomicAdd(&ctr[KVQ_CTR_RECORDS_], (unsigned long long)S.records);
    }
}

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
kvq_expand_tiles(uint32_t nchunks, const uint32_t *__restrict__ chunk_off, const uint32_t *__restrict__ tile_first, uint4 *__restrict__ tile_tab, unsigned int *__restrict__ redo_count)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && redo_count) *redo_count = 0;             // (records that skipped tiles leave are counted afresh for this launch)
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

