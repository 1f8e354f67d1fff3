// kvarq_amd/csrc/kernels_bp.hip -- the fused seed-filter scan on bit-planes, the scan kernel (kernels_seeded.hip
// holds the seed index, the tile geometry and the tile reports; kvq_launch.hip launches it).
//
// Why planes: round 1's kernel kept the tile's text in LDS (80 KB per workgroup, 117 VGPRs), which
// capped a CU at 16 waves, and it was latency bound (profiles/round1_pmc_sq_counters.txt: waves
// parked 48 % of their cycles, vector pipes 38 % busy).  Here every byte of the text is looked at
// once, in the registers it was fetched into, and LDS holds three bits per byte: "score >= Amin"
// (1 bit) and the 2-bit base code (byte >> 1) & 3.  That is 15 KB instead of 40 KB per tile, so a
// workgroup fits 40 KB of LDS / 64 VGPRs and a CU holds four of them: 32 waves, 8 per SIMD.
//
//   P0  every thread turns the 80 contiguous bytes it fetched (one tile ahead, as in v1) into
//       80 newline flags (registers), 80 good-score bits and 160 code bits (LDS)
//   P1  workgroup prefix sum -> sorted newline offsets in LDS        (as v1)
//   P2  first record of the tile, speculated and validated after the kernel   (as v1; the '@' / '+'
//       bytes it looks at come from global memory: the text is not in LDS)
//   P3  G lanes per read: quality trim = longest run of ones in a bit range of the good plane
//       (workhorse.c:1055-1068), histogram; seed filter: the read's packed bases ARE a bit range
//       of the code plane, each 8-mer code is one v_alignbit away
//   P4  candidates -> (candidate, index entry) items -> first 16 bases compared as 2-bit codes
//       (plane against a 2-bit copy of the table: equal bytes have equal codes, so this only ever
//       rejects), survivors byte-exact against the text in global memory (1112-1174)
#include "kvq_host.h"

// a comment line in the ISA text (tools/isa_marks.py counts the instructions between two of them)
#ifndef KVQ_TEXT_AUX
#define KVQ_TEXT_AUX 0          // cache policy of the tile's text loads (experiments: 1 sc0, 2 nt, 16 sc1)
#endif
#define KVQ_MARK(name) asm volatile("; KVQMARK " name)
#define BP_WIN (ST_BUF + ST_BLK)   // bytes of text a tile's planes cover: 512 blocks of 80
#define BP_Q1_KIND 25               // bit of a candidate word that says "fixed block, all-positions index"
#define BP_QW 96                   // candidates per wave and stretch
#define BP_Q2W 96                  // work items per wave and stretch
#define BP_NLCAP 1920              // newlines per window (beyond: the tile raises its fallback flag)
#define BP_P2_LATE 0xFFFEu          // "wave 0 could not tell the tile's first record"
#define BP_HEAD 2048u              // bytes of raw text kept for P2
#define BP_QCAP (BP_QW * ST_WAVES)
#define BP_Q2CAP (BP_Q2W * ST_WAVES)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

struct BpLds {
    uint32_t cdp[BP_WIN / 16 + 8];       // code plane: 2 bits per byte, byte o of the window in bits 2o, 2o+1
    uint32_t gdp[BP_WIN / 32 + 8];       // good plane: 1 bit per byte (score byte >= Amin)
    uint16_t nl[BP_NLCAP];               // window offsets of every '\n', ascending
    uint8_t  bmA[8192];                  // one bit per 8-mer code: an anchor block of some sequence
    uint32_t hist[KVQ_RL_BINS / 2];      // read-length histogram, two 16-bit bins per word
    union {
        struct {
            uint32_t q1[BP_QCAP];        // candidate: read (9 bits) | position in the read << 9 (16 bits: a read may fill the window) | kind << 25
            uint32_t q2[BP_Q2CAP];       // work item: candidate << 22 | index entry
        };
        uint32_t nlp[BP_WIN / 32 + 8];   // newline plane, 1 bit per byte: lives from P0 to the newline list (P1), the queues from P3 on
    };
    uint32_t rinfo[ST_RCAP];             // read offset in the window | rl << 16
    __attribute__((aligned(16))) uint8_t head[BP_HEAD];   // the first bytes of the window as text (P2 looks at line starts there)
    __attribute__((aligned(16))) uint32_t wtot[ST_WAVES];
    uint32_t longest_p1, records, fallback, n_owned, next_tile, first_tile, p2_jn;
    uint32_t sv_at[ST_WAVES], sv_end[ST_WAVES];      // each wave's chunk of the survivors' list: next free slot, end
};
static_assert(sizeof(BpLds) <= 40 * 1024, "four workgroups per CU: at most 40 KB of LDS each");
static_assert(offsetof(BpLds, cdp) == 0, "cdp[] first");
#define BP_LDS_GDP ((uint32_t)offsetof(BpLds, gdp))
#define BP_LDS_BMA ((uint32_t)offsetof(BpLds, bmA))
#define BP_LDS_NLP ((uint32_t)offsetof(BpLds, nlp))
#define BP_LDS_HEAD ((uint32_t)offsetof(BpLds, head))
static_assert(sizeof(uint32_t) * (BP_QCAP + BP_Q2CAP) >= sizeof(uint32_t) * (BP_WIN / 32 + 8), "the newline plane fits the queues it shares LDS with");

// (the kernel's only LDS object starts at LDS address 0: checked at kernel start)
__device__ __forceinline__ uint32_t lds_u32_at(uint32_t addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>((uintptr_t)addr);
}
// 32 bits of the code plane from window byte `pos` on (16 bases) / its 8-mer code
__device__ __forceinline__ uint32_t cdp32(uint32_t pos)
{
    const uint32_t a = (pos >> 2) & ~3u, sh = (pos & 15u) * 2u;
    return __builtin_amdgcn_alignbit(lds_u32_at(a + 4u), lds_u32_at(a), sh);
}
template <int KK> __device__ __forceinline__ uint32_t cdp_code(uint32_t pos) { return cdp32(pos) & ((1u << (2 * KK)) - 1u); }      // the K-mer code of the bases from window byte `pos` on

// the same for the 2-bit copy of the sequence table in global memory
typedef const __attribute__((address_space(1))) uint32_t *GlbWords;
__device__ __forceinline__ uint32_t tab2_32(GlbWords tab2, uint32_t tpos)
{
    const uint32_t w = tpos >> 4, sh = (tpos & 15u) * 2u;
    return __builtin_amdgcn_alignbit(tab2[w + 1u], tab2[w], sh);
}

// bases that differ between two words of 16 2-bit codes
__device__ __forceinline__ int diff_codes(uint32_t x, uint32_t y)
{
    const uint32_t v = x ^ y;
    return __popc((v | (v >> 1)) & 0x55555555u);
}

template <int KK>
__device__ __forceinline__ bool seed_live_bp(GlbWords tab2, uint32_t roff, int rl, int rp, uint32_t toff, int seql, int sq)
{
    if (rp < 0 || rp + KK > rl || sq < 0 || sq + KK > seql) return false;
    return cdp_code<KK>(roff + (uint32_t)rp) == (tab2_32(tab2, toff + (uint32_t)sq) & ((1u << (2 * KK)) - 1u));
}

// What is left of a work item -- one (candidate, index entry) pair = one diagonal of one read against one sequence -- once its
// sixteen bases have passed as 2-bit codes (0.02 per read, a true hit as a rule): which loops of the reference visit the diagonal,
// the byte-exact count over the whole overlap (the read's bytes come from global memory, `text` = address of window offset 0),
// the canonical discoverer, the emit.  Must be called by every lane of the wave.
struct BpHot {                       // what the verification needs of the parameters
    const KvqParams *cold; GlbBytes tab; GlbWords tab2;
    int maxerrors, minoverlap, pitch;
};
template <int KK>
__device__ __forceinline__ void bp_verify_rest(const BpHot &P, GlbBytes text, bool alive, uint32_t roff, int rl, int p, uint32_t kind,
                                               uint32_t en_lo, uint32_t en_hi, int64_t tile_fpos, int stride)
{
    bool hitAB = false, hitC = false;
    int s = 0, lenAB = 0, lenC = 0, sposAB = 0, sposC = 0; uint32_t keyAB = 0, keyC = 0;
    const int64_t fpos = tile_fpos + (int64_t)roff - (int64_t)ST_PRE;
    if (alive) {
        const int q = (int)(en_lo & 4095u);
        s = (int)(en_lo >> 12);
        const uint32_t toff = en_hi & 0xFFFFFu;
        const int seql = (int)(en_hi >> 20);
        const int mo = P.minoverlap, me = P.maxerrors;
        const int d = q - p;                             // sequence index = read index + d
        const int a = d < 0 ? -d : 0;
        const int L = (rl < seql - d ? rl : seql - d) - a;
        // which reference loops visit this diagonal
        bool canAB = false, canC = false;
        const bool guard = rl > mo && seql > mo;
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
        if (canAB || canC) {
            // byte-exact mismatch count of the whole overlap
            int mism = 0, j = 0;
            const GlbBytes x = text + roff + (uint32_t)a, y = P.tab + toff + (uint32_t)(a + d);
            // (eight bytes a step while they last: what comes here is a true hit as a rule, the whole overlap gets compared, and a step is one
            // trip to memory -- four bytes a step made a 150-base hit 38 dependent trips, for which the other seven waves waited at the tile's end;
            // sixteen a step cost the kernel four registers it does not have)
            for (; j + 8 <= L && mism <= me; j += 8) {
                typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                typedef u32x2_t __attribute__((aligned(1))) u32x2_any;
                const u32x2_t xv = *reinterpret_cast<const __attribute__((address_space(1))) u32x2_any *>(x + j);
                const u32x2_t yv = *reinterpret_cast<const __attribute__((address_space(1))) u32x2_any *>(y + j);
                mism += diff_bytes(xv.x, yv.x) + diff_bytes(xv.y, yv.y);
            }
            for (; j + 4 <= L && mism <= me; j += 4) mism += diff_bytes(glb_u32(x + j), glb_u32(y + j));
            for (; j < L && mism <= me; j++) mism += (x[j] != y[j]);
            if (mism <= me) {
                // canonical discoverer: no live seed earlier in the order
                // [ALL-index read blocks by position] then [ANCHOR blocks by number]
                bool earlier = false;
                for (int jj = 0; jj <= me && !earlier; jj++) {
                    const int ph = jj * KK, pt = rl - (jj + 1) * KK;
                    if (ph + KK <= rl && (kind == 0u || ph < p)) earlier = seed_live_bp<KK>(P.tab2, roff, rl, ph, toff, seql, ph + d);
                    if (!earlier && pt >= 0 && (kind == 0u || pt < p)) earlier = seed_live_bp<KK>(P.tab2, roff, rl, pt, toff, seql, pt + d);
                }
                if (kind == 0u) {
                    for (int jj = 0; jj <= me && !earlier; jj++)
                        for (int sft = 0; sft < stride && !earlier; sft++) {
                            const int o = jj * P.pitch + sft;
                            if (o < q && ((o - d) & (stride - 1)) == 0) earlier = seed_live_bp<KK>(P.tab2, roff, rl, o - d, toff, seql, o);
                        }
                }
                if (!earlier) { hitAB = canAB; hitC = canC; }
            }
        }
    }
    emit_cold(P.cold, hitAB, fpos, s, sposAB, lenAB, rl, keyAB);
    emit_cold(P.cold, hitC, fpos, s, sposC, lenC, rl, keyC);
}

// one 16-byte vector of text -> 16 newline bits, 16 good bits, 32 code bits
__device__ __forceinline__ void bp_vector(const uint4 v, uint32_t addk, uint32_t &nl16, uint32_t &g16, uint32_t &c32)
{
    const uint32_t x[4] = { v.x, v.y, v.z, v.w };
    uint32_t nf[4], gf[4], cc[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        nf[d] = kvq_nl_flags(x[d]);
        gf[d] = good_flags(x[d], addk);
        cc[d] = __builtin_amdgcn_udot4(x[d] & 0x06060606u, 0x40100401u, 0u, false);     // twice the 8-bit code of the four bytes
    }
#ifdef KVQ_ABL_JUNK
    // (ablation builds only: KVQ_ABL_JUNK extra vector instructions per 16 bytes of text -- what one more instruction costs in place)
    {
        uint32_t junk = x[0];
#pragma unroll
        for (int i = 0; i < KVQ_ABL_JUNK; i++) {
#if KVQ_ABL_SLOW == 2
            { uint32_t sj = 0; asm volatile("s_add_u32 %0, %0, 1" : "+s"(sj)); asm volatile("" :: "s"(sj)); }
#elif KVQ_ABL_SLOW == 3
            { uint32_t lj; asm volatile("ds_read_b32 %0, %1" : "=v"(lj) : "v"(junk & 0x3FCu)); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm volatile("" :: "v"(lj)); }
#elif KVQ_ABL_SLOW
            asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(junk) : "v"(x[1]));
#else
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(junk) : "v"(x[1]));
#endif
        }
        asm volatile("" :: "v"(junk));
    }
#endif
    nl16 = kvq_flags16(nf[0], nf[1], nf[2], nf[3]);
    g16 = kvq_flags16(gf[0], gf[1], gf[2], gf[3]);
    c32 = ((cc[0] | (cc[1] << 8)) >> 1) | ((cc[2] | (cc[3] << 8)) << 15);
}

// everything the kernel needs lives in one block of device memory; the kernel reads a field where it is
// wanted (scalar loads) instead of holding thirty kernel arguments in scalar registers for the whole
// launch: at eight waves per SIMD a wave has 80 of them, and what does not fit is kept in VGPR lanes
struct BpArgs {
    KvqParams P;
    SeedTables X;
    const uint8_t *data; int64_t fpos_base;
    const uint4 *tiles; uint32_t *tile_report; unsigned int *tile_ctr;
    void *redo;                       // KvqRedo: where a read that floods its wave's queues is put for the exhaustive matcher
    unsigned int *fail;               // the batch's fail word (bit 1: such reads exist)
    void *surv;                       // KvqSurvivors: work items that passed the 16-base test, verified behind the kernel (kvq_verify_survivors)
    uint32_t ntiles, tile_bytes, dbg, pad_, redo_cap;
    uint32_t surv_cap;                // slots of the survivors' list (read from here, not from the list's header: that line is busy with the slot counter's atomics)
};
// Tiles are handed out by counters.  One counter for the whole launch is a ceiling by itself: a word in
// memory takes about 88 atomic adds per microsecond (MI355X_MICROARCH.md, "dequeue"), i.e. 3.5 TB/s of
// 40 KB tiles.  So the tiles are dealt into BP_SHARDS contiguous shares with a counter each (on a
// 128-byte line of its own); a workgroup starts in share blockIdx.x % BP_SHARDS -- workgroups with equal
// blockIdx.x % 8 tend to sit on one XCD, so a share is streamed through one L2 -- and moves on to the
// next share when its own has run out.
#define BP_SHARDS 64u
#define BP_SHARD_STRIDE 32u          // counters are this many words apart
__host__ __device__ static inline uint32_t bp_shard_begin(uint32_t sh, uint32_t ntiles) { return (uint32_t)(((uint64_t)sh * ntiles) / BP_SHARDS); }
// one lane: the next tile of share `sh`, or of the shares behind it; ntiles when there is none left
__device__ __forceinline__ uint32_t bp_draw(unsigned int *ctr, uint32_t ntiles, uint32_t &sh, uint32_t &left)
{
    while (left) {
        const uint32_t t = atomicAdd(&ctr[sh * BP_SHARD_STRIDE], 1u);
        if (t < bp_shard_begin(sh + 1u, ntiles)) return t;
        sh = sh + 1u == BP_SHARDS ? 0u : sh + 1u; left--;
    }
    return ntiles;
}
// the whole wave: the next tile of share `sh` or, when that has run out, of the first share behind it that has tiles left;
// ntiles when there is none.  All BP_SHARDS counters are looked at with ONE load (lane i reads the counter of share sh + i)
// and an atomic goes only to a share that still has tiles: when the text runs out a workgroup learns it from one round
// trip, not from sixty-four atomics one behind the other (which was a fixed 80 us at the end of every launch).
static_assert(BP_SHARDS == 64u, "one counter per lane of the drawing wave");
// The shares behind `sh` are tried in an order of the workgroup's own (sh + m i for its odd m): with a common order all
// the workgroups whose shares run out together -- and they do: a grid that is no multiple of 64 leaves half the shares
// with one workgroup more -- would fall on the same next share, empty it at once and move on as a herd (measured with
// 992 workgroups: + 11 % kernel time).
__device__ __forceinline__ uint32_t bp_draw_wave(unsigned int *ctr, uint32_t ntiles, uint32_t &sh, uint32_t lane)
{
    const uint32_t m = (((uint32_t)blockIdx.x >> 6) * 2u + 37u) | 1u;
    for (;;) {
        const uint32_t mine = (sh + m * lane) & (BP_SHARDS - 1u);
        const uint32_t seen = __hip_atomic_load(&ctr[mine * BP_SHARD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t open = __ballot(seen < bp_shard_begin(mine + 1u, ntiles));
        if (open == 0) return ntiles;
        const uint32_t pick = (sh + m * (uint32_t)(__ffsll((long long)open) - 1)) & (BP_SHARDS - 1u);
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&ctr[pick * BP_SHARD_STRIDE], 1u);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        sh = pick;
        if (t < bp_shard_begin(pick + 1u, ntiles)) return t;
        // (another workgroup took the share's last tile in between: look again -- every round closes a share for good)
    }
}
typedef const __attribute__((address_space(4))) BpArgs *BpArgsPtr;
// the block's address, opaque to the compiler from here on: loads through it are issued where they are
// written, not hoisted to the top of the kernel
__device__ __forceinline__ BpArgsPtr bp_args(const BpArgs *A)
{
    BpArgsPtr p = (BpArgsPtr)A;
    asm volatile("" : "+s"(p));
    return p;
}

#ifndef BP_OCC
#define BP_OCC 8                   // waves per SIMD the kernel is built for (8: four workgroups per CU, 64 VGPRs)
#endif

// longest run of ones in the n (0..64) low bits of m (bits from n on are zero): summary of the slice
// for the merge over lanes (first-longest-wins, workhorse.c:1062)
__device__ __forceinline__ void bp_runs64(uint64_t m, int n, uint32_t dbg, int &pre, int &suf, int &best, int &bstart)
{
    const uint64_t nmask = n < 64 ? (1ull << n) - 1ull : ~0ull;
    uint64_t zz = ~m & nmask;                                 // the bad bytes of the slice
    const int c = __popc((uint32_t)zz) + __popc((uint32_t)(zz >> 32));      // (as 32-bit counts: the comparisons below stay 32-bit ones)
    if (!__any(c > 3) && !(dbg & 4u)) {
        // the usual case, at most three bad bytes in any lane's slice: their positions in closed form
        // (first, second and last one), the runs are the four gaps between them
        // (v_ffbl / v_ffbh answer -1 for zero by themselves: written as `__ffs(x) - 1` or `__clz` they come with a 64-bit compare and
        // selects, or an exec-mask branch, around them -- a third of this block's instructions)
        const uint32_t lo = (uint32_t)zz, hi = (uint32_t)(zz >> 32);
        auto ffbl = [](uint32_t x) -> uint32_t { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; };
        auto ffbh = [](uint32_t x) -> uint32_t { uint32_t r; asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(x)); return r; };
        auto ctz64 = [&](uint32_t l, uint32_t h) -> uint32_t {        // n when there is no bit
            const uint32_t pl = ffbl(l), ph = ffbl(h) | 32u;
            const uint32_t r = pl < ph ? pl : ph; return r < (uint32_t)n ? r : (uint32_t)n;
        };
        const uint32_t p1 = ctz64(lo, hi);
        const uint64_t z2 = zz & (zz - 1ull);
        const uint32_t p2 = ctz64((uint32_t)z2, (uint32_t)(z2 >> 32));
        // one behind the last bad byte (0 when there is none): 64 - leading zeros
        const uint32_t ch = ffbh(hi), cl = ffbh(lo) | 32u;
        uint32_t lz = ch < cl ? ch : cl; lz = lz < 64u ? lz : 64u;
        const uint32_t e = 64u - lz;
        const int g0 = (int)p1, g1 = (int)p2 - (int)p1 - 1, g2 = (int)e - (int)p2 - 2, g3 = n - (int)e;
        int bl = g0, bs = 0;
        if (g1 > bl) { bl = g1; bs = (int)p1 + 1; }
        if (g2 > bl) { bl = g2; bs = (int)p2 + 1; }
        if (g3 > bl) { bl = g3; bs = (int)e; }
        pre = g0; suf = g3; best = bl; bstart = bs;
    } else if (!__any(c > 8) && !(dbg & 4u)) {
        // a few more: walk them (branch-free: a lane that has run out of bad bytes sees "one at n", which closes its last run)
        int prev = 0, first = n, p, bl = 0, bs = 0;
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
        pre = first; suf = n - prev; best = bl; bstart = bs;
    } else {
        const uint64_t inv = ~m;
        pre = inv ? __ffsll((long long)inv) - 1 : 64; if (pre > n) pre = n;
        const uint64_t top = ~(m << (64 - n));                 // leading ones of the n-bit mask = trailing run
        suf = top ? __clzll((long long)top) : 64; if (suf > n) suf = n;
        longest_run64(m, n, best, bstart);
    }
}

// LG >= 0: the lane group of a read is 1 << LG lanes wide whatever the tile holds (a tile with more than
// 512 >> LG reads takes several passes); LG < 0: the widest group that gives every read of the tile its
// own lanes in one pass, worked out per tile.  The launch picks LG = 2 when that is what the records of
// the text ask for (100 to 250 bases), the general kernel otherwise.
// DIAG: the kernel honours the KVQ_DBG switches (ablations, forced paths); the production instantiations do not carry them
// KK: the seed length the table's index was built with (kvq_seed_k: 8 at the product settings, 5 to 7 where (maxerrors + 1) * 8
// does not fit the shortest accepted overlap)
// DENSE: the instantiation for tables and settings that give candidates by the dozen per read (kvq_seed_index_build estimates them from the
// bitmaps' occupancy: seeds shorter than 8, the MTBC table x 8 and beyond): the queues are drained as they fill instead of the stretch being
// halved and filtered again
template <int SS, int LG, bool STAMPS, bool DIAG = STAMPS, int KK = 8, bool DENSE = (KK < 8)>
__global__ void __launch_bounds__(ST_THREADS, DENSE ? 8 : BP_OCC)
kvq_scan_bp(const BpArgs *__restrict__ A_)
{
    __shared__ __align__(16) BpLds S;
    uint8_t *const lds_raw = reinterpret_cast<uint8_t *>(&S);
    int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = rfl((uint32_t)tid >> 6);
    if ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t *)lds_raw) != 0u) __builtin_trap();   // lds_u32_at
    unsigned long long stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stamp_t = 0, wave_p34 = 0;
    // (instrumented build: the workgroup's life on the constant 100 MHz clock -- entry, end of the prologue, end of its first tile, exit)
    unsigned long long rt_in = 0, rt_pro = 0, rt_first = 0;
    // (build with -DKVQ_TALLY, instrumented instantiation: what P4 eats -- per-lane tallies, summed into spare counter slots at the end:
    // 0 anchor candidates, 1 fixed-block candidates, 2 work items (index entries tested), 3 items that passed the 16-base test,
    // 4 rounds of the work-item loop (per wave), 5 candidate batches (per wave), 6 stretches (per wave), 7 calls of the byte-exact part (per wave))
    uint32_t tally[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#ifdef KVQ_TALLY          // (a build of its own: the tallies cost the instrumented kernel registers it does not have)
#define BTALLY(i, cond) do { if constexpr (STAMPS) { if (cond) tally[i]++; } } while (0)
#else
#define BTALLY(i, cond) ((void)0)
#endif
    if constexpr (STAMPS) rt_in = __builtin_amdgcn_s_memrealtime();
#define BSTAMP(i) do { if constexpr (STAMPS) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } } while (0)

    uint32_t ntiles, tile_bytes, dbg, amin;
    {
        const BpArgsPtr A = bp_args(A_);
        ntiles = A->ntiles; tile_bytes = A->tile_bytes; dbg = DIAG ? A->dbg : 0u; amin = (uint32_t)A->P.amin;
        const GlbWords bm1 = (GlbWords)A->X.bm1;
        for (int i = tid; i < (1 << (2 * KK)) / 32; i += ST_THREADS) reinterpret_cast<uint32_t *>(S.bmA)[i] = bm1[i];
    }
    // the drawing wave (the last one, which holds the fewest reads) and its share of the tiles
    uint32_t my_shard = blockIdx.x % BP_SHARDS; bool no_more = false;
    if (wave == ST_WAVES - 1u) {
        unsigned int *const ctr = bp_args(A_)->tile_ctr;
        const uint32_t t0 = bp_draw_wave(ctr, ntiles, my_shard, (uint32_t)lane);
        const uint32_t t1 = t0 < ntiles ? bp_draw_wave(ctr, ntiles, my_shard, (uint32_t)lane) : ntiles;
        no_more = t1 >= ntiles;
        if (lane == 0) { S.first_tile = t0; S.next_tile = t1; }
    }
    for (int i = tid; i < KVQ_RL_BINS / 2; i += ST_THREADS) S.hist[i] = 0;
    if (tid == 0) { S.longest_p1 = 0; S.records = 0; S.fallback = 0; }
    if (tid < (int)ST_WAVES) { S.sv_at[tid] = 0; S.sv_end[tid] = 0; }
    if (tid < 8) { S.cdp[BP_WIN / 16 + tid] = 0; S.gdp[BP_WIN / 32 + tid] = 0; }    // slack behind the planes

    static_assert(ST_ROUNDS * 16u == ST_BLK, "one scan block per thread");
    static_assert(ST_WAVES == 8, "eight wave totals");
    // The tile's text travels HBM -> registers -> planes.  The loads are coalesced: wave w covers the
    // window bytes [5120 w, 5120 (w + 1)), in round r lane l takes vector (16 bytes) 320 w + 64 r + l
    // of the window, so one load instruction of a wave fetches 1 KiB of consecutive text.  A vector's
    // three plane pieces (16 newline bits, 16 good bits, 32 code bits) go to its place in the planes;
    // the newline list then needs every thread's own 80 consecutive newline bits: thread t re-reads
    // bits [80 t, 80 t + 80) of the newline plane, which its own wave has written.
    __syncthreads();

    // (experiment, KVQ_STAGGER: the workgroups that share a compute unit start a fraction of a tile apart, so that one's wait
    // for its text meets another's arithmetic: workgroup b sits in slot b / 256 of its compute unit under round-robin dispatch)
    if (const uint32_t stg = bp_args(A_)->pad_) { for (uint32_t i = 0; i < (blockIdx.x >> 8) * stg; i++) __builtin_amdgcn_s_sleep(127); }
    if constexpr (STAMPS) rt_pro = __builtin_amdgcn_s_memrealtime();
    unsigned long long *const ctr_stamps = bp_args(A_)->P.ctr + KVQ_CTR_RL_ + 924;      // (instrumented build only; copy 0 of the staged counters)
    const uint32_t addk = (0x80u - amin) * 0x01010101u;
    uint32_t tiles_done = 0;
    // The barrier that ends a tile stands at the top of the next one, BEHIND the issue of that tile's loads:
    // registers are free there, and the wait for the slowest wave hides the way to memory and back.
    uint32_t g_done = 0xFFFFFFFFu;                                          // the tile this workgroup has just finished
    uint32_t my_longest = 0;                                                // 1 + the longest read this lane has trimmed
    for (uint32_t g = rfl(S.first_tile); ; ) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        const uint32_t blk = (uint32_t)tid * ST_BLK;                           // the thread's newline block in the window
        const uint32_t wv = wave * (ST_BLK * 64u / 16u) + (uint32_t)lane;      // the lane's vector of round 0
        if constexpr (STAMPS) stamp_t = __builtin_amdgcn_s_memtime();
        // tile geometry (the per-tile table kvq_expand_tiles wrote: chunk begin, chunk end, tile number)
        uint32_t Ja = 0, Jb = 0, Jt = 0, g0 = 0; GlbBytes text = nullptr;
        uint4 pre[ST_ROUNDS];
#pragma unroll
        for (int r = 0; r < (int)ST_ROUNDS; r++) pre[r] = make_uint4(0, 0, 0, 0);      // (no "value of the last tile" for the compiler to keep alive through the whole tile)
        if (g < ntiles) {
            const BpArgsPtr A = bp_args(A_);
            // (diagnostic 64: every tile scans the text of one of the first 64 tiles -- the same work from L2 instead of HBM; results are wrong)
            const u32x4_t q = ((const __attribute__((address_space(4))) u32x4_t *)A->tiles)[(dbg & 64u) ? (g & 63u) : g];
            Ja = q.x; Jb = q.y; Jt = q.z;
            g0 = (Ja & ~15u) + Jt * tile_bytes;
            const uint8_t *const data = A->data;
            text = (GlbBytes)data + g0 - ST_PRE;                               // window offset 0 (never read below ST_PRE)
            const uint32_t load_hi = g0 + ST_TILE + ST_OV < Jb ? g0 + ST_TILE + ST_OV : Jb;
            // byte offset of the lane's vectors in the tile's text (the window starts ST_PRE bytes in front of
            // the text: vectors 0..4 never hold text, an offset beyond any tile makes their load return zeros;
            // all of the offset in the checked part: the range check leaves the scalar offset out)
            const uint32_t vo = 16u * wv - ST_PRE, vo0 = wv >= ST_PRE / 16u ? vo : ST_NO_BLOCK;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(data + g0), 0, (int)(((load_hi + 15u) & ~15u) - g0), 0x00020000);
#pragma unroll
            for (int r = 0; r < (int)ST_ROUNDS; r++) {
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(r ? vo + 1024u * r : vo0), 0, KVQ_TEXT_AUX);
                pre[r] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
        // everyone is done with the last tile's planes; its loose ends
        __syncthreads();
        unsigned long long sub_t0 = 0, sub_t1 = 0;
        if constexpr (STAMPS) sub_t0 = __builtin_amdgcn_s_memtime();          // (instrumented build: the first phase in three parts -- the wait at this barrier, the wait for the text, P0 + the scan)
        if (g_done != 0xFFFFFFFFu) {
            if (tid == 0 && S.fallback) { atomicOr(&bp_args(A_)->tile_report[g_done], TR_FLAG_FALLBACK); S.fallback = 0; }
        }
        if (g >= ntiles) break;
        const uint32_t gn = rfl(S.next_tile);
        const uint32_t own_end = g0 + tile_bytes < Jb ? g0 + tile_bytes : Jb;
        const uint32_t load_hi = g0 + ST_TILE + ST_OV < Jb ? g0 + ST_TILE + ST_OV : Jb;
        const uint32_t own_begin_l = (Jt == 0 ? Ja : g0) - g0 + ST_PRE;       // window offsets: first owned byte,
        const uint32_t own_end_l = own_end - g0 + ST_PRE;                     // ownership ends here,
        const uint32_t end_l = load_hi - g0 + ST_PRE;                         // end of the loaded text

        // ---- P0: registers -> planes; the thread's 80 newline flags ----
        // bytes in front of the chunk's first byte (when that is not 16-byte aligned) and behind its last
        // one are zeroed in place by the wave whose stretch holds them (rare: scalar test)
        {
            const uint32_t w_lo = wave * (ST_BLK * 64u), w_hi = w_lo + ST_BLK * 64u;
            const bool cut = ((own_begin_l & 15u) && own_begin_l > w_lo && own_begin_l < w_hi) || ((end_l & 15u) && end_l > w_lo && end_l < w_hi);
            if (cut) {
                uint32_t wv_ = wv;
                asm volatile("" : "+v"(wv_));                               // (worked out here, not hoisted out of the tile loop into twenty spilled registers)
#pragma unroll
                for (int r = 0; r < (int)ST_ROUNDS; r++) {
                    const uint32_t o = 16u * (wv_ + 64u * r);               // window offset of the vector
                    uint32_t x[4] = { pre[r].x, pre[r].y, pre[r].z, pre[r].w };
#pragma unroll
                    for (int d = 0; d < 4; d++) x[d] &= (kvq_range_flags(o + 4u * d, own_begin_l, end_l) >> 7) * 0xFFu;
                    pre[r] = make_uint4(x[0], x[1], x[2], x[3]);
                }
            }
        }
        if constexpr (STAMPS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); sub_t1 = __builtin_amdgcn_s_memtime(); }
        KVQ_MARK("P0 vectors");
        {
            const uint32_t pa = 2u * wv, ca = 4u * wv;
#pragma unroll
            for (int r = 0; r < (int)ST_ROUNDS; r++) {
                uint32_t n16, g16, c32;
                bp_vector(pre[r], addk, n16, g16, c32);
                *reinterpret_cast<__attribute__((address_space(3))) uint16_t *>((uintptr_t)(BP_LDS_NLP + pa + 128u * r)) = (uint16_t)n16;
                *reinterpret_cast<__attribute__((address_space(3))) uint16_t *>((uintptr_t)(BP_LDS_GDP + pa + 128u * r)) = (uint16_t)g16;
                *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)(ca + 256u * r)) = c32;
            }
            // the head of the window as text, for P2 (wave 0 holds it)
            if (wave == 0u) {
#pragma unroll
                for (int r = 0; r < (int)(BP_HEAD / 1024u); r++)
                    *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>((uintptr_t)(BP_LDS_HEAD + 16u * (uint32_t)lane + 1024u * r)) = u32x4_t{ pre[r].x, pre[r].y, pre[r].z, pre[r].w };
            }
        }
        KVQ_MARK("P0 vectors end");
        // this thread's 80 newline bits: bits [80 t, 80 t + 80) of the newline plane = its bytes 10 t ..
        uint32_t m0, m1, m2;
        {
            const uint32_t a = BP_LDS_NLP + ((10u * (uint32_t)tid) & ~3u), sh = ((uint32_t)tid & 1u) * 16u;
            const uint32_t d0 = lds_u32_at(a), d1 = lds_u32_at(a + 4u), d2 = lds_u32_at(a + 8u);
            m0 = __builtin_amdgcn_alignbit(d1, d0, sh); m1 = __builtin_amdgcn_alignbit(d2, d1, sh); m2 = (d2 >> sh) & 0xFFFFu;
        }
        static_assert(ST_BLK == 80u, "five vectors per block");
        const uint32_t cnt = (uint32_t)(__popc(m0) + __popc(m1) + __popc(m2));
        const uint32_t incl = kvq_wave_incl_scan(cnt);
        if (lane == 63) S.wtot[wave] = incl;
        KVQ_MARK("P0 scan end");
        if constexpr (STAMPS) { if (tid == 0) { atomicAdd(&ctr_stamps[0], sub_t0 - stamp_t); atomicAdd(&ctr_stamps[1], sub_t1 - sub_t0); } }
        BSTAMP(0);
        __syncthreads();
        KVQ_MARK("P1b");
        BSTAMP(1);
        uint32_t n_all = 0;
        {
            uint32_t run = S.wtot[lane & 7];
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x111, 0xf, 0xf, false);
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x112, 0xf, 0xf, false);
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x114, 0xf, 0xf, false);
            n_all = (uint32_t)__builtin_amdgcn_readlane((int)run, 7);
            const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)run, (int)wave);      // ... including this wave
            uint32_t n = upto - (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) + incl - cnt;
            if (blk < own_end_l && blk + ST_BLK >= own_end_l) S.n_owned = n + cnt;
            if (__any(cnt != 0u)) {
                // (the newlines of sequencer records lie 21, 151, 2, 151 bytes apart: no thread has more than two in its 80 bytes.  Then the
                // first and the second one come in closed form -- lowest set bit of the 80, lowest of what is left -- in two thirds of the
                // instruction time of two rounds of the general loop below, which picks the first non-empty word with compares and selects)
                if (n_all <= BP_NLCAP && !__any(cnt > 2u)) {
                    auto low80 = [](uint32_t w0, uint32_t w1, uint32_t w2) -> uint32_t {     // (an empty word gives 0xFFFFFFFF, with or without the OR)
                        // (the instruction itself answers -1 for zero; `__ffs(x) - 1` is compiled with a compare and a select around it)
                        auto ffbl = [](uint32_t x) -> uint32_t { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; };
                        const uint32_t f0 = ffbl(w0), f1 = ffbl(w1) | 32u, f2 = ffbl(w2) | 64u;
                        const uint32_t f01 = f0 < f1 ? f0 : f1;
                        return f01 < f2 ? f01 : f2;
                    };
                    const uint32_t p1 = low80(m0, m1, m2);
                    const uint32_t k0 = m0 & (m0 - 1u), k1 = m0 ? m1 : m1 & (m1 - 1u), k2 = (m0 | m1) ? m2 : m2 & (m2 - 1u);
                    const uint32_t p2 = low80(k0, k1, k2);
                    if (cnt) {
                        S.nl[n] = (uint16_t)(blk + p1);
                        if (cnt > 1u) S.nl[n + 1u] = (uint16_t)(blk + p2);
                    }
                    m0 = m1 = m2 = 0u;
                }
                while (__any((m0 | m1 | m2) != 0u)) {
                    const bool in0 = m0 != 0u, in1 = m1 != 0u;
                    const uint32_t w = in0 ? m0 : in1 ? m1 : m2;
                    if (w) {
                        const uint32_t pos = blk + (in0 ? 0u : in1 ? 32u : 64u) + (uint32_t)(__ffs((int)w) - 1);
                        if (n < BP_NLCAP) S.nl[n] = (uint16_t)pos;
                        n++;
                        const uint32_t w1 = w & (w - 1u);
                        if (in0) m0 = w1; else if (in1) m1 = w1; else m2 = w1;
                    }
                }
            }
        }
        // Which record is the tile's first?  Speculated from the text: an '@' line followed two lines later by
        // a '+' line, among the first eight lines (P2 below turns it into the records the tile owns, and
        // kvq_validate_tiles checks it against the exact newline count after the kernel).  The ten newlines
        // this looks at nearly always lie in wave 0's stretch, whose entries of the list are its own: it works
        // the answer out here, once, and the other seven waves read it behind the barrier.
        if (wave == 0u) {
            const uint32_t cnt0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t early = BP_P2_LATE;
            if (Jt != 0 && cnt0 >= 10u) {
                const uint32_t m = (uint32_t)lane;
                bool ok = false;
                if (m >= 1 && m <= 8) {
                    const uint32_t e0 = S.nl[m - 1], e2 = S.nl[m + 1];
                    const uint32_t ls0 = e0 + 1u, ls2 = e2 + 1u;
                    if (e0 < own_end_l && ls0 < end_l && ls2 < end_l) {
                        const uint32_t c0 = ls0 < BP_HEAD ? lds_byte_at(BP_LDS_HEAD + ls0) : (uint32_t)text[ls0];
                        const uint32_t c2 = ls2 < BP_HEAD ? lds_byte_at(BP_LDS_HEAD + ls2) : (uint32_t)text[ls2];
                        ok = c0 == '@' && c2 == '+';
                    }
                }
                const uint64_t mk = __ballot(ok);
                early = mk ? (uint32_t)(__ffsll((long long)mk) - 1) : TR_NONE;
            }
            if (lane == 0) S.p2_jn = early;
        }
        __syncthreads();
        KVQ_MARK("P1b end / P2");
        BSTAMP(2);

        // ---- P2 (every wave, redundantly): which records does this tile own? ----
        uint32_t nrec = 0, jn = TR_NONE, drawn;                     // (drawn: the drawing lane's only)
        {
            const uint32_t n_nl = n_all < BP_NLCAP ? n_all : BP_NLCAP;
            const uint32_t n_owned = rfl(S.n_owned);
            uint32_t fallback = n_all > BP_NLCAP ? 1u : 0u;
            if (Jt == 0) jn = 0;                                            // chunk start: exact
            else if (const uint32_t early = rfl(S.p2_jn); early != BP_P2_LATE) jn = early;      // wave 0 has seen it already (below)
            else {
                // (the general form: the tile's first ten newlines did not all lie in wave 0's stretch)
                const uint32_t m = (uint32_t)lane;
                bool ok = false;
                if (m >= 1 && m <= 8 && m <= n_owned && m + 2 <= n_nl) {
                    const uint32_t ls0 = (uint32_t)S.nl[m - 1] + 1u;
                    const uint32_t ls2 = (uint32_t)S.nl[m + 1] + 1u;
                    if (ls0 < end_l && ls2 < end_l) {
                        // (line starts inside the head of the window are looked up in LDS; further on, rare, in global memory)
                        const uint32_t c0 = ls0 < BP_HEAD ? lds_byte_at(BP_LDS_HEAD + ls0) : (uint32_t)text[ls0];
                        const uint32_t c2 = ls2 < BP_HEAD ? lds_byte_at(BP_LDS_HEAD + ls2) : (uint32_t)text[ls2];
                        ok = c0 == '@' && c2 == '+';
                    }
                }
                const uint64_t mk = __ballot(ok);
                if (mk) jn = (uint32_t)(__ffsll((long long)mk) - 1);
            }
            // a tile whose LAST records do not fit its window (a record longer than the look-ahead) keeps the
            // records that do and leaves the rest; one that overflows its tables (newlines, records) keeps none.
            // What it leaves is scanned again after the kernel (kvq_collect_skipped): TR_FLAG_SKIPPED, and where the
            // first record it left begins (batch offset + 1; 0: it kept none) in the second report word
            uint32_t left = 0u;
            if (jn != TR_NONE) {
                if (jn <= n_owned) nrec = (n_owned - jn) / 4u + 1u;
                if (nrec > 0 && jn + 4u * nrec > n_nl) {
                    const uint32_t fit = n_nl >= jn ? (n_nl - jn) / 4u : 0u;
                    if (load_hi < Jb) left = 1u;
                    if (n_all > BP_NLCAP) fallback = 1u;
                    nrec = fit;
                }
                if (nrec > ST_RCAP) { nrec = ST_RCAP; fallback = 1u; }
            }
            if (fallback) { nrec = 0; left = 1u; }
            // thread 0 reports; the last wave (which holds the fewest reads) draws the tile after next: wanted
            // at the end of this tile, and the answer is not waited for before that
            if (tid == 0 || tid == ST_THREADS - 64) {
                const BpArgsPtr A = bp_args(A_);
                if (tid == 0) {
                    A->tile_report[g] = (n_owned & 0xFFFFu) | ((jn & 0xFFu) << 16) | (left ? TR_FLAG_SKIPPED : 0u) | (left && nrec ? TR_FLAG_PARTIAL : 0u);
                    if (left) A->tile_report[ntiles + g] = nrec ? g0 - ST_PRE + (uint32_t)S.nl[jn + 4u * nrec - 1u] + 2u : 0u;
                    S.records += nrec;
                } else if (!no_more) drawn = atomicAdd(&A->tile_ctr[my_shard * BP_SHARD_STRIDE], 1u);
            }
        }

        // now and then the read-length histogram goes to the global counters (16-bit bins); waves that are
        // already counting this tile's reads may add to a bin at any time: it is taken and cleared in one step
        if (++tiles_done == ST_HIST_TILES) {
            unsigned long long *const ctr = (bp_args(A_)->P.ctr + (size_t)(blockIdx.x % KVQ_STAGE_COPIES) * KVQ_STAGE_SLOTS);
            for (int i = tid; i < KVQ_RL_BINS / 2; i += ST_THREADS) {
                const uint32_t w = atomicExch(&S.hist[i], 0u);
                if (w & 0xFFFFu) atomicAdd(&ctr[KVQ_CTR_RL_ + 2 * i], (unsigned long long)(w & 0xFFFFu));
                if (w >> 16) atomicAdd(&ctr[KVQ_CTR_RL_ + 2 * i + 1], (unsigned long long)(w >> 16));
            }
            tiles_done = 0;
        }
        if (dbg & 32u) nrec = 0;                                     // diagnostic: front end only
        KVQ_MARK("P2 end / P3 setup");
        BSTAMP(3);
        if (wave >= 4u) KVQ_SETPRIO(2); else KVQ_SETPRIO(1);
        unsigned long long wave_t3 = 0;
        if constexpr (STAMPS) wave_t3 = __builtin_amdgcn_s_memtime();
        // ---- P3 / P4 passes: reads -> candidates, then candidates -> hits ----
        // G lanes share one read, a wave owns its 64/G reads from here to the end of the tile
        static_assert(ST_THREADS == 512, "lg = 9 - ceil(log2(nrec))");
        uint32_t lg;
        if constexpr (LG >= 0) lg = (uint32_t)LG;
        else {
            const int lg_ = 9 - (nrec > 1u ? 32 - __builtin_clz(nrec - 1u) : 0);
            lg = lg_ < 0 ? 0u : lg_ > 6 ? 6u : (uint32_t)lg_;
        }
        const uint32_t G = 1u << lg, RP = ST_THREADS >> lg;
        const uint32_t gl = (uint32_t)tid & (G - 1u), gr = (uint32_t)tid >> lg;
        for (uint32_t pass0 = 0; pass0 < nrec; pass0 += RP) {
            const uint32_t k = pass0 + gr;
            const bool have = k < nrec;
            uint32_t roff = 0; int rl = 0;
            uint32_t c0, cp;                                        // the record's '@' and '+' (1037-1048): fetched here by the group's first lane, looked at behind the pass
            if (have) {
                const uint32_t m = jn + 4u * k;
                const uint32_t rstart = m == 0 ? ST_PRE + (Ja & 15u) : (uint32_t)S.nl[m - 1] + 1u;
                const uint32_t n0 = S.nl[m], n1 = S.nl[m + 1], n2 = S.nl[m + 2], n3 = S.nl[m + 3];
                const uint32_t sread = n0 + 1u, plus = n1 + 1u, sscore = n2 + 1u;
        KVQ_MARK("trim");
                if (gl == 0) { if (dbg & 8u) { c0 = '@'; cp = '+'; } else { c0 = text[rstart]; cp = text[plus]; } }   // (diagnostic 8: no '@' / '+' probes -- ablation only)
                // quality trim (1055-1068): this lane's slice of the score line is a bit range of the good plane
                const int Q = (int)(n3 - sscore);                   // the closing '\n' is implied
                const int per = (Q + (int)G - 1) >> lg;
                Seg sg; sg.beg = (int)mul_u24(gl, (uint32_t)per); if (sg.beg > Q) sg.beg = Q;
                int s1 = sg.beg + per; if (s1 > Q) s1 = Q;
                sg.len = 0; sg.pre = 0; sg.suf = 0; sg.best = 0; sg.bstart = sg.beg;
                {
                    // one round of at most 64 score bits
                    auto round = [&](int c0_) -> Seg {
                        const int n = s1 - c0_ < 64 ? s1 - c0_ : 64;
                        const uint32_t bit = sscore + (uint32_t)c0_;
                        const uint32_t a = BP_LDS_GDP + ((bit >> 5) << 2), sh = bit & 31u;
                        const uint32_t d0 = lds_u32_at(a), d1 = lds_u32_at(a + 4u), d2 = lds_u32_at(a + 8u);
                        const uint64_t nmask = n < 64 ? (1ull << n) - 1ull : ~0ull;
                        const uint64_t mm = (((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32) | (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh)) & nmask;
                        Seg sub; sub.beg = c0_; sub.len = n;
                        int bs;
                        bp_runs64(mm, n, dbg, sub.pre, sub.suf, sub.best, bs);
                        sub.bstart = c0_ + bs;
                        return sub;
                    };
                    if (!__any(per > 64)) sg = round(sg.beg);
                    else {
                        for (int cc = sg.beg; cc < s1; cc += 64) {
                            const Seg sub = round(cc);
                            sg = (cc == sg.beg) ? sub : seg_merge(sg, sub);
                        }
                    }
                }
                // ordered tree merge over the G lanes of the read
                if (G == 4u) {
#define KVQ_QUAD(v, ctl) __builtin_amdgcn_update_dpp(0, (v), (ctl), 0xf, 0xf, true)
                    {
                        Seg B;                                                       // lane ^ 1: quad_perm [1,0,3,2]
                        B.len = KVQ_QUAD(sg.len, 0xB1); B.pre = KVQ_QUAD(sg.pre, 0xB1); B.suf = KVQ_QUAD(sg.suf, 0xB1);
                        B.best = KVQ_QUAD(sg.best, 0xB1); B.bstart = KVQ_QUAD(sg.bstart, 0xB1); B.beg = 0;
                        sg = seg_merge(sg, B);
                    }
                    {
                        Seg B;                                                       // lane ^ 2: quad_perm [2,3,0,1]
                        B.len = KVQ_QUAD(sg.len, 0x4E); B.pre = KVQ_QUAD(sg.pre, 0x4E); B.suf = KVQ_QUAD(sg.suf, 0x4E);
                        B.best = KVQ_QUAD(sg.best, 0x4E); B.bstart = KVQ_QUAD(sg.bstart, 0x4E); B.beg = 0;
                        sg = seg_merge(sg, B);
                    }
                    rl = KVQ_QUAD(sg.best, 0x00);
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
                // (the longest read: a running maximum per lane, put together once at the end of the kernel -- an
                // atomic maximum in LDS per read is turned into a scalar loop over the wave's lanes by the compiler)
                my_longest = my_longest > (uint32_t)(rl + 1) ? my_longest : (uint32_t)(rl + 1);
                if (gl == 0) {
                    if (rl < KVQ_RL_BINS) atomicAdd(&S.hist[rl >> 1], 1u << (16 * (rl & 1)));          // 394-402
                    S.rinfo[k] = roff | ((uint32_t)rl << 16);
                }
            }
        KVQ_MARK("trim end / filter");
            BSTAMP(4);
            KVQ_SETPRIO(2);
            // filter + verify, wave by wave (the wave's reads, candidates and work items are its own)
            const uint32_t rpw = 64u >> lg, grw = (uint32_t)lane >> lg;
            const uint32_t wfirst = pass0 + wave * rpw;
            const uint32_t npass = wfirst < nrec ? (nrec - wfirst < rpw ? nrec - wfirst : rpw) : 0u;
            uint32_t *const q1 = S.q1 + wave * BP_QW; uint32_t *const q2 = S.q2 + wave * BP_Q2W;
            uint32_t sub = (dbg & 128u) ? npass : 0u, step = rpw;             // (diagnostic 128: trim only)
            while (sub < npass) {
                BTALLY(6, lane == 0);
                // (a stretch is one turn of this loop unless a queue overflows; what depends on the read only is worked out here, where it is
                // wanted, not hoisted in front of the loop into a dozen registers and spilled wave masks: the compiler cannot know the trip count)
                asm volatile("" : "+v"(rl), "+v"(roff));
                int minrl, me_; const __attribute__((address_space(1))) uint8_t *bmL;
                {
                    const BpArgsPtr A = bp_args(A_);
                    minrl = A->P.minreadlength; me_ = A->P.maxerrors;
                    bmL = (const __attribute__((address_space(1))) uint8_t *)A->X.bm1 + (1 << (2 * KK)) / 8;
                }
                const bool mine = have && rl >= minrl && !(dbg & 2u) && grw >= sub && grw - sub < step;       // 1100
                uint32_t qn = 0;
                int e0 = 0, e1 = 0;
                if (mine) {
                    const int NPe = (rl - KK) / SS + 1;
                    const int per = (NPe + (int)G - 1) >> lg;
                    e0 = (int)mul_u24(gl, (uint32_t)per); if (e0 > NPe) e0 = NPe;
                    e1 = e0 + per; if (e1 > NPe) e1 = NPe;
                }
                // is the 8-mer at read position pp anywhere in a sequence?  (bitmap of all sequence 8-mers: global memory)
                // (the load is issued whether the block is wanted or not -- any code has its byte in the bitmap --
                // so that the two lookups of a lane travel together instead of each behind a branch of its own)
                auto fixed_block = [&](int pp, bool ok) -> bool {
                    const uint32_t code = cdp_code<KK>(roff + (uint32_t)(ok ? pp : 0));
                    const uint32_t bits = bmL[code >> 3];
                    return ok & (bool)((bits >> (code & 7u)) & 1u);
                };
                auto head_ok = [&](int jj) { return mine && jj <= me_ && (jj + 1) * KK <= rl; };
                auto tail_ok = [&](int jj) { const int pp = rl - (jj + 1) * KK; return mine && jj <= me_ && pp >= 0 && !((pp % KK) == 0 && pp <= me_ * KK); };
                const bool hhit = fixed_block((int)gl * KK, head_ok((int)gl));
                const bool thit = fixed_block(rl - ((int)gl + 1) * KK, tail_ok((int)gl));
                constexpr int NR = SS == 8 ? 6 : SS == 4 ? 12 : (LG == 3 || LG == 1) ? 24 : 18;  // lookups per lane and round (18: a 150-base read's 72 even positions over four lanes; 24: a 300-base read's 147 over eight -- 19 a lane -- or a 100-base read's 47 over two, one round instead of two)
                bool first = true;
                if constexpr (!DENSE) {
                for (int ee = e0; __any(ee < e1); ee += NR, first = false) {
                    const bool act = ee < e1;
#include "kernels_bp_lookup.inc"
                    const int nv = act ? (e1 - ee < NR ? e1 - ee : NR) : 0;
                    hA &= (1u << nv) - 1u;                                      // nv <= 24
                    const bool hh = first && hhit, th = first && thit;
                    const uint32_t c = (uint32_t)__popc(hA) + (hh ? 1u : 0u) + (th ? 1u : 0u);
                    const uint32_t inc = kvq_wave_incl_scan(c);
                    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    if (tot) {
                        uint32_t idx = qn + inc - c;
                        while (hA) {
                            const int j = __ffs((int)hA) - 1; hA &= hA - 1u;
                            if (idx < BP_QW) q1[idx] = k | ((uint32_t)(SS * (ee + j)) << 9);        // beyond the cap: dropped, the stretch is redone in halves
                            idx++;
                        }
                        if (hh) { if (idx < BP_QW) q1[idx] = k | ((gl * KK) << 9) | (1u << BP_Q1_KIND); idx++; }
                        if (th && idx < BP_QW) { const uint32_t pt = (uint32_t)(rl - ((int)gl + 1) * KK); q1[idx] = k | (pt << 9) | (1u << BP_Q1_KIND); }
                        qn += tot;
                    }
                }
                // groups narrower than e+1 lanes: the remaining head and tail blocks, one push round each
                for (int t = (int)G; t <= me_; t += (int)G) {
                    const int jj = t + (int)gl;
#pragma unroll
                    for (int side = 0; side < 2; side++) {
                        const int pp = side ? rl - (jj + 1) * KK : jj * KK;
                        const bool hit = fixed_block(pp, side ? tail_ok(jj) : head_ok(jj));
                        const uint64_t mm = __ballot(hit);
                        if (mm) {
                            const uint32_t idx = qn + (uint32_t)__popcll(mm & kvq_lanemask_lt());
                            if (hit && idx < BP_QW) q1[idx] = k | ((uint32_t)pp << 9) | (1u << BP_Q1_KIND);
                            qn += (uint32_t)__popcll(mm);
                        }
                    }
                }
                }   // !DENSE
        KVQ_MARK("filter end / P4a");
                BSTAMP(5);
                KVQ_SETPRIO(3);

                // ---- P4: candidates -> (candidate, index entry) work items, one per lane -> a window of sixteen bases of the diagonal
                // as 2-bit codes against the entry's own copy of the sequence around the seed (SeedEntry::ctx; equal bytes have equal
                // codes, so this only ever rejects, and nearly every false candidate ends here: TWO dependent loads, code -> range of
                // entries -> entry; round 3 went on to the 2-bit table for the bases) -> what is left goes through bp_verify_rest ----
                if constexpr (!DENSE) {
                const bool over = qn > BP_QW;                             // candidates were dropped
                const uint32_t qn_ok = (over || (dbg & 1u)) ? 0u : qn;
                if (over && step > 1u) { step >>= 1; continue; }
                if (over) {
                    // ONE read has more candidates than the queue holds (step == 1): it alone goes to the exhaustive matcher behind the scan (the redo's
                    // list: trimmed and counted here, KVQ_REDO_TRIMMED), its tile carries on; a full list fails the batch
                    if (mine && gl == 0) {
                        const BpArgsPtr A = bp_args(A_);
                        const KvqRedo Rd(A->redo);
                        const uint32_t boff = g0 - ST_PRE + roff;
                        bool put = false;
                        if (A->redo) {
                            if (rl >= KVQ_LONG_READ) {
                                const unsigned int i = atomicAdd(Rd.count + 1, 1u);
                                if (i < KVQ_LONG_CAP) { Rd.read_off[KVQ_REDO_CAP - 1u - i] = boff; Rd.read_len[KVQ_REDO_CAP - 1u - i] = rl; put = true; }
                            }
                            if (!put) {
                                const unsigned int i = atomicAdd(Rd.count, 1u);
                                if (i < A->redo_cap) { Rd.rec_start[i] = KVQ_REDO_TRIMMED; Rd.read_off[i] = boff; Rd.read_len[i] = rl; }
                                put = true;                                  // (beyond the cap: kvq_dev_count fails the batch)
                            }
                        }
                        if (!put) S.fallback = 1u;
                        else atomicOr(A->fail, 2u);                      // (reported as records that went through the exhaustive kernels; the next launches of this scan object use the wide grids)
                    }
                }
#include "kernels_bp_p4.inc"
                } else {
                    // ---- dense tables, short seeds: candidates by the dozen per read (round 4).  Nothing is dropped and nothing filtered twice:
                    // the lanes push what a round of lookups found as far as the queue has room, the queue is drained (the work items of its
                    // candidates verified), and the pushing goes on where it stopped; the sparse kernel halves the stretch and filters it again
                    // when a queue overflows -- up to six times per stretch with the MTBC table x 8 ----
                    uint32_t hA = 0, hpos = 0, tpos = 0; bool hh = false, th = false;
                    int ee = e0, xt = (int)G;
                    int ee_of_hA = e0;                                          // the round the pending anchor bits belong to
                    bool done = false;
                    while (!done) {
                        bool full = false;
                        while (!full) {
                            if (!__any(hA != 0u || hh || th)) {
                                // nothing pending: the next round of lookups (spelled out here, not a lambda: what a lambda captures by reference lives in scratch memory)
                                ee_of_hA = ee;
                                if (__any(ee < e1)) {
                                    const bool act = ee < e1;
                                    uint32_t hA_;
                                    {
#include "kernels_bp_lookup.inc"
                                        hA_ = hA;
                                    }
                                    const int nv = act ? (e1 - ee < NR ? e1 - ee : NR) : 0;
                                    hA = hA_ & ((1u << nv) - 1u);
                                    if (first) { hh = hhit; th = thit; hpos = gl * KK; tpos = (uint32_t)(rl - ((int)gl + 1) * KK); first = false; }
                                    ee += NR;
                                } else if (xt <= me_) {                             // groups narrower than e + 1 lanes: the remaining head and tail blocks
                                    const int jj = xt + (int)gl;
                                    hpos = (uint32_t)(jj * KK); tpos = (uint32_t)(rl - (jj + 1) * KK);
                                    hh = fixed_block((int)hpos, head_ok(jj)); th = fixed_block((int)tpos, tail_ok(jj));
                                    xt += (int)G;
                                } else { done = true; break; }
                                continue;
                            }
                            const uint32_t c = (uint32_t)__popc(hA) + (hh ? 1u : 0u) + (th ? 1u : 0u);
                            const uint32_t inc = kvq_wave_incl_scan(c);
                            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                            uint32_t idx = qn + inc - c;
                            while (hA && idx < BP_QW) {
                                const int j = __ffs((int)hA) - 1; hA &= hA - 1u;
                                q1[idx++] = k | ((uint32_t)(SS * (ee_of_hA + j)) << 9);
                            }
                            if (hh && idx < BP_QW) { q1[idx++] = k | (hpos << 9) | (1u << BP_Q1_KIND); hh = false; }
                            if (th && idx < BP_QW) { q1[idx++] = k | (tpos << 9) | (1u << BP_Q1_KIND); th = false; }
                            full = qn + tot >= BP_QW;
                            qn = qn + tot < BP_QW ? qn + tot : BP_QW;
                        }
                        {
                            const uint32_t qn_ok = (dbg & 1u) ? 0u : qn;
#include "kernels_bp_p4.inc"
                        }
                        qn = 0;
                    }
                }
                KVQ_SETPRIO(2);
                sub += step;
            }
            KVQ_SETPRIO(0);
            // the '@' / '+' checks of the pass's records (the bytes have come back long ago)
            if (have && gl == 0 && (c0 != '@' || cp != '+')) {          // (lanes that fetched nothing never look)
                const BpArgsPtr A = bp_args(A_);
                const uint32_t m = jn + 4u * k;
                const uint32_t rstart = m == 0 ? ST_PRE + (Ja & 15u) : (uint32_t)S.nl[m - 1] + 1u, plus = (uint32_t)S.nl[m + 1] + 1u;
                const int64_t tf = A->fpos_base + (int64_t)g0;
                if (c0 != '@') atomicMin(A->P.err, ((unsigned long long)(tf + rstart - ST_PRE) << 16) | (0ull << 8) | c0);
                else atomicMin(A->P.err, ((unsigned long long)(tf + plus - ST_PRE) << 16) | (1ull << 8) | cp);
            }
        KVQ_MARK("P4 end");
            BSTAMP(6);
        }
        if constexpr (STAMPS) wave_p34 += __builtin_amdgcn_s_memtime() - wave_t3;
        // everyone is done with the tile's planes before the next tile's fill
        if (wave == ST_WAVES - 1u) {
            // (the tile drawn behind P2 is the first lane's; when the share had run out, the wave looks for another one, now and here)
            uint32_t d = no_more ? ntiles : rfl(drawn);
            if (!no_more && d >= bp_shard_begin(my_shard + 1u, ntiles)) {
                d = bp_draw_wave(bp_args(A_)->tile_ctr, ntiles, my_shard, (uint32_t)tid & 63u);
                no_more = d >= ntiles;
            }
            if ((tid & 63) == 0) S.next_tile = d;
        }
        KVQ_MARK("tile end");
        BSTAMP(7);
        if constexpr (STAMPS) { if (rt_first == 0) rt_first = __builtin_amdgcn_s_memrealtime(); }
        g_done = g; g = gn;
    }

    unsigned long long rt_loop = 0;
    if constexpr (STAMPS) rt_loop = __builtin_amdgcn_s_memrealtime();
    {
        // what is left of the wave's chunk of the survivors' list is marked dead (read length 0: kvq_verify_survivors skips it)
        const uint32_t at = S.sv_at[wave], end = S.sv_end[wave];
        if (at + (uint32_t)lane < end) KvqSurvivors(bp_args(A_)->surv).item[at + (uint32_t)lane].rl = 0;
    }
    atomicMax(&S.longest_p1, my_longest);
    __syncthreads();
    unsigned long long *const ctr = (bp_args(A_)->P.ctr + (size_t)(blockIdx.x % KVQ_STAGE_COPIES) * KVQ_STAGE_SLOTS);
    if constexpr (STAMPS) {
        if (tid == 0) for (int i = 0; i < 8; i++) atomicAdd(&ctr[KVQ_CTR_RL_ + 900 + i], stamp_acc[i]);
        for (int i = 0; i < 8; i++) if (tally[i]) atomicAdd(&ctr[KVQ_CTR_RL_ + 930 + i], (unsigned long long)tally[i]);
        if (lane == 0) atomicAdd(&ctr[KVQ_CTR_RL_ + 908 + wave], wave_p34);
        if (tid == 0) {
            const unsigned long long rt_out = __builtin_amdgcn_s_memrealtime();
            atomicAdd(&ctr[KVQ_CTR_RL_ + 916], rt_pro - rt_in); atomicAdd(&ctr[KVQ_CTR_RL_ + 917], rt_first ? rt_first - rt_pro : 0ull);
            atomicAdd(&ctr[KVQ_CTR_RL_ + 918], rt_loop - rt_in); atomicAdd(&ctr[KVQ_CTR_RL_ + 919], 1ull);
            atomicMax(&ctr[KVQ_CTR_RL_ + 920], rt_out); atomicMax(&ctr[KVQ_CTR_RL_ + 921], rt_in);
            atomicMax(&ctr[KVQ_CTR_RL_ + 922], ~rt_in);              // ~(earliest entry)
            atomicAdd(&ctr[KVQ_CTR_RL_ + 923], rt_out - rt_loop);
        }
    }
    for (int i = tid; i < KVQ_RL_BINS / 2; i += ST_THREADS) {
        const uint32_t w = S.hist[i];
        if (w & 0xFFFFu) atomicAdd(&ctr[KVQ_CTR_RL_ + 2 * i], (unsigned long long)(w & 0xFFFFu));
        if (w >> 16) atomicAdd(&ctr[KVQ_CTR_RL_ + 2 * i + 1], (unsigned long long)(w >> 16));
    }
    if (tid == 0) {
        if (S.longest_p1) atomicMax(&ctr[KVQ_CTR_LONGEST_], (unsigned long long)S.longest_p1);
        if (S.records) atomicAdd(&ctr[KVQ_CTR_RECORDS_], (unsigned long long)S.records);
    }
}

// ---------------------------------------------------------------------------
// kvq_verify_survivors: the byte-exact part for what passed the scan kernel's 16-base test, a lane per work item (round 4)
// ---------------------------------------------------------------------------
// The rules are bp_verify_rest's (which loops of the reference visit the diagonal, the mismatch count over the whole overlap, the
// canonical discoverer among the seeds that find the diagonal, the emit); the read's bases come from the batch's text in global
// memory instead of the tile's planes, which are gone by now: a seed is live when the K codes (byte >> 1) & 3 of the read equal the
// sequence's -- the very codes the planes held.
// do the K bases at x and at y have equal codes?  (bounds are the caller's.)  K <= 8 bases as two (unaligned) 4-byte words a side; bytes
// behind the K-th are masked off (they exist: a newline follows the read, the table has 64 bytes of slack)
__device__ __forceinline__ bool seed_same_text(GlbBytes x, GlbBytes y, int K)
{
    const uint32_t x0 = glb_u32(x), y0 = glb_u32(y), x1 = glb_u32(x + 4), y1 = glb_u32(y + 4);
    const uint32_t m0 = K >= 4 ? 0x06060606u : (0x06060606u & ((1u << (8 * K)) - 1u));
    const uint32_t m1 = K >= 8 ? 0x06060606u : K > 4 ? (0x06060606u & ((1u << (8 * (K - 4))) - 1u)) : 0u;
    return (((x0 ^ y0) & m0) | ((x1 ^ y1) & m1)) == 0u;
}
extern "C" __global__ void __launch_bounds__(1024)
kvq_verify_survivors(KvqParams P, const uint8_t *__restrict__ data, int64_t fpos_base, const void *surv_, const unsigned int *__restrict__ fail,
                     int K, int stride, int pitch)
{
    const KvqSurvivors Sv(const_cast<void *>(surv_));
    if (fail && (*fail & 1u)) return;                          // (the batch is redone as a whole: its hits are rolled back anyway)
    const uint32_t n = Sv.count[0] < Sv.count[1] ? Sv.count[0] : Sv.count[1];      // (slots handed out, the list's size)
    const int me = P.maxerrors, mo = P.minoverlap;
    const uint32_t step = gridDim.x * blockDim.x;
    // The hits of a round take their arena slots together: the waves add their counts up in LDS, ONE add on the arena's counter for the
    // workgroup and round (300 of them on the headline input).  Two adds per wave and round -- kvq_emit's way -- were 9 400 on a word
    // that takes some 88 a microsecond: the kernel's whole 108 us.
    __shared__ uint32_t n_round, arena_at;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += step) {          // (the same rounds for every wave of the workgroup)
        if (threadIdx.x == 0) n_round = 0;
        __syncthreads();
        const uint32_t i = base + threadIdx.x;
        bool hitAB = false, hitC = false;
        int s = 0, rl = 0, lenAB = 0, lenC = 0, sposAB = 0, sposC = 0; uint32_t keyAB = 0, keyC = 0; int64_t fpos = 0;
        if (i < n) {
            const KvqSurvivor v = Sv.item[i];
            rl = (int)v.rl; const int p = (int)v.p; const uint32_t kind = v.kind;      // (rl 0: a slot of some wave's chunk that was never filled)
            fpos = fpos_base + (int64_t)v.boff;
            const int q = (int)(v.en & 4095u);
            s = (int)((v.en >> 12) & 0xFFFFFu);
            const uint32_t toff = (uint32_t)((v.en >> 32) & 0xFFFFFu);
            const int seql = (int)(v.en >> 52);
            const GlbBytes read = (GlbBytes)data + v.boff, seq = (GlbBytes)P.tab + toff;
            const int d = q - p;                             // sequence index = read index + d
            const int a = d < 0 ? -d : 0;
            const int L = (rl < seql - d ? rl : seql - d) - a;
            bool canAB = false, canC = false;
            const bool guard = rl > mo && seql > mo;
            if (d < 0) {
                const int ii = -d;
                if (ii <= rl - seql) { canC = true; lenC = seql; sposC = -ii; keyC = (2u << 30) | (uint32_t)ii; }              // 1147
                else if (guard && ii <= rl - mo) { canAB = true; lenAB = rl - ii; sposAB = -ii; keyAB = (0u << 30) | (uint32_t)(rl - mo - ii); }   // 1116
            } else if (d == 0) {
                canC = true; lenC = rl > seql ? seql : rl; sposC = 0; keyC = 2u << 30;                                         // 1147 / 1163
            } else {
                const int ii = d;
                if (guard && ii <= seql - mo && ii >= seql - rl) { canAB = true; lenAB = seql - ii; sposAB = ii; keyAB = (1u << 30) | (uint32_t)(seql - mo - ii); }   // 1130
                if (rl <= seql && ii <= seql - rl) { canC = true; lenC = rl; sposC = ii; keyC = (2u << 30) | (uint32_t)ii; }       // 1163
            }
            if ((canAB || canC) && L > 0 && rl > 0) {
                // canonical discoverer FIRST: no live seed earlier in the order [ALL-index read blocks by position] then [ANCHOR blocks by
                // number].  A true hit's diagonal is met by half a dozen seeds, so five survivors in six end here, after a load or two,
                // and only the one that will emit compares the whole overlap (the in-place form counts first: it has the bytes at hand).
                // (every seed of the order is looked at, with no exit in between: the loads of all of them are in flight together -- the one
                // survivor in six that IS the first discoverer went through nine dependent trips to memory otherwise, and its wave with it.
                // A seed that does not exist reads offset 0 of the read and of the sequence and is then left out of the answer.)
                bool earlier = false;
                auto live = [&](bool want, int rp, int sq) -> bool {
                    const bool ok = want && rp >= 0 && rp + K <= rl && sq >= 0 && sq + K <= seql;
                    return seed_same_text(read + (ok ? rp : 0), seq + (ok ? sq : 0), K) && ok;
                };
                const int sft = d & (stride - 1);                         // the one anchor offset whose read position is a probed one ((o - d) % stride == 0)
                constexpr int JJ = 4;
#pragma unroll
                for (int jj = 0; jj < JJ; jj++) {
                    const int ph = jj * K, pt = rl - (jj + 1) * K, o = jj * pitch + sft;
                    const bool in = jj <= me;
                    const bool e0 = live(in && (kind == 0u || ph < p), ph, ph + d);
                    const bool e1 = live(in && (kind == 0u || pt < p), pt, pt + d);
                    const bool e2 = live(in && kind == 0u && o < q, o - d, o);
                    earlier = earlier || e0 || e1 || e2;
                }
                for (int jj = JJ; jj <= me && !earlier; jj++) {           // (more than three errors allowed: the rest of the order, one seed at a time)
                    const int ph = jj * K, pt = rl - (jj + 1) * K, o = jj * pitch + sft;
                    earlier = live(kind == 0u || ph < p, ph, ph + d) || live(kind == 0u || pt < p, pt, pt + d) || live(kind == 0u && o < q, o - d, o);
                }
                if (!earlier) {
                    // the whole overlap, 128 bytes a step (sixteen loads in flight: this kernel has the registers), then 64, then 16 at a time, and what is
                    // left behind the last whole 16 as ONE more load that reaches back over bytes already counted (they are masked off)
                    int mism = 0, j = 0;
                    const GlbBytes x = read + a, y = seq + (a + d);
                    typedef u32x4_t __attribute__((aligned(1))) u32x4_any;
                    typedef const __attribute__((address_space(1))) u32x4_any *GlbVec;
                    auto diff16 = [](const u32x4_t xv, const u32x4_t yv) { return diff_bytes(xv.x, yv.x) + diff_bytes(xv.y, yv.y) + diff_bytes(xv.z, yv.z) + diff_bytes(xv.w, yv.w); };
                    for (; j + 128 <= L && mism <= me; j += 128) {
                        u32x4_t xv[8], yv[8];
#pragma unroll
                        for (int t = 0; t < 8; t++) { xv[t] = *(GlbVec)(x + j + 16 * t); yv[t] = *(GlbVec)(y + j + 16 * t); }
#pragma unroll
                        for (int t = 0; t < 8; t++) mism += diff16(xv[t], yv[t]);
                    }
                    if (j + 64 <= L && mism <= me) {
                        u32x4_t xv[4], yv[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) { xv[t] = *(GlbVec)(x + j + 16 * t); yv[t] = *(GlbVec)(y + j + 16 * t); }
#pragma unroll
                        for (int t = 0; t < 4; t++) mism += diff16(xv[t], yv[t]);
                        j += 64;
                    }
                    for (; j + 16 <= L && mism <= me; j += 16) mism += diff16(*(GlbVec)(x + j), *(GlbVec)(y + j));
                    if (j < L && mism <= me) {
                        if (L >= 16) {
                            const u32x4_t xv = *(GlbVec)(x + L - 16), yv = *(GlbVec)(y + L - 16);
                            const int keep = L - j;                                   // the last `keep` (1..15) bytes of the vector are new
                            uint32_t dx[4] = { xv.x ^ yv.x, xv.y ^ yv.y, xv.z ^ yv.z, xv.w ^ yv.w };
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                const int first_new = 16 - keep - 4 * t;              // bytes of dword t in front of this index are old
                                const uint32_t m = first_new <= 0 ? 0xFFFFFFFFu : first_new >= 4 ? 0u : 0xFFFFFFFFu << (8 * first_new);
                                mism += diff_bytes(dx[t] & m, 0u);
                            }
                        } else {
                            for (; j + 4 <= L && mism <= me; j += 4) mism += diff_bytes(glb_u32(x + j), glb_u32(y + j));
                            for (; j < L && mism <= me; j++) mism += (x[j] != y[j]);
                        }
                    }
                    if (mism <= me) { hitAB = canAB; hitC = canC; }
                }
            }
        }
        const uint64_t mAB = __ballot(hitAB), mC = __ballot(hitC);
        const uint32_t nAB = (uint32_t)__popcll(mAB), nC = (uint32_t)__popcll(mC);
        uint32_t at = 0;
        if (nAB + nC) {
            if (lane == 0u) at = atomicAdd(&n_round, nAB + nC);
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        }
        __syncthreads();
        const uint32_t total = n_round;
        if (total) {                                               // (uniform over the workgroup)
            if (threadIdx.x == 0) arena_at = atomicAdd(P.arena_n, total);      // (counts on beyond the arena's end, as kvq_emit does: the host sees the overflow there)
            __syncthreads();
            const uint32_t o = arena_at + at;
            KvqHit h;
            h.fpos = fpos; h.seq_nr = s; h.readlength = rl; h.blob_off = 0;
            if (hitAB) {
                const uint32_t idx = o + (uint32_t)__popcll(mAB & kvq_lanemask_lt());
                h.seq_pos = sposAB; h.length = lenAB; h.key = keyAB;
                if (idx < P.arena_cap) P.arena[idx] = h;
            }
            if (hitC) {
                const uint32_t idx = o + nAB + (uint32_t)__popcll(mC & kvq_lanemask_lt());
                h.seq_pos = sposC; h.length = lenC; h.key = keyC;
                if (idx < P.arena_cap) P.arena[idx] = h;
            }
        }
    }
}

// kvq_trim_records' long score lines (declared in kernels_general.hip): lane l sums up scores [1024 r + 16 l, + 16) of
// every KiB r as a Seg (bp_runs64), an ordered tree merge over the wave gives the KiB's Seg, the KiBs are merged in turn
__device__ void kvq_long_line_run(const uint8_t *line, uint32_t Q, int amin, int lane, int &best, uint32_t &best_start)
{
    Seg all; all.beg = 0; all.len = 0; all.pre = 0; all.suf = 0; all.best = 0; all.bstart = 0;
    for (uint32_t r0 = 0; r0 < Q; r0 += 4096u) {
        uint32_t w[4][4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t o = r0 + 1024u * (uint32_t)r + 16u * (uint32_t)lane;
            const uint32_t oo = o + 16u <= Q + 1u ? o : (Q + 1u >= 16u ? Q + 1u - 16u : 0u);      // (loads stay inside the line and its newline; every load is issued)
#pragma unroll
            for (int d = 0; d < 4; d++) __builtin_memcpy(&w[r][d], line + oo + 4u * d, 4);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t row = r0 + 1024u * (uint32_t)r;
            if (row >= Q) break;
            const uint32_t o = row + 16u * (uint32_t)lane;
            const uint32_t oo = o + 16u <= Q + 1u ? o : (Q + 1u >= 16u ? Q + 1u - 16u : 0u);
            const int n = o < Q ? (Q - o < 16u ? (int)(Q - o) : 16) : 0;
            uint32_t m = 0;
#pragma unroll
            for (int d = 0; d < 4; d++)
#pragma unroll
                for (int b = 0; b < 4; b++) m |= (uint32_t)((int)(int8_t)(w[r][d] >> (8 * b)) >= amin) << (4 * d + b);
            // (a vector that was loaded from further down -- the line's last one -- is shifted into place)
            const uint32_t shift = o - oo;                                 // 0 unless the vector was moved
            m = shift < 16u ? m >> shift : 0u;
            m &= n > 0 ? (1u << n) - 1u : 0u;
            Seg sg; sg.beg = (int)o; sg.len = n;
            int bs;
            bp_runs64((uint64_t)m, n, 0u, sg.pre, sg.suf, sg.best, bs);
            sg.bstart = (int)o + bs;
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                Seg B;
                B.len = __shfl_xor(sg.len, (int)d, 64); B.pre = __shfl_xor(sg.pre, (int)d, 64); B.suf = __shfl_xor(sg.suf, (int)d, 64);
                B.best = __shfl_xor(sg.best, (int)d, 64); B.bstart = __shfl_xor(sg.bstart, (int)d, 64); B.beg = 0;
                if (((uint32_t)lane & d) == 0) sg = seg_merge(sg, B);
            }
            Seg R;
            R.beg = (int)row; R.len = __shfl(sg.len, 0, 64); R.pre = __shfl(sg.pre, 0, 64); R.suf = __shfl(sg.suf, 0, 64);
            R.best = __shfl(sg.best, 0, 64); R.bstart = __shfl(sg.bstart, 0, 64);
            all = (row == 0u) ? R : seg_merge(all, R);
        }
    }
    best = all.best; best_start = (uint32_t)all.bstart;
}
