// kvarq_amd/csrc/kernels_pool.hip -- kvq_scan_pool: the bit-plane scan (kernels_bp.hip) with the two things
// round 2's counters asked for (profiles/round2_pmc_phases.txt: trim 15.4, verification 12.7 of the 65 vector
// instructions per read):
//
//   * the quality trim (workhorse.c:1055-1070) as ONE ordered key per lane -- (run length << 16 | 0xFFFF - start),
//     the larger the better, which is exactly "longest, first of equals" (1062) -- instead of five-word segment
//     summaries merged pairwise: a lane looks at the runs that START in its slice of the score line, the last of
//     them grows by the good bytes that lead the following slices (a two-step suffix scan of one packed word),
//     and the lanes of a read agree on the maximum with two DPP steps;
//   * candidates and work items POOLED over the workgroup: every wave appends to one queue, and behind a barrier
//     the waves take the queue 64 entries at a time.  In kvq_scan_bp each wave verified the candidates of its
//     own sixteen reads -- about thirty work items on sixty-four lanes, twice per tile stretch; here a tile's
//     two to three hundred items fill four or five wave rounds to the last lane.
//
// Everything in front of the trim (tile hand-out, loads, planes, newline list, the speculated first record) and
// everything behind the kernel (kvq_validate_tiles, the skip list, the exhaustive redo) is kvq_scan_bp's.
#include "kvq_host.h"

#ifndef PO_POOLED
#define PO_POOLED 1                // 1: candidates and work items pooled over the workgroup (two more barriers per sub-pass); 0: every wave keeps its own
#endif
#ifndef PO_STATIC
#define PO_STATIC 0                // 1: a workgroup walks its own contiguous share of the tiles (no counters); 0: tiles are drawn from the sharded counters
#endif
#ifndef PO_EARLY
#define PO_EARLY 0                 // 1: the next tile's text is fetched before the verification of this one (its twenty registers stay live through it)
#endif
#if PO_POOLED
#define PO_QCAP 768u               // candidates per sub-pass of a workgroup
#define PO_Q2CAP 768u              // work items per sub-pass
#else
#define PO_QCAP 96u                // candidates per sub-pass of a wave
#define PO_Q2CAP 96u               // work items per sub-pass of a wave
#endif
#define PO_QTOT 768u

struct PoLds {
    uint32_t cdp[BP_WIN / 16 + 8];       // code plane (as BpLds)
    uint32_t gdp[BP_WIN / 32 + 8];       // good plane
    uint16_t nl[BP_NLCAP];
    uint8_t  bmA[8192];
    uint32_t hist[KVQ_RL_BINS / 2];
    union {
        struct {
            uint32_t q1[PO_QTOT];        // candidate: read (9 bits) | position in the read << 9 (16 bits) | kind << 25
            uint32_t q2[PO_QTOT];       // work item: candidate << 22 | index entry
        };
        uint32_t nlp[BP_WIN / 32 + 8];
    };
    uint32_t rinfo[ST_RCAP];             // read offset in the window | rl << 16
    __attribute__((aligned(16))) uint8_t head[BP_HEAD];
    __attribute__((aligned(16))) uint32_t wtot[ST_WAVES];
    uint32_t qn[4];                      // entries in q1 (slots 0, 1) and q2 (slots 2, 3), by the parity of the sub-pass
    uint32_t longest_p1, records, fallback, n_owned, next_tile, first_tile, p2_jn;
};
static_assert(sizeof(PoLds) <= 40 * 1024, "four workgroups per CU: at most 40 KB of LDS each");
static_assert(offsetof(PoLds, cdp) == 0 && offsetof(PoLds, gdp) == BP_LDS_GDP && offsetof(PoLds, bmA) == BP_LDS_BMA &&
              offsetof(PoLds, nlp) == BP_LDS_NLP && offsetof(PoLds, head) == BP_LDS_HEAD, "the plane helpers of kernels_bp.hip address LDS by these offsets");
static_assert(sizeof(uint32_t) * (PO_QTOT + PO_QTOT) >= sizeof(uint32_t) * (BP_WIN / 32 + 8), "the newline plane fits the queues it shares LDS with");
static_assert(PO_QTOT <= 1024u && PO_QCAP * (PO_POOLED ? 1u : ST_WAVES) <= PO_QTOT, "a work item holds its candidate's number in ten bits");

// ---- lanes of a read ----------------------------------------------------------------------------------------
// LG = log2 of the lanes per read when the build fixes it (0, 1, 2: DPP inside a quad), -1: worked out per tile
// the value `d` lanes up (garbage where that lane belongs to another read: the caller masks)
template <int LG>
__device__ __forceinline__ uint32_t po_down(uint32_t v, uint32_t d)
{
    if constexpr (LG == 2) {
        // quad_perm [1,2,3,3] / [2,3,2,3]
        return d == 1u ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xF9, 0xf, 0xf, true)
                       : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xEE, 0xf, 0xf, true);
    } else if constexpr (LG == 1) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xF5, 0xf, 0xf, true);                  // quad_perm [1,1,3,3]
    } else if constexpr (LG == 0) {
        return 0u;
    } else {
        return (uint32_t)__shfl_down((int)v, (int)d, 64);
    }
}
// maximum over the lanes of a read, in every one of them
template <int LG>
__device__ __forceinline__ uint32_t po_group_max(uint32_t v, uint32_t G)
{
    if constexpr (LG == 0) return v;
    else if constexpr (LG == 1 || LG == 2) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);     // lane ^ 1
        v = v > a ? v : a;
        if constexpr (LG == 2) {
            const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); // lane ^ 2
            v = v > b ? v : b;
        }
        return v;
    } else {
        for (uint32_t d = 1; d < G; d <<= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)v, (int)d, 64);
            v = v > o ? v : o;
        }
        return v;
    }
}

__device__ __forceinline__ uint32_t po_ctz64(uint32_t lo, uint32_t hi, uint32_t n)        // n when no bit below n is set
{
    const uint32_t pl = (uint32_t)(__ffs((int)lo) - 1), ph = (uint32_t)(__ffs((int)hi) - 1);
    const uint32_t phi = ph > 0xFFFFFFDFu ? 0xFFFFFFFFu : ph + 32u;
    const uint32_t r = pl < phi ? pl : phi;
    return r < n ? r : n;
}

// ---- quality trim (workhorse.c:1055-1070) -------------------------------------------------------------------
// The longest run of good scores, the first one among equals, as the maximum of ONE ordered key per run:
// length << 16 | 0xFFFF - start.  A lane owns up to two chunks of at most 63 scores each; a chunk reports the
// runs that START in it, the run that touches its end grows by `ext`, the good scores that lead what follows.
struct PoChunk { uint32_t zlo, zhi, n, c, p1; };        // bad scores of the chunk (bits), its length, their number, the first one (n: none)
__device__ __forceinline__ PoChunk po_chunk(uint32_t bit, uint32_t n)
{
    PoChunk k; k.n = n;                                                            // 0 .. 63
    const uint32_t a = BP_LDS_GDP + ((bit >> 5) << 2), sh = bit & 31u;
    const uint32_t d0 = lds_u32_at(a), d1 = lds_u32_at(a + 4u), d2 = lds_u32_at(a + 8u);
    const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
    const uint64_t above = ~0ull << n;
    k.zlo = ~(lo | (uint32_t)above); k.zhi = ~(hi | (uint32_t)(above >> 32));
    k.c = (uint32_t)__popc(k.zlo) + (uint32_t)__popc(k.zhi);
    k.p1 = po_ctz64(k.zlo, k.zhi, n);
    return k;
}
// at most three bad scores in the chunk: the first, the second and the last one in closed form, the runs are the
// four gaps between them (a gap that does not exist comes out with length 0 and never beats a real run; an empty
// chunk hands `ext` on under the position the next chunk reports it at)
__device__ __forceinline__ uint32_t po_chunk_key(const PoChunk &k, uint32_t base, uint32_t ext)
{
    const uint32_t n = k.n, p1 = k.p1;
    const uint32_t lo2 = k.zlo & (k.zlo - 1u), hi2 = k.zlo ? k.zhi : (k.zhi & (k.zhi - 1u));
    const uint32_t p2 = po_ctz64(lo2, hi2, n);
    const uint32_t e = k.zhi ? 64u - (uint32_t)__clz((int)k.zhi) : k.zlo ? 32u - (uint32_t)__clz((int)k.zlo) : 0u;   // one behind the last bad score
    const uint32_t g1 = p2 > p1 ? p2 - p1 - 1u : 0u;
    const uint32_t g2 = e > p2 + 1u ? e - p2 - 2u : 0u;
    const uint32_t g3 = n - e + ((e < n || n == 0u) ? ext : 0u);
    const uint32_t k0 = (p1 << 16) + base;
    const uint32_t k1 = (g1 << 16) + (base - p1 - 1u);
    const uint32_t k2 = (g2 << 16) + (base - p2 - 1u);
    const uint32_t k3 = (g3 << 16) + (base - e);
    const uint32_t ka = k0 > k1 ? k0 : k1, kb = k2 > k3 ? k2 : k3;
    return ka > kb ? ka : kb;
}
// any number of bad scores: walk the GOOD runs of the chunk (a tail of bad scores is one step); a wave-wide loop
__device__ __forceinline__ uint32_t po_chunk_walk(const PoChunk &k, uint32_t base, uint32_t ext)
{
    const uint64_t below = ~(~0ull << k.n);
    uint32_t mlo = ~k.zlo & (uint32_t)below, mhi = ~k.zhi & (uint32_t)(below >> 32);
    uint32_t key = k.n == 0u ? (ext << 16) + base : base;                          // (nothing found: length 0 at the chunk's first score)
    while (__any((mlo | mhi) != 0u)) {
        const uint32_t s = po_ctz64(mlo, mhi, 64u);                                // start of the lowest run
        const uint64_t m = ((uint64_t)mhi << 32) | mlo;
        const uint64_t t = s < 64u ? m + (1ull << s) : 0ull;                       // the carry clears the run and lands behind it (n <= 63: no wrap)
        const uint32_t tl = (uint32_t)t, th = (uint32_t)(t >> 32);
        if (s < 64u) {
            const uint32_t en = po_ctz64(tl, th, 64u);
            const uint32_t len = en - s + (en == k.n ? ext : 0u);
            const uint32_t kk = (len << 16) + (base - s);
            key = key > kk ? key : kk;
            mlo &= tl; mhi &= th;
        }
    }
    return key;
}
// the G lanes of a read's group -> the key of its trimmed read, in every one of them.  Every lane of the wave must
// call it (groups without a read pass Q = 0); the caller has checked that no lane's share exceeds 126 scores.
template <int LG>
__device__ __forceinline__ uint32_t po_trim_key(uint32_t sscore, int Q, uint32_t gl, uint32_t lg, bool force_walk)
{
    const uint32_t G = 1u << lg;
    const int per = (Q + (int)G - 1) >> lg;
    int beg = (int)mul_u24(gl, (uint32_t)per); if (beg > Q) beg = Q;
    int end = beg + per; if (end > Q) end = Q;
    const uint32_t n = (uint32_t)(end - beg);                                       // 0 .. 126
    const uint32_t na = n < 63u ? n : 63u, nb = n - na;
    const bool two = __any(nb != 0u);                                              // (wave-uniform: reads of up to 63 G scores need one chunk per lane)
    const uint32_t bit = sscore + (uint32_t)beg;
    const PoChunk A = po_chunk(bit, na);
    PoChunk B; B.zlo = 0; B.zhi = 0; B.n = 0; B.c = 0; B.p1 = 0;
    if (two) B = po_chunk(bit + na, nb);
    // good scores that lead the FOLLOWING lanes' shares, as far as they are unbroken: suffix scan over the lanes
    // of (leading good scores, "the whole share is good"), packed as count | whole << 15
    uint32_t cnext;
    {
        const bool fa = A.c == 0u, fb = B.c == 0u;
        const uint32_t V = (A.p1 + (fa ? B.p1 : 0u)) | (fa && fb ? 0x8000u : 0u);
        uint32_t W = po_down<LG>(V, 1u);
        if (gl + 1u >= G) W = 0u;
        for (uint32_t d = 1; d + 1u < G; d <<= 1) {
            uint32_t nx = po_down<LG>(W, d);
            if (gl + d >= G) nx = 0u;
            W = (W & 0x8000u) ? (W & 0x7FFFu) + nx : W;
        }
        cnext = W & 0x7FFFu;
    }
    const uint32_t ext_a = B.p1 + (B.c == 0u ? cnext : 0u);                         // (an empty second chunk: p1 = 0, c = 0)
    const uint32_t base_a = 0xFFFFu - (uint32_t)beg, base_b = base_a - na;
    uint32_t key;
    if (!__any((A.c > 3u) | (B.c > 3u)) && !force_walk) {
        key = po_chunk_key(A, base_a, ext_a);
        if (two) { const uint32_t kb = po_chunk_key(B, base_b, cnext); key = key > kb ? key : kb; }
    } else {
        key = po_chunk_walk(A, base_a, ext_a);
        if (two) { const uint32_t kb = po_chunk_walk(B, base_b, cnext); key = key > kb ? key : kb; }
    }
    return po_group_max<LG>(key, G);
}

// One work item = one (candidate, index entry) pair = one diagonal of one read against one sequence (the rules
// of verify_item_bp; roff / rl are the read's, the read's bases come from the code plane, its bytes from global
// memory).  Must be called by every lane of the wave.  `ixb`: the index blob (SeedTables::blob), whose 2-bit
// copy of the table begins at word `o_tab2`; everything a survivor needs beyond that is read from the
// parameter block where it is wanted.
__device__ __forceinline__ void po_verify(const BpArgs *A_, GlbWords ixb, uint32_t o_tab2, int me, bool active, uint32_t roff, int rl,
                                          int p, uint32_t kind, uint64_t en, uint32_t g0, int stride)
{
    const GlbWords tab2 = ixb + o_tab2;
    int s = 0, d = 0, a = 0, L = 0, seql = 0, q = 0; uint32_t toff = 0;
    bool alive = false;
    if (active) {
        q = (int)(en & 4095u);
        s = (int)((en >> 12) & 0xFFFFFu);
        toff = (uint32_t)((en >> 32) & 0xFFFFFu);
        seql = (int)(en >> 52);
        d = q - p;                                       // sequence index = read index + d
        a = d < 0 ? -d : 0;
        L = (rl < seql - d ? rl : seql - d) - a;
        // most false candidates die here, on the first 16 bases of the diagonal: compared as 2-bit codes (bytes
        // that are equal have equal codes: this never rejects what the bytes accept)
        alive = L > 0;
        if (alive && L >= 16) alive = diff_codes(cdp32(roff + (uint32_t)a), tab2_32(tab2, toff + (uint32_t)(a + d))) <= me;
    }
    if (!__any(alive)) return;
    // ---- the survivors (a wave in ten has one) ----
    const BpArgsPtr A = bp_args(A_);
    const int mo = A->P.minoverlap;
    bool hitAB = false, hitC = false, canAB = false, canC = false;
    int lenAB = 0, lenC = 0, sposAB = 0, sposC = 0; uint32_t keyAB = 0, keyC = 0;
    if (alive) {
        // which reference loops visit this diagonal
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
    }
    if (canAB || canC) {
        // byte-exact mismatch count of the whole overlap
        const GlbBytes text = (GlbBytes)A->data + g0 - ST_PRE;
        const GlbBytes tab = (GlbBytes)A->P.tab;
        int mism = 0, j = 0;
        const GlbBytes x = text + roff + (uint32_t)a, y = tab + toff + (uint32_t)(a + d);
        for (; j + 4 <= L && mism <= me; j += 4) mism += diff_bytes(glb_u32(x + j), glb_u32(y + j));
        for (; j < L && mism <= me; j++) mism += (x[j] != y[j]);
        if (mism <= me) {
            // canonical discoverer: no live seed earlier in the order
            // [ALL-index read blocks by position] then [ANCHOR blocks by number]
            bool earlier = false;
            for (int jj = 0; jj <= me && !earlier; jj++) {
                const int ph = jj * SK, pt = rl - (jj + 1) * SK;
                if (ph + SK <= rl && (kind == 0u || ph < p)) earlier = seed_live_bp(tab2, roff, rl, ph, toff, seql, ph + d);
                if (!earlier && pt >= 0 && (kind == 0u || pt < p)) earlier = seed_live_bp(tab2, roff, rl, pt, toff, seql, pt + d);
            }
            if (kind == 0u) {
                for (int jj = 0; jj <= me && !earlier; jj++)
                    for (int sft = 0; sft < stride && !earlier; sft++) {
                        const int o = jj * SK + sft;
                        if (o < q && ((o - d) & (stride - 1)) == 0) earlier = seed_live_bp(tab2, roff, rl, o - d, toff, seql, o);
                    }
            }
            if (!earlier) { hitAB = canAB; hitC = canC; }
        }
    }
    if (__any(hitAB || hitC)) {
        const int64_t fpos = A->fpos_base + (int64_t)g0 + (int64_t)roff - (int64_t)ST_PRE;
        emit_cold(&A_->P, hitAB, fpos, s, sposAB, lenAB, rl, keyAB);
        emit_cold(&A_->P, hitC, fpos, s, sposC, lenC, rl, keyC);
    }
}

// MODE 0: the production build (the diagnostic word of the parameter block is not even read); 1: the diagnostic
// switches of KVQ_DBG work; 2: ... and wave 0's cycles are summed per phase (KVQ_DBG=16, tools/phase_stamps.py)
template <int SS, int LG, int MODE>
__global__ void __launch_bounds__(ST_THREADS, BP_OCC)
kvq_scan_pool(const BpArgs *__restrict__ A_)
{
    constexpr bool DIAG = MODE >= 1, STAMPS = MODE == 2;
    __shared__ __align__(16) PoLds S;
    uint8_t *const lds_raw = reinterpret_cast<uint8_t *>(&S);
    int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = rfl((uint32_t)tid >> 6);
    if ((uint32_t)(uintptr_t)((__attribute__((address_space(3))) uint8_t *)lds_raw) != 0u) __builtin_trap();   // lds_u32_at
    unsigned long long stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stamp_t = 0;
#define PSTAMP(i) do { if constexpr (STAMPS) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } } while (0)

    uint32_t ntiles, tile_bytes, dbg = 0, amin;
    {
        const BpArgsPtr A = bp_args(A_);
        ntiles = A->ntiles; tile_bytes = A->tile_bytes; amin = (uint32_t)A->P.amin;
        if constexpr (DIAG) dbg = A->dbg;
        const GlbWords bm1 = (GlbWords)A->X.bm1;
        for (int i = tid; i < 2048; i += ST_THREADS) reinterpret_cast<uint32_t *>(S.bmA)[i] = bm1[i];
    }
    uint32_t my_shard = blockIdx.x % BP_SHARDS, shards_left = BP_SHARDS;
    const uint32_t share_lo = (uint32_t)(((uint64_t)blockIdx.x * ntiles) / gridDim.x), share_hi = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * ntiles) / gridDim.x);
    if constexpr (PO_STATIC) { if (tid == 0) S.first_tile = share_lo < share_hi ? share_lo : ntiles; }
    else if (tid == ST_THREADS - 64) {
        unsigned int *const ctr = bp_args(A_)->tile_ctr;
        S.first_tile = bp_draw(ctr, ntiles, my_shard, shards_left);
        S.next_tile = bp_draw(ctr, ntiles, my_shard, shards_left);
    }
    for (int i = tid; i < KVQ_RL_BINS / 2; i += ST_THREADS) S.hist[i] = 0;
    if (tid == 0) { S.longest_p1 = 0; S.records = 0; S.fallback = 0; }
    if (tid < 4) S.qn[tid] = 0;
    if (tid < 8) { S.cdp[BP_WIN / 16 + tid] = 0; S.gdp[BP_WIN / 32 + tid] = 0; }    // slack behind the planes
    __syncthreads();

    // (experiment, KVQ_STAGGER: the workgroups that share a compute unit start a fraction of a tile apart, so that one's wait
    // for its text meets another's arithmetic: workgroup b sits in slot b / 256 of its compute unit under round-robin dispatch)
    if (const uint32_t stg = bp_args(A_)->pad_) { for (uint32_t i = 0; i < (blockIdx.x >> 8) * stg; i++) __builtin_amdgcn_s_sleep(127); }
    const uint32_t addk = (0x80u - amin) * 0x01010101u;
    uint32_t tiles_done = 0;
    uint32_t pc = 0;                                                        // sub-passes so far (their parity picks the queue counters)
    uint32_t g_done = 0xFFFFFFFFu;
    uint32_t my_longest = 0;
    // the text of tile `gg` -> twenty registers: coalesced (one load of a wave fetches 1 KiB), range-checked by the
    // buffer descriptor (bytes outside the chunk come back as zeros, a tile that does not exist as all zeros)
    uint4 pre[ST_ROUNDS];
    auto fetch_tile = [&](uint32_t gg) {
        uint32_t tid_ = (uint32_t)threadIdx.x;
        asm volatile("" : "+v"(tid_));
        const uint32_t wv = wave * (ST_BLK * 64u / 16u) + (tid_ & 63u);
        const BpArgsPtr A = bp_args(A_);
        const uint32_t ga = gg < ntiles ? gg : 0u;                          // (behind the last tile: a descriptor of no bytes)
        const u32x4_t q = ((const __attribute__((address_space(4))) u32x4_t *)A->tiles)[(dbg & 64u) ? (ga & 63u) : ga];
        const uint32_t jb = q.y, t0 = (q.x & ~15u) + q.z * A->tile_bytes;
        const uint8_t *const data = A->data;
        const uint32_t load_hi = t0 + ST_TILE + ST_OV < jb ? t0 + ST_TILE + ST_OV : jb;
        const uint32_t vo = 16u * wv - ST_PRE, vo0 = wv >= ST_PRE / 16u ? vo : ST_NO_BLOCK;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(data + t0), 0, gg < ntiles ? (int)(((load_hi + 15u) & ~15u) - t0) : 0, 0x00020000);
#pragma unroll
        for (int r = 0; r < (int)ST_ROUNDS; r++) {
            const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(r ? vo + 1024u * r : vo0), 0, 0);
            pre[r] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    fetch_tile(rfl(S.first_tile));
    for (uint32_t g = rfl(S.first_tile); ; ) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        const uint32_t blk = (uint32_t)tid * ST_BLK;
        const uint32_t wv = wave * (ST_BLK * 64u / 16u) + (uint32_t)lane;
        if constexpr (STAMPS) stamp_t = __builtin_amdgcn_s_memtime();
        uint32_t Ja = 0, Jb = 0, Jt = 0, g0 = 0;
        if (g < ntiles) {
            const BpArgsPtr A = bp_args(A_);
            const u32x4_t q = ((const __attribute__((address_space(4))) u32x4_t *)A->tiles)[(dbg & 64u) ? (g & 63u) : g];
            Ja = q.x; Jb = q.y; Jt = q.z;
            g0 = (Ja & ~15u) + Jt * tile_bytes;
        }
        // everyone is done with the last tile's planes and queues; its loose ends
        __syncthreads();
        if (g_done != 0xFFFFFFFFu) {
            if (tid == 0 && S.fallback) { atomicOr(&bp_args(A_)->tile_report[g_done], TR_FLAG_FALLBACK); S.fallback = 0; }
        }
        if (g >= ntiles) break;
        const uint32_t gn = PO_STATIC ? (g + 1u < share_hi ? g + 1u : ntiles) : rfl(S.next_tile);
        const uint32_t own_end = g0 + tile_bytes < Jb ? g0 + tile_bytes : Jb;
        const uint32_t load_hi = g0 + ST_TILE + ST_OV < Jb ? g0 + ST_TILE + ST_OV : Jb;
        const uint32_t own_begin_l = (Jt == 0 ? Ja : g0) - g0 + ST_PRE;
        const uint32_t own_end_l = own_end - g0 + ST_PRE;
        const uint32_t end_l = load_hi - g0 + ST_PRE;
        // (the window's text in global memory, for the few bytes that are looked at there; worked out where it is wanted)
#define PO_TEXT() ((GlbBytes)bp_args(A_)->data + g0 - ST_PRE)

        // ---- P0: registers -> planes (as kvq_scan_bp) ----
        {
            const uint32_t w_lo = wave * (ST_BLK * 64u), w_hi = w_lo + ST_BLK * 64u;
            const bool cut = (Jt == 0 || load_hi == Jb) &&                    // (only a chunk's first and last tiles have such a place)
                             (((own_begin_l & 15u) && own_begin_l > w_lo && own_begin_l < w_hi) || ((end_l & 15u) && end_l > w_lo && end_l < w_hi));
            if (cut) {
                uint32_t wv_ = wv;
                asm volatile("" : "+v"(wv_));
#pragma unroll
                for (int r = 0; r < (int)ST_ROUNDS; r++) {
                    const uint32_t o = 16u * (wv_ + 64u * r);
                    uint32_t x[4] = { pre[r].x, pre[r].y, pre[r].z, pre[r].w };
#pragma unroll
                    for (int d = 0; d < 4; d++) x[d] &= (kvq_range_flags(o + 4u * d, own_begin_l, end_l) >> 7) * 0xFFu;
                    pre[r] = make_uint4(x[0], x[1], x[2], x[3]);
                }
            }
        }
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
            if (wave == 0u) {
#pragma unroll
                for (int r = 0; r < (int)(BP_HEAD / 1024u); r++)
                    *reinterpret_cast<__attribute__((address_space(3))) u32x4_t *>((uintptr_t)(BP_LDS_HEAD + 16u * (uint32_t)lane + 1024u * r)) = u32x4_t{ pre[r].x, pre[r].y, pre[r].z, pre[r].w };
            }
        }
        KVQ_MARK("P0 vectors end");
        uint32_t m0, m1, m2;
        {
            const uint32_t a = BP_LDS_NLP + ((10u * (uint32_t)tid) & ~3u), sh = ((uint32_t)tid & 1u) * 16u;
            const uint32_t d0 = lds_u32_at(a), d1 = lds_u32_at(a + 4u), d2 = lds_u32_at(a + 8u);
            m0 = __builtin_amdgcn_alignbit(d1, d0, sh); m1 = __builtin_amdgcn_alignbit(d2, d1, sh); m2 = (d2 >> sh) & 0xFFFFu;
        }
        const uint32_t cnt = (uint32_t)(__popc(m0) + __popc(m1) + __popc(m2));
        const uint32_t incl = kvq_wave_incl_scan(cnt);
        if (lane == 63) S.wtot[wave] = incl;
        KVQ_MARK("P0 scan end");
        PSTAMP(0);
        __syncthreads();
        KVQ_MARK("P1b");
        PSTAMP(1);
        uint32_t n_all = 0;
        {
            uint32_t run = S.wtot[lane & 7];
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x111, 0xf, 0xf, false);
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x112, 0xf, 0xf, false);
            run += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)run, 0x114, 0xf, 0xf, false);
            n_all = (uint32_t)__builtin_amdgcn_readlane((int)run, 7);
            const uint32_t upto = (uint32_t)__builtin_amdgcn_readlane((int)run, (int)wave);
            uint32_t n = upto - (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) + incl - cnt;
            if (blk < own_end_l && blk + ST_BLK >= own_end_l) S.n_owned = n + cnt;
            if (__any(cnt != 0u)) {
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
                        const uint32_t c0 = ls0 < BP_HEAD ? lds_byte_at(BP_LDS_HEAD + ls0) : (uint32_t)PO_TEXT()[ls0];
                        const uint32_t c2 = ls2 < BP_HEAD ? lds_byte_at(BP_LDS_HEAD + ls2) : (uint32_t)PO_TEXT()[ls2];
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
        PSTAMP(2);

        // ---- P2 (every wave, redundantly): which records does this tile own?  (as kvq_scan_bp) ----
        uint32_t nrec = 0, jn = TR_NONE, drawn;
        {
            const uint32_t n_nl = n_all < BP_NLCAP ? n_all : BP_NLCAP;
            const uint32_t n_owned = rfl(S.n_owned);
            uint32_t fallback = n_all > BP_NLCAP ? 1u : 0u;
            if (Jt == 0) jn = 0;
            else if (const uint32_t early = rfl(S.p2_jn); early != BP_P2_LATE) jn = early;
            else {
                const uint32_t m = (uint32_t)lane;
                bool ok = false;
                if (m >= 1 && m <= 8 && m <= n_owned && m + 2 <= n_nl) {
                    const uint32_t ls0 = (uint32_t)S.nl[m - 1] + 1u;
                    const uint32_t ls2 = (uint32_t)S.nl[m + 1] + 1u;
                    if (ls0 < end_l && ls2 < end_l) {
                        const uint32_t c0 = ls0 < BP_HEAD ? lds_byte_at(BP_LDS_HEAD + ls0) : (uint32_t)PO_TEXT()[ls0];
                        const uint32_t c2 = ls2 < BP_HEAD ? lds_byte_at(BP_LDS_HEAD + ls2) : (uint32_t)PO_TEXT()[ls2];
                        ok = c0 == '@' && c2 == '+';
                    }
                }
                const uint64_t mk = __ballot(ok);
                if (mk) jn = (uint32_t)(__ffsll((long long)mk) - 1);
            }
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
            if (tid == 0 || tid == ST_THREADS - 64) {
                const BpArgsPtr A = bp_args(A_);
                if (tid == 0) {
                    A->tile_report[g] = (n_owned & 0xFFFFu) | ((jn & 0xFFu) << 16) | (left ? TR_FLAG_SKIPPED : 0u) | (left && nrec ? TR_FLAG_PARTIAL : 0u);
                    if (left) A->tile_report[ntiles + g] = nrec ? g0 - ST_PRE + (uint32_t)S.nl[jn + 4u * nrec - 1u] + 2u : 0u;
                    S.records += nrec;
                } else if (!PO_STATIC && shards_left) drawn = atomicAdd(&A->tile_ctr[my_shard * BP_SHARD_STRIDE], 1u);
            }
        }
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
        PSTAMP(3);
        if (wave >= 4u) KVQ_SETPRIO(2); else KVQ_SETPRIO(1);
        // ---- P3 / P4: reads -> candidates (per wave), candidates -> work items -> hits (pooled) ----
        // G lanes share a read: as few as give every lane at most 126 scores to trim (two 63-bit chunks); a wave
        // takes 64 / G reads per pass, the first waves of the workgroup first (a tile of 122 reads of 150 bases at
        // two lanes per read: waves 0 to 3) -- what a pass costs besides its reads is paid 32 reads at a time
        uint32_t lg;
        if constexpr (LG >= 0) lg = (uint32_t)LG;
        else {
            // (from the tile's average record: its score line is about (bytes - 25) / 2 long)
            const uint32_t rec = nrec ? (own_end_l - own_begin_l) / nrec : 0u, q_est = rec > 25u ? (rec - 25u) / 2u : 1u;
            lg = 0u; while (lg < 6u && (126u << lg) < q_est) lg++;
        }
        const uint32_t G = 1u << lg, RP = ST_THREADS >> lg;
        bool fetched = false;                                                // the next tile's text is on its way
        for (uint32_t pass0 = 0; pass0 < nrec; pass0 += RP) {
            uint32_t tid_ = (uint32_t)tid;
            asm volatile("" : "+v"(tid_));                                   // (what follows from the lane number is worked out here, not kept in scalar registers across the tile)
            const uint32_t gl = tid_ & (G - 1u), gr = tid_ >> lg;
            const uint32_t k = pass0 + gr;
            const bool have = k < nrec;
            const bool wave_has = pass0 + wave * (64u >> lg) < nrec;         // (the waves behind the tile's last read have nothing to trim or look up)
            uint32_t roff = 0; int rl = 0;
            uint32_t c0 = '@', cp = '+';
            if (wave_has) {
                uint32_t sscore = 0, sread = 0; int Q = 0;
                if (have) {
                    const uint32_t m = jn + 4u * k;
                    const uint32_t rstart = m == 0 ? ST_PRE + (Ja & 15u) : (uint32_t)S.nl[m - 1] + 1u;
                    const uint32_t n0 = S.nl[m], n1 = S.nl[m + 1], n2 = S.nl[m + 2], n3 = S.nl[m + 3];
                    sread = n0 + 1u; sscore = n2 + 1u;
                    if (gl == 0 && !(dbg & 8u)) { const GlbBytes text = PO_TEXT(); c0 = text[rstart]; cp = text[n1 + 1u]; }     // the record's '@' and '+' (1037-1048): looked at behind the pass
                    Q = (int)(n3 - sscore);
                }
        KVQ_MARK("trim");
                const int per = (Q + (int)G - 1) >> lg;
                if (!__any(per > 126)) {
                    const uint32_t key = po_trim_key<LG>(sscore, Q, gl, lg, (dbg & 4u) != 0u);
                    rl = (int)(key >> 16);
                    roff = sread + (0xFFFFu - (key & 0xFFFFu));
                } else {
                    // shares beyond two 63-bit chunks (rare: reads of more than 126 G scores): summaries merged pairwise, as kvq_scan_bp
                    Seg sg; sg.beg = (int)mul_u24(gl, (uint32_t)per); if (sg.beg > Q) sg.beg = Q;
                    int s1 = sg.beg + per; if (s1 > Q) s1 = Q;
                    sg.len = 0; sg.pre = 0; sg.suf = 0; sg.best = 0; sg.bstart = sg.beg;
                    for (int cc = sg.beg; __any(cc < s1); cc += 64) {
                        if (cc < s1) {
                            const int n = s1 - cc < 64 ? s1 - cc : 64;
                            const uint32_t bit = sscore + (uint32_t)cc;
                            const uint32_t a = BP_LDS_GDP + ((bit >> 5) << 2), sh = bit & 31u;
                            const uint32_t d0 = lds_u32_at(a), d1 = lds_u32_at(a + 4u), d2 = lds_u32_at(a + 8u);
                            const uint64_t nmask = n < 64 ? (1ull << n) - 1ull : ~0ull;
                            const uint64_t mm = (((uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32) | (uint64_t)__builtin_amdgcn_alignbit(d1, d0, sh)) & nmask;
                            Seg sub; sub.beg = cc; sub.len = n;
                            const uint64_t inv = ~mm;
                            sub.pre = inv ? __ffsll((long long)inv) - 1 : 64; if (sub.pre > n) sub.pre = n;
                            const uint64_t top = ~(mm << (64 - n));
                            sub.suf = top ? __clzll((long long)top) : 64; if (sub.suf > n) sub.suf = n;
                            int bs; longest_run64(mm, n, sub.best, bs);
                            sub.bstart = cc + bs;
                            sg = (cc == sg.beg) ? sub : seg_merge(sg, sub);
                        }
                    }
                    for (uint32_t d = 1; d < G; d <<= 1) {
                        Seg B;
                        B.len = __shfl_xor(sg.len, (int)d, 64); B.pre = __shfl_xor(sg.pre, (int)d, 64); B.suf = __shfl_xor(sg.suf, (int)d, 64);
                        B.best = __shfl_xor(sg.best, (int)d, 64); B.bstart = __shfl_xor(sg.bstart, (int)d, 64); B.beg = 0;
                        if ((gl & d) == 0) sg = seg_merge(sg, B);
                    }
                    rl = __shfl(sg.best, lane & ~(int)(G - 1u), 64);
                    roff = sread + (uint32_t)__shfl(sg.bstart, lane & ~(int)(G - 1u), 64);             // 1070
                }
                if (have) {
                    my_longest = my_longest > (uint32_t)(rl + 1) ? my_longest : (uint32_t)(rl + 1);
                    if (gl == 0) {
                        if (rl < KVQ_RL_BINS) atomicAdd(&S.hist[rl >> 1], 1u << (16 * (rl & 1)));          // 394-402
                        S.rinfo[k] = roff | ((uint32_t)rl << 16);
                    }
                } else { rl = 0; roff = 0; }
            }
        KVQ_MARK("trim end / filter");
            PSTAMP(4);
            KVQ_SETPRIO(2);
            const uint32_t rpw = 64u >> lg;
            uint32_t sub = (dbg & 128u) ? rpw : 0u, step = rpw;               // (diagnostic 128: trim only)
            uint32_t *const q1 = S.q1 + (PO_POOLED ? 0u : wave * PO_QCAP), *const q2 = S.q2 + (PO_POOLED ? 0u : wave * PO_Q2CAP);
            while (sub < rpw) {
                const uint32_t par = pc & 1u; pc++;
                uint32_t qn1w = 0, qn2w = 0;                                  // (the wave's own counts, when the queues are not pooled)
                if (wave_has) {
                    uint32_t lane_ = (uint32_t)tid & 63u;
                    asm volatile("" : "+v"(lane_));
                    const uint32_t grw = lane_ >> lg;
                    int minrl, me_; const __attribute__((address_space(1))) uint8_t *bmL;
                    {
                        const BpArgsPtr A = bp_args(A_);
                        minrl = A->P.minreadlength; me_ = A->P.maxerrors;
                        bmL = (const __attribute__((address_space(1))) uint8_t *)A->X.bm1 + 8192;
                    }
                    const bool mine = have && rl >= minrl && !(dbg & 2u) && grw >= sub && grw - sub < step;       // 1100
                    int e0 = 0, e1 = 0;
                    if (mine) {
                        const int NPe = (rl - SK) / SS + 1;
                        const int per = (NPe + (int)G - 1) >> lg;
                        e0 = (int)mul_u24(gl, (uint32_t)per); if (e0 > NPe) e0 = NPe;
                        e1 = e0 + per; if (e1 > NPe) e1 = NPe;
                    }
                    // is the 8-mer at read position pp anywhere in a sequence?  (bitmap of all sequence 8-mers: global memory;
                    // the load is issued whether the block is wanted or not, so that a lane's lookups travel together)
                    auto fixed_block = [&](int pp, bool ok) -> bool {
                        const uint32_t code = cdp_code8(roff + (uint32_t)(ok ? pp : 0));
                        const uint32_t bits = bmL[code >> 3];
                        return ok & (bool)((bits >> (code & 7u)) & 1u);
                    };
                    // the read's 2 (e + 1) fixed blocks -- head block f for f <= e, then the tail blocks -- are dealt to the lanes
                    // of its group: lane gl looks up blocks gl, gl + G, ... (the first rounds here, the rest, if any, below)
                    const int nfix = 2 * (me_ + 1);
                    auto fixed_pos = [&](int f) -> int { return f <= me_ ? f * SK : rl - (f - me_) * SK; };
                    auto fixed_ok = [&](int f) -> bool {
                        if (!mine || f >= nfix) return false;
                        if (f <= me_) return (f + 1) * SK <= rl;
                        const int pp = rl - (f - me_) * SK;
                        return pp >= 0 && !((pp % SK) == 0 && pp <= me_ * SK);                                      // (a tail block that is a head block too is looked up as that)
                    };
                    // a wave's candidates of one round go to the pooled queue together: one LDS atomic per wave and round
                    auto reserve = [&](uint32_t tot) -> uint32_t {
                        if constexpr (!PO_POOLED) { const uint32_t b = qn1w; qn1w += tot; return b; }
                        uint32_t base = 0;
                        if (lane_ == 0) base = atomicAdd(&S.qn[par], tot);
                        return (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    };
                    // (every slot's load is issued before the first answer is looked at: they travel together)
                    constexpr int NSLOT = LG == 2 ? 2 : LG == 1 ? 3 : 4;
                    uint32_t fx = 0;
                    {
                        uint32_t fcode[NSLOT], fbits[NSLOT]; bool fokv[NSLOT];
#pragma unroll
                        for (int u = 0; u < NSLOT; u++) {
                            const int f = (int)gl + u * (int)G;
                            fokv[u] = fixed_ok(f);
                            fcode[u] = cdp_code8(roff + (uint32_t)(fokv[u] ? fixed_pos(f) : 0));
                        }
#pragma unroll
                        for (int u = 0; u < NSLOT; u++) fbits[u] = bmL[fcode[u] >> 3];
#pragma unroll
                        for (int u = 0; u < NSLOT; u++) if (fokv[u] && ((fbits[u] >> (fcode[u] & 7u)) & 1u)) fx |= 1u << u;
                    }
                    constexpr int NR = SS == 8 ? 6 : SS == 4 ? 12 : 18;
                    bool first = true;
                    for (int ee = e0; __any(ee < e1) || (first && __any(fx != 0u)); ee += NR, first = false) {
                        const bool act = ee < e1;
                        uint32_t hA = 0;
                        if (__any(act)) {
                            const uint32_t pos = roff + (uint32_t)SS * (uint32_t)(act ? ee : 0);
                            const uint32_t a = (pos >> 2) & ~3u, bo = (pos & 15u) * 2u;
                            constexpr int NW = (2 * SS * (NR - 1) + 16 + 31) / 32 + 1;
                            uint32_t W[NW];
#pragma unroll
                            for (int t = 0; t < NW; t++) W[t] = lds_u32_at(a + 4u * (uint32_t)t);
                            uint32_t R[NW - 1];
#pragma unroll
                            for (int t = 0; t < NW - 1; t++) R[t] = __builtin_amdgcn_alignbit(W[t + 1], W[t], bo);
                            constexpr int NB = 6;
#pragma unroll
                            for (int j0 = 0; j0 < NR; j0 += NB) {
                                uint32_t bi[NB], bb[NB];
#pragma unroll
                                for (int u = 0; u < NB; u++) {
                                    const int b = 2 * SS * (j0 + u), wj = b >> 5, o = b & 31;
                                    const uint32_t word = o <= 16 ? R[wj] : __builtin_amdgcn_alignbit(R[wj + 1 < NW - 1 ? wj + 1 : wj], R[wj], 16);
                                    const uint32_t off = (uint32_t)(o <= 16 ? o : o - 16);
                                    bi[u] = __builtin_amdgcn_ubfe(word, off, 3u);
                                    bb[u] = lds_byte_at(BP_LDS_BMA + __builtin_amdgcn_ubfe(word, off + 3u, 13u));
                                }
                                asm volatile("" ::: "memory");
#pragma unroll
                                for (int u = 0; u < NB; u++) hA |= __builtin_amdgcn_ubfe(bb[u], bi[u], 1u) << (j0 + u);
                            }
                            const int nv = act ? (e1 - ee < NR ? e1 - ee : NR) : 0;
                            hA &= (1u << nv) - 1u;
                        }
                        uint32_t fxn = first ? fx : 0u;
                        const uint32_t c = (uint32_t)__popc(hA) + (uint32_t)__popc(fxn);
                        const uint32_t inc = kvq_wave_incl_scan(c);
                        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                        if (tot) {
                            uint32_t idx = reserve(tot) + inc - c;
                            while (hA) {
                                const int j = __ffs((int)hA) - 1; hA &= hA - 1u;
                                if (idx < PO_QCAP) q1[idx] = k | ((uint32_t)(SS * (ee + j)) << 9);        // beyond the cap: dropped, the stretch is redone in halves
                                idx++;
                            }
                            while (fxn) {
                                const int u = __ffs((int)fxn) - 1; fxn &= fxn - 1u;
                                if (idx < PO_QCAP) q1[idx] = k | ((uint32_t)fixed_pos((int)gl + u * (int)G) << 9) | (1u << BP_Q1_KIND);
                                idx++;
                            }
                        }
                    }
                    // more fixed blocks than those rounds of the group's lanes (many errors allowed, narrow groups): one push round each
                    for (int f0 = NSLOT * (int)G; f0 < nfix; f0 += (int)G) {
                        const int f = f0 + (int)gl;
                        const bool hit = fixed_block(fixed_pos(f), fixed_ok(f));
                        const uint64_t mm = __ballot(hit);
                        if (mm) {
                            const uint32_t idx = reserve((uint32_t)__popcll(mm)) + (uint32_t)__popcll(mm & kvq_lanemask_lt());
                            if (hit && idx < PO_QCAP) q1[idx] = k | ((uint32_t)fixed_pos(f) << 9) | (1u << BP_Q1_KIND);
                        }
                    }
                }
        KVQ_MARK("filter end / P4a");
                if constexpr (PO_EARLY) {
                    // the next tile's text sets out now: the verification below needs few registers, and the way to memory
                    // and back is as long as it takes
                    if (!fetched && sub + step >= rpw && pass0 + RP >= nrec) { fetch_tile(gn); fetched = true; }
                }
                if constexpr (PO_POOLED) {
                    KVQ_SETPRIO(0);
                    __syncthreads();                                        // every wave's candidates are in
                    PSTAMP(5);
                    if (tid == 0) { S.qn[par ^ 1u] = 0; S.qn[2u + (par ^ 1u)] = 0; }     // the next sub-pass's counters (nobody looks at them now)
                }
                KVQ_SETPRIO(3);

                // ---- P4a: one candidate per lane: index range -> (candidate, entry) work items ----
                const uint32_t n1 = PO_POOLED ? rfl(S.qn[par]) : qn1w;
                const bool over1 = n1 > PO_QCAP;                            // candidates were dropped
                const uint32_t n1_ok = (over1 || (dbg & 1u)) ? 0u : n1;
                // (the rounds go to the waves in turn, starting with another wave every sub-pass)
                const uint32_t qa0 = PO_POOLED ? ((wave - pc) & 7u) * 64u : 0u, qstep = PO_POOLED ? 64u * ST_WAVES : 64u;
                if (qa0 < n1_ok) {
                    GlbWords ixb; uint32_t o_all;
                    { const BpArgsPtr A = bp_args(A_); ixb = (GlbWords)A->X.blob; o_all = A->X.off_start_all; }
                    const uint32_t lane_ = (uint32_t)tid & 63u;
                    for (uint32_t q0 = qa0; q0 < n1_ok; q0 += qstep) {
                        const uint32_t qi = q0 + lane_;
                        uint32_t en0 = 0, ne = 0;
                        if (qi < n1_ok) {
                            const uint32_t cd = q1[qi];
                            const uint32_t code = cdp_code8((S.rinfo[cd & 511u] & 0xFFFFu) + ((cd >> 9) & 0xFFFFu));
                            const uint32_t at = code + ((cd >> BP_Q1_KIND) ? o_all : 0u);
                            en0 = ixb[at]; ne = ixb[at + 1u] - en0;
                        }
                        const uint32_t inc = kvq_wave_incl_scan(ne);
                        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                        if (tot) {
                            uint32_t base = 0;
                            if constexpr (PO_POOLED) {
                                if (lane_ == 0) base = atomicAdd(&S.qn[2u + par], tot);
                                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                            } else { base = qn2w; qn2w += tot; }
                            base += inc - ne;
                            for (uint32_t j = 0; j < ne; j++)
                                if (base + j < PO_Q2CAP) q2[base + j] = (qi << 22) | (en0 + j);
                        }
                    }
                }
                if constexpr (PO_POOLED) {
                    KVQ_SETPRIO(0);
                    __syncthreads();                                        // every work item is in
                    PSTAMP(6);
                    KVQ_SETPRIO(3);
                }
                const uint32_t n2 = PO_POOLED ? rfl(S.qn[2u + par]) : qn2w;
                const bool over = over1 || n2 > PO_Q2CAP;
                if (over && step > 1u) { step >>= 1; continue; }            // (the barrier above stands between this sub-pass's queues and the next one's)
                if (over && (PO_POOLED ? tid == 0 : (tid & 63) == 0)) S.fallback = 1u;   // one read floods the queues: the batch goes to the exhaustive kernels

        KVQ_MARK("P4b");
                // ---- P4b: one work item per lane ----
                {
                    const uint32_t n2_ok = over ? 0u : n2;
                    const uint32_t qb0 = PO_POOLED ? ((wave + pc) & 7u) * 64u : 0u;
                    if (qb0 < n2_ok) {
                        GlbWords ixb; uint32_t o_tab2, o_anc, o_all; int me_;
                        { const BpArgsPtr A = bp_args(A_); ixb = (GlbWords)A->X.blob; o_tab2 = A->X.off_tab2; o_anc = A->X.off_ent_anc; o_all = A->X.off_ent_all; me_ = A->P.maxerrors; }
                        const uint32_t lane_ = (uint32_t)tid & 63u;
                        for (uint32_t i0 = qb0; i0 < n2_ok; i0 += qstep) {
                            const uint32_t ii = i0 + lane_;
                            const bool active = ii < n2_ok;
                            uint32_t kind = 0, ro = 0; int p = 0, rl_ = 0; uint64_t en = 0;
                            if (active) {
                                const uint32_t it = q2[ii];
                                const uint32_t cd = q1[it >> 22];
                                const uint32_t ri = S.rinfo[cd & 511u];
                                ro = ri & 0xFFFFu; rl_ = (int)(ri >> 16);
                                p = (int)((cd >> 9) & 0xFFFFu); kind = cd >> BP_Q1_KIND;
                                const uint32_t at = (kind ? o_all : o_anc) + 2u * (it & 0x3FFFFFu);
                                en = *reinterpret_cast<const __attribute__((address_space(1))) uint64_t *>(ixb + at);
                            }
                            po_verify(A_, ixb, o_tab2, me_, active, ro, rl_, p, kind, en, g0, SS);
                        }
                    }
                }
                KVQ_SETPRIO(2);
                sub += step;
                if constexpr (PO_POOLED) { if (sub < rpw || pass0 + RP < nrec) __syncthreads(); }        // the queues are filled again within this tile
            }
            KVQ_SETPRIO(0);
            // the '@' / '+' checks of the pass's records (the bytes have come back long ago)
            if (have && (c0 != '@' || cp != '+')) {                          // (lanes that fetched nothing never look)
                const BpArgsPtr A = bp_args(A_);
                const uint32_t m = jn + 4u * k;
                const uint32_t rstart = m == 0 ? ST_PRE + (Ja & 15u) : (uint32_t)S.nl[m - 1] + 1u, plus = (uint32_t)S.nl[m + 1] + 1u;
                const int64_t tf = A->fpos_base + (int64_t)g0;
                if (c0 != '@') atomicMin(A->P.err, ((unsigned long long)(tf + rstart - ST_PRE) << 16) | (0ull << 8) | c0);
                else atomicMin(A->P.err, ((unsigned long long)(tf + plus - ST_PRE) << 16) | (1ull << 8) | cp);
            }
        KVQ_MARK("P4 end");
        }
        KVQ_SETPRIO(0);
        if (!PO_STATIC && tid == ST_THREADS - 64) {
            if (!shards_left) drawn = ntiles;
            else if (drawn >= bp_shard_begin(my_shard + 1u, ntiles)) {
                my_shard = my_shard + 1u == BP_SHARDS ? 0u : my_shard + 1u; shards_left--;
                drawn = bp_draw(bp_args(A_)->tile_ctr, ntiles, my_shard, shards_left);
            }
            S.next_tile = drawn;
        }
        if (!fetched) fetch_tile(gn);
        KVQ_MARK("tile end");
        PSTAMP(7);
        g_done = g; g = gn;
    }

    atomicMax(&S.longest_p1, my_longest);
    __syncthreads();
    unsigned long long *const ctr = (bp_args(A_)->P.ctr + (size_t)(blockIdx.x % KVQ_STAGE_COPIES) * KVQ_STAGE_SLOTS);
    if constexpr (STAMPS) {
        if (tid == 0) for (int i = 0; i < 8; i++) atomicAdd(&ctr[KVQ_CTR_RL_ + 900 + i], stamp_acc[i]);
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
#undef PO_TEXT
#undef PSTAMP
}
