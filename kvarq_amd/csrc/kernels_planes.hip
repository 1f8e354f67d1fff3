// kvarq_amd/csrc/kernels_planes.hip -- second generation of the fused seed-filter scan
// (same contract as kvq_scan_seeded in kernels_seeded.hip, which stays selectable with
// KVQ_KERNEL=v1).  Every byte of the text is handled exactly once, in registers:
//
//   P1  512 threads x 64 contiguous bytes straight from HBM (registers, fetched one tile
//       ahead; no copy of the text in LDS).  SWAR per dword gives three bit-planes of the
//       32 KiB window: newline flags (kept in registers), "score >= Amin" flags and 2-bit
//       base codes (both to LDS, 1 and 2 bits per byte).  Prefix scan -> sorted newline
//       offsets + the byte that follows each newline ('@' / '+' checks need nothing else).
//   P2  first record of the tile, speculated and validated as in v1.
//   P3a quality trim, one lane per read: the score line is a bit range of the plane;
//       longest run of ones by shifts (64 bits a step), first-longest-wins merge.
//   P3b seed filter, four lanes per read: a read's packed bases ARE a bit range of the
//       code plane, so each 8-mer code is one v_alignbit away.  Only even read positions
//       are looked up: the anchor blocks of a sequence are indexed at offsets
//       {0,8,..,8e} and {1,9,..,8e+1}, so an alignment at an odd offset is found through
//       the shifted set (needs sequence length >= 8(e+1)+1).  The 2(e+1) fixed head / tail
//       blocks of the read are looked up separately (index of all sequence positions).
//   P4  candidates -> (candidate, entry) items -> byte-exact verification against the text
//       in global memory (just streamed: L2 hits) -> hits.
//
// LDS per workgroup 52 KiB -> three workgroups (24 waves) per CU.
#include "kvq_host.h"

#define PT_TILE 28672u             // bytes a tile owns (448 blocks of 64)
#define PT_OV 4096u                // look-ahead (64 blocks): 512 threads x 64 B = the whole window
#define PT_WIN (PT_TILE + PT_OV)
#define PT_THREADS 512
#define PT_WAVES (PT_THREADS / 64)
#define PT_NLCAP 2048
#define PT_RCAP 512
#define PT_QCAP 512
#define PT_Q2CAP 1024

struct PlanesLds {
    uint32_t gdp[PT_WIN / 32 + 4];       // 1 bit per byte: score byte is >= Amin
    uint32_t cdp[PT_WIN / 16 + 4];       // 2 bits per byte: (byte >> 1) & 3
    uint16_t nl[PT_NLCAP];               // window offsets of every '\n', ascending
    uint8_t  nlnext[PT_NLCAP];           // the byte behind that newline
    uint8_t  firstb[PT_THREADS + 8];     // first byte of every thread's block
    uint32_t bm2[4096];                  // 2 bits per 8-mer code (anchor / anywhere)
    uint32_t hist[KVQ_RL_BINS];
    uint2    q1[PT_QCAP];                // candidate: x = rec | pos << 16, y = code | kind << 16
    uint32_t q2[PT_Q2CAP];               // work item: candidate << 22 | index entry
    uint32_t rinfo[PT_RCAP];             // read offset in the window | rl << 16
    uint32_t wtot[PT_WAVES];
    uint32_t n_owned, qn, q2n, longest_p1, records, fallback;
};

struct PTile {
    uint32_t a, b, t, g0, own_begin, own_end, load_hi;
};

__device__ __forceinline__ PTile ptile(uint32_t g, const uint32_t *chunk_off, const uint32_t *tile_chunk, const uint32_t *tile_first)
{
    (void)chunk_off; (void)tile_first;
    PTile J;
    const uint4 q = reinterpret_cast<const uint4 *>(tile_chunk)[g];     // the per-tile table of kvq_expand_tiles
    J.a = q.x; J.b = q.y; J.t = q.z;
    J.g0 = (J.a & ~15u) + J.t * PT_TILE;
    J.own_end = J.g0 + PT_TILE < J.b ? J.g0 + PT_TILE : J.b;
    J.own_begin = J.t == 0 ? J.a : J.g0;
    J.load_hi = J.g0 + PT_WIN < J.b ? J.g0 + PT_WIN : J.b;
    return J;
}

// is the seed (read block at rp, sequence block at sq) live: in range, equal 2-bit codes
__device__ __forceinline__ bool seed_live_g(const uint8_t *rd, int rl, int rp, const uint8_t *seq, int seql, int sq)
{
    if (rp < 0 || rp + SK > rl || sq < 0 || sq + SK > seql) return false;
    return glb_code8(rd + rp) == glb_code8(seq + sq);
}

// one (candidate, index entry) pair = one diagonal of one read against one sequence
// (same rules as verify_item of v1; the read's bytes come from global memory and the
// seed set is the planes variant's: fixed head/tail blocks, then anchors by sequence offset,
// anchors counting only at even read positions)
__device__ __forceinline__ void verify_item_g(const KvqParams &P, const PlanesLds &S, const uint8_t *win, bool active,
                                              uint32_t rec, int p, uint32_t kind, uint64_t en, int64_t tile_fpos)
{
    bool hitAB = false, hitC = false;
    int s = 0, rl = 0, lenAB = 0, lenC = 0, sposAB = 0, sposC = 0; uint32_t keyAB = 0, keyC = 0;
    int64_t fpos = 0;
    if (active) {
        const uint32_t ri = S.rinfo[rec];
        const uint32_t roff = ri & 0xFFFFu; rl = (int)(ri >> 16);
        const uint8_t *rd = win + roff;
        fpos = tile_fpos + (int64_t)roff;
        const int q = (int)(en & 4095u);
        s = (int)((en >> 12) & 0xFFFFFu);
        const uint8_t *seq = P.tab + (uint32_t)((en >> 32) & 0xFFFFFu);
        const int seql = (int)(en >> 52);
        const int mo = P.minoverlap, me = P.maxerrors;
        const int d = q - p;                             // sequence index = read index + d
        const int a = d < 0 ? -d : 0;
        const int L = (rl < seql - d ? rl : seql - d) - a;
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
        if ((canAB || canC) && L > 0) {
            int mism = 0, j = 0;
            const uint8_t *x = rd + a, *y = seq + a + d;
            for (; j + 4 <= L && mism <= me; j += 4) {
                uint32_t rw, sw;
                __builtin_memcpy(&rw, x + j, 4); __builtin_memcpy(&sw, y + j, 4);
                mism += diff_bytes(rw, sw);
            }
            for (; j < L && mism <= me; j++) mism += (x[j] != y[j]);
            if (mism <= me) {
                // canonical discoverer: no live seed earlier in the order
                // [fixed read blocks by position] then [anchor blocks by sequence offset]
                bool earlier = false;
                for (int jj = 0; jj <= me && !earlier; jj++) {
                    const int ph = jj * SK, pt = rl - (jj + 1) * SK;
                    if (ph + SK <= rl && (kind == 0u || ph < p)) earlier = seed_live_g(rd, rl, ph, seq, seql, ph + d);
                    if (!earlier && pt >= 0 && (kind == 0u || pt < p)) earlier = seed_live_g(rd, rl, pt, seq, seql, pt + d);
                }
                if (kind == 0u) {
                    // anchors sit at sequence offsets 8j and 8j+1 and are looked up at even read positions only
                    for (int jj = 0; jj <= me && !earlier; jj++)
                        for (int sft = 0; sft < 2 && !earlier; sft++) {
                            const int o = jj * SK + sft;
                            if (o < q && ((o - d) & 1) == 0) earlier = seed_live_g(rd, rl, o - d, seq, seql, o);
                        }
                }
                if (!earlier) { hitAB = canAB; hitC = canC; }
            }
        }
    }
    kvq_emit(P, hitAB, fpos, s, sposAB, lenAB, rl, keyAB);
    kvq_emit(P, hitC, fpos, s, sposC, lenC, rl, keyC);
}

// 32 bits of a bit-plane starting at bit `bit`
__device__ __forceinline__ uint32_t plane32(const uint32_t *pl, uint32_t bit)
{
    const uint32_t w = bit >> 5;
    return __builtin_amdgcn_alignbit(pl[w + 1], pl[w], bit & 31u);
}

extern "C" __global__ void __launch_bounds__(PT_THREADS, 4)
kvq_scan_planes(KvqParams P, SeedTables X, const uint8_t *__restrict__ data, int64_t fpos_base,
                const uint32_t *__restrict__ chunk_off, const uint32_t *__restrict__ tile_chunk,
                const uint32_t *__restrict__ tile_first, uint32_t ntiles, uint32_t *__restrict__ tile_report, uint32_t dbg)
{
    extern __shared__ __align__(16) uint8_t lds_raw[];
    PlanesLds &S = *reinterpret_cast<PlanesLds *>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = rfl((uint32_t)tid >> 6);
    unsigned long long stamp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, stamp_t = 0;
#define PSTAMP(i) do { if (dbg & 16u) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } } while (0)

    for (int i = tid; i < 4096; i += PT_THREADS) S.bm2[i] = X.bm2[i];
    for (int i = tid; i < KVQ_RL_BINS; i += PT_THREADS) S.hist[i] = 0;
    if (tid == 0) { S.longest_p1 = 0; S.records = 0; }
    if (tid < 8) { S.gdp[PT_WIN / 32 + (tid & 3)] = 0; S.cdp[PT_WIN / 16 + (tid & 3)] = 0; }

    // this thread's 64 bytes of the next tile
    uint4 pre[4];
    const uint32_t blk = (uint32_t)tid * 64u;                  // window offset of the block
    if (blockIdx.x < ntiles) {
        const PTile J = ptile(blockIdx.x, chunk_off, tile_chunk, tile_first);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t gp = J.g0 + blk + 16u * r;
            pre[r] = gp < J.load_hi ? *reinterpret_cast<const uint4 *>(data + gp) : make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();

    for (uint32_t g = blockIdx.x; g < ntiles; g += gridDim.x) {
        const PTile J = ptile(g, chunk_off, tile_chunk, tile_first);
        const uint8_t *win = data + J.g0;                       // window offset 0
        if (dbg & 16u) stamp_t = __builtin_amdgcn_s_memtime();

        // ---- P1: the block in registers -> planes ----
        uint32_t x[16];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            uint32_t v[4] = { pre[r].x, pre[r].y, pre[r].z, pre[r].w };
            const uint32_t gp = J.g0 + blk + 16u * r;
            if (gp >= J.load_hi) { v[0] = v[1] = v[2] = v[3] = 0; }
            else if (gp < J.own_begin || gp + 16u > J.load_hi) {             // the text's ends: zero what lies outside (rare)
#pragma unroll
                for (int d = 0; d < 4; d++) v[d] &= (kvq_range_flags(gp + 4u * d, J.own_begin, J.load_hi) >> 7) * 0xFFu;
            }
            x[4 * r] = v[0]; x[4 * r + 1] = v[1]; x[4 * r + 2] = v[2]; x[4 * r + 3] = v[3];
        }
        uint32_t nlo = 0, nhi = 0;
        {
            const uint32_t addk = (uint32_t)(0x80 - P.amin) * 0x01010101u;
            uint32_t glo = 0, ghi = 0, cw[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int d = 0; d < 16; d++) {
                const uint32_t n4 = (kvq_nl_flags(x[d]) * 0x00204081u) >> 28;
                const uint32_t g4 = good4(x[d], addk);
                const uint32_t c8 = (((x[d] >> 1) & 0x03030303u) * 0x01041040u) >> 24;
                if (d < 8) { nlo |= n4 << (4 * d); glo |= g4 << (4 * d); } else { nhi |= n4 << (4 * (d - 8)); ghi |= g4 << (4 * (d - 8)); }
                cw[d >> 2] |= c8 << (8 * (d & 3));
            }
            *reinterpret_cast<uint2 *>(&S.gdp[2 * tid]) = make_uint2(glo, ghi);
            *reinterpret_cast<uint4 *>(&S.cdp[4 * tid]) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            S.firstb[tid] = (uint8_t)(x[0] & 0xFFu);
        }
        const uint32_t cnt = __popc(nlo) + __popc(nhi);
        const uint32_t own_end_w = J.own_end - J.g0;             // window offset where ownership ends (block boundary or end of text)
        const uint32_t end_w = J.load_hi - J.g0;
        const uint32_t incl = kvq_wave_incl_scan(cnt);
        if (lane == 63) S.wtot[wave] = incl;
        if (tid == 0) { S.n_owned = 0; S.fallback = 0; S.qn = 0; S.q2n = 0; }
        __syncthreads();
        PSTAMP(0);

        // ---- P1b: sorted newline offsets, the byte behind each newline ----
        uint32_t n_all = 0;
        {
            uint32_t mine = 0;
#pragma unroll
            for (int w = 0; w < PT_WAVES; w++) { const uint32_t t = S.wtot[w]; if (w == (int)wave) mine = n_all; n_all += t; }
            uint32_t n = mine + incl - cnt;
            {
                // owned newlines: one LDS atomic per wave
                const uint32_t o = kvq_wave_incl_scan(blk < own_end_w ? cnt : 0u);
                if (lane == 63 && o) atomicAdd(&S.n_owned, o);
            }
            uint64_t nm = (uint64_t)nlo | ((uint64_t)nhi << 32);
            while (nm) {
                const int bit = __ffsll((long long)nm) - 1; nm &= nm - 1ull;
                uint32_t nb;
                if (bit == 63) nb = S.firstb[tid + 1];
                else {
                    // dword (bit + 1) >> 2 of the block, picked with selects (no indexed registers)
                    const int dd = (bit + 1) >> 2;
                    uint32_t q0 = dd & 8 ? x[8] : x[0], q1 = dd & 8 ? x[9] : x[1], q2 = dd & 8 ? x[10] : x[2], q3 = dd & 8 ? x[11] : x[3];
                    uint32_t q4 = dd & 8 ? x[12] : x[4], q5 = dd & 8 ? x[13] : x[5], q6 = dd & 8 ? x[14] : x[6], q7 = dd & 8 ? x[15] : x[7];
                    q0 = dd & 4 ? q4 : q0; q1 = dd & 4 ? q5 : q1; q2 = dd & 4 ? q6 : q2; q3 = dd & 4 ? q7 : q3;
                    q0 = dd & 2 ? q2 : q0; q1 = dd & 2 ? q3 : q1;
                    q0 = dd & 1 ? q1 : q0;
                    nb = (q0 >> (8 * ((bit + 1) & 3))) & 0xFFu;
                }
                if (n < PT_NLCAP) { S.nl[n] = (uint16_t)(blk + (uint32_t)bit); S.nlnext[n] = (uint8_t)nb; }
                n++;
            }
        }
        n_all = rfl(n_all);
        __syncthreads();
        PSTAMP(1);
        // the block's registers are free now: fetch this thread's 64 bytes of the next tile
        if (g + gridDim.x < ntiles) {
            const PTile N = ptile(g + gridDim.x, chunk_off, tile_chunk, tile_first);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t gp = N.g0 + blk + 16u * r;
                pre[r] = gp < N.load_hi ? *reinterpret_cast<const uint4 *>(data + gp) : make_uint4(0, 0, 0, 0);
            }
        }

        // ---- P2 (every wave, redundantly): which records does this tile own? ----
        uint32_t nrec = 0, jn = TR_NONE;
        {
            const uint32_t n_nl = n_all < PT_NLCAP ? n_all : PT_NLCAP;
            const uint32_t n_owned = rfl(S.n_owned);
            uint32_t fallback = n_all > PT_NLCAP ? 1u : 0u;
            if (J.t == 0) jn = 0;                                            // chunk start: exact
            else {
                // lane m (m >= 1): the line behind the tile's m-th newline starts with '@' and the
                // line two further on with '+'?
                const uint32_t m = (uint32_t)lane;
                bool ok = false;
                if (m >= 1 && m <= 8 && m <= n_owned && m + 2 <= n_nl)
                    ok = (uint32_t)S.nl[m - 1] + 1u < end_w && (uint32_t)S.nl[m + 1] + 1u < end_w && S.nlnext[m - 1] == '@' && S.nlnext[m + 1] == '+';
                const uint64_t mk = __ballot(ok);
                if (mk) jn = (uint32_t)(__ffsll((long long)mk) - 1);
            }
            if (jn != TR_NONE) {
                if (jn <= n_owned) nrec = (n_owned - jn) / 4u + 1u;
                if (nrec > 0 && jn + 4u * nrec > n_nl) {                     // chunk end (partial record, 1033) or a record beyond the look-ahead
                    const uint32_t fit = n_nl >= jn ? (n_nl - jn) / 4u : 0u;
                    if (J.load_hi < J.b || n_all > PT_NLCAP) fallback = 1u;
                    nrec = fit;
                }
                if (nrec > PT_RCAP) { nrec = PT_RCAP; fallback = 1u; }
            }
            if (tid == 0) {
                tile_report[g] = (n_owned & 0xFFFFu) | ((jn & 0xFFu) << 16) | (fallback ? TR_FLAG_FALLBACK : 0u);
                S.records += nrec;
            }
        }
        PSTAMP(2);

        // ---- P3a: checks + quality trim, one lane per read ----
        const int64_t tile_fpos = fpos_base + (int64_t)J.g0;
        for (uint32_t k = (uint32_t)tid; k < nrec; k += PT_THREADS) {
            const uint32_t m = jn + 4u * k;
            const uint32_t rstart = m == 0 ? J.a - J.g0 : (uint32_t)S.nl[m - 1] + 1u;
            const uint32_t sread = (uint32_t)S.nl[m] + 1u, plus = (uint32_t)S.nl[m + 1] + 1u, sscore = (uint32_t)S.nl[m + 2] + 1u, n3 = S.nl[m + 3];
            {
                const uint32_t c0 = m == 0 ? (uint32_t)win[rstart] : (uint32_t)S.nlnext[m - 1], cp = S.nlnext[m + 1];
                if (c0 != '@') atomicMin(P.err, ((unsigned long long)(tile_fpos + rstart) << 16) | (0ull << 8) | c0);
                else if (cp != '+') atomicMin(P.err, ((unsigned long long)(tile_fpos + plus) << 16) | (1ull << 8) | cp);
            }
            const int Q = (int)(n3 - sscore);                               // the closing '\n' is implied
            Seg sg; sg.beg = 0; sg.len = 0; sg.pre = 0; sg.suf = 0; sg.best = 0; sg.bstart = 0;
            for (int c0 = 0; c0 < Q; c0 += 64) {
                const int n = Q - c0 < 64 ? Q - c0 : 64;
                const uint32_t bit = sscore + (uint32_t)c0;
                uint64_t mk = (uint64_t)plane32(S.gdp, bit) | ((uint64_t)plane32(S.gdp, bit + 32u) << 32);
                if (n < 64) mk &= (1ull << n) - 1ull;
                Seg sub; sub.beg = c0; sub.len = n;
                const uint64_t inv = ~mk;
                sub.pre = inv ? __ffsll((long long)inv) - 1 : 64; if (sub.pre > n) sub.pre = n;
                const uint64_t top = ~(mk << (64 - n));
                sub.suf = top ? __clzll((long long)top) : 64; if (sub.suf > n) sub.suf = n;
                int bl, bs; longest_run64(mk, n, bl, bs);
                sub.best = bl; sub.bstart = c0 + bs;
                sg = c0 == 0 ? sub : seg_merge(sg, sub);
            }
            const int rl = sg.best;
            const uint32_t roff = sread + (uint32_t)sg.bstart;                                         // 1070
            if (rl < KVQ_RL_BINS) atomicAdd(&S.hist[rl], 1u);                                         // 394-402
            atomicMax(&S.longest_p1, (uint32_t)(rl + 1));
            S.rinfo[k] = roff | ((uint32_t)rl << 16);
        }
        __syncthreads();
        PSTAMP(3);

        // ---- P3b / P4: seed filter (G lanes per read) and verification, in stretches ----
        uint32_t lg = 0;
        while (lg < 6u && (2u << lg) * nrec <= PT_THREADS) lg++;
        const uint32_t G = 1u << lg, RP = PT_THREADS >> lg;
        const uint32_t gl = (uint32_t)tid & (G - 1u), gr = (uint32_t)tid >> lg;
        for (uint32_t pass0 = 0; pass0 < nrec; pass0 += RP) {
            const uint32_t k = pass0 + gr;
            const bool have = k < nrec;
            uint32_t roff = 0; int rl = 0;
            if (have) { const uint32_t ri = S.rinfo[k]; roff = ri & 0xFFFFu; rl = (int)(ri >> 16); }
            const uint32_t npass = nrec - pass0 < RP ? nrec - pass0 : RP;
            uint32_t sub = 0, step = RP;
            for (;;) {
                const bool mine = have && rl >= P.minreadlength && !(dbg & 2u) && gr >= sub && gr - sub < step;       // 1100
                // even positions 0, 2, .. of the read, a slice per lane, 16 at a time
                int e0 = 0, e1 = 0;
                if (mine) {
                    const int NPe = (rl - (SK - 1) + 1) >> 1;
                    const int per = (NPe + (int)G - 1) >> lg;
                    e0 = (int)gl * per; if (e0 > NPe) e0 = NPe;
                    e1 = e0 + per; if (e1 > NPe) e1 = NPe;
                }
                for (int ee = e0; __any(ee < e1); ee += 16) {
                    const bool act = ee < e1;
                    uint32_t hbits = 0;                                      // bit j: position 2(ee + j) is an anchor code
                    if (__any(act)) {
                        const uint32_t bit = 2u * (roff + 2u * (uint32_t)ee);   // code plane: 2 bits per byte
                        const uint32_t w = bit >> 5, sh = bit & 31u;
                        const uint32_t W0 = S.cdp[w], W1 = S.cdp[w + 1], W2 = S.cdp[w + 2], W3 = S.cdp[w + 3];
                        const uint32_t R0 = __builtin_amdgcn_alignbit(W1, W0, sh), R1 = __builtin_amdgcn_alignbit(W2, W1, sh),
                                       R2 = __builtin_amdgcn_alignbit(W3, W2, sh);
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const uint32_t wn = j < 8 ? __builtin_amdgcn_alignbit(R1, R0, 4 * j) : __builtin_amdgcn_alignbit(R2, R1, 4 * (j - 8));
                            const uint32_t wv = S.bm2[(wn >> 4) & 0xFFFu];
                            hbits |= ((wv >> ((wn << 1) & 31u)) & 1u) << j;
                        }
                        const int nv = act ? (e1 - ee < 16 ? e1 - ee : 16) : 0;
                        hbits &= nv >= 16 ? 0xFFFFu : ((1u << nv) - 1u);
                    }
                    for (;;) {
                        const uint64_t mm = __ballot(hbits != 0);
                        if (!mm) break;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(&S.qn, (uint32_t)__popcll(mm));
                        base = rfl(base);
                        if (hbits) {
                            const int j = __ffs((int)hbits) - 1; hbits &= hbits - 1u;
                            const int pp = 2 * (ee + j);
                            const uint32_t cd = plane32(S.cdp, 2u * (roff + (uint32_t)pp)) & 0xFFFFu;
                            const uint32_t idx = base + (uint32_t)__popcll(mm & kvq_lanemask_lt());
                            if (idx < PT_QCAP) S.q1[idx] = make_uint2(k | ((uint32_t)pp << 16), cd);       // beyond the cap: dropped, the stretch is redone in halves
                        }
                    }
                }
                // the 2(e+1) fixed head / tail blocks of the read against the index of all sequence positions:
                // lane 0 of the group takes the head blocks, the last lane the tail blocks
                {
                    const bool head = mine && gl == 0, tail = mine && gl == G - 1u && G > 1u;
                    const bool both = mine && G == 1u;
                    for (int j = 0; j <= P.maxerrors; j++) {
                        for (int side = 0; side < 2; side++) {
                            const bool on = side == 0 ? (head || both) : (tail || both);
                            const int pp = side == 0 ? j * SK : rl - (j + 1) * SK;
                            bool hit = false; uint32_t cd = 0;
                            if (on && pp >= 0 && pp + SK <= rl && !(side == 1 && (pp % SK) == 0 && pp <= P.maxerrors * SK)) {
                                cd = plane32(S.cdp, 2u * (roff + (uint32_t)pp)) & 0xFFFFu;
                                hit = (S.bm2[cd >> 4] >> (((cd & 15u) << 1) + 1u)) & 1u;
                            }
                            const uint64_t mm = __ballot(hit);
                            if (mm) {
                                uint32_t base = 0;
                                if (lane == 0) base = atomicAdd(&S.qn, (uint32_t)__popcll(mm));
                                base = rfl(base);
                                const uint32_t idx = base + (uint32_t)__popcll(mm & kvq_lanemask_lt());
                                if (hit && idx < PT_QCAP) S.q1[idx] = make_uint2(k | ((uint32_t)pp << 16), cd | (1u << 16));
                            }
                        }
                    }
                }
                PSTAMP(4);
                __syncthreads();
                PSTAMP(5);

                // ---- P4a: one candidate per lane: index range -> (candidate, entry) work items ----
                const uint32_t qall = rfl(S.qn);
                const bool over1 = qall > PT_QCAP;
                const uint32_t qn = over1 ? 0u : qall;
                if (!(dbg & 1u))
                for (uint32_t q0 = wave * 64u; q0 < qn; q0 += PT_THREADS) {
                    const uint32_t qi = q0 + lane;
                    uint32_t en0 = 0, ne = 0;
                    if (qi < qn) {
                        const uint2 cd = S.q1[qi];
                        const uint32_t *st = (cd.y >> 16) ? X.start_all : X.start_anc;
                        en0 = st[cd.y & 0xFFFFu]; ne = st[(cd.y & 0xFFFFu) + 1u] - en0;
                    }
                    const uint32_t inc = kvq_wave_incl_scan(ne);
                    uint32_t base = 0;
                    if (lane == 63 && inc) base = atomicAdd(&S.q2n, inc);
                    base = __shfl(base, 63, 64) + inc - ne;
                    for (uint32_t j = 0; j < ne; j++)
                        if (base + j < PT_Q2CAP) S.q2[base + j] = (qi << 22) | (en0 + j);
                }
                __syncthreads();
                const uint32_t q2all = rfl(S.q2n);
                const bool over = over1 || q2all > PT_Q2CAP;
                if (over && step > 1u) {
                    __syncthreads();
                    if (tid == 0) { S.qn = 0; S.q2n = 0; }
                    step >>= 1;
                    __syncthreads();
                    continue;
                }
                if (over && tid == 0) S.fallback = 1u;                    // one read floods the queues: the batch goes to the exhaustive kernels

                // ---- P4b: one work item per lane ----
                {
                    const uint32_t q2n = over ? 0u : q2all;
                    if (!(dbg & 1u))
                    for (uint32_t i0 = wave * 64u; i0 < q2n; i0 += PT_THREADS) {
                        const uint32_t ii = i0 + lane;
                        const bool active = ii < q2n;
                        uint32_t rec = 0, kind = 0; int p = 0; uint64_t en = 0;
                        if (active) {
                            const uint32_t it = S.q2[ii];
                            const uint2 cd = S.q1[it >> 22];
                            rec = cd.x & 0xFFFFu; p = (int)(cd.x >> 16); kind = cd.y >> 16;
                            en = (kind ? X.ent_all : X.ent_anc)[it & 0x3FFFFFu];
                        }
                        verify_item_g(P, S, win, active, rec, p, kind, en, tile_fpos);
                    }
                }
                __syncthreads();                                   // everyone is done with the queues
                PSTAMP(6);
                sub += step;
                if (sub >= npass) break;
                if (tid == 0) { S.qn = 0; S.q2n = 0; }
                __syncthreads();
            }
            if (pass0 + RP < nrec) {
                if (tid == 0) { S.qn = 0; S.q2n = 0; }
                __syncthreads();
            }
        }
        // (planes, newline list and rinfo of this tile are dead now; the barrier that ended the last
        // stretch -- or the one below when the tile owns no record -- protects them from the next P1)
        if (tid == 0 && S.fallback) atomicOr(&tile_report[g], TR_FLAG_FALLBACK);
        if (nrec == 0) __syncthreads();
        PSTAMP(7);
    }

    if ((dbg & 16u) && tid == 0) for (int i = 0; i < 8; i++) atomicAdd(&P.ctr[KVQ_CTR_RL_ + 900 + i], stamp_acc[i]);
    for (int i = tid; i < KVQ_RL_BINS; i += PT_THREADS)
        if (S.hist[i]) atomicAdd(&P.ctr[KVQ_CTR_RL_ + i], (unsigned long long)S.hist[i]);
    if (tid == 0) {
        if (S.longest_p1) atomicMax(&P.ctr[KVQ_CTR_LONGEST_], (unsigned long long)S.longest_p1);
        if (S.records) atomicAdd(&P.ctr[KVQ_CTR_RECORDS_], (unsigned long long)S.records);
    }
}

size_t kvq_planes_lds_bytes() { return sizeof(PlanesLds); }
uint32_t kvq_planes_tile_bytes() { return PT_TILE; }
