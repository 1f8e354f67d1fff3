// kvarq_amd/csrc/kernels_general.hip -- the exhaustive scan path: fully general
// (any bytes, any read/sequence length, any config), one kernel per stage of
// the reference's scan_filepart (workhorse.c:976-1197).  It serves sequences
// and configs the seed-filter kernel cannot (kernels_seeded.hip), and is the
// on-device cross-check of that kernel in the tests.
//
//   kvq_count_lines    newline count per 4 KiB segment of every chunk          (1018-1030)
//   kvq_scan_segments  per-chunk exclusive scan of those counts, records/chunk (1033: four '\n' = one record)
//   kvq_index_records  positions of the four '\n' of every complete record     (1024-1029)
//   kvq_trim_records   '@'/'+' checks, longest quality run, length histogram   (1037-1070)
//   kvq_match_all      every read against every listed sequence, classes A/B/C (1107-1175)
//   kvq_fold_batch     per-sequence counters, coverage/mutation fold, hit bytes (434-439, analyse.py:57-78)
#include "kvq_device.h"

// ---------------------------------------------------------------------------
// newline indexing
// ---------------------------------------------------------------------------

// flags of the newlines among the 16 bytes a lane owns in one round
struct Lane16 { uint32_t f[4]; uint32_t p; };

__device__ __forceinline__ Lane16 load_lane16(const uint8_t *data, uint32_t p, uint32_t a, uint32_t b)
{
    Lane16 r; r.p = p;
    if (p >= b || p + 16u <= a) { r.f[0] = r.f[1] = r.f[2] = r.f[3] = 0u; return r; }
    const uint4 v = *reinterpret_cast<const uint4 *>(data + p);      // p is a multiple of 16
    const uint32_t x[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t f = kvq_nl_flags(x[d]);
        if (p + 4u * d < a || p + 4u * d + 4u > b) f &= kvq_range_flags(p + 4u * d, a, b);
        r.f[d] = f;
    }
    return r;
}

// grid (gx, nchunks), 256 threads: wave w of block bx handles segments
// bx*4+w, bx*4+w+4*gx, ... of chunk blockIdx.y
extern "C" __global__ void __launch_bounds__(256)
kvq_count_lines(const uint8_t *__restrict__ data, const uint32_t *__restrict__ chunk_off,
                const uint32_t *__restrict__ chunk_seg_base, uint32_t *__restrict__ seg_cnt)
{
    const uint32_t c = blockIdx.y;
    const uint32_t a = chunk_off[c], b = chunk_off[c + 1];
    const uint32_t A = a & ~15u;
    const uint32_t nseg = chunk_seg_base[c + 1] - chunk_seg_base[c];
    const int lane = kvq_lane();
    for (uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6); k < nseg; k += gridDim.x * 4u) {
        uint32_t cnt = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const Lane16 l = load_lane16(data, A + k * KVQ_SEG_BYTES + r * 1024u + lane * 16u, a, b);
            cnt += __popc(l.f[0]) + __popc(l.f[1]) + __popc(l.f[2]) + __popc(l.f[3]);
        }
        cnt = kvq_wave_incl_scan(cnt);
        if (lane == 63) seg_cnt[chunk_seg_base[c] + k] = cnt;
    }
}

// one wave per chunk: exclusive scan of its segment counts (in place), number
// of complete records of the chunk
extern "C" __global__ void __launch_bounds__(256)
kvq_scan_segments(uint32_t nchunks, const uint32_t *__restrict__ chunk_seg_base,
                  uint32_t *__restrict__ seg_cnt, uint32_t *__restrict__ chunk_nrec)
{
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int lane = kvq_lane();
    const uint32_t s0 = chunk_seg_base[c], s1 = chunk_seg_base[c + 1];
    uint32_t run = 0;
    for (uint32_t s = s0; s < s1; s += 64u) {
        const uint32_t v = (s + lane < s1) ? seg_cnt[s + lane] : 0u;
        const uint32_t inc = kvq_wave_incl_scan(v);
        if (s + lane < s1) seg_cnt[s + lane] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) chunk_nrec[c] = run >> 2;
}

// same geometry as kvq_count_lines.  nl4[4*g + k] = batch offset of the k-th
// '\n' of record g; rec_start[g] = batch offset of its first byte
extern "C" __global__ void __launch_bounds__(256)
kvq_index_records(const uint8_t *__restrict__ data, const uint32_t *__restrict__ chunk_off,
                  const uint32_t *__restrict__ chunk_seg_base, const uint32_t *__restrict__ seg_excl,
                  const uint32_t *__restrict__ chunk_nrec, const uint32_t *__restrict__ chunk_rec_base,
                  uint32_t *__restrict__ nl4, uint32_t *__restrict__ rec_start)
{
    const uint32_t c = blockIdx.y;
    const uint32_t a = chunk_off[c], b = chunk_off[c + 1];
    const uint32_t A = a & ~15u;
    const uint32_t nseg = chunk_seg_base[c + 1] - chunk_seg_base[c];
    const uint32_t nrec = chunk_nrec[c];
    const uint32_t limit = nrec * 4u;
    const uint32_t g0 = chunk_rec_base[c];
    const int lane = kvq_lane();
    if (blockIdx.x == 0 && threadIdx.x == 0 && nrec > 0) rec_start[g0] = a;
    for (uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6); k < nseg; k += gridDim.x * 4u) {
        uint32_t base = seg_excl[chunk_seg_base[c] + k];
        if (base >= limit) continue;              // only the dropped partial tail lives here
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
            const Lane16 l = load_lane16(data, A + k * KVQ_SEG_BYTES + r * 1024u + lane * 16u, a, b);
            const uint32_t cnt = __popc(l.f[0]) + __popc(l.f[1]) + __popc(l.f[2]) + __popc(l.f[3]);
            const uint32_t inc = kvq_wave_incl_scan(cnt);
            uint32_t n = base + inc - cnt;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                uint32_t f = l.f[d];
                while (f) {
                    const int bit = __ffs((int)f) - 1;          // 7, 15, 23 or 31
                    f &= f - 1u;
                    const uint32_t pos = l.p + 4u * d + (uint32_t)(bit >> 3);
                    if (n < limit) {
                        nl4[(size_t)g0 * 4u + n] = pos;
                        if ((n & 3u) == 3u && n + 1u < limit) rec_start[g0 + ((n + 1u) >> 2)] = pos + 1u;
                    }
                    n++;
                }
            }
            base += __shfl(inc, 63, 64);
        }
    }
}

// The records of tiles that the fused scan kernel has skipped (kernels_seeded.hip, TR_FLAG_SKIPPED): one wave
// per tile walks the text from the tile's first owned byte and finds, from the exact number of newlines of the
// chunk in front of the tile, the records that start behind a newline the tile owns (tile 0 of a chunk: also
// the chunk's first record) -- the rule of the fused kernels -- with their four newlines, wherever those
// lie (a record may run to the end of the chunk).  nl4 / rec_start as kvq_index_records writes them.
struct KvqSkippedTile { uint32_t a, b, own_begin, own_end, seen, first; };

// How many items a kernel has to deal with, when that number is only known on the device (the redo of skipped tiles is
// enqueued behind every seed-filter launch, without the host looking: kvq_runtime.hip): `n` when d_n is null, else
// *d_n >> shift -- nothing when the batch's fail word says the whole batch is redone anyway (bit 0), and a number beyond
// `cap` (the tables are full) raises that bit, so that the batch IS redone as a whole.  The kernels walk their items with a
// stride of their grid: a launch of fixed size serves any count.
struct KvqDevCount { const unsigned int *d_n; unsigned int shift, cap; unsigned int *fail; };
__device__ __forceinline__ uint32_t kvq_dev_count(uint32_t n, const KvqDevCount &c)
{
    if (!c.d_n) return n;
    if (c.fail && (*c.fail & 1u)) return 0u;
    const uint32_t v = *c.d_n >> c.shift;
    if (v > c.cap) { if (c.fail && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) atomicOr(c.fail, 1u); return 0u; }
    return v;
}

// sixteen bytes of text -> bit i: byte i is a newline
__device__ __forceinline__ uint32_t kvq_newlines16(const uint4 v)
{
    const uint32_t x[4] = { v.x, v.y, v.z, v.w };
    uint32_t m = 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t t = x[d] ^ 0x0A0A0A0Au;
        const uint32_t z = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu);     // 0x80 in every byte that is '\n' (exact)
        m |= ((((z >> 7) * 0x00204081u) >> 21) & 0xFu) << (4 * d);
    }
    return m;
}

extern "C" __global__ void __launch_bounds__(256)
kvq_collect_skipped(const uint8_t *__restrict__ data, const KvqSkippedTile *__restrict__ tiles, uint32_t ntiles_, KvqDevCount dc,
                    uint32_t *__restrict__ nl4, uint32_t *__restrict__ rec_start, unsigned int *__restrict__ rec_count, uint32_t rec_cap)
{
    KVQ_BESIDE_SCAN();
    const uint32_t ntiles = kvq_dev_count(ntiles_, dc);
    const int lane = kvq_lane();
    for (uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6); w < ntiles; w += gridDim.x * 4u) {
    const KvqSkippedTile T = tiles[w];
    uint32_t idx = T.seen;                       // number (within the chunk) of the next newline met
    bool collecting = T.first != 0u;             // tile 0: the chunk's first record
    uint32_t rstart = T.a, cnt = 0, nlb[4] = { 0, 0, 0, 0 };
    // 4 KiB a round: four 16-byte vectors a lane, all four loads in flight (a record of thousands of bytes is a chain of
    // memory round trips otherwise; the text is 16-byte aligned, check_batch), the newlines of a vector as a 16-bit mask;
    // the wave then walks the masks that are not empty, in text order
    const uint32_t last_v = (T.b - 1u) & ~15u;
    for (uint32_t p4 = T.own_begin & ~15u; p4 < T.b; p4 += 4096u) {
        if (p4 >= T.own_end && !collecting) break;
        uint32_t m16[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t q = p4 + 1024u * (uint32_t)k + 16u * (uint32_t)lane;
            const uint4 v = *reinterpret_cast<const uint4 *>(data + (q < T.b ? q : last_v));      // (every load is issued whatever its place: a load behind a branch is waited for before the next one goes out)
            const uint32_t lo = q < T.own_begin ? (T.own_begin - q < 16u ? T.own_begin - q : 16u) : 0u;
            const uint32_t hi = q < T.b ? (T.b - q < 16u ? T.b - q : 16u) : 0u;
            m16[k] = hi > lo ? kvq_newlines16(v) & ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t row = p4 + 1024u * (uint32_t)k;
            if (row >= T.b || (row >= T.own_end && !collecting)) break;
            unsigned long long any = __ballot(m16[k] != 0u);
            while (any) {                            // (the same for every lane: the mask is the wave's)
                const int l = __ffsll((long long)any) - 1; any &= any - 1ull;
                uint32_t mm = (uint32_t)__shfl((int)m16[k], l, 64);
                while (mm) {
                    const uint32_t pos = row + 16u * (uint32_t)l + (uint32_t)(__ffs((int)mm) - 1); mm &= mm - 1u;
                    if (collecting) {
                        nlb[cnt++] = pos;
                        if (cnt == 4u) {
                            if (lane == 0) {
                                const unsigned int r = atomicAdd(rec_count, 1u);
                                if (r < rec_cap) { rec_start[r] = rstart; nl4[4 * (size_t)r] = nlb[0]; nl4[4 * (size_t)r + 1] = nlb[1]; nl4[4 * (size_t)r + 2] = nlb[2]; nl4[4 * (size_t)r + 3] = nlb[3]; }
                            }
                            collecting = false;
                        }
                    }
                    if ((idx & 3u) == 3u && pos < T.own_end) { collecting = true; rstart = pos + 1u; cnt = 0; }     // (an owned newline that ends a record: the next one starts behind it)
                    idx++;
                }
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------
// quality trim
// ---------------------------------------------------------------------------

// running state of the longest-run search over the quality line, fed 64 bits
// at a time (bit i = score byte >= Amin).  A run only counts when a low byte
// closes it; the first of equally long runs wins (workhorse.c:1055-1068).
struct RunState { int in_run; uint32_t run_start; int best; uint32_t best_start; };

__device__ __forceinline__ void run_feed(RunState &st, uint64_t good, int nbits, uint32_t base)
{
    int pos = 0;
    while (pos < nbits) {
        if (st.in_run) {
            const uint64_t z = (~good) >> pos;
            const int d = z ? __ffsll((long long)z) - 1 : 64;
            if (pos + d >= nbits) break;                       // run continues in the next word
            const int len = (int)(base + pos + d - st.run_start);
            if (len > st.best) { st.best = len; st.best_start = st.run_start; }
            st.in_run = 0;
            pos += d + 1;
        } else {
            const uint64_t o = good >> pos;
            const int d = o ? __ffsll((long long)o) - 1 : 64;
            if (pos + d >= nbits) break;
            st.run_start = base + pos + d;
            st.in_run = 1;
            pos += d;
        }
    }
}

#define KVQ_TRIM_RPW 16   // records per wave
// the longest run of scores >= amin among line[0 .. Q) (the byte behind it closes the last run), the first of equally long
// ones, by all 64 lanes of a wave; the answer in every lane (kernels_bp.hip)
__device__ void kvq_long_line_run(const uint8_t *line, uint32_t Q, int amin, int lane, int &best, uint32_t &best_start);

// one wave per record, KVQ_TRIM_RPW consecutive records per wave.
// read_off[g] = batch offset of the first base of the trimmed read,
// read_len[g] = its length, or -1 when shorter than minreadlength (workhorse.c:1100)
extern "C" __global__ void __launch_bounds__(256)
kvq_trim_records(KvqParams P, const uint8_t *__restrict__ data, int64_t fpos_base, uint32_t nrec_, KvqDevCount dc,
                 const uint32_t *__restrict__ nl4, const uint32_t *__restrict__ rec_start,
                 uint32_t *__restrict__ read_off, int32_t *__restrict__ read_len, int32_t count, uint32_t rpw,
                 unsigned int *__restrict__ long_count, uint32_t long_top)
{
    KVQ_BESIDE_SCAN();
    // long_count (the redo of skipped tiles only): a read of KVQ_LONG_READ bases or more goes to a list of its own, filled
    // from read_off[long_top] / read_len[long_top] downwards, at most KVQ_LONG_CAP of them: the matcher is launched once for
    // the many ordinary reads (a wave a read) and once for the few long ones (a read's sequences and alignments spread
    // over hundreds of waves) -- one grid cannot serve both without the host knowing the numbers
    const uint32_t nrec = kvq_dev_count(nrec_, dc);
    if (nrec == 0) return;
    // rpw: consecutive records per wave (KVQ_TRIM_RPW for a batch of ordinary reads; 1 for the few, possibly very
    // long records of skipped tiles)
    __shared__ unsigned int hist[KVQ_RL_BINS];
    __shared__ int longest;
    __shared__ unsigned int nproc;
    for (int i = threadIdx.x; i < KVQ_RL_BINS; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) { longest = -1; nproc = 0; }
    __syncthreads();

    const int lane = kvq_lane();
    for (uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6); (uint64_t)wave * rpw < nrec; wave += gridDim.x * 4u) {
    const uint32_t g_begin = wave * rpw;
    for (uint32_t g = g_begin; g < g_begin + rpw && g < nrec; g++) {
        const uint32_t rstart = rec_start[g];
        // (the redo's list also holds reads that the scan kernel has trimmed itself and only could not match -- one read
        // flooding its wave's queues: marked by this record start, read_off / read_len already in place, counted there)
        if (rstart == KVQ_REDO_TRIMMED) continue;
        if (lane == 0) atomicAdd(&nproc, 1u);
        const uint32_t n0 = nl4[4 * (size_t)g], n1 = nl4[4 * (size_t)g + 1], n2 = nl4[4 * (size_t)g + 2], n3 = nl4[4 * (size_t)g + 3];
        const uint32_t sread = n0 + 1u, plus = n1 + 1u, sscore = n2 + 1u;
        if (lane == 0) {
            const uint8_t c0 = data[rstart];
            if (c0 != '@')
                atomicMin(P.err, ((unsigned long long)(fpos_base + rstart) << 16) | (0ull << 8) | c0);
            else if (data[plus] != '+')
                atomicMin(P.err, ((unsigned long long)(fpos_base + plus) << 16) | (1ull << 8) | data[plus]);
        }
        // score line including its '\n': bytes [sscore, n3]
        const uint32_t qlen = n3 - sscore + 1u;
        RunState st; st.in_run = 1; st.run_start = 0; st.best = 0; st.best_start = 0;   // qtr starts at startscore (1055)
        if (qlen > 1024u) {
            // a line of thousands of scores: every lane sums up sixteen scores of each KiB, the sums are merged over the
            // wave (kvq_long_line_run, kernels_bp.hip) -- fed 64 scores at a time it is a chain of thousands of steps
            kvq_long_line_run(data + sscore, qlen - 1u, P.amin, lane, st.best, st.best_start);
        } else
        for (uint32_t o4 = 0; o4 < qlen; o4 += 1024u) {
            // (sixteen loads in flight; ordinary reads are done with the first few)
            uint8_t by[16]; bool ok[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t i = o4 + 64u * (uint32_t)k + (uint32_t)lane;
                by[k] = data[sscore + (i < qlen ? i : qlen - 1u)];                 // (unconditional loads travel together; qlen >= 1: the closing newline)
            }
#pragma unroll
            for (int k = 0; k < 16; k++) ok[k] = (o4 + 64u * (uint32_t)k + (uint32_t)lane < qlen) && ((int)(int8_t)by[k] >= P.amin);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t o = o4 + 64u * (uint32_t)k;
                if (o >= qlen) break;
                const uint64_t good = __ballot(ok[k]);
                const int nbits = (qlen - o) < 64u ? (int)(qlen - o) : 64;
                run_feed(st, good, nbits, o);
            }
        }
        const int rl = st.best;
        if (lane == 0) {
            if (count) {
                if (rl < KVQ_RL_BINS) atomicAdd(&hist[rl], 1u);          // add_rl, 394-402
                atomicMax(&longest, rl);
            }
            read_off[g] = sread + st.best_start;                          // 1070
            read_len[g] = rl >= P.minreadlength ? rl : -1;
            if (long_count && rl >= KVQ_LONG_READ && rl >= P.minreadlength) {
                const unsigned int i = atomicAdd(long_count, 1u);
                if (i < KVQ_LONG_CAP) { read_off[long_top - i] = sread + st.best_start; read_len[long_top - i] = rl; read_len[g] = -1; }
            }
        }
    }
    }
    __syncthreads();
    if (!count) return;
    for (int i = threadIdx.x; i < KVQ_RL_BINS; i += blockDim.x)
        if (hist[i]) atomicAdd(&P.ctr[KVQ_CTR_RL_ + i], (unsigned long long)hist[i]);
    if (threadIdx.x == 0) {
        if (longest >= 0) atomicMax(&P.ctr[KVQ_CTR_LONGEST_], (unsigned long long)(longest + 1));
        // records of this block (add_records_parsed, 1187)
        if (nproc) atomicAdd(&P.ctr[KVQ_CTR_RECORDS_], (unsigned long long)nproc);
    }
}

// ---------------------------------------------------------------------------
// exhaustive matcher
// ---------------------------------------------------------------------------

// mismatches of x[0..n) vs y[0..n) stay within the budget?
__device__ __forceinline__ bool within_budget(const uint8_t *x, const uint8_t *y, int n, int budget)
{
    int e = 0, j = 0;
    // four bytes at a time (any alignment), two such pairs of loads in flight: most alignments are over after the first
    for (; j + 8 <= n; j += 8) {
        uint32_t a0, a1, b0, b1;
        __builtin_memcpy(&a0, x + j, 4); __builtin_memcpy(&a1, x + j + 4, 4); __builtin_memcpy(&b0, y + j, 4); __builtin_memcpy(&b1, y + j + 4, 4);
        const uint32_t v0 = a0 ^ b0, v1 = a1 ^ b1;
        e += __popc((((v0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v0) & 0x80808080u) + __popc((((v1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v1) & 0x80808080u);
        if (e > budget) return false;
    }
    for (; j < n; j++) {
        e += (x[j] != y[j]);
        if (e > budget) return false;
    }
    return e <= budget;
}

// The alignments ("jobs") of one read against one sequence, in emission order: class A (tail of the read over the head of
// the sequence, 1116), class B (head of the read over the tail of the sequence, 1130), class C (one inside the other,
// 1147 / 1163).  Job j compares `len` bases from read offset ri and sequence offset si.
struct KvqJobs { int nA, nB, njobs, iA_hi, iB_hi; };
__device__ __forceinline__ KvqJobs kvq_jobs_of(int rl, int seql, int mo)
{
    KvqJobs J; J.nA = 0; J.nB = 0; J.iA_hi = 0; J.iB_hi = 0;
    if (rl > mo && seql > mo) {
        J.iA_hi = rl - mo;                                     // A: i = iA_hi .. iA_lo (1116)
        const int iA_lo = (rl - seql + 1) > 1 ? (rl - seql + 1) : 1;
        J.nA = J.iA_hi - iA_lo + 1; if (J.nA < 0) J.nA = 0;
        J.iB_hi = seql - mo;                                   // B: i = iB_hi .. iB_lo (1130)
        const int iB_lo = (seql - rl) > 1 ? (seql - rl) : 1;
        J.nB = J.iB_hi - iB_lo + 1; if (J.nB < 0) J.nB = 0;
    }
    J.njobs = J.nA + J.nB + (rl > seql ? rl - seql : seql - rl) + 1;   // 1147 / 1163
    return J;
}
__device__ __forceinline__ void kvq_job(const KvqJobs &J, int j, int rl, int seql, int &ri, int &si, int &len, int &spos, uint32_t &key)
{
    if (j < J.nA) { const int i = J.iA_hi - j; ri = i; si = 0; len = rl - i; spos = -i; key = (0u << 30) | (uint32_t)j; }
    else if (j < J.nA + J.nB) { const int i = J.iB_hi - (j - J.nA); ri = 0; si = i; len = seql - i; spos = i; key = (1u << 30) | (uint32_t)(j - J.nA); }
    else {
        const int i = j - J.nA - J.nB;
        key = (2u << 30) | (uint32_t)i;
        if (rl > seql) { ri = i; si = 0; len = seql; spos = -i; }
        else           { ri = 0; si = i; len = rl;   spos = i; }
    }
}

// one wave per read; lanes share out the jobs of the read against one sequence
extern "C" __global__ void __launch_bounds__(256)
kvq_match_all(KvqParams P, const uint8_t *__restrict__ data, int64_t fpos_base, uint32_t nrec_, KvqDevCount dc,
              const uint32_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
              const int32_t *__restrict__ seq_list, int32_t nlist)
{
    KVQ_BESIDE_SCAN();
    const int lane = kvq_lane();
    const uint32_t nrec = kvq_dev_count(nrec_, dc);
    for (uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6); g < nrec; g += gridDim.x * 4u) {
    const int rl = read_len[g];
    if (rl < 0) continue;
    const uint8_t *read = data + read_off[g];
    const int64_t fpos = fpos_base + read_off[g];
    const int mo = P.minoverlap, me = P.maxerrors;

    for (int q = (int)blockIdx.y; q < nlist; q += (int)gridDim.y) {      // (gridDim.y > 1: the sequences of a read shared out)
        const int s = seq_list[q];
        const uint8_t *seq = P.tab + P.tab_off[s];
        const int seql = P.tab_off[s + 1] - P.tab_off[s];
        const KvqJobs J = kvq_jobs_of(rl, seql, mo);
        for (int j0 = 64 * (int)blockIdx.z; j0 < J.njobs; j0 += 64 * (int)gridDim.z) {
            const int j = j0 + lane;
            bool hit = false; int ri = 0, si = 0, spos = 0, len = 0; uint32_t key = 0;
            if (j < J.njobs) {
                kvq_job(J, j, rl, seql, ri, si, len, spos, key);
                hit = within_budget(read + ri, seq + si, len, me);
            }
            kvq_emit(P, hit, fpos, s, spos, len, rl, key);
        }
    }
    }
}

// The same for the few LONG reads that the redo of skipped tiles meets (kvq_trim_records puts them on a list of their own,
// read_off / read_len [long_top], [long_top - 1], ...): a workgroup per read, sequence group (blockIdx.y) and share of the
// alignments (blockIdx.z), the read's bases in LDS -- the thousands of alignments of such a read against every sequence
// are a chain of memory round trips otherwise.  A read that does not fit the LDS buffer is compared out of global memory.
#define KVQ_LONG_LDS 16384u
__device__ __forceinline__ bool within_budget_lds(const uint32_t *R, uint32_t x, const uint8_t *y, int n, int budget)
{
    // x: byte offset of the read's bases in R (any alignment: two words and a funnel shift)
    auto word = [&](uint32_t o) { const uint32_t w = o >> 2; return __builtin_amdgcn_alignbit(R[w + 1], R[w], (o & 3u) * 8u); };
    int e = 0, j = 0;
    for (; j + 8 <= n; j += 8) {
        uint32_t b0, b1;
        __builtin_memcpy(&b0, y + j, 4); __builtin_memcpy(&b1, y + j + 4, 4);
        const uint32_t v0 = word(x + (uint32_t)j) ^ b0, v1 = word(x + (uint32_t)j + 4u) ^ b1;
        e += __popc((((v0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v0) & 0x80808080u) + __popc((((v1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v1) & 0x80808080u);
        if (e > budget) return false;
    }
    for (; j < n; j++) {
        e += ((word(x + (uint32_t)j) & 0xFFu) != y[j]);
        if (e > budget) return false;
    }
    return e <= budget;
}

#define KVQ_LONG_NZ 2u              // a read's alignments against one sequence are shared out over so many waves
extern "C" __global__ void __launch_bounds__(256)
kvq_match_long(KvqParams P, const uint8_t *__restrict__ data, int64_t fpos_base, KvqDevCount dc,
               const uint32_t *__restrict__ read_off, const int32_t *__restrict__ read_len,
               const int32_t *__restrict__ seq_list, int32_t nlist, uint32_t long_top)
{
    KVQ_BESIDE_SCAN();
    // work units (read, group of four sequences -- a wave each --, share of the alignments), dealt out over the grid in
    // turn: a launch of fixed size serves one long read as evenly as a thousand
    __shared__ uint32_t R[KVQ_LONG_LDS / 4u + 2u];
    const int lane = kvq_lane(), wave = (int)(threadIdx.x >> 6);
    uint32_t nrec = kvq_dev_count(0u, dc);
    if (nrec > KVQ_LONG_CAP) nrec = KVQ_LONG_CAP;
    const uint32_t ngroups = ((uint32_t)nlist + 3u) / 4u, per_read = ngroups * KVQ_LONG_NZ;
    const uint32_t units = nrec * per_read;                           // (<= 2048 x 16384 x 2)
    const int mo = P.minoverlap, me = P.maxerrors;
    for (uint32_t u = blockIdx.x; u < units; u += gridDim.x) {
        const uint32_t g = long_top - u / per_read, rem = u % per_read;
        const int q = (int)(rem / KVQ_LONG_NZ) * 4 + wave, z = (int)(rem % KVQ_LONG_NZ);
        const int rl = read_len[g];
        const uint32_t ro = read_off[g];
        // the wave's sequence: its number, its place in the table, its first eight bases (on their way while the read
        // is being staged) -- nearly all alignments of a long read start at the head of the sequence, and one that has too
        // many mismatches in those eight bases, nearly every one, never touches global memory
        int s = 0, seql = 0; const uint8_t *seq = P.tab; uint32_t s0 = 0, s1 = 0;
        if (q < nlist) {
            s = seq_list[q];
            const int off = P.tab_off[s]; seql = P.tab_off[s + 1] - off; seq = P.tab + off;
            if (seql >= 8) { __builtin_memcpy(&s0, seq, 4); __builtin_memcpy(&s1, seq + 4, 4); }
        }
        const uint8_t *read = data + ro;
        const int64_t fpos = fpos_base + ro;
        // the read's bases, from the aligned word in front of them on (the text is 16-byte aligned, check_batch)
        const uint32_t sh = ro & 3u;
        const bool in_lds = rl >= 0 && (uint32_t)rl + sh <= KVQ_LONG_LDS;
        __syncthreads();                                                // (the last unit's comparisons are over)
        if (in_lds) {
            const uint32_t words = ((uint32_t)rl + sh + 3u) / 4u;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(data + (ro - sh));
            for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) R[w] = src[w];
            if (threadIdx.x < 2u) R[words + threadIdx.x] = 0u;
        }
        __syncthreads();
        if (rl < 0 || q >= nlist) continue;
        const KvqJobs J = kvq_jobs_of(rl, seql, mo);
        for (int j0 = 64 * z; j0 < J.njobs; j0 += 64 * (int)KVQ_LONG_NZ) {
            const int j = j0 + lane;
            bool hit = false; int ri = 0, si = 0, spos = 0, len = 0; uint32_t key = 0;
            if (j < J.njobs) {
                kvq_job(J, j, rl, seql, ri, si, len, spos, key);
                if (!in_lds) hit = within_budget(read + ri, seq + si, len, me);
                else {
                    bool maybe = true;
                    if (si == 0 && len >= 8) {
                        const uint32_t x = sh + (uint32_t)ri, w = x >> 2, b = (x & 3u) * 8u;
                        const uint32_t v0 = __builtin_amdgcn_alignbit(R[w + 1], R[w], b) ^ s0, v1 = __builtin_amdgcn_alignbit(R[w + 2], R[w + 1], b) ^ s1;
                        maybe = __popc((((v0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v0) & 0x80808080u) + __popc((((v1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v1) & 0x80808080u) <= me;
                    }
                    hit = maybe && within_budget_lds(R, sh + (uint32_t)ri, seq + si, len, me);
                }
            }
            kvq_emit(P, hit, fpos, s, spos, len, rl, key);
        }
    }
}

// ---------------------------------------------------------------------------
// hit fold: counters, coverage/mutations, hit bytes
// ---------------------------------------------------------------------------

__device__ __forceinline__ int base_class(uint8_t c)
{
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : c == 'N' ? 4 : 5;
}

// The fold of one batch's hits [*range_begin, min(*range_end, cap)) of the arena, a LANE per hit: per-sequence
// counters, coverage marks, the place of the hit's bytes in the blob (prefix sum over the wave's 64 hits, one
// reservation per wave -- a reservation per hit would queue up on blob_n), then the hit's bases one by one:
// hit bytes into the blob, mutation counters (analyse.py:57-78).  (Hits are a few dozen bytes: 64 of them side
// by side keep 64 independent byte streams in flight, where a wave per hit waits out the way to memory and
// back once per hit.)
extern "C" __global__ void __launch_bounds__(256)
kvq_fold_batch(KvqParams P, const uint8_t *__restrict__ data, int64_t fpos_base,
               const unsigned int *__restrict__ range_begin, const unsigned int *__restrict__ range_end)
{
    KVQ_BESIDE_SCAN();
    const uint32_t h0 = *range_begin;
    uint32_t h1 = *range_end; if (h1 > P.arena_cap) h1 = P.arena_cap;
    const int lane = kvq_lane();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    if (wave == 0 && lane == 0 && h1 > h0) atomicAdd(&P.ctr[KVQ_CTR_HITS_], (unsigned long long)(h1 - h0));
    for (uint32_t hb = h0 + wave * 64u; hb < h1; hb += nwaves * 64u) {
        const uint32_t h = hb + (uint32_t)lane;
        const bool valid = h < h1;
        int s = 0, len = 0, seq_pos = 0; int64_t fpos = 0;
        if (valid) { const KvqHit hit = P.arena[h]; s = hit.seq_nr; len = hit.length; seq_pos = hit.seq_pos; fpos = hit.fpos; }
        const uint32_t ulen = len > 0 ? (uint32_t)len : 0u;
        const int start = seq_pos > 0 ? seq_pos : 0;                                  // analyse.py:70
        const int64_t at = (int64_t)(valid ? P.tab_off[s] : 0) + start;
        if (valid) {
            atomicAdd(&P.ctr[P.off_nseqhits + s], 1ull);                              // 435
            atomicAdd(&P.ctr[P.off_nseqbasehits + s], (unsigned long long)len);       // 434
            if (len > 0) {
                // coverage[start .. start + len) += 1 (analyse.py:76) as two marks; kvq_cov_apply sums them up
                atomicAdd(&P.covdiff[at + s], 1ull);
                atomicAdd(&P.covdiff[at + s + len], ~0ull);
            }
        }
        const uint32_t incl = kvq_wave_incl_scan(ulen);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        unsigned long long base = 0;
        if (lane == 0 && tot) base = atomicAdd(P.blob_n, (unsigned long long)tot);
        base = ((unsigned long long)__shfl((unsigned int)(base >> 32), 0, 64) << 32) | __shfl((unsigned int)base, 0, 64);
        const unsigned long long boff = base + incl - ulen;
        const bool fits = boff + ulen <= P.blob_cap;
        if (valid) P.arena[h].blob_off = fits ? (uint32_t)boff : 0xFFFFFFFFu;
        const uint8_t *src = data + (fpos - fpos_base) + (seq_pos < 0 ? -seq_pos : 0);
        const uint8_t *seq = P.tab + at;
        uint8_t *dst = P.blob + boff;
        // sixteen bases a step (unaligned vectors; the buffers have slack behind their ends), the loads of a step in flight together:
        // a 150-base hit is ten trips to memory one behind the other (four bases a step were 38 of them, and the kernel's whole time)
        typedef uint32_t u32x4_any __attribute__((ext_vector_type(4), aligned(1)));
        for (uint32_t q0 = 0; __any(q0 < ulen); q0 += 16u) {
            if (q0 < ulen) {
                const u32x4_any xv = *reinterpret_cast<const u32x4_any *>(src + q0), yv = *reinterpret_cast<const u32x4_any *>(seq + q0);
                const uint32_t xs[4] = { xv.x, xv.y, xv.z, xv.w }, ys[4] = { yv.x, yv.y, yv.z, yv.w };
                const uint32_t left = ulen - q0;                                       // bases of the hit from q0 on
#pragma unroll
                for (uint32_t t = 0; t < 4u; t++) {
                    const uint32_t q = q0 + 4u * t;
                    if (4u * t < left) {
                        const uint32_t x = xs[t]; uint32_t y = ys[t];
                        const uint32_t nq = left - 4u * t < 4u ? left - 4u * t : 4u;
                        if (nq < 4u) { const uint32_t keep = (1u << (8u * nq)) - 1u; y = (y & keep) | (x & ~keep); }      // (bytes behind the hit: "equal")
                        if (x != y)
                            for (uint32_t j = 0; j < nq; j++) {
                                const uint8_t c = (uint8_t)(x >> (8u * j));
                                if (c != (uint8_t)(y >> (8u * j))) atomicAdd(&P.ctr[P.off_mut + (at + q + j) * 6 + base_class(c)], 1ull);   // analyse.py:77-78
                            }
                    }
                }
                if (fits) {                                                           // 437
                    if (left >= 16u) *reinterpret_cast<u32x4_any *>(dst + q0) = xv;
                    else for (uint32_t j = 0; j < left; j++) dst[q0 + j] = (uint8_t)(xs[j >> 2] >> (8u * (j & 3u)));
                }
            }
        }
    }
}

// one wave per sequence: running sum of the coverage marks kvq_fold_batch left, added to the
// coverage counters; the marks are cleared on the way
extern "C" __global__ void __launch_bounds__(256)
kvq_cov_apply(KvqParams P)
{
    KVQ_BESIDE_SCAN();
    const int s = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (s >= P.nseq) return;
    const int lane = kvq_lane();
    const int64_t t0 = P.tab_off[s], len = (int64_t)P.tab_off[s + 1] - t0;
    unsigned long long *d = P.covdiff + t0 + s;
    unsigned long long carry = 0;
    for (int64_t i0 = 0; i0 <= len; i0 += 64) {
        const int64_t i = i0 + lane;
        unsigned long long v = 0;
        if (i <= len) { v = d[i]; if (v) d[i] = 0; }
        // inclusive scan of the marks over the wave (64-bit, modulo 2^64: -1 marks are ~0)
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long u = ((unsigned long long)__shfl_up((unsigned int)(v >> 32), o, 64) << 32) | __shfl_up((unsigned int)v, o, 64);
            if (lane >= o) v += u;
        }
        v += carry;
        if (i < len && v) atomicAdd(&P.ctr[P.off_cov + t0 + i], v);
        carry = ((unsigned long long)__shfl((unsigned int)(v >> 32), 63, 64) << 32) | __shfl((unsigned int)v, 63, 64);
    }
}
