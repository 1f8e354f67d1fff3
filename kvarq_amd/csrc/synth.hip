// kvarq_amd/csrc/synth.hip -- synthetic FastQ workload of SURVEY.md 8(d), the
// byte-identical twin of kvarq_amd/synth.py (counter-based generator: one
// splitmix64 finalisation per draw, no shared state, so any record range can be
// produced independently on any device).
#include "kvq_host.h"

#define SYN_GOLD  0x9E3779B97F4A7C15ull
#define SYN_C_REC 0xD1B54A32D192ED03ull
#define SYN_C_GEN 0xA0761D6478BD642Full
#define SYN_THR_ERR  21474836u      /* int(0.005 * 2^32) */
#define SYN_THR_BADQ 167772u        /* int(0.01 * 2^24)  */

__host__ __device__ static inline uint64_t syn_mix64(uint64_t x)
{
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ static inline int syn_code(uint8_t c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }

// byte `col` of record `rec` (record layout: "@SYN.%09d 1:N:0\n" bases "\n+\n" quals "\n")
__host__ __device__ static inline uint8_t syn_byte(int64_t rec, int col, int L, uint64_t seed,
                                                   const uint8_t *genome, int64_t G)
{
    const char letters[4] = { 'A', 'C', 'G', 'T' };
    if (col < 21) {
        if (col < 5) { const char h[5] = { '@', 'S', 'Y', 'N', '.' }; return (uint8_t)h[col]; }
        if (col < 14) {                      // 9 decimal digits, most significant first
            int64_t v = rec; const int digit = 13 - col;
            for (int i = 0; i < digit; i++) v /= 10;
            return (uint8_t)('0' + (v % 10));
        }
        const char t[7] = { ' ', '1', ':', 'N', ':', '0', '\n' };
        return (uint8_t)t[col - 14];
    }
    const uint64_t state = seed + (uint64_t)rec * SYN_C_REC;
    if (col < 21 + L) {
        const int j = col - 21;
        const uint64_t d0 = syn_mix64(state);
        const int64_t start = (int64_t)((d0 & 0xFFFFFFFFull) % (uint64_t)(G - L + 1));
        const bool minus = (d0 >> 63) != 0;
        const uint64_t dj = syn_mix64(state + (uint64_t)(j + 1) * SYN_GOLD);
        int code = syn_code(genome[minus ? start + (L - 1 - j) : start + j]);
        if (minus) code = 3 - code;
        if ((uint32_t)(dj & 0xFFFFFFFFull) < SYN_THR_ERR) code = (code + (int)(((dj >> 32) & 3ull) % 3ull) + 1) & 3;
        return (uint8_t)letters[code];
    }
    if (col == 21 + L) return '\n';
    if (col == 22 + L) return '+';
    if (col == 23 + L) return '\n';
    if (col < 24 + 2 * L) {
        const int j = col - (24 + L);
        const uint64_t dj = syn_mix64(state + (uint64_t)(j + 1) * SYN_GOLD);
        return ((dj >> 40) < SYN_THR_BADQ) ? '#' : 'I';
    }
    return '\n';
}

// one thread per output dword; the stream is written with coalesced 4-byte stores
extern "C" __global__ void __launch_bounds__(256)
kvq_synth_kernel(uint8_t *out, int64_t first, int64_t n, int L, uint64_t seed, const uint8_t *genome, int64_t G)
{
    const int rb = 2 * L + 25;
    const int64_t total = n * rb;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w * 4 < total; w += (int64_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        for (int k = 0; k < 4; k++) {
            const int64_t at = w * 4 + k;
            if (at < total) v |= (uint32_t)syn_byte(first + at / rb, (int)(at % rb), L, seed, genome, G) << (8 * k);
        }
        if (w * 4 + 4 <= total) *reinterpret_cast<uint32_t *>(out + w * 4) = v;
        else for (int k = 0; w * 4 + k < total; k++) out[w * 4 + k] = (uint8_t)(v >> (8 * k));
    }
}

extern "C" int32_t kvq_synth_reads_device(void *d_out, int64_t first, int64_t n, int32_t L, uint64_t seed,
                                          const void *d_genome, int64_t genome_size)
{
    kvq_clear_error();
    if (n <= 0) return KVQ_OK;
    hipLaunchKernelGGL(kvq_synth_kernel, dim3(256 * 16), dim3(256), 0, 0, (uint8_t *)d_out, first, n, (int)L, seed,
                       (const uint8_t *)d_genome, genome_size);
    KVQ_HIP(hipGetLastError());
    KVQ_HIP(hipDeviceSynchronize());
    return KVQ_OK;
}

extern "C" void kvq_synth_reads_host(uint8_t *out, int64_t first, int64_t n, int32_t L, uint64_t seed,
                                     const uint8_t *genome, int64_t genome_size)
{
    const int rb = 2 * L + 25;
    for (int64_t r = 0; r < n; r++)
        for (int c = 0; c < rb; c++) out[r * rb + c] = syn_byte(first + r, c, L, seed, genome, genome_size);
}

extern "C" void kvq_synth_genome_host(uint8_t *out, int64_t size, uint64_t seed)
{
    const char letters[4] = { 'A', 'C', 'G', 'T' };
    for (int64_t i = 0; i < size; i++) out[i] = (uint8_t)letters[syn_mix64(seed + SYN_C_GEN + (uint64_t)i * SYN_GOLD) >> 62];
}
