
// Issue rate of scalar and mixed instruction streams on gfx950 (companion of valu_rate.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i * seed;
    uint32_t b = seed | 1u;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    asm volatile("" : "+v"(b), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i += 4) {
                if (OP == 0) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc"); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) :: "scc"); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s2) :: "scc"); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s3) :: "scc"); }
                if (OP == 1) { asm volatile("s_and_b32 %0, %0, 0xff" : "+s"(s0) :: "scc"); asm volatile("s_lshl_b32 %0, %0, 1" : "+s"(s1) :: "scc"); asm volatile("s_or_b32 %0, %0, 3" : "+s"(s2) :: "scc"); asm volatile("s_mov_b32 %0, 5" : "=s"(s3)); }
                // one VALU + one SALU alternating
                if (OP == 2) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) :: "scc"); }
                // two VALU per SALU
                if (OP == 3) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(b)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) :: "scc"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 2]) : "v"(b)); }
                if (OP == 4) { asm volatile("s_nop 0"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("s_nop 0"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(b)); }
                if (OP == 5) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 2]) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 3]) : "v"(b)); }
                // dependent VALU chain (each instruction needs the previous result)
                if (OP == 6) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(b)); }
                if (OP == 7) { asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a[0]) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a[0]) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a[0]) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(a[0]) : "v"(b)); }
                // v_cmp -> s_and_saveexec-like scalar consumer
                if (OP == 8) { asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, exec" :: "v"(a[i]), "v"(b) : "vcc", "scc"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(b)); asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, exec" :: "v"(a[i + 2]), "v"(b) : "vcc", "scc"); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i + 3]) : "v"(b)); }
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    if (s == 0x12345678u || s0 + s1 + s2 + s3 == 77) out[0] = s;
}
typedef void (*K)(uint32_t *, int, uint32_t);
int main()
{
    uint32_t *d; (void)hipMalloc(&d, 64);
    const char *names[] = { "s_add_u32 x4", "s_and/lshl/or/mov", "valu,salu alternating", "3 valu : 1 salu", "s_nop,valu alternating", "valu x4 indep", "v_add dependent chain", "v_alignbit dep chain", "v_cmp+s_and,valu" };
    const int slots[] = { 4, 4, 4, 4, 4, 4, 4, 4, 6 };
    K ks[] = { k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8> };
    const int iters = 2000;
    printf("%-26s", "waves/SIMD:");
    for (int w = 1; w <= 8; w *= 2) printf("  %8d", w);
    printf("   (wall time x 2.4 GHz per instruction of one SIMD's combined stream)\n");
    for (int op = 0; op < 9; op++) {
        printf("%-26s", names[op]);
        for (int w = 1; w <= 8; w *= 2) {
            hipLaunchKernelGGL(ks[op], dim3(256 * w), dim3(256), 0, 0, d, 10, 12345u);
            (void)hipDeviceSynchronize();
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[op], dim3(256 * w), dim3(256), 0, 0, d, iters, 12345u);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            const double ninstr = (double)iters * (REP / 4) * slots[op] * w;
            printf("  %8.2f", (double)ms * 1e-3 * 2.4e9 / ninstr);
        }
        printf("\n"); fflush(stdout);
    }
    return 0;
}
