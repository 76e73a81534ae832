// VALU issue-rate microbenchmark for gfx950 (review item: "a wave64 VALU instruction occupies its
// SIMD for four cycles whatever its type" -- DESIGN.md 5.1 -- had no measurement behind it).
//
// One block of 64 * W threads per CU-slot, W waves per SIMD chosen by the grid; every wave runs a
// dependent-free stream of N instructions of one kind (8 independent accumulators, so neither the
// dependent-issue latency nor register ports limit it) and reports s_memtime ticks.  Printed:
// cycles per wave-instruction per SIMD = ticks * waves_per_simd_resident / N ... measured two ways:
//   (a) one wave alone on its SIMD  -> issue interval seen by ONE wave (latency-limited)
//   (b) 2 and 4 waves per SIMD      -> the pipe's throughput (cycles per instruction per SIMD)
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/microbench/valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(1024) void k_issue(unsigned long long *ticks, double *sink, int iters) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    unsigned i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const unsigned k = 3;
    double m2 = 1.0000001; unsigned k2 = 3;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {   // v_fma_f64
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                              "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (KIND == 1) {   // v_add_f64
            REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                              "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (KIND == 2) {   // v_mul_f64
            REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                              "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == 3) {   // v_add_u32
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k));)
        } else if (KIND == 4) {   // v_mul_lo_u32
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                              "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k));)
        } else if (KIND == 5) {   // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(1.0000001f), "v"(1e-9f));)
        } else if (KIND == 6) {   // v_mov_b32 (a 64-bit register move is two of these, or one v_mov_b64)
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));)
        } else if (KIND == 7) {   // v_mov_b64
            REP8(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n"
                              "v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 8) {   // v_cndmask_b32 (vcc)
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k));)
        } else if (KIND == 13) {  // v_cndmask_b32_e64 with an SGPR-pair mask
            unsigned long long msk = 0x5555555555555555ull;
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                              "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k), "s"(msk));)
        } else if (KIND == 14) {  // v_cmp_lt_f64 -> SGPR pair, then v_cndmask on it (compare + select pairs)
            REP8(asm volatile("v_cmp_lt_f64_e64 s[40:41], %0, %8\n v_cndmask_b32_e64 %4, %4, %9, s[40:41]\n v_cmp_lt_f64_e64 s[42:43], %1, %8\n v_cndmask_b32_e64 %5, %5, %9, s[42:43]\n"
                              "v_cmp_lt_f64_e64 s[44:45], %2, %8\n v_cndmask_b32_e64 %6, %6, %9, s[44:45]\n v_cmp_lt_f64_e64 s[46:47], %3, %8\n v_cndmask_b32_e64 %7, %7, %9, s[46:47]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m), "v"(k)
                              : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");)
        } else if (KIND == 15) {  // what hipcc emits for a select: v_cmp -> vcc, v_cndmask_b32_e32 reading vcc
            REP8(asm volatile("v_cmp_lt_f64_e32 vcc, %0, %8\n v_cndmask_b32_e32 %4, %4, %9, vcc\n v_cmp_lt_f64_e32 vcc, %1, %8\n v_cndmask_b32_e32 %5, %5, %9, vcc\n"
                              "v_cmp_lt_f64_e32 vcc, %2, %8\n v_cndmask_b32_e32 %6, %6, %9, vcc\n v_cmp_lt_f64_e32 vcc, %3, %8\n v_cndmask_b32_e32 %7, %7, %9, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m), "v"(k) : "vcc");)
        } else if (KIND == 16) {  // one compare, then eight selects on the same vcc
            asm volatile("v_cmp_lt_f64_e32 vcc, %0, %1" :: "v"(a0), "v"(m) : "vcc");
            REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(k));)
        } else if (KIND == 17) {  // a select of a DOUBLE as hipcc emits it: one v_cmp -> vcc, two v_cndmask_b32_e32 on that vcc
            REP8(asm volatile("v_cmp_lt_f64_e32 vcc, %0, %4\n v_cndmask_b32_e32 %1, %1, %5, vcc\n v_cndmask_b32_e32 %2, %2, %5, vcc\n"
                              "v_cmp_lt_f64_e32 vcc, %3, %4\n v_cndmask_b32_e32 %6, %6, %5, vcc\n v_cndmask_b32_e32 %7, %7, %5, vcc\n"
                              "v_cmp_lt_f64_e32 vcc, %0, %4\n v_cndmask_b32_e32 %8, %8, %5, vcc\n"
                              : "+v"(a0), "+v"(i0), "+v"(i1), "+v"(a1), "+v"(m2), "+v"(k2), "+v"(i2), "+v"(i3), "+v"(i4) :: "vcc");)
        } else if (KIND == 18) {  // the same select through an SGPR pair: v_cmp_e64 -> s[..], two v_cndmask_b32_e64
            REP8(asm volatile("v_cmp_lt_f64_e64 s[40:41], %0, %4\n v_cndmask_b32_e64 %1, %1, %5, s[40:41]\n v_cndmask_b32_e64 %2, %2, %5, s[40:41]\n"
                              "v_cmp_lt_f64_e64 s[42:43], %3, %4\n v_cndmask_b32_e64 %6, %6, %5, s[42:43]\n v_cndmask_b32_e64 %7, %7, %5, s[42:43]\n"
                              "v_cmp_lt_f64_e64 s[44:45], %0, %4\n v_cndmask_b32_e64 %8, %8, %5, s[44:45]\n"
                              : "+v"(a0), "+v"(i0), "+v"(i1), "+v"(a1), "+v"(m2), "+v"(k2), "+v"(i2), "+v"(i3), "+v"(i4)
                              :: "s40", "s41", "s42", "s43", "s44", "s45");)
        } else if (KIND == 9) {   // v_lshl_add_u64 (64-bit address arithmetic)
            unsigned long long *p0 = (unsigned long long *)&a0, *p1 = (unsigned long long *)&a1;
            (void)p0; (void)p1;
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                              "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == 10) {  // v_pk_fma_f32 (two f32 lanes per register pair)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == 11) {  // v_rcp_f64 (transcendental pipe)
            REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n"
                              "v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 12) {  // 1 : 1 mix of v_fma_f64 and v_add_u32 (does integer work hide under FP64?)
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_add_u32 %4, %4, %10\n v_fma_f64 %1, %1, %8, %9\n v_add_u32 %5, %5, %10\n"
                              "v_fma_f64 %2, %2, %8, %9\n v_add_u32 %6, %6, %10\n v_fma_f64 %3, %3, %8, %9\n v_add_u32 %7, %7, %10\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m), "v"(c), "v"(k));)
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) +
                                                  (double)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) + m2 + (double)k2;
}

template <int KIND>
static void run(const char *name, int per_iter) {
    const int iters = 1000, rounds = 6;
    unsigned long long *d_t; double *d_s;
    (void)hipMalloc(&d_t, sizeof(unsigned long long) * 256 * 4 * rounds * 4);
    (void)hipMalloc(&d_s, sizeof(double) * 256 * 4 * rounds * 256);
    (void)hipFuncSetAttribute((const void *)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%-16s", name);
    // w waves per SIMD = w resident 256-thread blocks per CU, enforced by the LDS each block asks for;
    // rounds * 256 * w blocks, so that every CU works through the same number of blocks and the wall
    // clock divided by the wave-instructions per SIMD is the pipe's rate
    for (int w : {1, 2, 4}) {
        const int blocks = 256 * w * rounds;
        const size_t lds = (160 * 1024) / w - 1024;
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), lds, 0, d_t, d_s, 10);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_issue<KIND>, dim3(blocks), dim3(256), lds, 0, d_t, d_s, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t(blocks * 4);
        (void)hipMemcpy(t.data(), d_t, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : t) mean += (double)v; mean /= t.size();
        const double n = (double)iters * per_iter;
        const double insts_per_simd = n * (double)t.size() / 1024.0;
        const double ns = ms * 1e6 / insts_per_simd;
        // a wave's ticks per instruction / w = ticks per instruction per SIMD; ticks / wall = the clock
        const double ticks_per_simd_inst = mean / n / w;
        printf("  w=%d: %5.2f ns %5.2f ticks (%4.2f GHz)", w, ns, ticks_per_simd_inst, ticks_per_simd_inst / ns);
    }
    printf("\n");
    (void)hipFree(d_t); (void)hipFree(d_s);
}

int main() {
    printf("per wave64 instruction per SIMD: wall-clock ns, s_memtime ticks of one wave / waves per SIMD, and their ratio (the clock the ticks run at)\n");
    run<0>("v_fma_f64", 64); run<1>("v_add_f64", 64); run<2>("v_mul_f64", 64); run<3>("v_add_u32", 64);
    run<4>("v_mul_lo_u32", 64); run<5>("v_fma_f32", 64); run<6>("v_mov_b32", 64); run<7>("v_mov_b64", 64);
    run<8>("v_cndmask_b32", 64); run<9>("v_lshl_add_u64", 64); run<10>("v_pk_fma_f32", 64); run<11>("v_rcp_f64", 64);
    run<12>("fma64+add32 1:1", 64); run<13>("v_cndmask_e64 sgpr", 64); run<14>("cmp_f64+cndmask", 64); run<15>("cmp->vcc+cndmask", 64); run<16>("cndmask_e32 vcc", 64); run<17>("cmp+2cndmask vcc", 64); run<18>("cmp+2cndmask sgpr", 64);
    return 0;
}
