// Probe: issue cost of v_cndmask_b32 / v_cmp + v_cndmask pairs on gfx950, instructions in ONE asm block
// (no compiler-inserted s_nop between them), 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 32
template <int OP> __global__ void k(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    float d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5\n v_mov_b32 %0, %5\n v_mov_b32 %1, %4\n v_mov_b32 %2, %5\n v_mov_b32 %3, %4"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));
            if (OP == 1) asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %4, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n"
                                      "v_cndmask_b32 %0, %5, %4, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %5, %4, vcc\n v_cndmask_b32 %3, %4, %5, vcc"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c) : "vcc");
            if (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %5, %4, s[20:21]\n v_cndmask_b32_e64 %2, %4, %5, s[20:21]\n v_cndmask_b32_e64 %3, %5, %4, s[20:21]\n"
                                      "v_cndmask_b32_e64 %0, %5, %4, s[20:21]\n v_cndmask_b32_e64 %1, %4, %5, s[20:21]\n v_cndmask_b32_e64 %2, %5, %4, s[20:21]\n v_cndmask_b32_e64 %3, %4, %5, s[20:21]"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c) : "s20", "s21");
            if (OP == 3) asm volatile("v_cmp_le_f32 vcc, %6, %4\n v_cndmask_b32 %0, %4, %5, vcc\n v_cmp_le_f32 vcc, %6, %5\n v_cndmask_b32 %1, %5, %4, vcc\n"
                                      "v_cmp_le_f32 vcc, %6, %4\n v_cndmask_b32 %2, %4, %5, vcc\n v_cmp_le_f32 vcc, %6, %5\n v_cndmask_b32 %3, %5, %4, vcc"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c), "v"(a) : "vcc");
            if (OP == 4) asm volatile("v_cmp_le_f32 s[20:21], %6, %4\n v_cmp_le_f32 s[22:23], %6, %5\n v_cmp_le_f32 s[24:25], %6, %4\n v_cmp_le_f32 s[26:27], %6, %5\n"
                                      "v_cndmask_b32_e64 %0, %4, %5, s[20:21]\n v_cndmask_b32_e64 %1, %5, %4, s[22:23]\n v_cndmask_b32_e64 %2, %4, %5, s[24:25]\n v_cndmask_b32_e64 %3, %5, %4, s[26:27]"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c), "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            if (OP == 5) asm volatile("v_max_f32 %0, %4, %5\n v_max_f32 %1, %5, %4\n v_max_f32 %2, %4, %5\n v_max_f32 %3, %5, %4\n v_max_f32 %0, %5, %4\n v_max_f32 %1, %4, %5\n v_max_f32 %2, %5, %4\n v_max_f32 %3, %4, %5"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));
            if (OP == 6) asm volatile("v_cmp_le_f32 vcc, %6, %4\n v_cmp_le_f32 vcc, %6, %5\n v_cmp_le_f32 vcc, %6, %4\n v_cmp_le_f32 vcc, %6, %5\n v_cmp_le_f32 vcc, %6, %4\n v_cmp_le_f32 vcc, %6, %5\n v_cmp_le_f32 vcc, %6, %4\n v_cmp_le_f32 vcc, %6, %5"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c), "v"(a) : "vcc");
            if (OP == 7) asm volatile("v_add_f32 %0, %4, %5\n v_add_f32 %1, %5, %4\n v_add_f32 %2, %4, %5\n v_add_f32 %3, %5, %4\n v_add_f32 %0, %5, %4\n v_add_f32 %1, %4, %5\n v_add_f32 %2, %5, %4\n v_add_f32 %3, %4, %5"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));
            if (OP == 8) asm volatile("v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %5\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %5"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));
            if (OP == 9) asm volatile("ds_read_u8 %0, %6\n ds_read_u8 %1, %6 offset:1\n ds_read_u8 %2, %6 offset:64\n ds_read_u8 %3, %6 offset:65\n s_waitcnt lgkmcnt(0)"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c), "v"(threadIdx.x * 2));
            if (OP == 10) asm volatile("ds_read_u16 %0, %6\n ds_read_u16 %2, %6 offset:64\n s_waitcnt lgkmcnt(0)"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c), "v"(threadIdx.x * 2));
            if (OP == 11) asm volatile("ds_read_b32 %0, %6\n ds_read_b32 %1, %6 offset:4\n ds_read_b32 %2, %6 offset:1024\n ds_read_b32 %3, %6 offset:1028\n s_waitcnt lgkmcnt(0)"
                                      : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c), "v"(threadIdx.x * 4));
            if (OP == 12) asm volatile("ds_read2_b32 %0, %2 offset1:1\n ds_read2_b32 %1, %2 offset0:64 offset1:65\n s_waitcnt lgkmcnt(0)"
                                      : "+v"(*(double*)&d0), "+v"(*(double*)&d2) : "v"(threadIdx.x * 4));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d0 + d1 + d2 + d3;
}
template <int OP> void run(const char *name, float *out, int per_block) {
    for (int wps : {1, 2, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 8192, 0, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 8192, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = (double)iters * REP * per_block * wps;
        printf("%-34s %d waves/SIMD: %.2f ns = %.2f cycles @2.4GHz per instruction per SIMD\n", name, wps, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    run<0>("v_mov_b32 x8", out, 8); run<7>("v_add_f32 x8 independent", out, 8); run<8>("v_add_f32 x8 dependent chain", out, 8);
    run<5>("v_max_f32 x8", out, 8); run<6>("v_cmp_le_f32 vcc x8", out, 8);
    run<1>("v_cndmask_b32 vcc x8", out, 8); run<2>("v_cndmask_b32_e64 sgpr x8", out, 8);
    run<3>("(v_cmp vcc; v_cndmask vcc) x4", out, 8); run<4>("v_cmp sgpr x4; v_cndmask sgpr x4", out, 8);
    run<9>("ds_read_u8 x4 + wait", out, 4); run<10>("ds_read_u16 x2 + wait", out, 2); run<11>("ds_read_b32 x4 + wait", out, 4);
    return 0;
}
