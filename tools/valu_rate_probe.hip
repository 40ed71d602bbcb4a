// Probe: per-SIMD issue cost of a few VALU instructions on gfx950 (wave64), 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 256
template <int OP> __global__ void k(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1.f;
    double x = a, y = 1.0000001;
    unsigned u = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
            if (OP == 3) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a) : "v"(u));
            if (OP == 4) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c));
            if (OP == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(b));
            if (OP == 6) asm volatile("v_mul_lo_u32 %0, %1, %1" : "=v"(u) : "v"(u));
            if (OP == 7) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
            if (OP == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
            if (OP == 9) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
            if (OP == 10) asm volatile("v_add_u32 %0, %1, %1" : "=v"(u) : "v"(u));
            if (OP == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
            if (OP == 12) asm volatile("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(u) : "v"(u));
            if (OP == 13) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a) : "v"(u));
            if (OP == 14) asm volatile("v_floor_f32 %0, %1" : "=v"(a) : "v"(b));
            if (OP == 15) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
            if (OP == 16) asm volatile("v_min3_u32 %0, %1, %1, %1" : "=v"(u) : "v"(u));
            if (OP == 17) asm volatile("v_fract_f32 %0, %1" : "=v"(a) : "v"(b));
            if (OP == 18) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
            if (OP == 19) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
            if (OP == 20) asm volatile("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(u) : "v"(u));
            if (OP == 21) asm volatile("v_and_b32 %0, %1, %1" : "=v"(u) : "v"(u));
            if (OP == 22) asm volatile("v_mul_u32_u24 %0, %1, %1" : "=v"(u) : "v"(u));
            if (OP == 23) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(u) : "s20");
            if (OP == 24) asm volatile("v_mul_hi_u32 %0, %1, %1" : "=v"(u) : "v"(u));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + (float)x + (float)u;
}
template <int OP> void run(const char *name, float *out) {
    for (int wps : {1, 4}) {           // waves per SIMD: blocks of wps*256 threads, one block per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = (double)iters * REP * wps;
        printf("%-18s %d waves/SIMD: %.2f ns per instruction per SIMD\n", name, wps, ms * 1e6 / instr_per_simd);
    }
}
int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<11>("v_pk_add_f32", out); run<2>("v_fma_f64", out); run<7>("v_add_f64", out);
    run<3>("v_cvt_f32_ubyte0", out); run<5>("v_cvt_f64_f32", out); run<4>("v_cndmask_b32", out); run<9>("v_cmp_le_f32", out);
    run<8>("v_mov_b32", out); run<10>("v_add_u32", out); run<6>("v_mul_lo_u32", out);
    run<12>("v_mad_u32_u24", out); run<22>("v_mul_u32_u24", out); run<24>("v_mul_hi_u32", out); run<13>("v_cvt_f32_u32", out);
    run<14>("v_floor_f32", out); run<17>("v_fract_f32", out); run<15>("v_max_f32", out); run<18>("v_mul_f32", out);
    run<19>("v_sub_f32", out); run<16>("v_min3_u32", out); run<20>("v_lshl_add_u32", out); run<21>("v_and_b32", out);
    run<23>("v_readlane_b32", out);
    return 0;
}
