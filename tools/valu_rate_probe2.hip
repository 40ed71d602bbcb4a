// Probe 2: v_cndmask_b32 variants (mask source, dependency on a preceding compare).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 128
template <int OP> __global__ void k(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a + 1.f, e = 2.f, f = 3.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (OP == 0) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c));
            if (OP == 1) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a) : "v"(b), "v"(c) : "s20", "s21");
            if (OP == 2) asm volatile("v_cmp_le_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c) : "vcc");
            if (OP == 3) asm volatile("v_cmp_le_f32_e64 s[20:21], %1, %2\n v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(a) : "v"(b), "v"(c) : "s20", "s21");
            if (OP == 4) asm volatile("v_cmp_le_f32 vcc, %2, %3\n v_cmp_le_f32_e64 s[20:21], %3, %2\n v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32_e64 %1, %2, %3, s[20:21]" : "=v"(a), "=v"(d) : "v"(b), "v"(c) : "vcc", "s20", "s21");
            if (OP == 5) asm volatile("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %4, %5, vcc" : "=v"(a), "=v"(d) : "v"(b), "v"(c), "v"(e), "v"(f));
            if (OP == 6) asm volatile("v_cmp_le_f32 vcc, %1, %2\n v_mov_b32 %0, %1\n v_mov_b32 %0, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a) : "v"(b), "v"(c) : "vcc");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d;
}
template <int OP> void run(const char *name, float *out, int n_instr) {
    for (int wps : {1, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, out, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %d waves/SIMD: %.2f ns per instruction per SIMD\n", name, wps, ms * 1e6 / ((double)iters * REP * wps * n_instr));
    }
}
int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    run<0>("cndmask vcc (no writer)", out, 1);
    run<1>("cndmask_e64 sgpr pair (no writer)", out, 1);
    run<2>("cmp->vcc ; cndmask vcc", out, 2);
    run<3>("cmp_e64->sgpr ; cndmask_e64 sgpr", out, 2);
    run<4>("2 cmps ; 2 cndmasks (vcc + sgpr)", out, 4);
    run<5>("2 cndmask vcc back to back", out, 2);
    run<6>("cmp ; mov ; mov ; cndmask", out, 4);
    return 0;
}
