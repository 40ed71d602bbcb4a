// Probe: does gfx950 interlock "VALU writes an SGPR pair (v_cmp_*_e64 sdst) -> SALU reads it" without software wait
// states?  (compilers insert none on gfx9; this checks the hardware on the box.)  Each wave primes s[20:21] with a
// stale value, compares, reads the pair with a SALU instruction GAP instructions later, and checks the value against
// the ballot computed the ordinary way.  hipcc -O2 --offload-arch=gfx950 -o probe tools/sgpr_hazard_probe.hip && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int GAP, int PRE> __global__ void k(unsigned long long *bad, unsigned long long *first, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned long long nbad = 0;
    float x = lane * 0.5f + 1.0f;
    for (int i = 0; i < iters; ++i) {
        const int thr = (i * 7 + blockIdx.x) % 70;
        const int v = lane;
        const unsigned long long expect = __ballot(thr > v);
        const unsigned long long stale = ~expect;
        unsigned long long out;
        // PRE dependent VALU ops in front (a busy vector pipe), then the compare, then the scalar read
        if (PRE) asm volatile("v_mul_f32 %0, %0, %0\n v_rcp_f32 %0, %0\n v_mul_f32 %0, %0, %0\n v_sqrt_f32 %0, %0" : "+v"(x));
        if (GAP == 0)
            asm volatile("s_mov_b64 s[20:21], %3\n v_cmp_gt_i32_e64 s[20:21], %1, %2\n s_and_b64 %0, s[20:21], exec"
                         : "=s"(out) : "s"(thr), "v"(v), "s"(stale) : "s20", "s21", "scc");
        else if (GAP == 1)
            asm volatile("s_mov_b64 s[20:21], %3\n v_cmp_gt_i32_e64 s[20:21], %1, %2\n s_mov_b32 s22, 0\n s_and_b64 %0, s[20:21], exec"
                         : "=s"(out) : "s"(thr), "v"(v), "s"(stale) : "s20", "s21", "s22", "scc");
        else if (GAP == 2)
            asm volatile("s_mov_b64 s[20:21], %3\n v_cmp_gt_i32_e64 s[20:21], %1, %2\n s_nop 3\n s_and_b64 %0, s[20:21], exec"
                         : "=s"(out) : "s"(thr), "v"(v), "s"(stale) : "s20", "s21", "scc");
        else   // the cascade kernel's pattern: compare, one scalar op on OTHER registers, then 32-bit selects on the halves
            asm volatile("s_mov_b64 vcc, 0\n s_mov_b64 s[20:21], %3\n v_cmp_gt_i32_e64 s[20:21], %1, %2\n s_and_b64 s[22:23], vcc, exec\n s_cselect_b32 s21, -1, s21\n"
                         "s_cselect_b32 s20, -1, s20\n s_mov_b64 %0, s[20:21]"
                         : "=s"(out) : "s"(thr), "v"(v), "s"(stale) : "s20", "s21", "s22", "s23", "scc", "vcc");
        unsigned long long want = expect;
        if (GAP == 3) want = expect;   // (vcc & exec != 0 selects -1: vcc is zeroed below so the selects keep the pair)
        if (out != want) {
            if (nbad == 0 && lane == 0) first[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = out ^ want;
            ++nbad;
        }
    }
    if (lane == 0 && nbad) atomicAdd(bad, nbad);
    if (x == 123.0f) bad[1] = 1;
}
template <int GAP, int PRE> void run(const char *what) {
    unsigned long long *bad, *first;
    hipMalloc(&bad, 16); hipMalloc(&first, 8 * 4096 * 8);
    hipMemset(bad, 0, 16); hipMemset(first, 0, 8 * 4096 * 8);
    hipLaunchKernelGGL((k<GAP, PRE>), dim3(4096), dim3(512), 0, 0, bad, first, 2000);
    hipDeviceSynchronize();
    unsigned long long h[2], f[64];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(f, first, sizeof(f), hipMemcpyDeviceToHost);
    printf("%-58s mismatches %llu of %llu", what, h[0], 4096ull * 8 * 2000);
    for (int i = 0, n = 0; i < 64 && n < 3; ++i) if (f[i]) { printf("  xor %016llx", f[i]); ++n; }
    printf("\n");
    hipFree(bad); hipFree(first);
}
int main() {
    run<0, 0>("v_cmp sdst -> s_and next instruction");
    run<0, 1>("v_cmp sdst -> s_and next instruction, busy vector pipe");
    run<1, 0>("v_cmp sdst -> one scalar op -> s_and");
    run<1, 1>("v_cmp sdst -> one scalar op -> s_and, busy vector pipe");
    run<2, 1>("v_cmp sdst -> s_nop 3 -> s_and, busy vector pipe");
    run<3, 0>("v_cmp sdst -> s_and vcc -> s_cselect_b32 on the halves");
    run<3, 1>("v_cmp sdst -> s_and vcc -> s_cselect_b32 halves, busy pipe");
    return 0;
}
