// Probe: do unaligned global dword loads and unaligned LDS u16/b32 reads return the right bytes on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
__global__ void probe(const uint8_t *src, uint32_t *out_g, uint32_t *out_l16, uint32_t *out_l32) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[512];
    int t = threadIdx.x;
    for (int i = t; i < 512; i += 64) lds[i] = src[i];
    __syncthreads();
    // unaligned global dword at byte offset t (t = 0..63)
    uint32_t g;
    const uint8_t *p = src + t;
    asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(g) : "v"(p) : "memory");
    out_g[t] = g;
    uint32_t a = (uint32_t)(uintptr_t)(lds) + t, v16, v32;   // LDS byte address
    asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v16) : "v"(a) : "memory");
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v32) : "v"(a) : "memory");
    out_l16[t] = v16;
    out_l32[t] = v32;
}
int main() {
    uint8_t h[512];
    for (int i = 0; i < 512; ++i) h[i] = (uint8_t)(i * 7 + 3);
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, 512); hipMalloc(&o, 3 * 64 * 4);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, o + 64, o + 128);
    uint32_t r[192];
    if (hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAULT\n"); return 1; }
    int bad_g = 0, bad16 = 0, bad32 = 0;
    for (int t = 0; t < 64; ++t) {
        uint32_t e32; memcpy(&e32, h + t, 4);
        uint16_t e16; memcpy(&e16, h + t, 2);
        bad_g += r[t] != e32; bad16 += r[64 + t] != e16; bad32 += r[128 + t] != e32;
    }
    printf("unaligned global dword mismatches %d, lds u16 %d, lds b32 %d\n", bad_g, bad16, bad32);
    return 0;
}
