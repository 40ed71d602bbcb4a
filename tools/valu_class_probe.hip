// Probe: issue cost per wave64 VALU instruction on gfx950 at 4 waves per SIMD, eight independent
// instructions per asm block.  Establishes which opcodes are in the 2-cycle class and which in the 4-cycle class.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 32
#define X8(s) s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s "\n" s
#define OPS(F) \
  F(0, "v_mov_b32 %0, %4") F(1, "v_add_f32 %0, %4, %5") F(2, "v_sub_f32 %0, %4, %5") F(3, "v_mul_f32 %0, %4, %5") \
  F(4, "v_fma_f32 %0, %4, %5, %5") F(5, "v_fmac_f32 %0, %4, %5") F(6, "v_max_f32 %0, %4, %5") F(7, "v_min_f32 %0, %4, %5") \
  F(8, "v_med3_f32 %0, %4, %5, %5") F(9, "v_add_u32 %0, %4, %5") F(10, "v_sub_u32 %0, %4, %5") F(11, "v_and_b32 %0, %4, %5") \
  F(12, "v_or_b32 %0, %4, %5") F(13, "v_xor_b32 %0, %4, %5") F(14, "v_lshlrev_b32 %0, 3, %4") F(15, "v_lshrrev_b32 %0, 3, %4") \
  F(16, "v_ashrrev_i32 %0, 3, %4") F(17, "v_bfe_u32 %0, %4, 8, 8") F(18, "v_lshl_add_u32 %0, %4, 2, %5") F(19, "v_add3_u32 %0, %4, %5, %5") \
  F(20, "v_lshl_or_b32 %0, %4, 2, %5") F(21, "v_and_or_b32 %0, %4, %5, %5") F(22, "v_mad_u32_u24 %0, %4, %5, %5") F(23, "v_mul_u32_u24 %0, %4, %5") \
  F(24, "v_mul_lo_u32 %0, %4, %5") F(25, "v_cvt_f32_ubyte0 %0, %4") F(26, "v_cvt_f32_ubyte1 %0, %4") F(27, "v_cvt_f32_u32 %0, %4") \
  F(28, "v_cvt_f32_i32 %0, %4") F(29, "v_cvt_u32_f32 %0, %4") F(30, "v_cvt_i32_f32 %0, %4") F(31, "v_floor_f32 %0, %4") \
  F(32, "v_fract_f32 %0, %4") F(33, "v_rndne_f32 %0, %4") F(34, "v_trunc_f32 %0, %4") F(35, "v_max_u32 %0, %4, %5") \
  F(36, "v_min_u32 %0, %4, %5") F(37, "v_max_i32 %0, %4, %5") F(38, "v_perm_b32 %0, %4, %5, %5") F(39, "v_alignbit_b32 %0, %4, %5, %5") \
  F(40, "v_bfi_b32 %0, %4, %5, %5") F(41, "v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf") F(42, "v_add_f32_dpp %0, %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf") \
  F(43, "v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf") F(44, "v_and_b32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
  F(45, "v_add_f32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD") F(46, "v_mbcnt_lo_u32_b32 %0, -1, %4") F(47, "v_add_f32 %0, |%4|, %5") \
  F(48, "v_add_f32 %0, s20, %5") F(49, "v_mul_f32 %0, 0x3e800000, %5") F(50, "v_mul_f32 %0, 0.25, %5") F(51, "v_add_f32 %0, -%4, %5") \
  F(52, "v_sub_f32 %0, %4, %5 clamp") F(53, "v_mul_f32 %0, %4, %5 mul:2") F(54, "v_subrev_f32 %0, %4, %5") F(55, "v_sub_co_u32 %0, vcc, %4, %5") \
  F(56, "v_cmp_le_f32 vcc, %4, %5") F(57, "v_cmp_le_u32 vcc, %4, %5") F(58, "v_cmp_le_f32 s[20:21], %4, %5") F(59, "v_cndmask_b32_e64 %0, %4, %5, s[22:23]") \
  F(60, "v_readlane_b32 s20, %4, 3") F(61, "v_readfirstlane_b32 s20, %4") F(62, "v_mul_f64 %6, %7, %7") F(63, "v_add_f64 %6, %7, %7") \
  F(64, "v_fma_f64 %6, %7, %7, %7") F(65, "v_cvt_f64_f32 %6, %4") F(66, "v_cvt_f32_f64 %0, %7") F(67, "v_pk_add_f32 %6, %7, %7") \
  F(68, "v_pk_mul_f32 %6, %7, %7") F(69, "v_pk_fma_f32 %6, %7, %7, %7") F(70, "v_mad_i32_i24 %0, %4, %5, %5") F(71, "v_sad_u8 %0, %4, %5, %5") \
  F(72, "v_cvt_pk_u8_f32 %0, %4, 1, %5") F(73, "v_pk_add_u16 %0, %4, %5") F(74, "v_pk_mul_lo_u16 %0, %4, %5") F(75, "v_pk_mad_u16 %0, %4, %5, %5") \
  F(76, "v_dot4_u32_u8 %0, %4, %5, %5") F(77, "v_pk_max_f16 %0, %4, %5") F(78, "v_pk_fma_f16 %0, %4, %5, %5") F(79, "v_cvt_f16_f32 %0, %4") \
  F(80, "v_max3_f32 %0, %4, %5, %5") F(81, "v_mad_u64_u32 %6, vcc, %4, %5, %7") F(82, "v_lshl_add_u64 %6, %7, 2, %7") F(83, "v_sub_f32 %0, |%4|, %5") \
  F(84, "v_ldexp_f32 %0, %4, %5") F(85, "v_rcp_f32 %0, %4") F(86, "v_sqrt_f32 %0, %4") F(87, "v_mul_legacy_f32 %0, %4, %5") \
  F(88, "v_max_f64 %6, %7, %7") F(89, "v_cvt_i32_f64 %0, %7") F(90, "v_cvt_f64_u32 %6, %4") F(91, "v_floor_f64 %6, %7")
template <int OP> __global__ void k(float *out, int iters) {
    float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f;
    float d0 = 0; double x = 0, y = a;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#define CASE(n, s) if (OP == n) asm volatile(X8(s) : "+v"(d0), "+v"(d0), "+v"(d0), "+v"(d0), "+v"(x) : "v"(a), "v"(b), "v"(y) : "vcc", "s20", "s21");
#undef CASE
#define CASE(n, s) if (OP == n) asm volatile(X8(s) : "+v"(d0) : "v"(d0), "v"(d0), "v"(d0), "v"(a), "v"(b), "v"(x), "v"(y) : "vcc", "s20", "s21");
            OPS(CASE)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d0 + (float)x;
}
template <int OP> void run(const char *name, float *out) {
    double res[2]; int j = 0;
    for (int wps : {1, 4}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const int iters = 500;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, out, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        res[j++] = ms * 1e6 / ((double)iters * REP * 8 * wps) * 2.4;
    }
    printf("%-100s  1 wave/SIMD %5.2f   4 waves/SIMD %5.2f cycles\n", name, res[0], res[1]);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
#define RUN(n, s) run<n>(s, out);
    OPS(RUN)
    return 0;
}
