// Shared device helpers for libfeddat_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/feddat_hip.h"

// The 16-bit OPERAND FORMAT of the library is a build-time choice (feddat_operand_format(), include/feddat_hip.h):
//   libfeddat_hip.so      bf16 (8 exponent / 7 mantissa bits)  -- v_mfma_f32_16x16x32_bf16
//   libfeddat_hip_f16.so  IEEE half (5 / 10), -DFEDDAT_OPERANDS_F16 -- v_mfma_f32_16x16x32_f16, the same MFMA rate, 8x finer
//                         rounding of every frozen weight and activation operand (the reference's own GPU arithmetic is fp16
//                         autocast: accelerate_config.yaml:8); the caller keeps gradients in range with a power-of-two loss
//                         scale (engine.py), as the reference's GradScaler does.
// Every kernel is written against the type names below; staging, LDS images, fragment layouts and epilogues are byte-identical
// in both builds (16-bit elements), only the conversion instructions and the MFMA opcode differ.  `bf16` therefore reads
// "the operand type of this build"; code that needs bf16 whatever the build (the split-operand weight gradients) uses tbf16.
typedef __bf16 tbf16;
typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
#ifdef FEDDAT_OPERANDS_F16
typedef _Float16 bf16;
#define FD_OPERAND_FORMAT FEDDAT_OPERANDS_FP16
#define FD_MFMA_16X16X32_ASM "v_mfma_f32_16x16x32_f16"
#else
typedef __bf16 bf16;
#define FD_OPERAND_FORMAT FEDDAT_OPERANDS_BF16
#define FD_MFMA_16X16X32_ASM "v_mfma_f32_16x16x32_bf16"
#endif
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define FD_WAVE 64

// One target only.  Besides the K = 32 bf16 MFMA forms and the 160 KiB LDS plans, the fp8 quantisers assume OCP e4m3
// (finite max 448): on an fnuz-e4m3 part (gfx942, max 240) amax / 448 scaling would overflow to NaN.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libfeddat_hip.so is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

#define FD_CHECK_ARG(cond)                 \
    do {                                   \
        if (!(cond)) return FEDDAT_EINVAL; \
    } while (0)

#define FD_LAUNCH_RET()                                        \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        return e__ == hipSuccess ? FEDDAT_OK : FEDDAT_ELAUNCH; \
    } while (0)

// host side, device_state.hip: per-device caches (thread-safe, keyed by the calling thread's current HIP device)
int fd_device_cus(int* n_cu);                          // compute units of the current device
int fd_set_max_lds(const void* kernel, int bytes);     // hipFuncAttributeMaxDynamicSharedMemorySize once per (device, kernel)
// Wrong-result ablations (timing probes of tools/: skip-epilogue, no-store, no-compute, deferred-epilogue probe ...) exist
// only in the -DFEDDAT_ABLATE build (`python -m feddat_amd.build --ablate` -> libfeddat_hip_ablate.so, loaded explicitly by
// tools/ through lib.use_ablation_build()).  The production library compiles them out: FD_ABL(x) is the constant 0 there, and
// feddat_set_debug_flags rejects every bit outside FD_DEBUG_SELECT_BITS, which only choose among kernels that give the same
// (bit-identical) results: 1 / 2 everything on the two-group / one-wave-per-SIMD GEMM kernel (2 in attention.hip: the
// two-role backward), 1 | 2 together the DUAL form of the persistent GEMM (two independent 128 x 192 workgroups per CU; with 64:
// only where it has at least two rounds of tiles), 32 / 64 force 192- / 256-row tiles, 128 no small-tile kernel, 256 the K = 32 fp8 MFMA, bit 23 one
// attention-backward block per pair, bit 27 no 160-row GEMM tiles, bits 28..31 cap the persistent GEMM grid at 16 x value workgroups.
#ifdef FEDDAT_ABLATE
#define FD_ABL(x) (x)
#else
#define FD_ABL(x) 0
#endif
constexpr unsigned FD_DEBUG_SELECT_BITS = 1u | 2u | 32u | 64u | 128u | 256u | (1u << 23) | (1u << 27) | (0xfu << 28);
int fd_debug_flags();                                  // ablation flags (feddat_set_debug_flags), 0 in production
int fd_prepare_all_kernels();                          // sets every kernel's LDS attribute on the current device
int fd_prepare_gemm_kernels();
int fd_prepare_attn_kernels();

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based dropout mask (ALBEF BERT towers, xbert.py:216,333,360,440): element `idx` of the tensor a site drops is
// KEPT iff hash(idx; key0, key1, step) >= p * 2^32 -- two murmur3 finalisers with the keys injected; (key0, key1) name
// (seed, pass, site), `step` is the train-step counter read from device memory so that a captured hipGraph draws fresh masks
// on every replay.  The backward REGENERATES the mask, nothing is stored.  (The function is plain
// 32-bit integer arithmetic, so a host-side restatement draws the same bits: tests/test_dropout_gpu.py.)
__device__ __forceinline__ uint32_t fd_fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
struct FdDrop {
    uint32_t key0, key1s, thr;      // key1s = key1 + step * 0x632BE5AB (folded once per kernel)
    float scale;                    // 1 / (1 - p)
};
__device__ __forceinline__ FdDrop fd_drop_make(float p, uint32_t key0, uint32_t key1, const int* step) {
    FdDrop d;
    d.key0 = key0;
    d.key1s = key1 + (step ? (uint32_t)*step : 0u) * 0x632BE5ABu;
    d.thr = (uint32_t)((double)p * 4294967296.0);
    d.scale = 1.0f / (1.0f - p);
    return d;
}
__device__ __forceinline__ bool fd_drop_keep(const FdDrop& d, uint32_t idx) {
    return fd_fmix32(fd_fmix32(idx * 0x9E3779B1u + d.key0) + d.key1s) >= d.thr;
}

// erf-GELU and its derivative: HF ACT2FN["gelu"] / nn.GELU() (vilt.py:206).
// erf by Abramowitz-Stegun 7.1.26 (|abs error| < 1.5e-7, far below the bf16 output rounding of the GEMM epilogues it
// is fused into, and 2.5x fewer VALU ops than erff -- the GELU epilogue was ~30 % of the FFN1 GEMM's time).
// exp(-x^2) with x = u / sqrt(2) is also the Gaussian pdf factor of gelu', so one v_exp serves both.
__device__ __forceinline__ void erf_exp_f(float u, float& erf_x, float& exp_mx2) {
    const float x = u * 0.70710678118654752f;
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    exp_mx2 = __expf(-x * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * exp_mx2;
    erf_x = copysignf(e, x);
}
__device__ __forceinline__ float gelu_f(float u) {
    float er, ex;
    erf_exp_f(u, er, ex);
    return 0.5f * u * (1.0f + er);
}
__device__ __forceinline__ float gelu_grad_f(float u) {
    float er, ex;
    erf_exp_f(u, er, ex);
    return 0.5f * (1.0f + er) + u * 0.39894228040143268f * ex;
}

// Packed-math GELU / GELU' for the GEMM epilogues (bf16 outputs).  The A-S form above costs two quarter-rate
// transcendentals + ~12 scalar fp32 ops per element, which made the FFN1 epilogue VALU-bound (as long as its k-loop).
// Here: t = clamp(u / 4.5, -1, 1);  erf(u / sqrt 2) ~ t * P(t^2),  gelu'(u) - 1/2 ~ t * Q(t^2), 9 coefficients each
// (weighted-LSQ minimax fits, scratch fit script; fp32 Horner max abs error 5.3e-5 resp. 3.2e-4 over all u, i.e.
// 10-60x below the bf16 rounding of the value that is stored), evaluated two elements per v_pk_fma_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 odd_poly9_pk(f32x2 t, const float (&c)[9]) {
    const f32x2 t2 = t * t;
    f32x2 p = f32x2{c[8], c[8]};
#pragma unroll
    for (int k = 7; k >= 0; --k) p = p * t2 + f32x2{c[k], c[k]};
    return p * t;
}
__device__ __forceinline__ f32x2 clamp_unit_pk(f32x2 u) {
    f32x2 t = u * f32x2{1.0f / 4.5f, 1.0f / 4.5f};
    t[0] = __builtin_amdgcn_fmed3f(t[0], -1.0f, 1.0f);
    t[1] = __builtin_amdgcn_fmed3f(t[1], -1.0f, 1.0f);
    return t;
}
__device__ __forceinline__ f32x2 gelu_pk(f32x2 u) {
    constexpr float C[9] = {3.589798371e+00f, -1.207231863e+01f, 3.590728051e+01f, -8.046883329e+01f, 1.320808609e+02f,
                            -1.519935397e+02f, 1.145676310e+02f, -5.029529085e+01f, 9.684438904e+00f};
    const f32x2 e = odd_poly9_pk(clamp_unit_pk(u), C);
    const f32x2 h = u * f32x2{0.5f, 0.5f};
    return h * e + h;
}
__device__ __forceinline__ f32x2 gelu_grad_pk(f32x2 u) {
    constexpr float C[9] = {3.586068066e+00f, -2.393606056e+01f, 1.043871964e+02f, -2.983035769e+02f, 5.731157980e+02f,
                            -7.304086803e+02f, 5.886747112e+02f, -2.703110568e+02f, 5.369588787e+01f};
    return odd_poly9_pk(clamp_unit_pk(u), C) + f32x2{0.5f, 0.5f};
}
// gelu(u) AND gelu'(u) (the FFN1 epilogue that stores 8-bit gelu' codes): gelu' = 1/2 + erf(u / sqrt 2) / 2 + u phi(u) shares
// the erf polynomial with gelu; u phi(u) = u exp2(-u^2 log2(e) / 2) / sqrt(2 pi) costs one v_exp_f32 (~5/3 of a VALU slot on
// gfx950) + 4 packed ops per element pair instead of a second 9-coefficient polynomial (14 packed ops).  |error| of this
// gelu' <= 2.7e-5 + the exp's ulp (+ 7.2e-5 beyond |u| = 4.5, see below).
__device__ __forceinline__ void gelu_and_grad_pk(f32x2 u, f32x2& f, f32x2& gp) {
    constexpr float C[9] = {3.589798371e+00f, -1.207231863e+01f, 3.590728051e+01f, -8.046883329e+01f, 1.320808609e+02f,
                            -1.519935397e+02f, 1.145676310e+02f, -5.029529085e+01f, 9.684438904e+00f};
    const f32x2 t = clamp_unit_pk(u);
    const f32x2 t2 = t * t;
    f32x2 p = f32x2{C[8], C[8]};
#pragma unroll
    for (int k = 7; k >= 0; --k) p = p * t2 + f32x2{C[k], C[k]};
    const f32x2 e = p * t;
    const f32x2 h = u * f32x2{0.5f, 0.5f};
    f = h * e + h;
    // u phi(u) from the clamped t the polynomial already has: exp2(KE t^2) = exp(-(4.5 t)^2 / 2); beyond |u| = 4.5 the true
    // u phi(u) is < 7.2e-5 and the clamped one is 7.2e-5 (0.014 code steps)
    constexpr float KE = -0.72134752044448170f * 20.25f;
    const f32x2 a = t2 * f32x2{KE, KE};
    const f32x2 tx = t * f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};      // (u phi(u) / 4.5, |u| clamped)
    gp = tx * f32x2{4.5f * 0.39894228040143268f, 4.5f * 0.39894228040143268f} + (e * f32x2{0.5f, 0.5f} + f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ void gelu_and_grad4_pk(f32x4 u, f32x4& f, f32x4& gp) {
    f32x2 f0, f1, g0, g1;
    gelu_and_grad_pk(f32x2{u[0], u[1]}, f0, g0);
    gelu_and_grad_pk(f32x2{u[2], u[3]}, f1, g1);
    f = f32x4{f0[0], f0[1], f1[0], f1[1]};
    gp = f32x4{g0[0], g0[1], g1[0], g1[1]};
}
__device__ __forceinline__ f32x4 gelu4_pk(f32x4 u) {
    const f32x2 a = gelu_pk(f32x2{u[0], u[1]}), b = gelu_pk(f32x2{u[2], u[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}
__device__ __forceinline__ f32x4 gelu_grad4_pk(f32x4 u) {
    const f32x2 a = gelu_grad_pk(f32x2{u[0], u[1]}), b = gelu_grad_pk(f32x2{u[2], u[3]});
    return f32x4{a[0], a[1], b[0], b[1]};
}

// 8-bit gelu'(u) codes (FEDDAT_EPI_GELU_G8 / FEDDAT_EPI_MUL_G8, include/feddat_hip.h): gelu' lives in [-0.1290, 1.1290];
// code = round((g' - FEDDAT_G8_LO) / FEDDAT_G8_STEP) in 0..255 covers [-0.135, 1.14] with |error| <= STEP / 2 = 2.5e-3
// (the bf16 u it replaces carried |u| 2^-9 gelu''(u) <= 2e-3 at |u| ~ 1).  Encode: one fma onto 2^23 + offset puts the
// round-to-nearest-even code into the low byte of the float's bits (LO / STEP is an integer, so the offset is exact).
__device__ __forceinline__ unsigned fd_g8_encode4(const f32x4 gp) {
    constexpr float INV = 1.0f / FEDDAT_G8_STEP, OFF = 8388608.0f - FEDDAT_G8_LO / FEDDAT_G8_STEP;
    const f32x4 t = gp * f32x4{INV, INV, INV, INV} + f32x4{OFF, OFF, OFF, OFF};
    // (__float_as_uint, not __builtin_bit_cast: applied to a vector ELEMENT hipcc's bit_cast reads element 0 every time)
    const unsigned b0 = __float_as_uint(t[0]), b1 = __float_as_uint(t[1]), b2 = __float_as_uint(t[2]), b3 = __float_as_uint(t[3]);
    // v_perm_b32 D = perm(S0, S1, sel): selector 0..3 = byte of S1, 4..7 = byte of S0, 0x0c = zero
    return __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u) | __builtin_amdgcn_perm(b3, b2, 0x04000c0cu);
}
__device__ __forceinline__ f32x4 fd_g8_decode4(const unsigned w) {
    const f32x4 c = {(float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24)};
    return c * f32x4{FEDDAT_G8_STEP, FEDDAT_G8_STEP, FEDDAT_G8_STEP, FEDDAT_G8_STEP} +
           f32x4{FEDDAT_G8_LO, FEDDAT_G8_LO, FEDDAT_G8_LO, FEDDAT_G8_LO};
}

__device__ __forceinline__ bf16x8 cvt8(const f32x4 a, const f32x4 b) {
    bf16x8 r;
    r[0] = (bf16)a[0]; r[1] = (bf16)a[1]; r[2] = (bf16)a[2]; r[3] = (bf16)a[3];
    r[4] = (bf16)b[0]; r[5] = (bf16)b[1]; r[6] = (bf16)b[2]; r[7] = (bf16)b[3];
    return r;
}
__device__ __forceinline__ bf16x4 cvt4(const f32x4 a) {
    bf16x4 r;
    r[0] = (bf16)a[0]; r[1] = (bf16)a[1]; r[2] = (bf16)a[2]; r[3] = (bf16)a[3];
    return r;
}

// ds_read_b64_tr_b16 of 16-bit operand elements (either operand format: the instruction moves 16-bit lanes)
__device__ __forceinline__ bf16x4 fd_ds_read_tr16(const char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(bf16x4, v);
}

// MFMA wrappers.  Operand convention used everywhere in this library (gfx950
// v_mfma_f32_16x16x32_bf16): lane l supplies, for the non-contracted index (l & 15), the 8
// contraction slots (g = l >> 4, j = 0..7); the hardware pairs slot (g, j) of A with slot (g, j) of B.
// D: lane l holds D[row = 4 * (l >> 4) + r][col = l & 15], r = 0..3, rows indexed by A's
// non-contracted index, cols by B's.
__device__ __forceinline__ f32x4 mfma16x32(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef FEDDAT_OPERANDS_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}
// bf16 whatever the build's operand format (adapter_wgrad.hip: the hi / lo split needs bf16's fp32 exponent range)
__device__ __forceinline__ f32x4 mfma16x32_tbf16(tbf16x8 a, tbf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ tbf16x8 cvt8_tbf16(const f32x4 a, const f32x4 b) {
    tbf16x8 r;
    r[0] = (tbf16)a[0]; r[1] = (tbf16)a[1]; r[2] = (tbf16)a[2]; r[3] = (tbf16)a[3];
    r[4] = (tbf16)b[0]; r[5] = (tbf16)b[1]; r[6] = (tbf16)b[2]; r[7] = (tbf16)b[3];
    return r;
}
// fp8 (OCP e4m3 on gfx950) form of the same product: a 16-byte fragment read carries 16 contraction slots per lane = two
// K = 32 instructions (slots 0-7, then 8-15; A and B are split the same way, which is all the contraction needs).
typedef long i64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 mfma16x64_fp8(bf16x8 a, bf16x8 b, f32x4 c) {
    const i64x2 a2 = __builtin_bit_cast(i64x2, a), b2 = __builtin_bit_cast(i64x2, b);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a2[0], b2[0], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a2[1], b2[1], c, 0, 0, 0);
}
// CDNA4 block-scaled form, v_mfma_scale_f32_16x16x128_f8f6f4 with both formats e4m3 and unit (E8M0 = 127) block scales:
// 32 contraction slots per lane and operand = two 16-byte fragment reads, at TWICE the bf16 MFMA rate (the K = 32 fp8
// instruction above issues at the bf16 rate).  The per-row / per-channel fp32 scales of the quantisers stay on the
// accumulators.  Slot pairing: A and B take the same (fragment, byte) -> slot map, which is all the contraction needs.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16x128_fp8_mx(bf16x8 a_lo, bf16x8 a_hi, bf16x8 b_lo, bf16x8 b_hi, f32x4 c) {
    const i32x4 a0 = __builtin_bit_cast(i32x4, a_lo), a1 = __builtin_bit_cast(i32x4, a_hi);
    const i32x4 b0 = __builtin_bit_cast(i32x4, b_lo), b1 = __builtin_bit_cast(i32x4, b_hi);
    const i32x8 A = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const i32x8 B = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
// the same with real E8M0 block scales on the SECOND operand: lane (r = l & 15, g = l >> 4) passes in `scale_b` (low byte) the
// scale 2^(byte - 127) of the g-th 32-element k-block of row r (k = 32 g .. 32 g + 31, which lives in the first / last four
// registers of lanes (r, 2 (g & 1)), (r, 2 (g & 1) + 1): tools/mx_scale_probe.py); the first operand keeps unit scales
__device__ __forceinline__ f32x4 mfma16x128_fp8_mx_sb(bf16x8 a_lo, bf16x8 a_hi, bf16x8 b_lo, bf16x8 b_hi, f32x4 c, int scale_b) {
    const i32x4 a0 = __builtin_bit_cast(i32x4, a_lo), a1 = __builtin_bit_cast(i32x4, a_hi);
    const i32x4 b0 = __builtin_bit_cast(i32x4, b_lo), b1 = __builtin_bit_cast(i32x4, b_hi);
    const i32x8 A = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const i32x8 B = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 0, 0, 0, 0x7F7F7F7F, 0, scale_b);
}
__device__ __forceinline__ f32x4 mfma16x4_f32(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
