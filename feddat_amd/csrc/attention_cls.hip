// Self-attention of the LAST ViLT layer restricted to what the model consumes: HF ViltPooler takes hidden_states[:, 0]
// (reference call site src/modeling/vilt.py:127), so behind the top layer's attention only token 0 of every sample is
// live -- ONE query per (sample, head) in the forward, and in the backward a gradient that is non-zero on that one row of
// dctx.  Everything is rank-1 then: scores s_k = q0 . K_k / 8, p = softmax(s + mask), ctx_0 = sum_k p_k V_k;
//   D = dO . O,  dP_k = dO . V_k,  dS_k = p_k (dP_k - D),  dV_k = p_k dO,  dK_k = dS_k q0 / 8,  dQ_0 = sum_k dS_k K_k / 8,
// dQ_k = 0 for k > 0.  No MFMA: one wave per (sample, head), fp32 VALU on the bf16 operands, HBM-bound (K and V of the pair
// once in the forward; K, V once and the three dqkv slices once in the backward) -- against 21 / 41 us for the dense kernels
// computing 185 queries of which 184 are never read.  Same operand layout as attention.hip: qkv bf16 [B*S, 3*H],
// columns [Q | K | V], head h at columns 64 h of each part.
#include "common.hip.h"

namespace {

constexpr int D = 64;
constexpr int MAXK = 5;      // keys per lane: S <= 320

__device__ __forceinline__ void load_row64(const bf16* p, float (&o)[D]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[c * 8 + e] = (float)v[e];
    }
}

// 4 waves per block, one (sample, head) pair per wave
__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                           bf16* __restrict__ ctx, float* __restrict__ lse, int npairs,
                                                           int S, int heads) {
    __shared__ float pbuf[4][MAXK * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= npairs) return;
    const int b = pair / heads, h = pair - b * heads;
    const int H = heads * D;
    const long ld = 3L * H;
    const bf16* base = qkv + (size_t)b * S * ld + h * D;
    float q[D];
    load_row64(base, q);                    // token 0's query: every lane holds it
    float s[MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
        const int k = lane + 64 * kk;
        s[kk] = -INFINITY;
        if (k < S && (!kmask || kmask[(size_t)b * S + k])) {
            float kr[D];
            load_row64(base + (size_t)k * ld + H, kr);
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) acc += q[d] * kr[d];
            s[kk] = acc * 0.125f;
        }
        mx = fmaxf(mx, s[kk]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
        s[kk] = __expf(s[kk] - mx);         // exp(-inf) = 0 for masked / out-of-range keys
        sum += s[kk];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) pbuf[wave][lane + 64 * kk] = s[kk] * inv;
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): the wave's own LDS writes before its reads (no other wave shares pbuf[wave])
    __builtin_amdgcn_wave_barrier();
    // ctx_0[d]: lane = (key parity, feature pair): 32 lanes x 4 bytes = one 128-byte V row per key
    const int half = lane >> 5, dp = lane & 31;
    float o0 = 0.f, o1 = 0.f;
    const bf16* vb = base + 2 * H + 2 * dp;
#pragma unroll 8
    for (int k = half; k < S; k += 2) {
        const float p = pbuf[wave][k];
        const bf16x2 v = *reinterpret_cast<const bf16x2*>(vb + (size_t)k * ld);
        o0 += p * (float)v[0];
        o1 += p * (float)v[1];
    }
    o0 += __shfl_xor(o0, 32, 64);
    o1 += __shfl_xor(o1, 32, 64);
    if (half == 0) {
        bf16x2 o = {(bf16)o0, (bf16)o1};
        *reinterpret_cast<bf16x2*>(ctx + (size_t)b * S * H + h * D + 2 * dp) = o;
    }
    if (lane == 0 && lse) lse[((size_t)b * heads + h) * S] = mx + __logf(sum);
}

// Backward: dctx0 fp32 [B, H] = gradient of token 0's context row; writes the pair's complete dqkv slices (dQ rows 1..S-1
// are zeros: the dense QKV^T product that follows reads every row).
__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                           const bf16* __restrict__ ctx, const float* __restrict__ lse,
                                                           const float* __restrict__ dctx0, bf16* __restrict__ dqkv,
                                                           int npairs, int S, int heads) {
    __shared__ float red[4][D][17];         // per-wave transpose buffer for the dQ_0 reduction over keys
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= npairs) return;
    const int b = pair / heads, h = pair - b * heads;
    const int H = heads * D;
    const long ld = 3L * H;
    const bf16* base = qkv + (size_t)b * S * ld + h * D;
    bf16* dbase = dqkv + (size_t)b * S * ld + h * D;
    float q[D], g[D];
    load_row64(base, q);
    float Dv = 0.f;
    {
        float o[D];
        load_row64(ctx + (size_t)b * S * H + h * D, o);
        const float* gp = dctx0 + (size_t)b * H + h * D;
#pragma unroll
        for (int c = 0; c < D / 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gp + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g[4 * c + e] = v[e];
                Dv += v[e] * o[4 * c + e];
            }
        }
    }
    const float l0 = lse[((size_t)b * heads + h) * S];
    float dq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) dq[d] = 0.f;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int kk = 0; kk < MAXK; ++kk) {
        const int k = lane + 64 * kk;
        if (k >= S) break;
        float kr[D], vr[D];
        load_row64(base + (size_t)k * ld + H, kr);
        load_row64(base + (size_t)k * ld + 2 * H, vr);
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            sc += q[d] * kr[d];
            dp += g[d] * vr[d];
        }
        const bool ok = !kmask || kmask[(size_t)b * S + k];
        const float p = ok ? __expf(sc * 0.125f - l0) : 0.f;
        const float ds = p * (dp - Dv) * 0.125f;
        bf16* ok_ = dbase + (size_t)k * ld + H;
        bf16* ov_ = dbase + (size_t)k * ld + 2 * H;
        bf16* oq_ = dbase + (size_t)k * ld;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            bf16x8 dk8, dv8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dk8[e] = (bf16)(ds * q[c * 8 + e]);
                dv8[e] = (bf16)(p * g[c * 8 + e]);
                dq[c * 8 + e] += ds * kr[c * 8 + e];
            }
            *reinterpret_cast<bf16x8*>(ok_ + c * 8) = dk8;
            *reinterpret_cast<bf16x8*>(ov_ + c * 8) = dv8;
            if (k > 0) *reinterpret_cast<bf16x8*>(oq_ + c * 8) = zero8;
        }
    }
    // dQ_0[d] = sum over lanes of dq[d]: 16-lane partial sums by shuffles, then 4 partials per d through LDS
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float v = dq[d];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        if ((lane & 15) == 0) red[wave][d][lane >> 4] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const float tot = (red[wave][lane][0] + red[wave][lane][1]) + (red[wave][lane][2] + red[wave][lane][3]);
    dbase[lane] = (bf16)tot;                // row 0, feature d = lane
}

}  // namespace

extern "C" int feddat_attn_cls_fwd(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S, int heads,
                                   hipStream_t stream) {
    FD_CHECK_ARG(qkv && ctx && B > 0 && S > 0 && S <= 64 * MAXK && heads > 0);
    const int np = B * heads;
    hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3((np + 3) / 4), dim3(256), 0, stream, (const bf16*)qkv, key_mask, (bf16*)ctx,
                       lse, np, S, heads);
    FD_LAUNCH_RET();
}

extern "C" int feddat_attn_cls_bwd(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse,
                                   const float* dctx0, void* dqkv, int B, int S, int heads, hipStream_t stream) {
    FD_CHECK_ARG(qkv && ctx && lse && dctx0 && dqkv && B > 0 && S > 0 && S <= 64 * MAXK && heads > 0);
    const int np = B * heads;
    hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3((np + 3) / 4), dim3(256), 0, stream, (const bf16*)qkv, key_mask,
                       (const bf16*)ctx, lse, dctx0, (bf16*)dqkv, np, S, heads);
    FD_LAUNCH_RET();
}
