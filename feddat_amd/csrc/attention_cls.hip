// Self-attention of the LAST ViLT layer restricted to what the model consumes: HF ViltPooler takes hidden_states[:, 0]
// (reference call site src/modeling/vilt.py:127), so behind the top layer's attention only token 0 of every sample is
// live -- ONE query per (sample, head) in the forward, and in the backward a gradient that is non-zero on that one row of
// dctx.  Everything is rank-1 then: scores s_k = q0 . K_k / 8, p = softmax(s + mask), ctx_0 = sum_k p_k V_k;
//   D = dO . O,  dP_k = dO . V_k,  dS_k = p_k (dP_k - D),  dV_k = p_k dO,  dK_k = dS_k q0 / 8,  dQ_0 = sum_k dS_k K_k / 8,
// dQ_k = 0 for k > 0.  No MFMA: one block per (sample, head), 8 lanes per key row (whole 128-byte rows per instruction), fp32 VALU on the bf16 operands, HBM-bound (K and V of the pair
// once in the forward; K, V once and the three dqkv slices once in the backward) -- against 21 / 41 us for the dense kernels
// computing 185 queries of which 184 are never read.  Same operand layout as attention.hip: qkv bf16 [B*S, 3*H],
// columns [Q | K | V], head h at columns 64 h of each part.
#include "common.hip.h"

namespace {

constexpr int D = 64;
constexpr int SMAX = 320;

// Lane layout of both kernels: 8 lanes per key row (lane & 7 = which 16-byte chunk of the 128-byte head slice), 8 keys per
// wave-instruction (lane >> 3), so every load / store instruction moves 8 whole 128-byte rows; a dot product over the 64
// features is 8 FMAs per lane + 3 shuffles.  4 waves per block = 32 keys per sweep; one block per (sample, head).
__device__ __forceinline__ float sum8(float v) {        // over the 8 lanes of a key row
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ __forceinline__ float sum_keys(float v) {    // over the 8 key slots of a wave (same chunk lane)
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float block_reduce_max(float v, float* red, int tid) {
    v = wave_max(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void attn_cls_fwd_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                           bf16* __restrict__ ctx, float* __restrict__ lse, int S,
                                                           int heads) {
    __shared__ float pbuf[SMAX], red[4], part[4][D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = lane & 7, slot = tid >> 3;           // chunk of the row, key slot in the block (0..31)
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int H = heads * D;
    const long ld = 3L * H;
    const bf16* base = qkv + (size_t)b * S * ld + h * D + ch * 8;
    float q[8];
    {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(base);      // token 0's query, this lane's chunk
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = (float)v[e];
    }
    float mx = -INFINITY;
    for (int k0 = 0; k0 < S; k0 += 32) {
        const int k = k0 + slot;
        float sc = -INFINITY;
        if (k < S) {
            const bf16x8 kr = *reinterpret_cast<const bf16x8*>(base + (size_t)k * ld + H);
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a += q[e] * (float)kr[e];
            a = sum8(a) * 0.125f;
            if (!kmask || kmask[(size_t)b * S + k]) sc = a;
            if (ch == 0) pbuf[k] = sc;
        }
        mx = fmaxf(mx, sc);
    }
    mx = block_reduce_max(mx, red, tid);                // (its barriers publish pbuf)
    float sum = 0.f;
    for (int k = tid; k < S; k += 256) {
        const float e = __expf(pbuf[k] - mx);           // exp(-inf) = 0 for masked keys
        pbuf[k] = e;
        sum += e;
    }
    sum = block_reduce_sum(sum, red, tid);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    for (int k0 = 0; k0 < S; k0 += 32) {
        const int k = k0 + slot;
        if (k < S) {
            const float p = pbuf[k];
            const bf16x8 vr = *reinterpret_cast<const bf16x8*>(base + (size_t)k * ld + 2 * H);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += p * (float)vr[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = sum_keys(o[e]);
        if (lane < 8) part[wave][lane * 8 + e] = t;
    }
    __syncthreads();
    if (tid < D) {
        const float tot = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) / sum;
        ctx[(size_t)b * S * H + h * D + tid] = (bf16)tot;
        if (tid == 0 && lse) lse[((size_t)b * heads + h) * S] = mx + __logf(sum);
    }
}

// Backward: dctx0 fp32 [B, H] = gradient of token 0's context row; writes the pair's complete dqkv slices (dQ rows 1..S-1
// are zeros: the dense QKV^T product that follows reads every row).
__global__ __launch_bounds__(256) void attn_cls_bwd_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                           const bf16* __restrict__ ctx, const float* __restrict__ lse,
                                                           const float* __restrict__ dctx0, bf16* __restrict__ dqkv, int S,
                                                           int heads) {
    __shared__ float red[4], part[4][D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = lane & 7, slot = tid >> 3;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int H = heads * D;
    const long ld = 3L * H;
    const bf16* base = qkv + (size_t)b * S * ld + h * D + ch * 8;
    bf16* dbase = dqkv + (size_t)b * S * ld + h * D + ch * 8;
    float q[8], g[8], dpart = 0.f;
    {
        const bf16x8 qv = *reinterpret_cast<const bf16x8*>(base);
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(ctx + (size_t)b * S * H + h * D + ch * 8);
        const float* gp = dctx0 + (size_t)b * H + h * D + ch * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            q[e] = (float)qv[e];
            g[e] = e < 4 ? g0[e] : g1[e - 4];
            dpart += g[e] * (float)ov[e];
        }
    }
    const float Dv = sum8(dpart);                       // D = dO . O (every 8-lane group holds the whole row)
    const float l0 = lse[((size_t)b * heads + h) * S];
    float dq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dq[e] = 0.f;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k0 = 0; k0 < S; k0 += 32) {
        const int k = k0 + slot;
        if (k < S) {
            const bf16x8 kr = *reinterpret_cast<const bf16x8*>(base + (size_t)k * ld + H);
            const bf16x8 vr = *reinterpret_cast<const bf16x8*>(base + (size_t)k * ld + 2 * H);
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sc += q[e] * (float)kr[e];
                dp += g[e] * (float)vr[e];
            }
            sc = sum8(sc);
            dp = sum8(dp);
            const bool ok = !kmask || kmask[(size_t)b * S + k];
            const float p = ok ? __expf(sc * 0.125f - l0) : 0.f;
            const float ds = p * (dp - Dv) * 0.125f;
            bf16x8 dk8, dv8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dk8[e] = (bf16)(ds * q[e]);
                dv8[e] = (bf16)(p * g[e]);
                dq[e] += ds * (float)kr[e];
            }
            *reinterpret_cast<bf16x8*>(dbase + (size_t)k * ld + H) = dk8;
            *reinterpret_cast<bf16x8*>(dbase + (size_t)k * ld + 2 * H) = dv8;
            if (k > 0) *reinterpret_cast<bf16x8*>(dbase + (size_t)k * ld) = zero8;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float t = sum_keys(dq[e]);
        if (lane < 8) part[wave][lane * 8 + e] = t;
    }
    __syncthreads();
    if (tid < D) dqkv[(size_t)b * S * ld + h * D + tid] = (bf16)((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]));
}

}  // namespace

extern "C" int feddat_attn_cls_fwd(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S, int heads,
                                   hipStream_t stream) {
    FD_CHECK_ARG(qkv && ctx && B > 0 && S > 0 && S <= SMAX && heads > 0);
    hipLaunchKernelGGL(attn_cls_fwd_kernel, dim3(B * heads), dim3(256), 0, stream, (const bf16*)qkv, key_mask, (bf16*)ctx, lse, S,
                       heads);
    FD_LAUNCH_RET();
}

extern "C" int feddat_attn_cls_bwd(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse,
                                   const float* dctx0, void* dqkv, int B, int S, int heads, hipStream_t stream) {
    FD_CHECK_ARG(qkv && ctx && lse && dctx0 && dqkv && B > 0 && S > 0 && S <= SMAX && heads > 0);
    hipLaunchKernelGGL(attn_cls_bwd_kernel, dim3(B * heads), dim3(256), 0, stream, (const bf16*)qkv, key_mask,
                       (const bf16*)ctx, lse, dctx0, (bf16*)dqkv, S, heads);
    FD_LAUNCH_RET();
}
