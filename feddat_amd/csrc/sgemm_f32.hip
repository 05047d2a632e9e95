// Exact-fp32 small GEMM on v_mfma_f32_16x16x4_f32 with arbitrary strides and split-K partial sums.
// Serves the trainable fp32 pieces of the path where bf16 inputs would cost parity: the task head
// (src/modeling/vilt.py:202-209) forward/backward, the ViLT pooler, and the adapter weight gradients
// dW_up = dy^T z, dW_down = dz^T x (contraction over tokens; autograd of adapter.py:125-131).
// f32 MFMA: A operand lane l = A[i = l & 15][k = l >> 4], B operand = B[k = l >> 4][j = l & 15] -- one
// dword per lane per step, so no operand ever needs a transposed copy.
#include "common.hip.h"

namespace {

constexpr int JT = 4;  // j-tiles (16 columns each) per wave sharing one A operand

struct SgemmArgs {
    const float* A;
    const float* B;
    const float* bias_j;
    float* out;
    float* colsum;
    long sa_i, sa_k, sb_k, sb_j, ldo, out_split_stride, colsum_split_stride;
    int I, J, K, ksplit, kchunk, jgroups;
    float alpha;
};

// One block = one (i-tile, j-group) x one grid split; its 4 waves take interleaved K sub-ranges and are summed
// through LDS in fixed order.  Loads are unconditional (indices clamped, values masked) so that the unrolled
// loop keeps 8 steps of loads in flight -- these products are latency-bound, not bandwidth-bound.
__global__ __launch_bounds__(256) void sgemm_kernel(SgemmArgs p) {
    __shared__ __attribute__((aligned(16))) float red[3][64][JT * 4 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wid = blockIdx.x;
    const int itile = wid / p.jgroups, jgrp = wid - itile * p.jgroups;
    const int split = blockIdx.y;
    // grid split -> [k_lo, k_hi); inside it wave w owns k-steps w, w+4, w+8, ... (4 consecutive k each)
    const int k_lo = split * p.kchunk;
    const int k_hi = min(p.K, k_lo + p.kchunk);
    const int i16 = lane & 15, g = lane >> 4;
    const int i = itile * 16 + i16;
    const bool iv = i < p.I;
    const float* ap = p.A + (size_t)(iv ? i : 0) * p.sa_i;
    int j[JT];
    bool jv[JT];
    const float* bp[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t) {
        j[t] = (jgrp * JT + t) * 16 + i16;
        jv[t] = j[t] < p.J;
        bp[t] = p.B + (size_t)(jv[t] ? j[t] : 0) * p.sb_j;
    }
    f32x4 acc[JT];
#pragma unroll
    for (int t = 0; t < JT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;
    const int kmax = max(k_hi - 1, k_lo);
    constexpr int U = 4;   // manual unroll: 4 k-steps (20 loads) in flight per wave
    for (int kb = k_lo + 4 * wave; kb < k_hi; kb += 16 * U) {
        float a[U], b[U][JT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = kb + 16 * u + g;
            const int kc = min(k, kmax);
            a[u] = ap[(size_t)kc * p.sa_k];
#pragma unroll
            for (int t = 0; t < JT; ++t) b[u][t] = bp[t][(size_t)kc * p.sb_k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool kv = (kb + 16 * u + g) < k_hi;
            const float av = (iv && kv) ? a[u] : 0.f;
            asum += av;
#pragma unroll
            for (int t = 0; t < JT; ++t) acc[t] = mfma16x4_f32(av, (jv[t] && kv) ? b[u][t] : 0.f, acc[t]);
        }
    }
    asum += __shfl_xor(asum, 16, 64);
    asum += __shfl_xor(asum, 32, 64);
    if (wave > 0) {
        float* dst = red[wave - 1][lane];
#pragma unroll
        for (int t = 0; t < JT; ++t) *reinterpret_cast<f32x4*>(dst + 4 * t) = acc[t];
        dst[JT * 4] = asum;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float* src = red[w][lane];
#pragma unroll
        for (int t = 0; t < JT; ++t) acc[t] = acc[t] + *reinterpret_cast<const f32x4*>(src + 4 * t);
        asum += src[JT * 4];
    }
    float* o = p.out + (size_t)split * p.out_split_stride;
#pragma unroll
    for (int t = 0; t < JT; ++t) {
        if (!jv[t]) continue;
        const float bj = (p.bias_j && split == 0) ? p.bias_j[j[t]] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int io = itile * 16 + 4 * g + e;
            if (io < p.I) o[(size_t)io * p.ldo + j[t]] = p.alpha * acc[t][e] + bj;
        }
    }
    if (p.colsum && jgrp == 0 && g == 0 && iv) p.colsum[(size_t)split * p.colsum_split_stride + i] = p.alpha * asum;
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ in, long stride, int nsplit,
                                                              long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += in[(size_t)k * stride + i];
    out[i] = s;
}

}  // namespace

extern "C" int feddat_sgemm_f32(const float* A, long sa_i, long sa_k, const float* B, long sb_k, long sb_j, int I,
                                int J, int K, int ksplit, float alpha, const float* bias_j, float* out, long ldo,
                                long out_split_stride, float* colsum, long colsum_split_stride, hipStream_t stream) {
    FD_CHECK_ARG(A && B && out && I > 0 && J > 0 && K > 0 && ksplit > 0 && ksplit <= 65535);
    FD_CHECK_ARG(ldo >= J && (ksplit == 1 || out_split_stride >= (long)I * ldo));
    SgemmArgs p;
    p.A = A; p.B = B; p.bias_j = bias_j; p.out = out; p.colsum = colsum;
    p.sa_i = sa_i; p.sa_k = sa_k; p.sb_k = sb_k; p.sb_j = sb_j; p.ldo = ldo; p.out_split_stride = out_split_stride;
    p.colsum_split_stride = colsum_split_stride;
    p.I = I; p.J = J; p.K = K; p.ksplit = ksplit; p.alpha = alpha;
    int kchunk = (K + ksplit - 1) / ksplit;
    kchunk = (kchunk + 3) / 4 * 4;
    p.kchunk = kchunk;
    const int itiles = (I + 15) / 16;
    p.jgroups = (J + 16 * JT - 1) / (16 * JT);
    hipLaunchKernelGGL(sgemm_kernel, dim3(itiles * p.jgroups, ksplit), dim3(256), 0, stream, p);
    FD_LAUNCH_RET();
}

extern "C" int feddat_reduce_partials(const float* in, long stride, int nsplit, long n, float* out,
                                      hipStream_t stream) {
    FD_CHECK_ARG(in && out && nsplit > 0 && n > 0);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, stride,
                       nsplit, n, out);
    FD_LAUNCH_RET();
}
