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

__global__ __launch_bounds__(256) void sgemm_kernel(SgemmArgs p) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int itile = wid / p.jgroups, jgrp = wid - itile * p.jgroups;
    if (itile * 16 >= p.I) return;
    const int split = blockIdx.y;
    const int k_begin = split * p.kchunk;
    const int k_end = min(p.K, k_begin + p.kchunk);
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
#pragma unroll 4
    for (int k0 = k_begin; k0 < k_end; k0 += 4) {
        const int k = k0 + g;
        const bool kv = k < k_end;
        const float a = (iv && kv) ? ap[(size_t)k * p.sa_k] : 0.f;
        asum += a;
#pragma unroll
        for (int t = 0; t < JT; ++t) {
            const float b = (jv[t] && kv) ? bp[t][(size_t)k * p.sb_k] : 0.f;
            acc[t] = mfma16x4_f32(a, b, acc[t]);
        }
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
    if (p.colsum && jgrp == 0) {
        asum += __shfl_xor(asum, 16, 64);
        asum += __shfl_xor(asum, 32, 64);
        if (g == 0 && iv) p.colsum[(size_t)split * p.colsum_split_stride + i] = p.alpha * asum;
    }
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
    const int waves = itiles * p.jgroups;
    hipLaunchKernelGGL(sgemm_kernel, dim3((waves + 3) / 4, ksplit), dim3(256), 0, stream, p);
    FD_LAUNCH_RET();
}

extern "C" int feddat_reduce_partials(const float* in, long stride, int nsplit, long n, float* out,
                                      hipStream_t stream) {
    FD_CHECK_ARG(in && out && nsplit > 0 && n > 0);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, stride,
                       nsplit, n, out);
    FD_LAUNCH_RET();
}
