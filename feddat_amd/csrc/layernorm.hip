// K3: LayerNorm forward / dX-only backward (frozen affine) / full backward (trainable affine).
// One 64-lane wave per row, the row lives in registers (H <= 2048), float4 I/O.  HBM-bound.
#include "common.hip.h"

namespace {

// MAXC = float4 chunks per lane (H <= 256 * MAXC); instantiated for 3 (H=768), 6 (H=1536), 8 (H<=2048)

template <int MAXC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long x_stride,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, int rows, int H, bf16* __restrict__ y16,
                                                     float* __restrict__ y32, float* __restrict__ stats,
                                                     unsigned char* __restrict__ y8 = nullptr,
                                                     float* __restrict__ y8_scale = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nc = H >> 2;
    const float* xr = x + (size_t)row * x_stride;
    f32x4 v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            v[c] = *reinterpret_cast<const f32x4*>(xr + i * 4);
            s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[c][e] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    if (stats && lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + i * 4);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + i * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * g4[e] + b4[e];
            if (y16) *reinterpret_cast<bf16x4*>(y16 + (size_t)row * H + i * 4) = cvt4(o);
            if (y32) *reinterpret_cast<f32x4*>(y32 + (size_t)row * H + i * 4) = o;
            v[c] = o;
        }
    }
    if (!y8) return;
    // fp8 (e4m3) copy with a per-row scale: y8 = round(y / s), s = max|y| / 448 (the fp8 GEMM multiplies s back in)
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + c * 64 < nc)
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[c][e]));
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) y8_scale[row] = sc;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, pk, true);
            *reinterpret_cast<int*>(y8 + (size_t)row * H + i * 4) = pk;
        }
    }
}

template <int MAXC>
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const bf16* __restrict__ dy16, const float* __restrict__ dy32,
                                                        long dy_stride, const float* __restrict__ x, long x_stride,
                                                        const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const float* __restrict__ dres,
                                                        long dres_stride, int rows, int H, float* __restrict__ out32,
                                                        long out_stride, bf16* __restrict__ out16,
                                                        unsigned char* __restrict__ out8 = nullptr,
                                                        float* __restrict__ out8_scale = nullptr, const int dres_every = 0) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nc = H >> 2;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    f32x4 xh[MAXC], gg[MAXC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)row * x_stride + i * 4);
            f32x4 d;
            if (dy16) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(dy16 + (size_t)row * dy_stride + i * 4);
                d = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
            } else {
                d = *reinterpret_cast<const f32x4*>(dy32 + (size_t)row * dy_stride + i * 4);
            }
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[c][e] = (xv[e] - mean) * rstd;
                gg[c][e] = d[e] * g4[e];
                s1 += gg[c][e];
                s2 += gg[c][e] * xh[c][e];
            }
        }
    }
    const float m1 = wave_sum(s1) / (float)H;
    const float m2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rstd * (gg[c][e] - m1 - xh[c][e] * m2);
            // dres_every = E > 0: the residual gradient is non-zero on rows 0, E, 2 E, ... only and comes COMPACT (row r / E of
            // dres): the top layer's, whose gradient lives on token 0 of every sample (wave-uniform condition)
            if (dres && (dres_every == 0 || row % dres_every == 0)) {
                const size_t rr = dres_every ? (size_t)(row / dres_every) : (size_t)row;
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(dres + rr * dres_stride + i * 4);
                o = o + r4;
            }
            if (out32) *reinterpret_cast<f32x4*>(out32 + (size_t)row * out_stride + i * 4) = o;
            if (out16) *reinterpret_cast<bf16x4*>(out16 + (size_t)row * H + i * 4) = cvt4(o);
            gg[c] = o;
        }
    }
    if (!out8) return;
    // e4m3 copy with a per-row scale (configs[4]: the A operand of the fp8 dX product that follows), as in ln_fwd_kernel
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane + c * 64 < nc)
#pragma unroll
            for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(gg[c][e]));
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (lane == 0) out8_scale[row] = sc;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(gg[c][0] * inv, gg[c][1] * inv, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(gg[c][2] * inv, gg[c][3] * inv, pk, true);
            *reinterpret_cast<int*>(out8 + (size_t)row * H + i * 4) = pk;
        }
    }
}

// trainable-affine backward for the task head's clf_norm0 (rows <= 1024): one block per 64 columns,
// lanes over columns for dgamma/dbeta; dx by the wave-per-row kernel above with fp32 dy.
__global__ __launch_bounds__(256) void ln_bwd_affine_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ stats, int rows, int H,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sg[4][64], sb[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    float ag = 0.f, ab = 0.f;
    if (col < H) {
        for (int r = part; r < rows; r += 4) {
            const float d = dy[(size_t)r * H + col];
            const float xh = (x[(size_t)r * H + col] - stats[2 * r]) * stats[2 * r + 1];
            ag += d * xh;
            ab += d;
        }
    }
    sg[part][threadIdx.x & 63] = ag;
    sb[part][threadIdx.x & 63] = ab;
    __syncthreads();
    if (part == 0 && col < H) {
        const int l = threadIdx.x;
        dgamma[col] = (sg[0][l] + sg[1][l]) + (sg[2][l] + sg[3][l]);
        dbeta[col] = (sb[0][l] + sb[1][l]) + (sb[2][l] + sb[3][l]);
    }
}

}  // namespace

extern "C" int feddat_layernorm_fwd(const float* x, long x_stride, const float* gamma, const float* beta, float eps,
                                    int rows, int H, void* y_bf16, float* y_f32, float* stats, hipStream_t stream) {
    FD_CHECK_ARG(x && gamma && beta && rows > 0 && H > 0 && H % 4 == 0 && H <= 2048 && x_stride % 4 == 0);
    FD_CHECK_ARG(y_bf16 || y_f32);
#define LN_FWD(MC)                                                                                              \
    hipLaunchKernelGGL(ln_fwd_kernel<MC>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, x_stride, gamma, beta, eps, \
                       rows, H, (bf16*)y_bf16, y_f32, stats)
    if (H <= 768) LN_FWD(3); else if (H <= 1536) LN_FWD(6); else LN_FWD(8);
#undef LN_FWD
    FD_LAUNCH_RET();
}

extern "C" int feddat_layernorm_fwd_fp8(const float* x, long x_stride, const float* gamma, const float* beta, float eps,
                                        int rows, int H, void* y_fp8, float* y_scale, void* y_bf16, float* stats,
                                        hipStream_t stream) {
    FD_CHECK_ARG(x && gamma && beta && rows > 0 && H > 0 && H % 4 == 0 && H <= 2048 && x_stride % 4 == 0);
    FD_CHECK_ARG(y_fp8 && y_scale);
#define LN_FWD8(MC)                                                                                              \
    hipLaunchKernelGGL(ln_fwd_kernel<MC>, dim3((rows + 3) / 4), dim3(256), 0, stream, x, x_stride, gamma, beta, eps, \
                       rows, H, (bf16*)y_bf16, (float*)nullptr, stats, (unsigned char*)y_fp8, y_scale)
    if (H <= 768) LN_FWD8(3); else if (H <= 1536) LN_FWD8(6); else LN_FWD8(8);
#undef LN_FWD8
    FD_LAUNCH_RET();
}

// x fp32 [rows, cols] (row stride ld) -> e4m3 [rows, cols] + per-row scale (amax / 448): frozen weights (per output
// channel) at load time, or any activation
namespace {
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const float* __restrict__ x, long ld, int cols,
                                                             unsigned char* __restrict__ y, float* __restrict__ scale) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* xr = x + (size_t)row * ld;
    float amax = 0.f;
    for (int c = tid * 4; c < cols; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
    if (tid == 0) scale[row] = sc;
    for (int c = tid * 4; c < cols; c += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, pk, true);
        *reinterpret_cast<int*>(y + (size_t)row * cols + c) = pk;
    }
}
}  // namespace

extern "C" int feddat_quant_rows_fp8(const float* x, long ld, int rows, int cols, void* y_fp8, float* scale,
                                     hipStream_t stream) {
    FD_CHECK_ARG(x && y_fp8 && scale && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && ld >= cols);
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3(rows), dim3(256), 0, stream, x, ld, cols, (unsigned char*)y_fp8, scale);
    FD_LAUNCH_RET();
}

extern "C" int feddat_layernorm_bwd_dx(const void* dy_bf16, const float* dy_f32, long dy_stride, const float* x,
                                       long x_stride, const float* stats, const float* gamma, const float* dres,
                                       long dres_stride, int rows, int H, float* out_f32, long out_stride,
                                       void* out_bf16, hipStream_t stream) {
    FD_CHECK_ARG((dy_bf16 != nullptr) != (dy_f32 != nullptr));
    FD_CHECK_ARG(x && stats && gamma && rows > 0 && H > 0 && H % 4 == 0 && H <= 2048);
    FD_CHECK_ARG(dy_stride % 4 == 0 && x_stride % 4 == 0 && (!dres || dres_stride % 4 == 0));
    FD_CHECK_ARG((out_f32 && out_stride % 4 == 0) || out_bf16);
#define LN_BWD(MC)                                                                                                \
    hipLaunchKernelGGL(ln_bwd_dx_kernel<MC>, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)dy_bf16, dy_f32, \
                       dy_stride, x, x_stride, stats, gamma, dres, dres_stride, rows, H, out_f32, out_stride,          \
                       (bf16*)out_bf16)
    if (H <= 768) LN_BWD(3); else if (H <= 1536) LN_BWD(6); else LN_BWD(8);
#undef LN_BWD
    FD_LAUNCH_RET();
}

extern "C" int feddat_layernorm_bwd_dx_sparse(const void* dy_bf16, const float* dy_f32, long dy_stride, const float* x,
                                              long x_stride, const float* stats, const float* gamma, const float* dres,
                                              long dres_stride, int dres_every, int rows, int H, float* out_f32,
                                              long out_stride, void* out_bf16, hipStream_t stream) {
    FD_CHECK_ARG((dy_bf16 != nullptr) != (dy_f32 != nullptr));
    FD_CHECK_ARG(x && stats && gamma && dres && dres_every > 0 && rows > 0 && H > 0 && H % 4 == 0 && H <= 2048);
    FD_CHECK_ARG(dy_stride % 4 == 0 && x_stride % 4 == 0 && dres_stride % 4 == 0);
    FD_CHECK_ARG((out_f32 && out_stride % 4 == 0) || out_bf16);
#define LN_BWDS(MC)                                                                                               \
    hipLaunchKernelGGL(ln_bwd_dx_kernel<MC>, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)dy_bf16, dy_f32, \
                       dy_stride, x, x_stride, stats, gamma, dres, dres_stride, rows, H, out_f32, out_stride,          \
                       (bf16*)out_bf16, (unsigned char*)nullptr, (float*)nullptr, dres_every)
    if (H <= 768) LN_BWDS(3); else if (H <= 1536) LN_BWDS(6); else LN_BWDS(8);
#undef LN_BWDS
    FD_LAUNCH_RET();
}

extern "C" int feddat_layernorm_bwd_dx_fp8(const void* dy_bf16, const float* dy_f32, long dy_stride, const float* x,
                                           long x_stride, const float* stats, const float* gamma, const float* dres,
                                           long dres_stride, int rows, int H, float* out_f32, long out_stride,
                                           void* out_bf16, void* out_fp8, float* out_scale, hipStream_t stream) {
    FD_CHECK_ARG((dy_bf16 != nullptr) != (dy_f32 != nullptr));
    FD_CHECK_ARG(x && stats && gamma && rows > 0 && H > 0 && H % 4 == 0 && H <= 2048 && out_fp8 && out_scale);
    FD_CHECK_ARG(dy_stride % 4 == 0 && x_stride % 4 == 0 && (!dres || dres_stride % 4 == 0) && (!out_f32 || out_stride % 4 == 0));
#define LN_BWD8(MC)                                                                                               \
    hipLaunchKernelGGL(ln_bwd_dx_kernel<MC>, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16*)dy_bf16, dy_f32, \
                       dy_stride, x, x_stride, stats, gamma, dres, dres_stride, rows, H, out_f32, out_stride,          \
                       (bf16*)out_bf16, (unsigned char*)out_fp8, out_scale)
    if (H <= 768) LN_BWD8(3); else if (H <= 1536) LN_BWD8(6); else LN_BWD8(8);
#undef LN_BWD8
    FD_LAUNCH_RET();
}

extern "C" int feddat_layernorm_bwd_full(const float* dy, const float* x, const float* stats, const float* gamma,
                                         int rows, int H, float* dx, float* dgamma, float* dbeta,
                                         hipStream_t stream) {
    FD_CHECK_ARG(dy && x && stats && gamma && dx && dgamma && dbeta && rows > 0 && rows <= 1024);
    FD_CHECK_ARG(H > 0 && H % 4 == 0 && H <= 2048);
    hipLaunchKernelGGL(ln_bwd_affine_kernel, dim3((H + 63) / 64), dim3(256), 0, stream, dy, x, stats, rows, H, dgamma,
                       dbeta);
    return feddat_layernorm_bwd_dx(nullptr, dy, (long)H, x, (long)H, stats, gamma, nullptr, 0L, rows, H, dx, (long)H,
                                   nullptr, stream);
}
