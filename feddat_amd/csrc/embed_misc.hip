// K8 embeddings (HF ViltEmbeddings.forward restated for full pixel masks, raster patch order; reference call
// site src/modeling/vilt.py:127) and small element-wise helpers.  All HBM/latency-bound.
#include "common.hip.h"

namespace {

// one wave per text token: LN(word[id] + type[tt] + pos[s]) + modality[0]
__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ ids,
                                                         const int64_t* __restrict__ tts,
                                                         const float* __restrict__ word, const float* __restrict__ pos,
                                                         const float* __restrict__ type, const float* __restrict__ ln_g,
                                                         const float* __restrict__ ln_b, float eps,
                                                         const float* __restrict__ mod0, float* __restrict__ h, int B,
                                                         int Lt, int S, int H) {
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= B * Lt) return;
    const int b = tok / Lt, s = tok - b * Lt;
    const float* w = word + (size_t)ids[tok] * H;
    const float* ty = type + (size_t)tts[tok] * H;
    const float* po = pos + (size_t)s * H;
    const int nc = H >> 2;
    f32x4 v[8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(w + i * 4);
            const f32x4 t4 = *reinterpret_cast<const f32x4*>(ty + i * 4);
            const f32x4 p4 = *reinterpret_cast<const f32x4*>(po + i * 4);
            v[c] = (a + t4) + p4;
            sum += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
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
    float* out = h + ((size_t)b * S + s) * H;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(ln_g + i * 4);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(ln_b + i * 4);
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(mod0 + i * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * g4[e] + b4[e] + m4[e];
            *reinterpret_cast<f32x4*>(out + i * 4) = o;
        }
    }
}

// pixels [B,C,Hi,Wi] fp32 -> patches [B*gh*gw, C*P*P] bf16; one thread = 4 consecutive px of one patch row
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ px, bf16* __restrict__ out, int B,
                                                     int C, int Hi, int Wi, int P) {
    const int gw = Wi / P, gh = Hi / P;
    const long total = (long)B * C * Hi * Wi / 4;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long e = i * 4;              // linear index into pixels
    const int x = (int)(e % Wi);
    const int y = (int)((e / Wi) % Hi);
    const int c = (int)((e / ((long)Hi * Wi)) % C);
    const int b = (int)(e / ((long)Hi * Wi * C));
    const f32x4 v = *reinterpret_cast<const f32x4*>(px + e);
    const int py = y / P, iy = y - py * P, pxi = x / P, ix = x - pxi * P;
    const size_t row = ((size_t)b * gh + py) * gw + pxi;
    const size_t col = ((size_t)c * P + iy) * P + ix;
    *reinterpret_cast<bf16x4*>(out + row * ((size_t)C * P * P) + col) = cvt4(v);
}

__global__ __launch_bounds__(256) void image_assemble_kernel(const float* __restrict__ proj,
                                                             const float* __restrict__ cls,
                                                             const float* __restrict__ pos0,
                                                             const float* __restrict__ pos_img,
                                                             const float* __restrict__ mod1, float* __restrict__ h,
                                                             int B, int Lt, int np, int S, int H, long pos_bstride) {
    const int nc = H >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * (np + 1) * nc;
    if (i >= total) return;
    const int c = (int)(i % nc);
    const int t = (int)((i / nc) % (np + 1));
    const int b = (int)(i / ((long)nc * (np + 1)));
    const f32x4 m4 = *reinterpret_cast<const f32x4*>(mod1 + c * 4);
    f32x4 o;
    if (t == 0) {
        o = (*reinterpret_cast<const f32x4*>(cls + c * 4) + *reinterpret_cast<const f32x4*>(pos0 + c * 4)) + m4;
    } else {
        const int p = t - 1;
        o = (*reinterpret_cast<const f32x4*>(proj + ((size_t)b * np + p) * H + c * 4) +
             *reinterpret_cast<const f32x4*>(pos_img + (size_t)b * pos_bstride + (size_t)p * H + c * 4)) + m4;
    }
    *reinterpret_cast<f32x4*>(h + ((size_t)b * S + Lt + t) * H + c * 4) = o;
}

// bilinear, align_corners=True: src = dst * (g - 1) / (gdst - 1)
__global__ __launch_bounds__(256) void pos_resize_kernel(const float* __restrict__ grid, float* __restrict__ out,
                                                         int g, int gh, int gw, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)gh * gw * H) return;
    const int c = (int)(i % H);
    const int x = (int)((i / H) % gw);
    const int y = (int)(i / ((long)H * gw));
    // torch area_pixel_compute_scale(align_corners=True): scale = (in - 1) / (out - 1); src = scale * dst
    const float sy = gh > 1 ? ((float)(g - 1) / (float)(gh - 1)) * (float)y : 0.f;
    const float sx = gw > 1 ? ((float)(g - 1) / (float)(gw - 1)) * (float)x : 0.f;
    const int y0 = min((int)sy, g - 1), x0 = min((int)sx, g - 1);
    const int y1 = min(y0 + 1, g - 1), x1 = min(x0 + 1, g - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float v00 = grid[((size_t)y0 * g + x0) * H + c], v01 = grid[((size_t)y0 * g + x1) * H + c];
    const float v10 = grid[((size_t)y1 * g + x0) * H + c], v11 = grid[((size_t)y1 * g + x1) * H + c];
    out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

// HF ViltEmbeddings.visual_embed for padded images: per sample the valid patch rectangle is h = #valid rows of patch
// column 0, w = #valid columns of patch row 0 of the pixel mask sampled at the patch origins (F.interpolate 'nearest');
// the g x g grid is resized to h x w (bilinear, align_corners=True), placed top-left, zero elsewhere.
__global__ __launch_bounds__(256) void pos_resize_masked_kernel(const float* __restrict__ grid,
                                                                const long* __restrict__ pmask,
                                                                float* __restrict__ out, int g, int Hi, int Wi, int P,
                                                                int H) {
    const int gh = Hi / P, gw = Wi / P;
    const int b = blockIdx.y;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long* pm = pmask + (size_t)b * Hi * Wi;
    // valid rows / columns of this sample: counted once per block (threads 0..gh-1 and 64..64+gw-1), not per element
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < gh && pm[(size_t)t * P * Wi] != 0) atomicAdd(&cnt[0], 1);
    if (t >= 64 && t - 64 < gw && pm[(size_t)(t - 64) * P] != 0) atomicAdd(&cnt[1], 1);
    __syncthreads();
    const int vh = cnt[0], vw = cnt[1];
    if (i >= (long)gh * gw * H) return;
    const int c = (int)(i % H);
    const int x = (int)((i / H) % gw);
    const int y = (int)(i / ((long)H * gw));
    float v = 0.f;
    if (y < vh && x < vw) {
        const float sy = vh > 1 ? ((float)(g - 1) / (float)(vh - 1)) * (float)y : 0.f;
        const float sx = vw > 1 ? ((float)(g - 1) / (float)(vw - 1)) * (float)x : 0.f;
        const int y0 = min((int)sy, g - 1), x0 = min((int)sx, g - 1);
        const int y1 = min(y0 + 1, g - 1), x1 = min(x0 + 1, g - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float v00 = grid[((size_t)y0 * g + x0) * H + c], v01 = grid[((size_t)y0 * g + x1) * H + c];
        const float v10 = grid[((size_t)y1 * g + x0) * H + c], v11 = grid[((size_t)y1 * g + x1) * H + c];
        v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
    out[(size_t)b * gh * gw * H + i] = v;
}

// image_assemble_kernel + pos_resize_masked_kernel + key_mask_kernel in ONE launch (round 4): the per-sample resized position
// grid is interpolated where it is consumed (same expressions in the same order: bit-identical hidden states) instead of
// being written to a [B, np, H] buffer and read back (2 x 14 MB at configs[1]), and block (0, b) also writes sample b's
// attention key mask.  pmask is the pixel mask sampled at the patch origins: [B, gh, gw].
__global__ __launch_bounds__(256) void image_assemble_masked_kernel(const float* __restrict__ proj,
                                                                    const float* __restrict__ cls,
                                                                    const float* __restrict__ pos0,
                                                                    const float* __restrict__ grid,
                                                                    const long* __restrict__ pmask,
                                                                    const long* __restrict__ amask,
                                                                    const float* __restrict__ mod1, float* __restrict__ h,
                                                                    uint8_t* __restrict__ key_mask, int B, int Lt, int gh,
                                                                    int gw, int g, int S, int H, int nrep) {
    const int np = gh * gw, nc = H >> 2;
    const int b = blockIdx.y;
    const long* pm = pmask + (size_t)b * np;
    __shared__ int cnt[2];
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int t_ = threadIdx.x;
    if (t_ < gh && pm[(size_t)t_ * gw] != 0) atomicAdd(&cnt[0], 1);
    if (t_ >= 64 && t_ - 64 < gw && pm[t_ - 64] != 0) atomicAdd(&cnt[1], 1);
    __syncthreads();
    const int vh = cnt[0], vw = cnt[1];
    if (blockIdx.x == 0 && key_mask) {
        for (int t = threadIdx.x; t < S; t += 256) {
            uint8_t v = 1;
            if (t < Lt) {
                if (amask) v = amask[(size_t)b * Lt + t] != 0;
            } else if (t > Lt) {
                v = pm[t - Lt - 1] != 0;
            }
            for (int r = 0; r < nrep; ++r) key_mask[(size_t)(b + r * B) * S + t] = v;
        }
    }
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)(np + 1) * nc) return;
    const int c = (int)(i % nc);
    const int t = (int)(i / nc);
    const f32x4 m4 = *reinterpret_cast<const f32x4*>(mod1 + c * 4);
    f32x4 o;
    if (t == 0) {
        o = (*reinterpret_cast<const f32x4*>(cls + c * 4) + *reinterpret_cast<const f32x4*>(pos0 + c * 4)) + m4;
    } else {
        const int p = t - 1;
        const int y = p / gw, x = p - y * gw;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (y < vh && x < vw) {
            const float sy = vh > 1 ? ((float)(g - 1) / (float)(vh - 1)) * (float)y : 0.f;
            const float sx = vw > 1 ? ((float)(g - 1) / (float)(vw - 1)) * (float)x : 0.f;
            const int y0 = min((int)sy, g - 1), x0 = min((int)sx, g - 1);
            const int y1 = min(y0 + 1, g - 1), x1 = min(x0 + 1, g - 1);
            const float ly = sy - (float)y0, lx = sx - (float)x0;
            const f32x4 v00 = *reinterpret_cast<const f32x4*>(grid + ((size_t)y0 * g + x0) * H + c * 4);
            const f32x4 v01 = *reinterpret_cast<const f32x4*>(grid + ((size_t)y0 * g + x1) * H + c * 4);
            const f32x4 v10 = *reinterpret_cast<const f32x4*>(grid + ((size_t)y1 * g + x0) * H + c * 4);
            const f32x4 v11 = *reinterpret_cast<const f32x4*>(grid + ((size_t)y1 * g + x1) * H + c * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = (1.f - ly) * ((1.f - lx) * v00[e] + lx * v01[e]) + ly * ((1.f - lx) * v10[e] + lx * v11[e]);
        }
        o = (*reinterpret_cast<const f32x4*>(proj + ((size_t)b * np + p) * H + c * 4) + v) + m4;
    }
    *reinterpret_cast<f32x4*>(h + ((size_t)b * S + Lt + t) * H + c * 4) = o;
}

// attention key mask of the [text | CLS | patches] sequence, written nrep times (rows b + rep * B)
__global__ __launch_bounds__(256) void key_mask_kernel(const long* __restrict__ amask, const long* __restrict__ pmask,
                                                       uint8_t* __restrict__ out, int B, int Lt, int Hi, int Wi, int P,
                                                       int S, int nrep) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * S) return;
    const int b = i / S, t = i - b * S;
    const int gw = Wi / P;
    uint8_t v = 1;
    if (t < Lt) {
        if (amask) v = amask[(size_t)b * Lt + t] != 0;
    } else if (t > Lt && pmask) {
        const int p = t - Lt - 1;
        v = pmask[(size_t)b * Hi * Wi + (size_t)(p / gw) * P * Wi + (size_t)(p % gw) * P] != 0;
    }
    for (int r = 0; r < nrep; ++r) out[(size_t)(b + r * B) * S + t] = v;
}

__global__ __launch_bounds__(256) void cvt_kernel(const float* __restrict__ in, bf16* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        *reinterpret_cast<bf16x4*>(out + i) = cvt4(*reinterpret_cast<const f32x4*>(in + i));
    } else {
        for (long k = i; k < n; ++k) out[k] = (bf16)in[k];
    }
}

__global__ __launch_bounds__(256) void transpose_cvt_kernel(const float* __restrict__ in, bf16* __restrict__ out,
                                                            int R, int C) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < C && r < R) out[(size_t)c * R + r] = (bf16)tile[tx][k];
    }
}

__global__ __launch_bounds__(256) void tanh_fwd_kernel(float* x, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = tanhf(x[i]);
}
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* y, const float* dy, float* dx, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = dy[i] * (1.f - y[i] * y[i]);
}
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* x, float* y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = gelu_f(x[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* x, const float* dy, float* dx, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dx[i] = dy[i] * gelu_grad_f(x[i]);
}

__global__ __launch_bounds__(256) void scatter_cls_kernel(const float* __restrict__ rows, float* __restrict__ o32,
                                                          bf16* __restrict__ o16, int B, int S, int H) {
    const int nc = H >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * S * nc) return;
    const int c = (int)(i % nc);
    const long tok = i / nc;
    const int s = (int)(tok % S);
    const int b = (int)(tok / S);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s == 0) v = *reinterpret_cast<const f32x4*>(rows + (size_t)b * H + c * 4);
    if (o32) *reinterpret_cast<f32x4*>(o32 + (size_t)tok * H + c * 4) = v;
    if (o16) *reinterpret_cast<bf16x4*>(o16 + (size_t)tok * H + c * 4) = cvt4(v);
}

}  // namespace

extern "C" int feddat_abi_version(void) { return FEDDAT_ABI_VERSION; }
extern "C" int feddat_operand_format(void) { return FD_OPERAND_FORMAT; }

extern "C" int feddat_text_embed(const int64_t* input_ids, const int64_t* token_type_ids, const float* word,
                                 const float* pos, const float* type, const float* ln_g, const float* ln_b, float eps,
                                 const float* modality0, float* h, int B, int Lt, int S, int H, hipStream_t stream) {
    FD_CHECK_ARG(input_ids && token_type_ids && word && pos && type && ln_g && ln_b && modality0 && h);
    FD_CHECK_ARG(B > 0 && Lt > 0 && S >= Lt && H % 4 == 0 && H <= 2048);
    hipLaunchKernelGGL(text_embed_kernel, dim3((B * Lt + 3) / 4), dim3(256), 0, stream, input_ids, token_type_ids,
                       word, pos, type, ln_g, ln_b, eps, modality0, h, B, Lt, S, H);
    FD_LAUNCH_RET();
}

extern "C" int feddat_im2col_patches(const float* pixels, void* patches_bf16, int B, int C, int Hi, int Wi, int P,
                                     hipStream_t stream) {
    FD_CHECK_ARG(pixels && patches_bf16 && B > 0 && C > 0 && Hi > 0 && Wi > 0 && P > 0);
    FD_CHECK_ARG(Hi % P == 0 && Wi % P == 0 && P % 4 == 0);
    const long total = (long)B * C * Hi * Wi / 4;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pixels,
                       (bf16*)patches_bf16, B, C, Hi, Wi, P);
    FD_LAUNCH_RET();
}

extern "C" int feddat_image_embed_assemble(const float* proj, const float* cls, const float* pos0,
                                           const float* pos_img, long pos_batch_stride, const float* modality1,
                                           float* h, int B, int Lt, int np, int S, int H, hipStream_t stream) {
    FD_CHECK_ARG(proj && cls && pos0 && pos_img && modality1 && h && B > 0 && np > 0 && S == Lt + 1 + np);
    FD_CHECK_ARG(pos_batch_stride == 0 || pos_batch_stride >= (long)np * H);
    FD_CHECK_ARG(H % 4 == 0);
    const long total = (long)B * (np + 1) * (H / 4);
    hipLaunchKernelGGL(image_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, proj, cls,
                       pos0, pos_img, modality1, h, B, Lt, np, S, H, pos_batch_stride);
    FD_LAUNCH_RET();
}

extern "C" int feddat_image_embed_assemble_masked(const float* proj, const float* cls, const float* pos0,
                                                  const float* pos_grid, const long* patch_mask,
                                                  const long* attention_mask, const float* modality1, float* h,
                                                  uint8_t* key_mask, int B, int Lt, int gh, int gw, int g, int H, int nrep,
                                                  hipStream_t stream) {
    FD_CHECK_ARG(proj && cls && pos0 && pos_grid && patch_mask && modality1 && h && B > 0 && B <= 65535 && Lt >= 0);
    FD_CHECK_ARG(gh > 0 && gw > 0 && gh <= 64 && gw <= 64 && g > 0 && H % 4 == 0 && nrep >= 1);
    const int np = gh * gw, S = Lt + 1 + np;
    const long per = (long)(np + 1) * (H / 4);
    hipLaunchKernelGGL(image_assemble_masked_kernel, dim3((unsigned)((per + 255) / 256), B), dim3(256), 0, stream, proj, cls,
                       pos0, pos_grid, patch_mask, attention_mask, modality1, h, key_mask, B, Lt, gh, gw, g, S, H, nrep);
    FD_LAUNCH_RET();
}

extern "C" int feddat_pos_embed_resize(const float* pos_grid, float* out, int g, int gh, int gw, int H,
                                       hipStream_t stream) {
    FD_CHECK_ARG(pos_grid && out && g > 0 && gh > 0 && gw > 0 && H > 0);
    const long total = (long)gh * gw * H;
    hipLaunchKernelGGL(pos_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, pos_grid, out,
                       g, gh, gw, H);
    FD_LAUNCH_RET();
}

extern "C" int feddat_pos_embed_resize_masked(const float* pos_grid, const long* pixel_mask, float* out, int g, int B,
                                              int Hi, int Wi, int P, int H, hipStream_t stream) {
    FD_CHECK_ARG(pos_grid && pixel_mask && out && g > 0 && B > 0 && Hi > 0 && Wi > 0 && P > 0 && H > 0);
    FD_CHECK_ARG(Hi % P == 0 && Wi % P == 0 && B <= 65535 && Hi / P <= 64 && Wi / P <= 64);
    const long total = (long)(Hi / P) * (Wi / P) * H;
    hipLaunchKernelGGL(pos_resize_masked_kernel, dim3((unsigned)((total + 255) / 256), B), dim3(256), 0, stream,
                       pos_grid, pixel_mask, out, g, Hi, Wi, P, H);
    FD_LAUNCH_RET();
}

extern "C" int feddat_vilt_key_mask(const long* attention_mask, const long* pixel_mask, uint8_t* key_mask, int B, int Lt,
                                    int Hi, int Wi, int P, int nrep, hipStream_t stream) {
    FD_CHECK_ARG(key_mask && B > 0 && Lt >= 0 && Hi > 0 && Wi > 0 && P > 0 && Hi % P == 0 && Wi % P == 0 && nrep >= 1);
    const int S = Lt + 1 + (Hi / P) * (Wi / P);
    hipLaunchKernelGGL(key_mask_kernel, dim3((B * S + 255) / 256), dim3(256), 0, stream, attention_mask, pixel_mask,
                       key_mask, B, Lt, Hi, Wi, P, S, nrep);
    FD_LAUNCH_RET();
}

// The small per-batch inputs of a ViLT step -> the engine's static buffers in ONE launch (five device-to-device copies of a
// few KB each cost 4-5 us apiece in front of every step).
struct StageInputs {
    const long *ids, *types, *amask, *pmask;
    const float* target;
    long *d_ids, *d_types, *d_amask, *d_pmask;
    float* d_target;
    int n_text, n_target, B, gh, gw, Hi, Wi, P;
};
__global__ __launch_bounds__(256) void stage_inputs_kernel(StageInputs a) {
    const int n_patch = a.B * a.gh * a.gw;
    const int n = max(max(a.n_text, a.n_target), n_patch);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (i < a.n_text) {
            a.d_ids[i] = a.ids[i];
            a.d_types[i] = a.types[i];
            a.d_amask[i] = a.amask ? a.amask[i] : 1;
        }
        if (i < a.n_target && a.target) a.d_target[i] = a.target[i];
        if (i < n_patch) {
            const int b = i / (a.gh * a.gw), r = i - b * (a.gh * a.gw), y = r / a.gw, x = r - y * a.gw;
            a.d_pmask[i] = a.pmask ? a.pmask[((size_t)b * a.Hi + (size_t)y * a.P) * a.Wi + (size_t)x * a.P] : 1;
        }
    }
}

extern "C" int feddat_vilt_stage_inputs(const long* input_ids, const long* token_type_ids, const long* attention_mask,
                                        const float* target, const long* pixel_mask, long* d_input_ids,
                                        long* d_token_type_ids, long* d_attention_mask, float* d_target, long* d_patch_mask,
                                        int B, int Lt, int n_labels, int Hi, int Wi, int P, hipStream_t stream) {
    FD_CHECK_ARG(input_ids && token_type_ids && d_input_ids && d_token_type_ids && d_attention_mask && d_patch_mask);
    FD_CHECK_ARG(B > 0 && Lt > 0 && n_labels >= 0 && Hi > 0 && Wi > 0 && P > 0 && Hi % P == 0 && Wi % P == 0 && (!target || d_target));
    StageInputs a{input_ids, token_type_ids, attention_mask, pixel_mask, target, d_input_ids, d_token_type_ids,
                  d_attention_mask, d_patch_mask, d_target, B * Lt, target ? B * n_labels : 0, B, Hi / P, Wi / P, Hi, Wi, P};
    const int n = max(max(a.n_text, a.n_target), B * a.gh * a.gw);
    hipLaunchKernelGGL(stage_inputs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a);
    FD_LAUNCH_RET();
}

extern "C" int feddat_cvt_f32_bf16(const float* in, void* out_bf16, long n, hipStream_t stream) {
    FD_CHECK_ARG(in && out_bf16 && n > 0);
    hipLaunchKernelGGL(cvt_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, stream, in, (bf16*)out_bf16,
                       n);
    FD_LAUNCH_RET();
}

extern "C" int feddat_transpose_f32_bf16(const float* in, void* out_bf16, int R, int C, hipStream_t stream) {
    FD_CHECK_ARG(in && out_bf16 && R > 0 && C > 0);
    hipLaunchKernelGGL(transpose_cvt_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, stream, in,
                       (bf16*)out_bf16, R, C);
    FD_LAUNCH_RET();
}

#define FD_EW(name, kern, ...)                                                                                   \
    hipLaunchKernelGGL(kern, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, __VA_ARGS__);              \
    FD_LAUNCH_RET();

extern "C" int feddat_tanh_fwd(float* x, long n, hipStream_t stream) {
    FD_CHECK_ARG(x && n > 0);
    FD_EW(tanh, tanh_fwd_kernel, x, n)
}
extern "C" int feddat_tanh_bwd(const float* y, const float* dy, float* dx, long n, hipStream_t stream) {
    FD_CHECK_ARG(y && dy && dx && n > 0);
    FD_EW(tanhb, tanh_bwd_kernel, y, dy, dx, n)
}
extern "C" int feddat_gelu_fwd(const float* x, float* y, long n, hipStream_t stream) {
    FD_CHECK_ARG(x && y && n > 0);
    FD_EW(gelu, gelu_fwd_kernel, x, y, n)
}
extern "C" int feddat_gelu_bwd(const float* x, const float* dy, float* dx, long n, hipStream_t stream) {
    FD_CHECK_ARG(x && dy && dx && n > 0);
    FD_EW(gelub, gelu_bwd_kernel, x, dy, dx, n)
}

extern "C" int feddat_scatter_cls_rows(const float* rows, float* out_f32, void* out_bf16, int B, int S, int H,
                                       hipStream_t stream) {
    FD_CHECK_ARG(rows && (out_f32 || out_bf16) && B > 0 && S > 0 && H % 4 == 0);
    const long n = (long)B * S * (H / 4);
    hipLaunchKernelGGL(scatter_cls_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rows, out_f32,
                       (bf16*)out_bf16, B, S, H);
    FD_LAUNCH_RET();
}
