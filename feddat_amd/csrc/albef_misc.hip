// Small kernels of the ALBEF path (configs[3]): the LM-head token loss with vocabulary-axis MKD, row gather / segment sum
// (answers <-> questions), and a 3-operand element-wise combine used around the BERT double-LayerNorm adapter variant.
#include "common.hip.h"

namespace {

__global__ __launch_bounds__(256) void axpby3_kernel(const float* __restrict__ a, float alpha, const float* __restrict__ b,
                                                     float beta, const float* __restrict__ c, float gamma,
                                                     float* __restrict__ out, bf16* __restrict__ out16, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = reinterpret_cast<const f32x4*>(a)[i] * f32x4{alpha, alpha, alpha, alpha};
    if (b) v = v + reinterpret_cast<const f32x4*>(b)[i] * f32x4{beta, beta, beta, beta};
    if (c) v = v + reinterpret_cast<const f32x4*>(c)[i] * f32x4{gamma, gamma, gamma, gamma};
    if (out) reinterpret_cast<f32x4*>(out)[i] = v;
    if (out16) reinterpret_cast<bf16x4*>(out16)[i] = cvt4(v);
}

// out = keep(idx) ? x * 1/(1-p) : 0  (+ resid): nn.Dropout in train mode on a [rows, cols] tensor (BertEmbeddings
// xbert.py:216, BertSelfOutput :360, BertOutput :440 -- there followed by the residual add of the fused forms), and the same
// mask applied to a gradient in the backward.  Input fp32 or bf16, outputs fp32 and / or bf16.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x32, const bf16* __restrict__ x16,
                                                      const float* __restrict__ resid, float* __restrict__ out,
                                                      bf16* __restrict__ out16, long n4, float p, uint32_t key0,
                                                      uint32_t key1, const int* __restrict__ step) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const FdDrop d = fd_drop_make(p, key0, key1, step);
    f32x4 v;
    if (x32) v = reinterpret_cast<const f32x4*>(x32)[i];
    else {
        const bf16x4 h = reinterpret_cast<const bf16x4*>(x16)[i];
        v = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fd_drop_keep(d, (uint32_t)(4 * i + e)) ? v[e] * d.scale : 0.f;
    if (resid) v = v + reinterpret_cast<const f32x4*>(resid)[i];
    if (out) reinterpret_cast<f32x4*>(out)[i] = v;
    if (out16) reinterpret_cast<bf16x4*>(out16)[i] = cvt4(v);
}

// dst[r] = src[idx[r]] (fp32 rows of `width` floats, width % 4 == 0; idx < 0 -> a zero row), optional bf16 copy
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          float* __restrict__ dst, bf16* __restrict__ dst16, int width) {
    const int r = blockIdx.x;
    const int j = idx[r];
    const float* s = src + (size_t)(j < 0 ? 0 : j) * width;
    for (int c = threadIdx.x * 4; c < width; c += 1024) {
        const f32x4 v = j < 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(s + c);
        if (dst) *reinterpret_cast<f32x4*>(dst + (size_t)r * width + c) = v;
        if (dst16) *reinterpret_cast<bf16x4*>(dst16 + (size_t)r * width + c) = cvt4(v);
    }
}

// dst[s] = (accumulate ? dst[s] : 0) + sum_{j in [off[s], off[s+1])} src[j]   (rows in index order: deterministic)
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const float* __restrict__ src, const int* __restrict__ off,
                                                               float* __restrict__ dst, int width, int accumulate) {
    const int s = blockIdx.x;
    const int j0 = off[s], j1 = off[s + 1];
    for (int c = threadIdx.x * 4; c < width; c += 1024) {
        f32x4 acc = accumulate ? *reinterpret_cast<const f32x4*>(dst + (size_t)s * width + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = j0; j < j1; ++j) acc = acc + *reinterpret_cast<const f32x4*>(src + (size_t)j * width + c);
        *reinterpret_cast<f32x4*>(dst + (size_t)s * width + c) = acc;
    }
}

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// One block per logits row (one answer token position).  ce = logsumexp(l) - l[label] (0 for label < 0);
// kl = sum_v q_v (log q_v - log p_v), p = softmax(l / T), q = softmax(teacher / T);
// dlogits = 0.5 * ( row_w * (softmax(l) - onehot) [label >= 0]  +  kl_scale / T * (p - q) ), stored as bf16 (the operand
// of the LM-head backward GEMM), zeros in the padding columns [V, ldd).
// 1024 threads per row, 16-byte loads (rows are 16-byte aligned: ldl % 4 == 0 is checked by the launcher): with 256 threads
// and scalar loads the three passes over 30 522 columns were 360 dependent load -> exp iterations per thread (206 us for
// the 96 rows of the bench configuration, on the critical path between the forward and the backward pass).
constexpr int LM_NT = 1024;
__device__ __forceinline__ float block_max16(float v, float* red) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < LM_NT / 64; ++i) r = fmaxf(r, red[i]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum16(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < LM_NT / 64; ++i) r += red[i];
    __syncthreads();
    return r;
}
__global__ __launch_bounds__(LM_NT) void lm_loss_kernel(const float* __restrict__ logits, const float* __restrict__ teacher,
                                                        long ldl, const long* __restrict__ labels,
                                                        const float* __restrict__ row_w, const float* __restrict__ row_kl,
                                                        int V, float T, float kl_scale, bf16* __restrict__ dlogits, long ldd,
                                                        float* __restrict__ row_terms, float grad_scale,
                                                        const float* __restrict__ grad_scale_dev) {
    __shared__ float red[LM_NT / 64];
    if (grad_scale_dev) grad_scale *= *grad_scale_dev;      // the dynamic loss scale (a power of two: exact)
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* l = logits + (size_t)r * ldl;
    const float* t = teacher ? teacher + (size_t)r * ldl : nullptr;
    const float invT = 1.0f / T;
    const int V4 = V & ~3;                      // whole 4-column groups; columns [V4, V) go to threads 0 .. V - V4 - 1
    const int vt = V4 + tid;                    // this thread's tail column (if < V)
    float m1 = -INFINITY, mt = -INFINITY;
    for (int v = 4 * tid; v < V4; v += 4 * LM_NT) {
        const f32x4 lv = *reinterpret_cast<const f32x4*>(l + v);
        m1 = fmaxf(fmaxf(m1, fmaxf(lv[0], lv[1])), fmaxf(lv[2], lv[3]));
        if (t) {
            const f32x4 tv = *reinterpret_cast<const f32x4*>(t + v);
            mt = fmaxf(fmaxf(mt, fmaxf(tv[0], tv[1])), fmaxf(tv[2], tv[3]));
        }
    }
    if (vt < V) {
        m1 = fmaxf(m1, l[vt]);
        if (t) mt = fmaxf(mt, t[vt]);
    }
    m1 = block_max16(m1, red);
    if (t) mt = block_max16(mt, red);
    float z1 = 0.f, zp = 0.f, zq = 0.f, a = 0.f;
    auto acc1 = [&](const float lv, const float tv) {
        const float d = lv - m1;
        z1 += __expf(d);
        if (t) {
            zp += __expf(d * invT);
            const float eq = __expf((tv - mt) * invT);
            zq += eq;
            a += eq * (tv - lv) * invT;
        }
    };
    for (int v = 4 * tid; v < V4; v += 4 * LM_NT) {
        const f32x4 lv = *reinterpret_cast<const f32x4*>(l + v);
        f32x4 tv = {0.f, 0.f, 0.f, 0.f};
        if (t) tv = *reinterpret_cast<const f32x4*>(t + v);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1(lv[e], tv[e]);
    }
    if (vt < V) acc1(l[vt], t ? t[vt] : 0.f);
    z1 = block_sum16(z1, red);
    float kl = 0.f;
    if (t) {
        zp = block_sum16(zp, red);
        zq = block_sum16(zq, red);
        a = block_sum16(a, red);
        kl = a / zq - (mt - m1) * invT - __logf(zq) + __logf(zp);
    }
    const long lab = labels[r];
    const float w = lab >= 0 ? row_w[r] : 0.f;
    const float ce = lab >= 0 ? (m1 + __logf(z1) - l[lab]) : 0.f;
    // row_kl: per-row factor on the MKD term (rows that exist only because the batch was padded to the engine's frame carry
    // 0; the others N_frame / n_batch, which turns kl_scale = T^2 / N_frame into the reference's batchmean over n_batch)
    const float rk = row_kl ? row_kl[r] : 1.f;
    if (tid == 0) {
        row_terms[2 * r] = w * ce;
        row_terms[2 * r + 1] = kl * rk;
    }
    if (!dlogits) return;
    bf16* dl = dlogits + (size_t)r * ldd;
    const float i1 = 1.0f / z1, ip = t ? 1.0f / zp : 0.f, iq = t ? 1.0f / zq : 0.f;
    const float ks = kl_scale * rk * invT;
    auto grad1 = [&](const int v, const float lv, const float tv) {
        const float d = lv - m1;
        float gv = w * (__expf(d) * i1 - (v == lab ? 1.f : 0.f));
        if (t) gv += ks * (__expf(d * invT) * ip - __expf((tv - mt) * invT) * iq);
        return (0.5f * grad_scale) * gv;      // grad_scale: the caller's power-of-two loss scale (fp16 operand build), else 1
    };
    for (int v = 4 * tid; v < V4; v += 4 * LM_NT) {          // (ldd % 4 == 0 and dlogits 8-byte aligned: launcher)
        const f32x4 lv = *reinterpret_cast<const f32x4*>(l + v);
        f32x4 tv = {0.f, 0.f, 0.f, 0.f};
        if (t) tv = *reinterpret_cast<const f32x4*>(t + v);
        f32x4 gv;
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[e] = grad1(v + e, lv[e], tv[e]);
        *reinterpret_cast<bf16x4*>(dl + v) = cvt4(gv);
    }
    for (int v = V4 + tid; v < (int)ldd; v += LM_NT)         // the last columns of the vocabulary, zeros in the padding
        dl[v] = (bf16)(v < V ? grad1(v, l[v], t ? t[v] : 0.f) : 0.f);
}

// scalars[0] = sum_r w_r ce_r (the loss ALBEF.forward returns), [1] = kl_scale * sum_r kl_r, [2] = ([0] + [1]) / 2
__global__ void lm_loss_finish(const float* __restrict__ row_terms, int R, float kl_scale, float* __restrict__ scalars,
                               int* __restrict__ nonfinite) {
    float ce = 0.f, kl = 0.f;
    for (int r = threadIdx.x; r < R; r += 64) {
        ce += row_terms[2 * r];
        kl += row_terms[2 * r + 1];
    }
    ce = wave_sum(ce);
    kl = wave_sum(kl);
    if (threadIdx.x == 0) {
        scalars[0] = ce;
        scalars[1] = kl * kl_scale;
        scalars[2] = 0.5f * (ce + kl * kl_scale);
        if (nonfinite && !(fabsf(ce + kl * kl_scale) <= 3.4e38f)) atomicOr(nonfinite, 1);      // GradScaler's inf check at its source
    }
}

// rank_answer, step 1 (albef_model.py:183-186): prob_first[b, j] = softmax(logits[b, :V])[first_ids[j]].  One block per
// question: log-sum-exp of the row over the vocabulary, then the gather.
__global__ __launch_bounds__(256) void softmax_gather_kernel(const float* __restrict__ logits, long row_stride, int V,
                                                             const long* __restrict__ ids, long id_stride, int n,
                                                             float* __restrict__ out) {
    __shared__ float red[8];
    const float* row = logits + (size_t)blockIdx.x * row_stride;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += 256) m = fmaxf(m, row[c]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = threadIdx.x; c < V; c += 256) sum += __expf(row[c] - m);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    const float lse = m + __logf(red[4] + red[5] + red[6] + red[7]);
    for (int j = threadIdx.x; j < n; j += 256) {
        const long id = ids[(size_t)j * id_stride];
        out[(size_t)blockIdx.x * n + j] = id >= 0 && id < V ? __expf(row[id] - lse) : 0.f;
    }
}

// rank_answer, steps 2 and 4 (albef_model.py:186,223-226): the k largest of a row of n values, sorted descending (equal
// values: lower index first).  Optional transforms ahead of the sort: flag 1: v = log(v); minus: v -= minus[row, j];
// flag 2: v = softmax over the row.  One block per row, bitonic sort of (value, index) pairs in LDS (n padded to a
// power of two with -inf).
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ vals, long ld, const float* __restrict__ minus,
                                                        int n, int n2, int k, int flags, float* __restrict__ out_vals,
                                                        long* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* v = reinterpret_cast<float*>(smem);
    int* ix = reinterpret_cast<int*>(smem + (size_t)n2 * 4);
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x;
    float m = -INFINITY;
    for (int j = tid; j < n2; j += 256) {
        float x = -INFINITY;
        if (j < n) {
            x = vals[(size_t)row * ld + j];
            if (flags & 1) x = __logf(x);
            if (minus) x -= minus[(size_t)row * n + j];
        }
        v[j] = x;
        ix[j] = j;
        m = fmaxf(m, x);
    }
    if (flags & 2) {
        m = wave_max(m);
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float sum = 0.f;
        for (int j = tid; j < n; j += 256) {
            const float e = __expf(v[j] - m);
            v[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
        __syncthreads();
        const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
        for (int j = tid; j < n; j += 256) v[j] *= inv;
    }
    __syncthreads();
    // "a before b": larger value first, ties by lower index
    auto before = [](float va, int ia, float vb, int ib) { return va > vb || (va == vb && ia < ib); };
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += 256) {
                const int lo = ((t / stride) * stride << 1) + (t % stride), hi = lo + stride;
                const bool desc = ((lo & size) == 0);        // this run sorts "before" first
                const float va = v[lo], vb = v[hi];
                const int ia = ix[lo], ib = ix[hi];
                const bool swap = desc ? before(vb, ib, va, ia) : before(va, ia, vb, ib);
                if (swap) {
                    v[lo] = vb; v[hi] = va;
                    ix[lo] = ib; ix[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < k; j += 256) {
        out_vals[(size_t)row * k + j] = v[j];
        out_idx[(size_t)row * k + j] = ix[j];
    }
}

}  // namespace

extern "C" int feddat_softmax_gather_rows(const float* logits, long row_stride, int rows, int V, const long* ids,
                                          long id_stride, int n, float* out, hipStream_t stream) {
    FD_CHECK_ARG(logits && ids && out && rows > 0 && V > 0 && n > 0 && row_stride >= V && id_stride >= 1);
    hipLaunchKernelGGL(softmax_gather_kernel, dim3(rows), dim3(256), 0, stream, logits, row_stride, V, ids, id_stride, n, out);
    FD_LAUNCH_RET();
}

extern "C" int feddat_topk_rows(const float* vals, long ld, const float* minus, int rows, int n, int k, int flags,
                                float* out_vals, long* out_idx, hipStream_t stream) {
    FD_CHECK_ARG(vals && out_vals && out_idx && rows > 0 && n > 0 && k > 0 && k <= n && ld >= n && n <= 8192 && !(flags & ~3));
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    const int lds = n2 * 8;
    if (lds > 48 * 1024 && fd_set_max_lds((const void*)topk_rows_kernel, lds) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(256), lds, stream, vals, ld, minus, n, n2, k, flags, out_vals, out_idx);
    FD_LAUNCH_RET();
}

extern "C" int feddat_axpby3(const float* a, float alpha, const float* b, float beta, const float* c, float gamma,
                             float* out_f32, void* out_bf16, long n, hipStream_t stream) {
    FD_CHECK_ARG(a && (out_f32 || out_bf16) && n > 0 && n % 4 == 0);
    const long n4 = n / 4;
    hipLaunchKernelGGL(axpby3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a, alpha, b, beta, c,
                       gamma, out_f32, (bf16*)out_bf16, n4);
    FD_LAUNCH_RET();
}

extern "C" int feddat_dropout(const float* x_f32, const void* x_bf16, const float* resid, float* out_f32, void* out_bf16,
                              long n, float p, unsigned key0, unsigned key1, const int* step_ctr, hipStream_t stream) {
    FD_CHECK_ARG((x_f32 != nullptr) != (x_bf16 != nullptr) && (out_f32 || out_bf16) && n > 0 && n % 4 == 0);
    FD_CHECK_ARG(p >= 0.f && p < 1.f && n < (1L << 32));
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, x_f32,
                       (const bf16*)x_bf16, resid, out_f32, (bf16*)out_bf16, n / 4, p, key0, key1, step_ctr);
    FD_LAUNCH_RET();
}

extern "C" int feddat_gather_rows(const float* src, const int* idx, float* dst_f32, void* dst_bf16, int rows, int width,
                                  hipStream_t stream) {
    FD_CHECK_ARG(src && idx && (dst_f32 || dst_bf16) && rows > 0 && width > 0 && width % 4 == 0);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, stream, src, idx, dst_f32, (bf16*)dst_bf16, width);
    FD_LAUNCH_RET();
}

extern "C" int feddat_segment_sum_rows(const float* src, const int* seg_offsets, float* dst, int nseg, int width,
                                       int accumulate, hipStream_t stream) {
    FD_CHECK_ARG(src && seg_offsets && dst && nseg > 0 && width > 0 && width % 4 == 0);
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3(nseg), dim3(256), 0, stream, src, seg_offsets, dst, width, accumulate);
    FD_LAUNCH_RET();
}

extern "C" int feddat_lm_loss_fwd_bwd(const float* logits, const float* teacher, long ldl, const long* labels,
                                      const float* row_weight, const float* row_kl, int R, int V, float temp, float kl_scale,
                                      float grad_scale, void* dlogits_bf16, long ldd, float* scalars, hipStream_t stream) {
    FD_CHECK_ARG(logits && labels && row_weight && scalars && R > 0 && V > 0 && ldl >= V && temp > 0.f && grad_scale > 0.f);
    FD_CHECK_ARG(!dlogits_bf16 || ldd >= V);
    float* row_terms = scalars + 4;       // scalars: 4 + 2 R floats
    FD_CHECK_ARG(ldl % 4 == 0 && ((uintptr_t)logits & 15) == 0 && (!teacher || ((uintptr_t)teacher & 15) == 0));
    FD_CHECK_ARG(!dlogits_bf16 || (ldd % 4 == 0 && ((uintptr_t)dlogits_bf16 & 7) == 0));
    hipLaunchKernelGGL(lm_loss_kernel, dim3(R), dim3(LM_NT), 0, stream, logits, teacher, ldl, labels, row_weight, row_kl, V, temp,
                       kl_scale, (bf16*)dlogits_bf16, ldd, row_terms, grad_scale, (const float*)nullptr);
    hipLaunchKernelGGL(lm_loss_finish, dim3(1), dim3(64), 0, stream, row_terms, R, kl_scale, scalars, (int*)nullptr);
    FD_LAUNCH_RET();
}

extern "C" int feddat_lm_loss_fwd_bwd_dyn(const float* logits, const float* teacher, long ldl, const long* labels,
                                          const float* row_weight, const float* row_kl, int R, int V, float temp, float kl_scale,
                                          float grad_scale, const float* grad_scale_dev, int* nonfinite, void* dlogits_bf16,
                                          long ldd, float* scalars, hipStream_t stream) {
    FD_CHECK_ARG(logits && labels && row_weight && scalars && R > 0 && V > 0 && ldl >= V && temp > 0.f && grad_scale > 0.f);
    FD_CHECK_ARG(!dlogits_bf16 || ldd >= V);
    float* row_terms = scalars + 4;       // scalars: 4 + 2 R floats
    FD_CHECK_ARG(ldl % 4 == 0 && ((uintptr_t)logits & 15) == 0 && (!teacher || ((uintptr_t)teacher & 15) == 0));
    FD_CHECK_ARG(!dlogits_bf16 || (ldd % 4 == 0 && ((uintptr_t)dlogits_bf16 & 7) == 0));
    hipLaunchKernelGGL(lm_loss_kernel, dim3(R), dim3(LM_NT), 0, stream, logits, teacher, ldl, labels, row_weight, row_kl, V, temp,
                       kl_scale, (bf16*)dlogits_bf16, ldd, row_terms, grad_scale, grad_scale_dev);
    hipLaunchKernelGGL(lm_loss_finish, dim3(1), dim3(64), 0, stream, row_terms, R, kl_scale, scalars, nonfinite);
    FD_LAUNCH_RET();
}
