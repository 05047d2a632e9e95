// K5 (DAT loss forward + dlogits), K6 (flat multi-tensor AdamW + device-side LR schedule), K7 (FedAvg
// accumulate).  All fp32, HBM/latency-bound, tiny.
#include "common.hip.h"

namespace {

// One wave per sample row.  L = (mean_bce * C + T^2 * KL_batchmean) / 2
// (src/train/visionlanguage_tasks/task_trainer.py:299-301, 506-516).
__global__ __launch_bounds__(64) void dat_loss_kernel(const float* __restrict__ logits,
                                                      const float* __restrict__ teacher,
                                                      const float* __restrict__ target, int B, int C, float temp,
                                                      float* __restrict__ dlogits, float* __restrict__ row_terms) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* x = logits + (size_t)b * C;
    const float* t = teacher + (size_t)b * C;
    const float* y = target + (size_t)b * C;
    const float it = 1.0f / temp;
    float mx = -INFINITY, mt = -INFINITY;
    for (int j = lane; j < C; j += 64) {
        mx = fmaxf(mx, x[j] * it);
        mt = fmaxf(mt, t[j] * it);
    }
    mx = wave_max(mx);
    mt = wave_max(mt);
    float sx = 0.f, st = 0.f;
    for (int j = lane; j < C; j += 64) {
        sx += expf(x[j] * it - mx);
        st += expf(t[j] * it - mt);
    }
    sx = wave_sum(sx);
    st = wave_sum(st);
    const float lsx = mx + logf(sx), lst = mt + logf(st);
    float bce = 0.f, kl = 0.f;
    for (int j = lane; j < C; j += 64) {
        const float xv = x[j];
        // BCEWithLogits: max(x,0) - x*y + log(1 + exp(-|x|))
        bce += fmaxf(xv, 0.f) - xv * y[j] + log1pf(expf(-fabsf(xv)));
        const float logp = xv * it - lsx;
        const float logq = t[j] * it - lst;
        const float q = expf(logq);
        kl += q * (logq - logp);
        const float sig = 1.0f / (1.0f + expf(-xv));
        dlogits[(size_t)b * C + j] = 0.5f / (float)B * ((sig - y[j]) + temp * (expf(logp) - q));
    }
    bce = wave_sum(bce);
    kl = wave_sum(kl);
    if (lane == 0) {
        row_terms[2 * b] = bce;
        row_terms[2 * b + 1] = kl;
    }
}

__global__ __launch_bounds__(64) void dat_loss_finish(const float* __restrict__ row_terms, int B, float temp,
                                                      float* __restrict__ scalars) {
    float bce = 0.f, kl = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
        bce += row_terms[2 * b];
        kl += row_terms[2 * b + 1];
    }
    bce = wave_sum(bce);
    kl = wave_sum(kl);
    if (threadIdx.x == 0) {
        const float l_bce = bce / (float)B;               // mean over B*C, times C
        const float l_kl = kl / (float)B * temp * temp;   // batchmean * T^2
        scalars[0] = l_bce;
        scalars[1] = l_kl;
        scalars[2] = 0.5f * (l_bce + l_kl);
    }
}

// VQA score of a batch (train_vqa_crossvqa.py:241-257, task_trainer.py:125-157): acc[0] += sum_b target[b, argmax_j logits[b, j]],
// acc[1] += B.  One block, one wave per row at a time; argmax takes the FIRST maximal index (torch.argmax); the row scores
// are added in row order by lane 0 of wave 0 (deterministic).
__global__ __launch_bounds__(256) void vqa_score_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                        int B, int C, float* __restrict__ acc) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mine = 0.f;
    for (int b = wave; b < B; b += 4) {
        const float* x = logits + (size_t)b * C;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = lane; j < C; j += 64) {
            const float v = x[j];
            if (v > best || (v == best && j < bi)) { best = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const float ov = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (bi < C) mine += target[(size_t)b * C + bi];       // (wave-uniform; rows b, b+4, ... in order)
    }
    if (lane == 0) part[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        acc[0] += (part[0] + part[1]) + (part[2] + part[3]);
        acc[1] += (float)B;
    }
}

// HF get_polynomial_decay_schedule_with_warmup(lr_end=0, power=1) multiplier (task_trainer.py:53-59).
__device__ __forceinline__ float poly_lambda(int t, int warmup, int total) {
    if (t < warmup) return (float)t / (float)max(1, warmup);
    if (t > total) return 0.f;
    return 1.0f - (float)(t - warmup) / (float)(total - warmup);
}

// VEC = 4: every thread owns 4 consecutive parameters (one 16-byte access per array; the host checks that all segment
// offsets are multiples of 4 so a quad never straddles two weight-decay groups); VEC = 1: the general element-wise form.
// The per-element arithmetic is identical in both (bit-identical results).
template <int VEC>
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long n,
                                                         const long* __restrict__ seg_off,
                                                         const float* __restrict__ seg_wd, int nseg,
                                                         const int* __restrict__ state, float base_lr, int warmup,
                                                         int total, float beta1, float beta2, float eps) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= n) return;
    const int sched_t = state[0];
    const int t = state[1] + 1;
    const float lr = base_lr * poly_lambda(sched_t, warmup, total);
    // per-tensor weight decay: binary search of the segment table
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    float wd[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) wd[e] = seg_wd[lo];
    if (VEC > 1 && lo + 1 < nseg && seg_off[lo + 1] < i + VEC) {       // the quad straddles a tensor boundary (rare)
        for (int e = 1; e < VEC; ++e) {
            int s2 = lo;
            while (s2 + 1 < nseg && seg_off[s2 + 1] <= i + e) ++s2;
            wd[e] = seg_wd[s2];
        }
    }
    const float bc1 = 1.0f - powf(beta1, (float)t);
    const float bc2 = 1.0f - powf(beta2, (float)t);
    float gi[VEC], pi[VEC], mi[VEC], vi[VEC];
    if (VEC == 4) {
        *reinterpret_cast<f32x4*>(gi) = *reinterpret_cast<const f32x4*>(g + i);
        *reinterpret_cast<f32x4*>(pi) = *reinterpret_cast<const f32x4*>(p + i);
        *reinterpret_cast<f32x4*>(mi) = *reinterpret_cast<const f32x4*>(m + i);
        *reinterpret_cast<f32x4*>(vi) = *reinterpret_cast<const f32x4*>(v + i);
    } else {
        gi[0] = g[i]; pi[0] = p[i]; mi[0] = m[i]; vi[0] = v[i];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        pi[e] = pi[e] * (1.0f - lr * wd[e]);
        mi[e] = mi[e] + (gi[e] - mi[e]) * (1.0f - beta1);   // lerp_(grad, 1 - beta1)
        vi[e] = vi[e] * beta2 + (1.0f - beta2) * gi[e] * gi[e];
        const float denom = sqrtf(vi[e]) / sqrtf(bc2) + eps;
        pi[e] -= (lr / bc1) * (mi[e] / denom);
    }
    if (VEC == 4) {
        *reinterpret_cast<f32x4*>(p + i) = *reinterpret_cast<const f32x4*>(pi);
        *reinterpret_cast<f32x4*>(m + i) = *reinterpret_cast<const f32x4*>(mi);
        *reinterpret_cast<f32x4*>(v + i) = *reinterpret_cast<const f32x4*>(vi);
    } else {
        p[i] = pi[0]; m[i] = mi[0]; v[i] = vi[0];
    }
}

__global__ void step_tick_kernel(int* state, int d_sched, int d_adam) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[0] += d_sched;
        state[1] += d_adam;
    }
}

__global__ __launch_bounds__(256) void fedavg_acc_kernel(float* __restrict__ acc, const float* __restrict__ x, long n,
                                                         float num, float total, int first) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float term = x[i] * num / total;   // main.py:62: net[key] * num / total_num_train
    acc[i] = first ? (0.0f + term) : (acc[i] + term);
}

}  // namespace

extern "C" int feddat_dat_loss_fwd_bwd(const float* logits, const float* teacher, const float* target, int B, int C,
                                       float temp, float* dlogits, float* scalars, hipStream_t stream) {
    FD_CHECK_ARG(logits && teacher && target && dlogits && scalars && B > 0 && B <= 4096 && C > 0 && temp > 0.f);
    // row_terms live behind the 3 scalars: scalars must hold 4 + 2*B floats
    float* row_terms = scalars + 4;
    hipLaunchKernelGGL(dat_loss_kernel, dim3(B), dim3(64), 0, stream, logits, teacher, target, B, C, temp, dlogits,
                       row_terms);
    hipLaunchKernelGGL(dat_loss_finish, dim3(1), dim3(64), 0, stream, row_terms, B, temp, scalars);
    FD_LAUNCH_RET();
}

extern "C" int feddat_vqa_score_accumulate(const float* logits, const float* target, int B, int C, float* acc,
                                           hipStream_t stream) {
    FD_CHECK_ARG(logits && target && acc && B > 0 && C > 0);
    hipLaunchKernelGGL(vqa_score_kernel, dim3(1), dim3(256), 0, stream, logits, target, B, C, acc);
    FD_LAUNCH_RET();
}

extern "C" int feddat_adamw_flat(float* p, const float* g, float* m, float* v, long n, const long* seg_off,
                                 const float* seg_wd, int nseg, const int* state, float base_lr, int warmup, int total,
                                 float beta1, float beta2, float eps, hipStream_t stream) {
    FD_CHECK_ARG(p && g && m && v && n > 0 && seg_off && seg_wd && nseg > 0 && state && total > warmup);
    // quads of 4 consecutive parameters per thread when n % 4 == 0 and the buffers are 16-byte aligned (a quad that
    // straddles a tensor boundary looks its weight decays up per element)
    const bool quads = (n % 4 == 0) && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    if (quads)
        hipLaunchKernelGGL(adamw_flat_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, p, g, m, v, n,
                           seg_off, seg_wd, nseg, state, base_lr, warmup, total, beta1, beta2, eps);
    else
        hipLaunchKernelGGL(adamw_flat_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, n,
                           seg_off, seg_wd, nseg, state, base_lr, warmup, total, beta1, beta2, eps);
    FD_LAUNCH_RET();
}

extern "C" int feddat_step_tick(int* state, int d_sched, int d_adam, hipStream_t stream) {
    FD_CHECK_ARG(state);
    hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, stream, state, d_sched, d_adam);
    FD_LAUNCH_RET();
}

extern "C" int feddat_fedavg_accumulate(float* acc, const float* x, long n, float num, float total, int first,
                                        hipStream_t stream) {
    FD_CHECK_ARG(acc && x && n > 0 && total > 0.f);
    hipLaunchKernelGGL(fedavg_acc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, acc, x, n, num,
                       total, first);
    FD_LAUNCH_RET();
}
