// The serial tail of a DAT train_step: ViltModel.layernorm + ViltPooler on the token-0 rows, the task head
// (src/modeling/vilt.py:202-209: fc0 -> LayerNorm(1536, eps 1e-5) -> GELU -> fc1) forward and backward for the P0 / P1 / P2
// passes, the DAT loss, and the AdamW / scheduler bookkeeping (src/train/visionlanguage_tasks/task_trainer.py:280-330,
// 477-504).  All exact fp32, ~0.2 GFLOP in total: the cost is LAUNCHES (round 3: 48 kernels of 2-10 us behind each other, a
// fifth of the step's graph nodes), so this file fuses what used to be separate kernels:
//   * ht_gemm_kernel: the strided exact-fp32 MFMA product of sgemm_f32.hip with the K range split over the 8 waves of a
//     block (no split-K partials in HBM, no reduce launch), an optional A-side prologue (LayerNorm of the rows with the
//     statistics computed in the block; multiply by 1 - y^2 = tanh'), an optional epilogue (bias, tanh, multiply by
//     gelu'(aux)), the column sum that gives a weight-gradient job its bias gradient, and up to TWO independent products
//     per launch (e.g. dW_fc1 = dl^T g0 next to dn0 = (dl W_fc1) * gelu'(n0));
//   * ht_ln_gelu_kernel, ht_ln_bwd_full_kernel: the head's LayerNorm + GELU, and its full backward (dx, dgamma, dbeta) in one
//     launch each; dat_loss_single_kernel: loss + dlogits + the batch reduction in one block;
//   * adamw_multi_kernel / step_tick_multi_kernel: the step's last three AdamW launches as one, their four counter ticks
//     as one (an AdamW launch can read its schedule / step counters at an offset, so the head's second update does not
//     need a tick between the two).
#include "common.hip.h"

namespace {

constexpr int HT_JT = 4;       // 16-column tiles per wave sharing one A operand (block tile: 16 rows x 64 columns)
constexpr int HT_NW = 8;       // waves per block
constexpr int HT_U = 3;        // 16-deep k-chunks whose loads are in flight together per wave

struct HtJob {
    const float* A; long sa_i, sa_k;
    const float* B; long sb_k, sb_j;
    int I, J, K;
    int mode;                  // 0: the block's waves split K (summed through LDS in wave order); 1: the waves take
                               //    consecutive j-groups, each over the whole K (short contractions: K = batch)
    float alpha;
    const float* alpha_dev;    // optional device factor on alpha (the dynamic loss scale)
    const float* bias_j;
    float* out; long ldo;
    float* colsum;             // colsum[i] = alpha * sum_k A'[i][k]  (A' = A after the prologue)
    int pro;                   // FEDDAT_HT_PRO_*
    const float* pro_a;        // LN: gamma [K]           TANH_BWD: y, indexed like A
    const float* pro_b;        // LN: beta [K]
    float pro_eps;
    float* stats_out;          // LN: [I, 2] = {mean, rstd} (optional)
    int epi;                   // FEDDAT_HT_EPI_*
    const float* aux; long ld_aux;
    int itiles, jblocks;       // grid decomposition of this job
    int avec, bvec;            // operand contiguous along k and 16-byte aligned: float4 loads
    int jt;                    // 16-column tiles per wave (1 or HT_JT)
};
struct HtLaunch {
    HtJob job[2];
    int blocks0;               // blocks [0, blocks0) run job 0, the rest job 1
};

// Main loop of one wave: k runs in chunks of 16; inside a chunk, MFMA step e (0..3) gives lane group g the index
// k = chunk + 4 g + e -- for BOTH operands, which is all the instruction needs (its four k slots are summed).  An operand
// that is contiguous along k is therefore fetched as ONE 16-byte load per lane and chunk (AV / BV; a product whose B is
// [J, K] row-major used to issue 16 separate 64-byte segments per dword load and was bound by the CU's address path, not by
// bytes: 44 us for the 64 x 1536 x 768 fc0 product); the other layout (contiguous along i / j) keeps dword loads, which
// are 64-byte runs per lane group there.
template <bool AV, bool BV, int JT>
__device__ __forceinline__ void ht_mainloop(const HtJob& p, const float* ap, const float* yp, const float* const (&bp)[JT],
                                            const bool iv, const bool (&jv)[JT], const int g, const float mean,
                                            const float rstd, const int c_first, const int c_step, f32x4 (&acc)[JT],
                                            float& asum) {
    const int nchunk = (p.K + 15) >> 4;
    for (int c0 = c_first; c0 < nchunk; c0 += c_step * HT_U) {
        f32x4 a[HT_U], y[HT_U], ga[HT_U], be[HT_U], b[HT_U][JT];
#pragma unroll
        for (int u = 0; u < HT_U; ++u) {
            const int k0 = (c0 + c_step * u) * 16 + 4 * g;           // this lane's four k of the chunk: k0 .. k0 + 3
            const bool in = k0 < p.K;                                // (K % 4 == 0 where a vector load is used)
            const int kc = in ? k0 : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = min(k0 + e, p.K - 1);
                if (!AV) a[u][e] = ap[(size_t)ke * p.sa_k];
                if (!AV && p.pro == FEDDAT_HT_PRO_TANH_BWD) y[u][e] = yp[(size_t)ke * p.sa_k];
            }
            if (AV) {
                a[u] = *reinterpret_cast<const f32x4*>(ap + kc);
                if (p.pro == FEDDAT_HT_PRO_TANH_BWD) y[u] = *reinterpret_cast<const f32x4*>(yp + kc);
            }
            if (p.pro == FEDDAT_HT_PRO_LN) {                         // (LN implies AV: checked on the host)
                ga[u] = *reinterpret_cast<const f32x4*>(p.pro_a + kc);
                be[u] = *reinterpret_cast<const f32x4*>(p.pro_b + kc);
            }
#pragma unroll
            for (int t = 0; t < JT; ++t) {
                if (BV) {
                    b[u][t] = *reinterpret_cast<const f32x4*>(bp[t] + kc);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[u][t][e] = bp[t][(size_t)min(k0 + e, p.K - 1) * p.sb_k];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < HT_U; ++u) {
            const int k0 = (c0 + c_step * u) * 16 + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool kv = (k0 + e) < p.K && (c0 + c_step * u) < nchunk;
                float av = a[u][e];
                if (p.pro == FEDDAT_HT_PRO_LN) av = (av - mean) * rstd * ga[u][e] + be[u][e];
                if (p.pro == FEDDAT_HT_PRO_TANH_BWD) av = av * (1.f - y[u][e] * y[u][e]);
                av = (iv && kv) ? av : 0.f;
                asum += av;
#pragma unroll
                for (int t = 0; t < JT; ++t) acc[t] = mfma16x4_f32(av, (jv[t] && kv) ? b[u][t][e] : 0.f, acc[t]);
            }
        }
    }
}

// JT = 16-column tiles per wave: 4 (64 columns per block) for the products with enough tiles to fill the chip, 1 for the
// few-tile ones (fc1: 64 x 100; d(pooled): 32 x 768 -- their time is one wave's chain of dependent load batches, so more,
// narrower blocks win).
template <int JT>
__device__ __forceinline__ void ht_body(const HtJob& p, const int wid, float* ht_smem) {
    float (*red)[64][HT_JT * 4 + 4] = reinterpret_cast<float (*)[64][HT_JT * 4 + 4]>(ht_smem);     // [HT_NW - 1]
    float (*srow)[2] = reinterpret_cast<float (*)[2]>(ht_smem + (HT_NW - 1) * 64 * (HT_JT * 4 + 4));     // [16]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int itile = wid / p.jblocks, jb = wid - itile * p.jblocks;
    const int i16 = lane & 15, g = lane >> 4;
    const int i = itile * 16 + i16;
    const bool iv = i < p.I;
    const float* ap = p.A + (size_t)(iv ? i : 0) * p.sa_i;
    const float* yp = p.pro == FEDDAT_HT_PRO_TANH_BWD ? p.pro_a + (size_t)(iv ? i : 0) * p.sa_i : nullptr;

    // LayerNorm prologue: wave w computes the statistics of rows 16 itile + w, + w + HT_NW
    float mean = 0.f, rstd = 0.f;
    if (p.pro == FEDDAT_HT_PRO_LN) {
        for (int rl = wave; rl < 16; rl += HT_NW) {
            const int r = itile * 16 + rl;
            if (r >= p.I) continue;
            // (sa_k == 1, K % 4 == 0, K <= 2048: checked on the host) the row sits in registers: one pass of loads
            const float* xr = p.A + (size_t)r * p.sa_i;
            const int nc = p.K >> 2;
            f32x4 v[8];
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int q4 = lane + c * 64;
                if (q4 < nc) {
                    v[c] = *reinterpret_cast<const f32x4*>(xr + q4 * 4);
                    s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
                }
            }
            const float m = wave_sum(s) / (float)p.K;
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (lane + c * 64 < nc)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float d = v[c][e] - m;
                        q += d * d;
                    }
            const float rs = rsqrtf(wave_sum(q) / (float)p.K + p.pro_eps);
            if (lane == 0) {
                srow[rl][0] = m;
                srow[rl][1] = rs;
                if (p.stats_out && jb == 0) {
                    p.stats_out[2 * r] = m;
                    p.stats_out[2 * r + 1] = rs;
                }
            }
        }
        __syncthreads();
        mean = srow[i16][0];
        rstd = srow[i16][1];
    }

    const int jgrp = p.mode == 0 ? jb : jb * HT_NW + wave;
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
    const int c_first = p.mode == 0 ? wave : 0, c_step = p.mode == 0 ? HT_NW : 1;
    if (p.avec && p.bvec) ht_mainloop<true, true, JT>(p, ap, yp, bp, iv, jv, g, mean, rstd, c_first, c_step, acc, asum);
    else if (p.avec) ht_mainloop<true, false, JT>(p, ap, yp, bp, iv, jv, g, mean, rstd, c_first, c_step, acc, asum);
    else if (p.bvec) ht_mainloop<false, true, JT>(p, ap, yp, bp, iv, jv, g, mean, rstd, c_first, c_step, acc, asum);
    else ht_mainloop<false, false, JT>(p, ap, yp, bp, iv, jv, g, mean, rstd, c_first, c_step, acc, asum);
    asum += __shfl_xor(asum, 16, 64);
    asum += __shfl_xor(asum, 32, 64);
    if (p.mode == 0) {
        if (wave > 0) {
            float* dst = red[wave - 1][lane];
#pragma unroll
            for (int t = 0; t < JT; ++t) *reinterpret_cast<f32x4*>(dst + 4 * t) = acc[t];
            dst[JT * 4] = asum;
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll 1
        for (int w = 0; w < HT_NW - 1; ++w) {
            const float* src = red[w][lane];
#pragma unroll
            for (int t = 0; t < JT; ++t) acc[t] = acc[t] + *reinterpret_cast<const f32x4*>(src + 4 * t);
            asum += src[JT * 4];
        }
    }
    const float alpha = p.alpha_dev ? p.alpha * *p.alpha_dev : p.alpha;      // (a power of two: the product is exact)
#pragma unroll
    for (int t = 0; t < JT; ++t) {
        if (!jv[t]) continue;
        const float bj = p.bias_j ? p.bias_j[j[t]] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int io = itile * 16 + 4 * g + e;
            if (io >= p.I) continue;
            float v = alpha * acc[t][e] + bj;
            if (p.epi == FEDDAT_HT_EPI_TANH) v = tanhf(v);
            if (p.epi == FEDDAT_HT_EPI_MUL_DGELU) v *= gelu_grad_f(p.aux[(size_t)io * p.ld_aux + j[t]]);
            p.out[(size_t)io * p.ldo + j[t]] = v;
        }
    }
    if (p.colsum && jgrp == 0 && g == 0 && iv) p.colsum[i] = alpha * asum;
}

__global__ __launch_bounds__(HT_NW * 64) void ht_gemm_kernel(const HtLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float ht_smem[];
    const bool second = (int)blockIdx.x >= L.blocks0;
    const HtJob& p = L.job[second ? 1 : 0];
    const int wid = second ? blockIdx.x - L.blocks0 : blockIdx.x;
    if (p.jt == 1) ht_body<1>(p, wid, ht_smem);
    else ht_body<HT_JT>(p, wid, ht_smem);
}

// y = LayerNorm(x) (fp32, stats), g = gelu(y): one wave per row, H <= 2048
__global__ __launch_bounds__(256) void ht_ln_gelu_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, int rows, int H,
                                                         float* __restrict__ y, float* __restrict__ stats,
                                                         float* __restrict__ gl) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nc = H >> 2;
    const float* xr = x + (size_t)row * H;
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            v[c] = *reinterpret_cast<const f32x4*>(xr + i * 4);
            s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (lane + c * 64 < nc)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[c][e] - mean;
                q += d * d;
            }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int i = lane + c * 64;
        if (i < nc) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + i * 4);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + i * 4);
            f32x4 o, ge;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[c][e] - mean) * rstd * g4[e] + b4[e];
                ge[e] = gelu_f(o[e]);
            }
            *reinterpret_cast<f32x4*>(y + (size_t)row * H + i * 4) = o;
            *reinterpret_cast<f32x4*>(gl + (size_t)row * H + i * 4) = ge;
        }
    }
}

// Full LayerNorm backward (trainable affine) in one launch: blocks [0, ceil(rows / 4)) give dx (one wave per row),
// the rest give dgamma / dbeta (64 columns per block, four row-interleaved partial sums, fixed order).
__global__ __launch_bounds__(256) void ht_ln_bwd_full_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, int rows, int H,
                                                             float* __restrict__ dx, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
    __shared__ float sg[4][64], sb[4][64];
    const int row_blocks = (rows + 3) / 4;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    if ((int)blockIdx.x < row_blocks) {
        const int row = blockIdx.x * 4 + part;
        if (row >= rows) return;
        const int nc = H >> 2;
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        f32x4 gd[8], xh[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = lane + c * 64;
            if (i < nc) {
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dy + (size_t)row * H + i * 4);
                const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + (size_t)row * H + i * 4);
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + i * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    gd[c][e] = d4[e] * g4[e];
                    xh[c][e] = (x4[e] - mean) * rstd;
                    s1 += gd[c][e];
                    s2 += gd[c][e] * xh[c][e];
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = lane + c * 64;
            if (i < nc) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (gd[c][e] - m1 - xh[c][e] * m2);
                *reinterpret_cast<f32x4*>(dx + (size_t)row * H + i * 4) = o;
            }
        }
        return;
    }
    const int col = ((int)blockIdx.x - row_blocks) * 64 + lane;
    float ag = 0.f, ab = 0.f;
    if (col < H) {
        for (int r = part; r < rows; r += 4) {
            const float d = dy[(size_t)r * H + col];
            ag += d * ((x[(size_t)r * H + col] - stats[2 * r]) * stats[2 * r + 1]);
            ab += d;
        }
    }
    sg[part][lane] = ag;
    sb[part][lane] = ab;
    __syncthreads();
    if (part == 0 && col < H) {
        dgamma[col] = (sg[0][lane] + sg[1][lane]) + (sg[2][lane] + sg[3][lane]);
        dbeta[col] = (sb[0][lane] + sb[1][lane]) + (sb[2][lane] + sb[3][lane]);
    }
}

// DAT loss (task_trainer.py:299-301, 506-516) in ONE launch: wave w takes rows w, w + 16, ...; the batch sums run over
// the row terms in LDS in the order of the two-kernel form (feddat_dat_loss_fwd_bwd), so both give the same bits.
__global__ __launch_bounds__(1024) void dat_loss_single_kernel(const float* __restrict__ logits,
                                                               const float* __restrict__ teacher,
                                                               const float* __restrict__ target, int B, int C, float temp,
                                                               float* __restrict__ dlogits, float* __restrict__ scalars,
                                                               int* __restrict__ nonfinite) {
    extern __shared__ float terms[];       // [2 B]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float it = 1.0f / temp;
    // C <= 128 (the VQA heads: 100 labels): two elements per lane, the rows of a wave are loaded up front (RPW at a time) so
    // that their three passes run from registers -- the launch is a chain of dependent reductions, not bytes
    constexpr int RPW = 4;
    for (int b0 = wave; b0 < B; b0 += 16 * RPW) {
        float xv[RPW][2], tv[RPW][2], yv[RPW][2];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int b = b0 + 16 * r;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = lane + 64 * h;
                const bool ok = b < B && j < C;
                xv[r][h] = ok ? logits[(size_t)b * C + j] : 0.f;
                tv[r][h] = ok ? teacher[(size_t)b * C + j] : 0.f;
                yv[r][h] = ok ? target[(size_t)b * C + j] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int b = b0 + 16 * r;
            if (b >= B) break;
            const bool v0 = lane < C, v1 = lane + 64 < C;
            float mx = fmaxf(v0 ? xv[r][0] * it : -INFINITY, v1 ? xv[r][1] * it : -INFINITY);
            float mt = fmaxf(v0 ? tv[r][0] * it : -INFINITY, v1 ? tv[r][1] * it : -INFINITY);
            mx = wave_max(mx);
            mt = wave_max(mt);
            float sx = (v0 ? expf(xv[r][0] * it - mx) : 0.f) + (v1 ? expf(xv[r][1] * it - mx) : 0.f);
            float st = (v0 ? expf(tv[r][0] * it - mt) : 0.f) + (v1 ? expf(tv[r][1] * it - mt) : 0.f);
            sx = wave_sum(sx);
            st = wave_sum(st);
            const float lsx = mx + logf(sx), lst = mt + logf(st);
            float bce = 0.f, kl = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!(h ? v1 : v0)) continue;
                const float x = xv[r][h];
                bce += fmaxf(x, 0.f) - x * yv[r][h] + log1pf(expf(-fabsf(x)));
                const float logp = x * it - lsx;
                const float logq = tv[r][h] * it - lst;
                const float q = expf(logq);
                kl += q * (logq - logp);
                const float sig = 1.0f / (1.0f + expf(-x));
                dlogits[(size_t)b * C + lane + 64 * h] = 0.5f / (float)B * ((sig - yv[r][h]) + temp * (expf(logp) - q));
            }
            bce = wave_sum(bce);
            kl = wave_sum(kl);
            if (lane == 0) {
                terms[2 * b] = bce;
                terms[2 * b + 1] = kl;
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    float bce = 0.f, kl = 0.f;
    for (int b = lane; b < B; b += 64) {
        bce += terms[2 * b];
        kl += terms[2 * b + 1];
    }
    bce = wave_sum(bce);
    kl = wave_sum(kl);
    if (lane == 0) {
        const float l_bce = bce / (float)B;
        const float l_kl = kl / (float)B * temp * temp;
        scalars[0] = l_bce;
        scalars[1] = l_kl;
        scalars[2] = 0.5f * (l_bce + l_kl);
        // GradScaler's inf check, at its source: a non-finite loss means non-finite gradients in every parameter it reaches
        if (nonfinite && !(fabsf(l_bce + l_kl) <= 3.4e38f)) atomicOr(nonfinite, 1);
    }
}

__device__ __forceinline__ float ht_poly_lambda(int t, int warmup, int total) {     // = poly_lambda of loss_optim.hip
    if (t < warmup) return (float)t / (float)max(1, warmup);
    if (t > total) return 0.f;
    return 1.0f - (float)(t - warmup) / (float)(total - warmup);
}

struct AdamwMulti {
    feddat_adamw_group grp[FEDDAT_ADAMW_MAX_GROUPS];
    int block_end[FEDDAT_ADAMW_MAX_GROUPS];      // group k owns blocks [block_end[k-1], block_end[k])
    int ngroups;
    float base_lr, beta1, beta2, eps;
    int warmup, total;
};

// adamw_flat_kernel<4>'s arithmetic (loss_optim.hip; bit-identical per element) for up to three parameter groups in one
// launch; a group reads its schedule index / Adam step count at (state[0] + d_sched, state[1] + d_adam).
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamwMulti a) {
    int k = 0;
    while (k + 1 < a.ngroups && (int)blockIdx.x >= a.block_end[k]) ++k;
    const feddat_adamw_group& G = a.grp[k];
    const int blk = (int)blockIdx.x - (k ? a.block_end[k - 1] : 0);
    const long i = ((long)blk * 256 + threadIdx.x) * 4;
    if (i >= G.n) return;
    // dynamic loss scale (ABI 8): restore / skip decisions are block-uniform device flags (head_tail.hip)
    if (G.bak_mode == 2 && G.restore_if && *G.restore_if) {
        *reinterpret_cast<f32x4*>(G.p + i) = *reinterpret_cast<const f32x4*>(G.bak + i);
        *reinterpret_cast<f32x4*>(G.m + i) = *reinterpret_cast<const f32x4*>(G.bak + G.n + i);
        *reinterpret_cast<f32x4*>(G.v + i) = *reinterpret_cast<const f32x4*>(G.bak + 2 * G.n + i);
        return;
    }
    const bool skip = (G.skip_if[0] && *G.skip_if[0]) || (G.skip_if[1] && *G.skip_if[1]);
    if (skip && G.bak_mode != 1) return;
    const int sched_t = G.state[0] + G.d_sched;
    const int t = G.state[1] + G.d_adam + 1;
    const float lr = a.base_lr * ht_poly_lambda(sched_t, a.warmup, a.total);
    int lo = 0, hi = G.nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (G.seg_off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    float wd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) wd[e] = G.seg_wd[lo];
    if (lo + 1 < G.nseg && G.seg_off[lo + 1] < i + 4) {
        for (int e = 1; e < 4; ++e) {
            int s2 = lo;
            while (s2 + 1 < G.nseg && G.seg_off[s2 + 1] <= i + e) ++s2;
            wd[e] = G.seg_wd[s2];
        }
    }
    const float bc1 = 1.0f - powf(a.beta1, (float)t);
    const float bc2 = 1.0f - powf(a.beta2, (float)t);
    f32x4 gi = *reinterpret_cast<const f32x4*>(G.g + i);
    f32x4 pi = *reinterpret_cast<const f32x4*>(G.p + i);
    f32x4 mi = *reinterpret_cast<const f32x4*>(G.m + i);
    f32x4 vi = *reinterpret_cast<const f32x4*>(G.v + i);
    if (G.bak_mode == 1) {
        *reinterpret_cast<f32x4*>(G.bak + i) = pi;
        *reinterpret_cast<f32x4*>(G.bak + G.n + i) = mi;
        *reinterpret_cast<f32x4*>(G.bak + 2 * G.n + i) = vi;
        if (skip) return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        pi[e] = pi[e] * (1.0f - lr * wd[e]);
        mi[e] = mi[e] + (gi[e] - mi[e]) * (1.0f - a.beta1);
        vi[e] = vi[e] * a.beta2 + (1.0f - a.beta2) * gi[e] * gi[e];
        const float denom = sqrtf(vi[e]) / sqrtf(bc2) + a.eps;
        pi[e] -= (lr / bc1) * (mi[e] / denom);
    }
    *reinterpret_cast<f32x4*>(G.p + i) = pi;
    *reinterpret_cast<f32x4*>(G.m + i) = mi;
    *reinterpret_cast<f32x4*>(G.v + i) = vi;
}

struct TickMulti {
    int* state[FEDDAT_ADAMW_MAX_GROUPS];
    int d_sched[FEDDAT_ADAMW_MAX_GROUPS], d_adam[FEDDAT_ADAMW_MAX_GROUPS];
    int n;
};
__global__ void step_tick_multi_kernel(const TickMulti t) {
    const int k = threadIdx.x;
    if (blockIdx.x == 0 && k < t.n) {
        t.state[k][0] += t.d_sched[k];
        t.state[k][1] += t.d_adam[k];
    }
}

// End of one dat train_step under a dynamic loss scale: counter ticks by what was applied, GradScaler.update(), flags cleared
// (semantics: include/feddat_hip.h, feddat_dat_step_finish).
__global__ void dat_step_finish_kernel(int* head_state, int* ad1_state, int* ad0_state, int* flags, float* scaler_f,
                                       int* scaler_i, float growth, float backoff, int growth_interval) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int fB = flags[0], fA = flags[1];
    const int applied = fA ? 0 : fB ? 1 : 2;
    head_state[0] += applied;
    head_state[1] += applied;
    ad1_state[0] += applied;
    ad1_state[1] += applied >= 1 ? 1 : 0;
    ad0_state[0] += applied;
    ad0_state[1] += applied == 2 ? 1 : 0;
    float scale = scaler_f[0];
    int tracker = scaler_i[0];
    if (applied < 2) {
        scale = fmaxf(scale * backoff, 6.103515625e-05f);
        tracker = 0;
        scaler_i[1] += 2 - applied;
        scaler_i[2] += 1;
    } else {
        tracker += 2;
        if (tracker >= growth_interval) {
            scale = fminf(scale * growth, 1073741824.0f);
            tracker = 0;
        }
    }
    scaler_f[0] = scale;
    scaler_f[1] = 1.0f / scale;
    scaler_i[0] = tracker;
    flags[0] = 0;
    flags[1] = 0;
}

int ht_fill(HtJob& j, const feddat_ht_job& s) {
    if (!(s.A && s.B && s.out && s.I > 0 && s.J > 0 && s.K > 0 && s.ldo >= s.J && (s.mode == 0 || s.mode == 1)))
        return FEDDAT_EINVAL;
    if (s.pro == FEDDAT_HT_PRO_LN && !(s.pro_a && s.pro_b && s.pro_eps > 0.f && s.sa_k == 1 && s.K % 4 == 0 && s.K <= 2048 &&
                                       s.sa_i % 4 == 0 && ((uintptr_t)s.A & 15) == 0))
        return FEDDAT_EINVAL;
    if (s.pro == FEDDAT_HT_PRO_TANH_BWD && !s.pro_a) return FEDDAT_EINVAL;
    if (s.epi == FEDDAT_HT_EPI_MUL_DGELU && !(s.aux && s.ld_aux >= s.J)) return FEDDAT_EINVAL;
    if (s.pro < 0 || s.pro > FEDDAT_HT_PRO_TANH_BWD || s.epi < 0 || s.epi > FEDDAT_HT_EPI_MUL_DGELU) return FEDDAT_EINVAL;
    j.A = s.A; j.sa_i = s.sa_i; j.sa_k = s.sa_k; j.B = s.B; j.sb_k = s.sb_k; j.sb_j = s.sb_j;
    j.I = s.I; j.J = s.J; j.K = s.K; j.mode = s.mode; j.alpha = s.alpha; j.alpha_dev = s.alpha_dev; j.bias_j = s.bias_j; j.out = s.out; j.ldo = s.ldo;
    j.colsum = s.colsum; j.pro = s.pro; j.pro_a = s.pro_a; j.pro_b = s.pro_b; j.pro_eps = s.pro_eps;
    j.stats_out = s.stats_out; j.epi = s.epi; j.aux = s.aux; j.ld_aux = s.ld_aux;
    j.avec = s.sa_k == 1 && s.K % 4 == 0 && s.sa_i % 4 == 0 && ((uintptr_t)s.A & 15) == 0 &&
             (s.pro != FEDDAT_HT_PRO_TANH_BWD || ((uintptr_t)s.pro_a & 15) == 0);
    j.bvec = s.sb_k == 1 && s.K % 4 == 0 && s.sb_j % 4 == 0 && ((uintptr_t)s.B & 15) == 0;
    if (s.pro == FEDDAT_HT_PRO_LN && !(j.avec && (((uintptr_t)s.pro_a | (uintptr_t)s.pro_b) & 15) == 0)) return FEDDAT_EINVAL;
    j.itiles = (s.I + 15) / 16;
    int jgroups = (s.J + 16 * HT_JT - 1) / (16 * HT_JT);
    j.jt = HT_JT;
    if (s.mode == 0 && j.itiles * jgroups < 64) {      // too few 16 x 64 tiles: 16 x 16 tiles, four times the blocks
        j.jt = 1;
        jgroups = (s.J + 15) / 16;
    }
    j.jblocks = s.mode == 0 ? jgroups : (jgroups + HT_NW - 1) / HT_NW;
    return FEDDAT_OK;
}

}  // namespace

extern "C" int feddat_head_gemm(const feddat_ht_job* jobs, int njobs, hipStream_t stream) {
    FD_CHECK_ARG(jobs && (njobs == 1 || njobs == 2));
    HtLaunch L{};
    int total = 0;
    for (int k = 0; k < njobs; ++k) {
        const int rc = ht_fill(L.job[k], jobs[k]);
        if (rc != FEDDAT_OK) return rc;
        const int nb = L.job[k].itiles * L.job[k].jblocks;
        if (k == 0) L.blocks0 = nb;
        total += nb;
    }
    constexpr int lds = ((HT_NW - 1) * 64 * (HT_JT * 4 + 4) + 32) * (int)sizeof(float);
    if (fd_set_max_lds((const void*)ht_gemm_kernel, lds) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    hipLaunchKernelGGL(ht_gemm_kernel, dim3(total), dim3(HT_NW * 64), lds, stream, L);
    FD_LAUNCH_RET();
}

extern "C" int feddat_head_ln_gelu(const float* x, const float* gamma, const float* beta, float eps, int rows, int H,
                                   float* y, float* stats, float* gelu_out, hipStream_t stream) {
    FD_CHECK_ARG(x && gamma && beta && y && stats && gelu_out && rows > 0 && H > 0 && H % 4 == 0 && H <= 2048 && eps > 0.f);
    hipLaunchKernelGGL(ht_ln_gelu_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, eps, rows, H, y, stats,
                       gelu_out);
    FD_LAUNCH_RET();
}

extern "C" int feddat_head_ln_bwd_full(const float* dy, const float* x, const float* stats, const float* gamma, int rows,
                                       int H, float* dx, float* dgamma, float* dbeta, hipStream_t stream) {
    FD_CHECK_ARG(dy && x && stats && gamma && dx && dgamma && dbeta && rows > 0 && rows <= 4096 && H > 0 && H % 4 == 0 &&
                 H <= 2048);
    hipLaunchKernelGGL(ht_ln_bwd_full_kernel, dim3((rows + 3) / 4 + (H + 63) / 64), dim3(256), 0, stream, dy, x, stats, gamma,
                       rows, H, dx, dgamma, dbeta);
    FD_LAUNCH_RET();
}

extern "C" int feddat_dat_loss_fwd_bwd_single(const float* logits, const float* teacher, const float* target, int B, int C,
                                              float temp, float* dlogits, float* scalars, hipStream_t stream) {
    FD_CHECK_ARG(logits && teacher && target && dlogits && scalars && B > 0 && B <= 4096 && C > 0 && C <= 128 && temp > 0.f);
    hipLaunchKernelGGL(dat_loss_single_kernel, dim3(1), dim3(1024), 2 * B * sizeof(float), stream, logits, teacher, target, B,
                       C, temp, dlogits, scalars, (int*)nullptr);
    FD_LAUNCH_RET();
}

extern "C" int feddat_dat_loss_fwd_bwd_checked(const float* logits, const float* teacher, const float* target, int B, int C,
                                               float temp, float* dlogits, float* scalars, int* nonfinite, hipStream_t stream) {
    FD_CHECK_ARG(logits && teacher && target && dlogits && scalars && B > 0 && B <= 4096 && C > 0 && C <= 128 && temp > 0.f);
    hipLaunchKernelGGL(dat_loss_single_kernel, dim3(1), dim3(1024), 2 * B * sizeof(float), stream, logits, teacher, target, B,
                       C, temp, dlogits, scalars, nonfinite);
    FD_LAUNCH_RET();
}

extern "C" int feddat_dat_step_finish(int* head_state, int* ad1_state, int* ad0_state, int* flags, float* scaler_f,
                                      int* scaler_i, float growth, float backoff, int growth_interval, hipStream_t stream) {
    FD_CHECK_ARG(head_state && ad1_state && ad0_state && flags && scaler_f && scaler_i && growth >= 1.0f && backoff > 0.f &&
                 backoff <= 1.0f && growth_interval > 0);
    hipLaunchKernelGGL(dat_step_finish_kernel, dim3(1), dim3(64), 0, stream, head_state, ad1_state, ad0_state, flags, scaler_f,
                       scaler_i, growth, backoff, growth_interval);
    FD_LAUNCH_RET();
}

extern "C" int feddat_adamw_multi(const feddat_adamw_group* groups, int ngroups, float base_lr, int warmup, int total,
                                  float beta1, float beta2, float eps, hipStream_t stream) {
    FD_CHECK_ARG(groups && ngroups >= 1 && ngroups <= FEDDAT_ADAMW_MAX_GROUPS && total > warmup);
    AdamwMulti a{};
    int blocks = 0;
    for (int k = 0; k < ngroups; ++k) {
        const feddat_adamw_group& G = groups[k];
        FD_CHECK_ARG(G.p && G.g && G.m && G.v && G.n > 0 && G.n % 4 == 0 && G.seg_off && G.seg_wd && G.nseg > 0 && G.state);
        FD_CHECK_ARG((((uintptr_t)G.p | (uintptr_t)G.g | (uintptr_t)G.m | (uintptr_t)G.v) & 15) == 0);
        FD_CHECK_ARG(G.bak_mode >= 0 && G.bak_mode <= 2 && (G.bak_mode == 0 || (G.bak && ((uintptr_t)G.bak & 15) == 0)) &&
                     (G.bak_mode != 2 || G.restore_if));
        a.grp[k] = G;
        blocks += (int)((G.n / 4 + 255) / 256);
        a.block_end[k] = blocks;
    }
    a.ngroups = ngroups;
    a.base_lr = base_lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.warmup = warmup; a.total = total;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks), dim3(256), 0, stream, a);
    FD_LAUNCH_RET();
}

extern "C" int feddat_step_tick_multi(int* const* states, const int* d_sched, const int* d_adam, int n, hipStream_t stream) {
    FD_CHECK_ARG(states && d_sched && d_adam && n >= 1 && n <= FEDDAT_ADAMW_MAX_GROUPS);
    TickMulti t{};
    for (int k = 0; k < n; ++k) {
        FD_CHECK_ARG(states[k]);
        t.state[k] = states[k];
        t.d_sched[k] = d_sched[k];
        t.d_adam[k] = d_adam[k];
    }
    t.n = n;
    hipLaunchKernelGGL(step_tick_multi_kernel, dim3(1), dim3(64), 0, stream, t);
    FD_LAUNCH_RET();
}
