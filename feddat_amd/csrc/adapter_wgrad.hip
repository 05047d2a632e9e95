// Adapter weight gradients (autograd of src/modeling/models/adapter.py:125-146 w.r.t. the trainable adapter):
//     dW_down[r][c] = sum_t dz[t][r] * x[t][c]          db_down[r] = sum_t dz[t][r]
//     dW_up  [c][r] = s * sum_t dy[t][c] * z[t][r]      db_up  [c] = s * sum_t dy[t][c]
// Both are "small^T x big" products contracted over the tokens: small = [T,48], big = [T,768], fp32.
// bf16 MFMA 16x16x32 on split operands (every fp32 value = bf16 head + bf16 remainder, all four products per pair: fp32
// accuracy to ~2^-16 relative per product, fp32 accumulation; details at the loop): lane (index, token group) =
// (lane & 15, lane >> 4) holds 8 consecutive tokens of its index, so the token-major activations are consumed as they
// lie in HBM, no transposes.  A wave owns 64 columns x all 48 bottleneck units for a range of tokens: 8 float4 loads of
// `big` feed 4 interleaved column tiles and (round 6) 8 three-dword loads of `small` feed 3 interleaved unit groups of 16 (unit
// 3 q + v is row q of group v) -> 48 MFMAs per 32 tokens.  The 24 single-dword loads per round that three CONTIGUOUS 16-unit tiles
// took were three quarters of the kernel's VMEM instructions for an eighth of its bytes.  The 4
// waves of a block own 4 consecutive token ranges and are summed through LDS; the NBLK blocks per column chunk leave NBLK
// partials that the (deterministic) reduce kernel folds straight into the flat gradient buffer [wd | bd | wu | bu].
// HBM-bound in bytes (x and dy are each read once: 2 x T x 768 x 4 B).
#include "common.hip.h"

namespace {

constexpr int H = 768, R = 48, NRT = 3, CW = 64;          // NRT interleaved unit groups (unit NRT q + v = row q of group v); columns per wave
constexpr int NCH = H / CW;                                 // 12 column chunks
constexpr int NBLK = 10;                                    // token-split blocks per column chunk: 12 x 10 x 4 problems = 480 blocks
constexpr int PSTRIDE = R * H + R + H;                      // one partial: [out r x c | colsum_small | colsum_big]

// x[0..7] (fp32) -> head = bf16(x) (round to nearest even: the remainder is zero-mean and <= 2^-9 |x|; truncated heads
// left a one-signed 2^-14 bias in the remainder x remainder product), rest = bf16(x - head)
__device__ __forceinline__ void split_bf16x2(const float (&x)[8], tbf16x8& head, tbf16x8& rest) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    head = cvt8_tbf16(f32x4{x[0], x[1], x[2], x[3]}, f32x4{x[4], x[5], x[6], x[7]});
    const u32x4_t hp = __builtin_bit_cast(u32x4_t, head);
    float r[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r[2 * k] = x[2 * k] - __uint_as_float(hp[k] << 16);
        r[2 * k + 1] = x[2 * k + 1] - __uint_as_float(hp[k] & 0xffff0000u);
    }
    rest = cvt8_tbf16(f32x4{r[0], r[1], r[2], r[3]}, f32x4{r[4], r[5], r[6], r[7]});
}

struct WgradLaunch {
    feddat_wgrad_seg seg[2];
    float* partials;
    int nseg;
};

__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradLaunch L) {   // 2 blocks per CU (225 registers, no scratch): 480 blocks = one round
    __shared__ __attribute__((aligned(16))) float red[3][64][NRT * 4 * 4 + 4];  // waves 1..3 -> wave 0
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the token loop must not be divergent
    const int g = lane >> 4, i16 = lane & 15;
    // token block = fast grid index
    const int blk = blockIdx.x, chunk = blockIdx.y;
    const int prob = blockIdx.z;            // 2 * seg + which (0: dW_down = dz^T x, 1: dW_up^T = z^T dy)
    const feddat_wgrad_seg& sg = L.seg[prob >> 1];
    const bool up = prob & 1;
    const float* big = up ? sg.dy : sg.x;
    const float* sm = up ? sg.z : sg.dz;
    float unscale = sg.grad_unscale != 0.0f ? sg.grad_unscale : 1.0f;      // 1 / loss scale (a power of two: exact)
    if (sg.grad_unscale_dev) unscale *= *sg.grad_unscale_dev;               // ... of the dynamic loss scale (device state)
    const float alpha = (up ? sg.scale : 1.0f) * unscale;
    const int T = sg.rows;
    int tps = (T + NBLK * 4 - 1) / (NBLK * 4);
    tps = (tps + 7) & ~7;
    const int t_begin = (blk * 4 + wave) * tps;
    const int t_end = min(T, t_begin + tps);
    const int c0 = chunk * CW;

    f32x4 acc[NRT][4];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[rt][v] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    float ssum[NRT] = {0.f, 0.f, 0.f};
    // 32 tokens per round.  Lane (i16, g) holds tokens t0 + 8 g .. + 7 of its 4 columns (one float4 per token) and of
    // its bottleneck unit per r-tile: exactly the K = 32 operand layout of the bf16 MFMA, no transposes.  Every fp32
    // value is split into its bf16 rounding and the bf16-rounded remainder (x = h + l to 2^-17 relative) and the
    // product is l l' + l h' + h l' + h h' on the bf16 pipe: 12 bf16 MFMAs (16 clk each) replace 24 fp32 MFMAs (32 clk each)
    // per (r-tile, column tile, 32 tokens); the fp32 form made this kernel MFMA-bound at 1/16 of the bf16 rate.
    // Accumulation stays fp32; the bias gradients (column sums) are exact fp32 adds.
    // the loads of round j + 1 are issued before the products of round j (addresses of masked tokens fall back to row 0, values
    // masked: the loop stays branch-free and the compiler's vmcnt counts exact)
    f32x4 braw[8];
    float sraw[8][NRT];
    const int su = NRT * i16;                           // this lane's 3 consecutive units: row i16 of the 3 unit groups
    auto load_round = [&](int t0, f32x4 (&bq)[8], float (&sq)[8][NRT]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int t = t0 + 8 * g + e;
            const int tc = t < t_end ? t : 0;          // masked tokens re-read row 0 (always inside the segment)
            bq[e] = *reinterpret_cast<const f32x4*>(big + (size_t)tc * H + c0 + 4 * i16);
            const float* sp = sm + (size_t)tc * R + su;          // 12 contiguous bytes: one global_load_dwordx3
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) sq[e][rt] = sp[rt];
        }
    };
    load_round(t_begin, braw, sraw);
    for (int t0 = t_begin; t0 < t_end; t0 += 32) {
        f32x4 bnext[8];
        float snext[8][NRT];
        load_round(t0 + 32, bnext, snext);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (t0 + 8 * g + e >= t_end) {
                braw[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) sraw[e][rt] = 0.f;
            }
            bsum = bsum + braw[e];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) ssum[rt] += sraw[e][rt];
        }
        tbf16x8 sh[NRT], sl[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = sraw[e][rt];
            split_bf16x2(x, sh[rt], sl[rt]);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = braw[e][v];
            tbf16x8 bh, bl;
            split_bf16x2(x, bh, bl);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                acc[rt][v] = mfma16x32_tbf16(sl[rt], bl, acc[rt][v]);        // smallest terms first
                acc[rt][v] = mfma16x32_tbf16(sl[rt], bh, acc[rt][v]);
                acc[rt][v] = mfma16x32_tbf16(sh[rt], bl, acc[rt][v]);
                acc[rt][v] = mfma16x32_tbf16(sh[rt], bh, acc[rt][v]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            braw[e] = bnext[e];
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) sraw[e][rt] = snext[e][rt];
        }
    }
    // column sums: reduce over the 4 token slots (g) of the wave
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        bsum[v] += __shfl_xor(bsum[v], 16, 64);
        bsum[v] += __shfl_xor(bsum[v], 32, 64);
    }
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        ssum[rt] += __shfl_xor(ssum[rt], 16, 64);
        ssum[rt] += __shfl_xor(ssum[rt], 32, 64);
    }
    // block reduction of the 4 waves through LDS (fixed order: deterministic)
    if (wave > 0) {
        float* dst = red[wave - 1][lane];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(dst + (rt * 4 + v) * 4) = acc[rt][v];
        // lanes with g == 0 carry the column sums: big (4 values) for lane i16, small (3 values)
        if (g == 0) *reinterpret_cast<f32x4*>(dst + NRT * 16) = bsum;
    }
    __shared__ float red_s[3][NRT][16];
    if (wave > 0 && g == 0) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) red_s[wave - 1][rt][i16] = ssum[rt];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        const float* src = red[w][lane];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[rt][v] = acc[rt][v] + *reinterpret_cast<const f32x4*>(src + (rt * 4 + v) * 4);
        if (g == 0) {
            bsum = bsum + *reinterpret_cast<const f32x4*>(src + NRT * 16);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) ssum[rt] += red_s[w][rt][i16];
        }
    }
    // D layout: acc[rt][v] row q = 4*(lane>>4) + e of unit group rt = unit NRT q + rt, col (lane & 15) -> column
    // c = c0 + 4*(lane & 15) + v
    float* P = L.partials + ((size_t)prob * NBLK + blk) * PSTRIDE;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = NRT * (4 * g + e) + rt;
            f32x4 o = {alpha * acc[rt][0][e], alpha * acc[rt][1][e], alpha * acc[rt][2][e], alpha * acc[rt][3][e]};
            *reinterpret_cast<f32x4*>(P + (size_t)r * H + c0 + 4 * i16) = o;
        }
    if (g == 0) {
        f32x4 o = {alpha * bsum[0], alpha * bsum[1], alpha * bsum[2], alpha * bsum[3]};
        *reinterpret_cast<f32x4*>(P + R * H + R + c0 + 4 * i16) = o;
        if (chunk == 0) {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) P[R * H + NRT * i16 + rt] = unscale * ssum[rt];
        }
    }
}

// grad layer layout: [wd (R x H) | bd (R) | wu (H x R) | bu (H)]
__device__ __forceinline__ bool wgrad_reduce_one(const float* __restrict__ partials, float* __restrict__ gl, int seg, int i) {
    const float* Pd = partials + (size_t)(2 * seg) * NBLK * PSTRIDE;       // dW_down problem
    const float* Pu = partials + (size_t)(2 * seg + 1) * NBLK * PSTRIDE;   // dW_up^T problem
    float sd = 0.f, su = 0.f;
#pragma unroll 4
    for (int b = 0; b < NBLK; ++b) {
        sd += Pd[(size_t)b * PSTRIDE + i];
        su += Pu[(size_t)b * PSTRIDE + i];
    }
    if (i < R * H) {
        gl[i] = sd;                                   // wd[r][c]
        const int r = i / H, c = i - r * H;
        gl[R * H + R + (size_t)c * R + r] = su;       // wu[c][r] (transpose of the computed [r][c])
    } else if (i < R * H + R) {
        gl[i] = sd;                                   // bd[r] = sum dz
    } else {
        gl[R * H + R + H * R + (i - R * H - R)] = su;  // bu[c] = s * sum dy
    }
    return !(fabsf(sd) <= 3.4e38f) || !(fabsf(su) <= 3.4e38f);      // inf / NaN in what was written (GradScaler's inf check)
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradLaunch L) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < PSTRIDE) wgrad_reduce_one(L.partials, L.seg[blockIdx.y].grad, blockIdx.y, i);
}

// the reductions of several launches (one per layer) in ONE: blockIdx.z = launch, its partials at + z * stride
struct WgradReduceBatch {
    float* const* grads;      // device array [n][nseg] of gradient pointers
    const float* partials;
    long stride;
    int nseg;
    int* nonfinite;           // [nseg] or NULL: OR-ed with 1 when a segment's gradients hold an inf / NaN
};
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(WgradReduceBatch B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    if (i < PSTRIDE)
        bad = wgrad_reduce_one(B.partials + (size_t)blockIdx.z * B.stride, B.grads[blockIdx.z * B.nseg + blockIdx.y], blockIdx.y, i);
    if (B.nonfinite && __ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(B.nonfinite + blockIdx.y, 1);
}

}  // namespace

extern "C" long feddat_adapter_wgrad_workspace_elems(int nseg) { return (long)nseg * 2 * NBLK * PSTRIDE; }

static int wgrad_launch(const feddat_wgrad_seg* segs, int nseg, float* partials, long partials_elems, int Hd, int r,
                        bool reduce_now, hipStream_t stream) {
    FD_CHECK_ARG(segs && nseg >= 1 && nseg <= 2 && partials && Hd == H && r == R);
    FD_CHECK_ARG(partials_elems >= (long)nseg * 2 * NBLK * PSTRIDE);
    WgradLaunch L;
    L.nseg = nseg;
    L.partials = partials;
    for (int s = 0; s < nseg; ++s) {
        FD_CHECK_ARG(segs[s].x && segs[s].dy && segs[s].z && segs[s].dz && segs[s].grad && segs[s].rows > 0);
        L.seg[s] = segs[s];
    }
    if (nseg == 1) L.seg[1] = L.seg[0];
    hipLaunchKernelGGL(wgrad_kernel, dim3(NBLK, NCH, 2 * nseg), dim3(256), 0, stream, L);
    if (reduce_now) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((PSTRIDE + 255) / 256, nseg), dim3(256), 0, stream, L);
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_wgrad(const feddat_wgrad_seg* segs, int nseg, float* partials, long partials_elems,
                                    int Hd, int r, hipStream_t stream) {
    return wgrad_launch(segs, nseg, partials, partials_elems, Hd, r, true, stream);
}

extern "C" int feddat_adapter_wgrad_partial(const feddat_wgrad_seg* segs, int nseg, float* partials, long partials_elems,
                                            int Hd, int r, hipStream_t stream) {
    return wgrad_launch(segs, nseg, partials, partials_elems, Hd, r, false, stream);
}

extern "C" int feddat_adapter_wgrad_reduce(float* const* grads_dev, int n, int nseg, const float* partials,
                                           long partials_stride, hipStream_t stream) {
    FD_CHECK_ARG(grads_dev && partials && n > 0 && n <= 65535 && nseg >= 1 && nseg <= 2);
    FD_CHECK_ARG(partials_stride >= (long)nseg * 2 * NBLK * PSTRIDE);
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((PSTRIDE + 255) / 256, nseg, n), dim3(256), 0, stream,
                       WgradReduceBatch{grads_dev, partials, partials_stride, nseg, nullptr});
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_wgrad_reduce_checked(float* const* grads_dev, int n, int nseg, const float* partials,
                                                   long partials_stride, int* nonfinite, hipStream_t stream) {
    FD_CHECK_ARG(grads_dev && partials && n > 0 && n <= 65535 && nseg >= 1 && nseg <= 2 && nonfinite);
    FD_CHECK_ARG(partials_stride >= (long)nseg * 2 * NBLK * PSTRIDE);
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((PSTRIDE + 255) / 256, nseg, n), dim3(256), 0, stream,
                       WgradReduceBatch{grads_dev, partials, partials_stride, nseg, nonfinite});
    FD_LAUNCH_RET();
}
