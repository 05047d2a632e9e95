// K2: fused self-attention forward / dX-only backward for the frozen ViLT layers
// (HF ViltSelfAttention: softmax(Q K^T / sqrt(64) + mask) V; reference call site src/modeling/vilt.py:127,
// backward = autograd through it from task_trainer.py:302,323).
//
// One workgroup (4 waves) per (sample, head) (two per (sample, head) in the backward).  S <= 320 so the operands
// of one head live in LDS for the whole kernel (row-major [S_pad][64] bf16, 16-byte chunks XOR-swizzled by (row & 7) so that the
// ds_read_b128 row-fragment reads are conflict-free).  No S x S matrix ever reaches HBM: scores stay in MFMA
// accumulators; the only saved statistic is the per-row log-sum-exp.
//
// MFMA operand plumbing (see common.hip.h for the slot convention): a score tile is always produced in the
// orientation whose accumulator layout (lane = one row/col index, 4 consecutive partner indices) IS the operand
// layout of the next product, with the contraction slots permuted -- the partner operand is then fetched with the
// same permutation: row fragments by ds_read_b128, "transposed" fragments (4 consecutive tokens for one feature)
// by ds_read_b64_tr_b16.
#include "common.hip.h"

namespace {

constexpr int D = 64;
constexpr int ROWB = D * 2;  // 128 bytes per LDS row
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int sw_off(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ bf16x8 row_frag(const char* m, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(m + sw_off(row, chunk));
}

// "transposed" fragment: for feature column (c0 + (lane & 15)) the 4 consecutive rows r0 .. r0+3, where r0 is
// uniform within each 16-lane group.  Lane i of the group supplies the address of row r0 + i/4, columns
// c0 + 4*(i%4) .. +3; ds_read_b64_tr_b16 hands lane i column i of that 4x16 block.
__device__ __forceinline__ bf16x4 tr_frag(const char* m, int r0, int c0, int lane) {
    const int i = lane & 15;
    const int row = r0 + (i >> 2);
    const int col = c0 + ((i & 3) << 2);
    const char* p = m + sw_off(row, col >> 3) + ((col & 7) << 1);
    return fd_ds_read_tr16(p);
}
__device__ __forceinline__ bf16x8 tr_frag8(const char* m, int r0a, int r0b, int c0, int lane) {
    const bf16x4 a = tr_frag(m, r0a, c0, lane);
    const bf16x4 b = tr_frag(m, r0b, c0, lane);
    return bf16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// cooperative load of one head's [S][64] bf16 slice (row stride ld elements) into swizzled LDS, zero-padding to S_pad
__device__ __forceinline__ void load_head(const bf16* __restrict__ src, long ld, int S, int S_pad, char* dst, int tid) {
    for (int idx = tid; idx < S_pad * 8; idx += 256) {
        const int row = idx >> 3, chunk = idx & 7;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < S) v = *reinterpret_cast<const bf16x8*>(src + (size_t)row * ld + chunk * 8);
        *reinterpret_cast<bf16x8*>(dst + sw_off(row, chunk)) = v;
    }
}

// row fragment of a [S][64] bf16 slice straight from global memory (zero beyond S)
__device__ __forceinline__ bf16x8 gload_frag(const bf16* base, long ld, int row, int S, int chunk) {
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < S) v = *reinterpret_cast<const bf16x8*>(base + (size_t)row * ld + chunk * 8);
    return v;
}

template <int NKS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                       bf16* __restrict__ ctx, float* __restrict__ lse, int S,
                                                       int heads) {
    constexpr int S_pad = NKS * 32, NKT = NKS * 2, NQT = NKS * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                 // K and V of the head live in LDS (3 blocks per CU at S <= 192); the two Q row
    char* Vs = Ks + S_pad * ROWB;    // fragments of a query tile are read once, straight from HBM
    float* mask_add = reinterpret_cast<float*>(Vs + S_pad * ROWB);  // 0 or -inf per key

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int H = heads * D;
    const long ld = 3L * H;
    const bf16* base = qkv + (size_t)b * S * ld + h * D;
    const int g = lane >> 4, i16 = lane & 15;
    // the Q fragments of a wave's NEXT query tile are requested before the current tile is computed (the first ones
    // before K / V are staged), so that only K / V's own latency is exposed
    bf16x8 qn[2];
    qn[0] = gload_frag(base, ld, wave * 16 + i16, S, g);
    qn[1] = gload_frag(base, ld, wave * 16 + i16, S, 4 + g);
    load_head(base + H, ld, S, S_pad, Ks, tid);
    load_head(base + 2 * H, ld, S, S_pad, Vs, tid);
    for (int k = tid; k < S_pad; k += 256) {
        const bool ok = k < S && (!kmask || kmask[(size_t)b * S + k]);
        mask_add[k] = ok ? 0.f : -INFINITY;
    }
    __syncthreads();

    for (int qt = wave; qt < NQT; qt += 4) {
        if (qt * 16 >= S) break;
        bf16x8 qf[2] = {qn[0], qn[1]};
        if (qt + 4 < NQT) {          // (rows beyond S come back as zeros from gload_frag)
            qn[0] = gload_frag(base, ld, (qt + 4) * 16 + i16, S, g);
            qn[1] = gload_frag(base, ld, (qt + 4) * 16 + i16, S, 4 + g);
        }
        // S^T tiles: rows = keys kt*16 + 4g + r, col = query i16
        f32x4 s[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = mfma16x32(row_frag(Ks, kt * 16 + i16, g), qf[0], a);
            a = mfma16x32(row_frag(Ks, kt * 16 + i16, 4 + g), qf[1], a);
            const f32x4 ma = *reinterpret_cast<const f32x4*>(mask_add + kt * 16 + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = a[e] * (0.125f * LOG2E) + ma[e];      // log2-domain scores: softmax via exp2
                mx = fmaxf(mx, a[e]);
            }
            s[kt] = a;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[kt][e] = __builtin_amdgcn_exp2f(s[kt][e] - mx);
                sum += s[kt][e];
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        // O^T[d][q] = sum_keys V^T[d][key] P^T[key][q]
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NKS; ++st) {
            const bf16x8 pb = cvt8(s[2 * st], s[2 * st + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vf = tr_frag8(Vs, st * 32 + 4 * g, st * 32 + 16 + 4 * g, dt * 16, lane);
                o[dt] = mfma16x32(vf, pb, o[dt]);
            }
        }
        const int q = qt * 16 + i16;
        if (q < S) {
            const float inv = 1.0f / sum;
            bf16* out = ctx + ((size_t)b * S + q) * H + h * D;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 v = o[dt];
                v[0] *= inv; v[1] *= inv; v[2] *= inv; v[3] *= inv;
                *reinterpret_cast<bf16x4*>(out + dt * 16 + 4 * g) = cvt4(v);
            }
            if (g == 0 && lse) lse[((size_t)b * heads + h) * S + q] = mx * (1.0f / LOG2E) + __logf(sum);
        }
    }
}

// Backward, split into two independent block roles (blockIdx.y) so that each block keeps only two of the four
// [S_pad x 64] operands in LDS (3 blocks / 12 waves per CU instead of 1 block / 4 waves):
//   role 0 (dK, dV): waves own 16-key tiles; Q and dO live in LDS (row + transposed fragments), the tile's K / V row
//                    fragments come straight from HBM into registers; scores in [q rows, key col] orientation.
//   role 1 (dQ):     waves own 16-query tiles; K and V live in LDS, the tile's Q / dO fragments, LSE and
//                    D = rowsum(dO * O) come from HBM; scores in [key rows, q col] orientation.

template <int NKS>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const bf16* __restrict__ qkv, const uint8_t* __restrict__ kmask,
                                                       const bf16* __restrict__ ctx, const float* __restrict__ lse,
                                                       const bf16* __restrict__ dctx, bf16* __restrict__ dqkv, int S,
                                                       int heads) {
    constexpr int S_pad = NKS * 32, NKT = NKS * 2, NQT = NKS * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* M0 = smem;                  // role 0: Q   | role 1: K
    char* M1 = M0 + S_pad * ROWB;     // role 0: dO  | role 1: V
    float* Dv = reinterpret_cast<float*>(M1 + S_pad * ROWB);
    float* Ls = Dv + S_pad;
    float* kvalid = Ls + S_pad;  // 1 / 0 per key

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
    const int role = blockIdx.y;
    const int H = heads * D;
    const long ld = 3L * H;
    const bf16* base = qkv + (size_t)b * S * ld + h * D;
    const bf16* gO = ctx + (size_t)b * S * H + h * D;
    const bf16* gG = dctx + (size_t)b * S * H + h * D;
    const int g = lane >> 4, i16 = lane & 15;
    bf16* dq_base = dqkv + (size_t)b * S * ld + h * D;
    for (int k = tid; k < S_pad; k += 256) {
        Ls[k] = k < S ? lse[((size_t)b * heads + h) * S + k] * LOG2E : 0.f;   // exp(x - lse) = exp2(x log2e - Ls)
        kvalid[k] = (k < S && (!kmask || kmask[(size_t)b * S + k])) ? 1.f : 0.f;
    }

    if (role == 0) {
        load_head(base, ld, S, S_pad, M0, tid);   // Q
        // dO into LDS, and Dv[q] = sum_d dO[q][d] * O[q][d]
        for (int idx = tid; idx < S_pad * 8; idx += 256) {
            const int row = idx >> 3, chunk = idx & 7;
            bf16x8 gv = {0, 0, 0, 0, 0, 0, 0, 0};
            float part = 0.f;
            if (row < S) {
                gv = *reinterpret_cast<const bf16x8*>(gG + (size_t)row * H + chunk * 8);
                const bf16x8 ov = *reinterpret_cast<const bf16x8*>(gO + (size_t)row * H + chunk * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) part += (float)gv[e] * (float)ov[e];
            }
            *reinterpret_cast<bf16x8*>(M1 + sw_off(row, chunk)) = gv;
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (chunk == 0) Dv[row] = part;
        }
        __syncthreads();
        const char* Qs = M0;
        const char* Gs = M1;
        for (int kt = wave; kt < NKT; kt += 4) {
            if (kt * 16 >= S) break;
            const int key = kt * 16 + i16;
            const bf16x8 kf0 = gload_frag(base + H, ld, key, S, g), kf1 = gload_frag(base + H, ld, key, S, 4 + g);
            const bf16x8 vf0 = gload_frag(base + 2 * H, ld, key, S, g), vf1 = gload_frag(base + 2 * H, ld, key, S, 4 + g);
            const float kv = kvalid[key];
            f32x4 dv[4], dk[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int qs = 0; qs < NKS; ++qs) {
                if (qs * 32 >= S) break;
                f32x4 p[2], ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int qrow = (2 * qs + t) * 16;
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sc = mfma16x32(row_frag(Qs, qrow + i16, g), kf0, sc);
                    sc = mfma16x32(row_frag(Qs, qrow + i16, 4 + g), kf1, sc);
                    dp = mfma16x32(row_frag(Gs, qrow + i16, g), vf0, dp);
                    dp = mfma16x32(row_frag(Gs, qrow + i16, 4 + g), vf1, dp);
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qrow + 4 * g);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(Dv + qrow + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = kv * __builtin_amdgcn_exp2f(sc[e] * (0.125f * LOG2E) - l4[e]);
                        p[t][e] = pe;
                        ds[t][e] = pe * (dp[e] - d4[e]);          // the 1/8 of dS is applied to dK at the end
                    }
                }
                const bf16x8 pb = cvt8(p[0], p[1]);
                const bf16x8 dsb = cvt8(ds[0], ds[1]);
                const int r0a = (2 * qs) * 16 + 4 * g, r0b = (2 * qs + 1) * 16 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = mfma16x32(tr_frag8(Gs, r0a, r0b, dt * 16, lane), pb, dv[dt]);
                    dk[dt] = mfma16x32(tr_frag8(Qs, r0a, r0b, dt * 16, lane), dsb, dk[dt]);
                }
            }
            if (key < S) {
                bf16* ok = dq_base + (size_t)key * ld + H;
                bf16* ov = dq_base + (size_t)key * ld + 2 * H;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *reinterpret_cast<bf16x4*>(ok + dt * 16 + 4 * g) = cvt4(dk[dt] * f32x4{0.125f, 0.125f, 0.125f, 0.125f});
                    *reinterpret_cast<bf16x4*>(ov + dt * 16 + 4 * g) = cvt4(dv[dt]);
                }
            }
        }
    } else {
        load_head(base + H, ld, S, S_pad, M0, tid);       // K
        load_head(base + 2 * H, ld, S, S_pad, M1, tid);   // V
        __syncthreads();
        const char* Ks = M0;
        const char* Vs = M1;
        for (int qt = wave; qt < NQT; qt += 4) {
            if (qt * 16 >= S) break;
            const int q = qt * 16 + i16;
            const bf16x8 qf0 = gload_frag(base, ld, q, S, g), qf1 = gload_frag(base, ld, q, S, 4 + g);
            const bf16x8 gf0 = gload_frag(gG, H, q, S, g), gf1 = gload_frag(gG, H, q, S, 4 + g);
            const bf16x8 of0 = gload_frag(gO, H, q, S, g), of1 = gload_frag(gO, H, q, S, 4 + g);
            float dq_ = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) dq_ += (float)gf0[e] * (float)of0[e] + (float)gf1[e] * (float)of1[e];
            dq_ += __shfl_xor(dq_, 16, 64);
            dq_ += __shfl_xor(dq_, 32, 64);
            const float lq = Ls[q];
            f32x4 dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks * 32 >= S) break;
                f32x4 ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int krow = (2 * ks + t) * 16;
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sc = mfma16x32(row_frag(Ks, krow + i16, g), qf0, sc);
                    sc = mfma16x32(row_frag(Ks, krow + i16, 4 + g), qf1, sc);
                    dp = mfma16x32(row_frag(Vs, krow + i16, g), gf0, dp);
                    dp = mfma16x32(row_frag(Vs, krow + i16, 4 + g), gf1, dp);
                    const f32x4 kv4 = *reinterpret_cast<const f32x4*>(kvalid + krow + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = kv4[e] * __builtin_amdgcn_exp2f(sc[e] * (0.125f * LOG2E) - lq);
                        ds[t][e] = pe * (dp[e] - dq_);            // the 1/8 of dS is applied to dQ at the end
                    }
                }
                const bf16x8 dsb = cvt8(ds[0], ds[1]);
                const int r0a = (2 * ks) * 16 + 4 * g, r0b = (2 * ks + 1) * 16 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    dq[dt] = mfma16x32(tr_frag8(Ks, r0a, r0b, dt * 16, lane), dsb, dq[dt]);
            }
            if (q < S) {
                bf16* oq = dq_base + (size_t)q * ld;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<bf16x4*>(oq + dt * 16 + 4 * g) = cvt4(dq[dt] * f32x4{0.125f, 0.125f, 0.125f, 0.125f});
            }
        }
    }
}

// Backward in ONE block per (sample, head) for S_pad <= 192 (the benchmark's S = 185): Q, dO and K live in LDS once,
// the probabilities are computed once (5 matmul units instead of the 7 of the two-role kernel, one exp per score instead of
// two) and every operand is read from HBM once (145 MB instead of 236 MB per launch at configs[1]).
//   phase A: wave w owns key tile w (16 keys; V fragments straight from HBM): for all queries S = Q K^T, P, dP = dO V^T,
//            dS = P (dP - D);  dV += P^T dO, dK += dS^T Q;  dS^T goes to LDS as bf16 panels [q tile][key][16 q];
//   phase B: wave w owns query tile w: dQ = sum over keys dS K, the dS operand read back transposed from the panels.
// 2 NKS waves per block; LDS = 3 S_pad x 128 B + S_pad^2 x 2 B + 3 S_pad x 4 B  (144 KiB at S_pad = 192).
// F8MX (configs[4]): dq | dk | dv leave as e4m3 with one E8M0 scale per (row, 32 columns) -- the MX block format the block-scaled
// MFMA of feddat_gemm_fp8mx_nt consumes -- instead of 16-bit values: every (sample, head) block owns whole 64-column slices of
// its rows, i.e. two whole scale blocks per row and part, so the quantisation needs nothing from other blocks (a per-ROW scale
// would need the maximum over all 36 slices of the row: a second pass over dqkv).  Half the bytes of the 16-bit form.
template <int NKS, bool F8MX = false>
__global__ __launch_bounds__(NKS * 128) void attn_bwd_fused_kernel(const bf16* __restrict__ qkv,
                                                                     const uint8_t* __restrict__ kmask,
                                                                     const bf16* __restrict__ ctx,
                                                                     const float* __restrict__ lse,
                                                                     const bf16* __restrict__ dctx,
                                                                     bf16* __restrict__ dqkv, int S, int heads,
                                                                     const int npairs, const int dbg_in,
                                                                     uint8_t* __restrict__ dq8 = nullptr,
                                                                     uint8_t* __restrict__ dqsc = nullptr) {
    // dbg (tools/attn_ablate.py, debug flags bits 20..22; -DFEDDAT_ABLATE build only, the constant 0 otherwise; timing only):
    // 1 no global stores, 2 no phase-A arithmetic, 4 no phase B
    const int dbg = FD_ABL(dbg_in);
    constexpr int S_pad = NKS * 32, NT = NKS * 2, NTHR = NKS * 128;
    constexpr int XROW = 144;                      // bytes per row of a wave's output staging tile (128 + 16: aligned b128 reads)
    static_assert(2 * NKS * 16 * XROW <= 2 * S_pad * ROWB, "the waves' staging tiles fit in the Q + dO space");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Gs = Qs + S_pad * ROWB;
    char* Ks = Gs + S_pad * ROWB;
    char* Pn = Ks + S_pad * ROWB;                  // dS^T panels: [NT q tiles][S_pad keys][16 q] bf16 (32-byte rows)
    float* Dv = reinterpret_cast<float*>(Pn + NT * S_pad * 32);
    float* Ls = Dv + S_pad;
    float* kvalid = Ls + S_pad;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = heads * D;
    const long ld = 3L * H;
    const int g = lane >> 4, i16 = lane & 15;
    constexpr int NIT = S_pad * 8 / NTHR;          // 16-byte chunks of one [S_pad][64] operand per thread (2)
    static_assert(S_pad * 8 % NTHR == 0, "operand chunks divide over the block");
    // PERSISTENT over (sample, head) pairs: blockIdx.x, + gridDim.x, ...  Everything the next pair needs from HBM is
    // requested into REGISTERS while the current pair computes (LDS is full), in two instalments so that the register
    // file (168 per thread at 12 waves per CU) holds: K chunks, this wave's two V row fragments, LSE and key mask (18
    // registers) before phase A; Q, dO and O chunks (24) before phase B, whose own footprint is small.  The loads fly
    // during the phases and the load phase of every pair but a block's first disappears (r03 ablation: loads 17 us of
    // the 50 us launch, serial with compute).  Nothing inside the phases consumes a prefetched register, so nothing
    // there waits on vmcnt.
    struct Pre {
        bf16x8 q[NIT], k[NIT], gd[NIT], o[NIT], vf0, vf1;
        float ls, kv;
    } P;
    auto prefetch1 = [&](int pair) {
        const int b = pair / heads, h = pair - b * heads;
        const bf16* base = qkv + (size_t)b * S * ld + h * D;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NTHR, row = idx >> 3, chunk = idx & 7;
            P.k[it] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (row < S) P.k[it] = *reinterpret_cast<const bf16x8*>(base + H + (size_t)row * ld + chunk * 8);
        }
        const int key = wave * 16 + i16;
        P.vf0 = gload_frag(base + 2 * H, ld, key, S, g);
        P.vf1 = gload_frag(base + 2 * H, ld, key, S, 4 + g);
        P.ls = 0.f;
        P.kv = 0.f;
        if (tid < S_pad && tid < S) {
            P.ls = lse[((size_t)b * heads + h) * S + tid] * LOG2E;
            P.kv = (!kmask || kmask[(size_t)b * S + tid]) ? 1.f : 0.f;
        }
    };
    auto prefetch2 = [&](int pair) {
        const int b = pair / heads, h = pair - b * heads;
        const bf16* base = qkv + (size_t)b * S * ld + h * D;
        const bf16* gO = ctx + (size_t)b * S * H + h * D;
        const bf16* gG = dctx + (size_t)b * S * H + h * D;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NTHR, row = idx >> 3, chunk = idx & 7;
            const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
            P.q[it] = P.gd[it] = P.o[it] = z8;
            if (row < S) {
                P.q[it] = *reinterpret_cast<const bf16x8*>(base + (size_t)row * ld + chunk * 8);
                P.gd[it] = *reinterpret_cast<const bf16x8*>(gG + (size_t)row * H + chunk * 8);
                P.o[it] = *reinterpret_cast<const bf16x8*>(gO + (size_t)row * H + chunk * 8);
            }
        }
    };
    prefetch1(blockIdx.x);
    prefetch2(blockIdx.x);
  for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
    // the lane-derived LDS offsets below are loop invariant; hoisted out of the loop they would all be live across both
    // phases (hipcc did exactly that: 54 spilled registers).  Re-deriving them from a laundered thread id keeps them local.
    int tid_l = threadIdx.x;
    asm volatile("" : "+v"(tid_l));
    const int tid = tid_l, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i16 = lane & 15;
    const int b = pair / heads, h = pair - b * heads;
    bf16* dq_base = dqkv + (size_t)b * S * ld + h * D;
    // hand the prefetched operands to LDS; D[q] = sum_d dO[q][d] O[q][d]
    if (tid < S_pad) {
        Ls[tid] = P.ls;
        kvalid[tid] = P.kv;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTHR, row = idx >> 3, chunk = idx & 7;
        *reinterpret_cast<bf16x8*>(Qs + sw_off(row, chunk)) = P.q[it];
        *reinterpret_cast<bf16x8*>(Ks + sw_off(row, chunk)) = P.k[it];
        *reinterpret_cast<bf16x8*>(Gs + sw_off(row, chunk)) = P.gd[it];
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) part += (float)P.gd[it][e] * (float)P.o[it][e];
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (chunk == 0) Dv[row] = part;
    }
    const bf16x8 vf0 = P.vf0, vf1 = P.vf1;
    __syncthreads();
    const bool more = pair + (int)gridDim.x < npairs;
    if (more) prefetch1(pair + gridDim.x);

    // ---------------- phase A: this wave's key tile ----------------
    bf16x4 dk16[4], dv16[4];
    bool have_kv = false;
    {
        const int kt = wave;
        const int key = kt * 16 + i16;
        const bf16x8 kf0 = row_frag(Ks, key, g), kf1 = row_frag(Ks, key, 4 + g);
        const float kv = kvalid[key];
        f32x4 dv[4], dk[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (kt * 16 < S) {
#pragma unroll
            for (int qs = 0; qs < NKS; ++qs) {
                if (qs * 32 >= S || (dbg & 2)) break;
                f32x4 p[2], ds[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int qrow = (2 * qs + t) * 16;
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sc = mfma16x32(row_frag(Qs, qrow + i16, g), kf0, sc);       // S: rows = queries qrow + 4g + e, col = key i16
                    sc = mfma16x32(row_frag(Qs, qrow + i16, 4 + g), kf1, sc);
                    dp = mfma16x32(row_frag(Gs, qrow + i16, g), vf0, dp);
                    dp = mfma16x32(row_frag(Gs, qrow + i16, 4 + g), vf1, dp);
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qrow + 4 * g);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(Dv + qrow + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = kv * __builtin_amdgcn_exp2f(sc[e] * (0.125f * LOG2E) - l4[e]);
                        p[t][e] = pe;
                        ds[t][e] = pe * (dp[e] - d4[e]);          // the 1/8 of dS is applied to dK / dQ at the end
                    }
                    // dS^T panel of query tile 2 qs + t: row = key, 4 consecutive queries 4g .. 4g+3 (8 bytes)
                    // (the 8-byte block index is XOR-ed with bits 2-3 of the key: the 16 lanes of a store then cover all 32
                    // banks instead of 8 -- the plain layout was a 4-way conflict on every panel store)
                    *reinterpret_cast<bf16x4*>(Pn + ((2 * qs + t) * S_pad + key) * 32 + ((g ^ ((key >> 2) & 3)) << 3)) = cvt4(ds[t]);
                }
                const bf16x8 pb = cvt8(p[0], p[1]);
                const bf16x8 dsb = cvt8(ds[0], ds[1]);
                const int r0a = (2 * qs) * 16 + 4 * g, r0b = (2 * qs + 1) * 16 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = mfma16x32(tr_frag8(Gs, r0a, r0b, dt * 16, lane), pb, dv[dt]);
                    dk[dt] = mfma16x32(tr_frag8(Qs, r0a, r0b, dt * 16, lane), dsb, dk[dt]);
                }
            }
            have_kv = true;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dk16[dt] = cvt4(dk[dt] * f32x4{0.125f, 0.125f, 0.125f, 0.125f});
                dv16[dt] = cvt4(dv[dt]);
            }
        } else {
            // key tile beyond S: its panel rows must still be defined (phase B reads all S_pad keys)
#pragma unroll
            for (int qt = 0; qt < NT; ++qt)
                *reinterpret_cast<bf16x4*>(Pn + (qt * S_pad + key) * 32 + g * 8) = bf16x4{0, 0, 0, 0};
        }
    }
    __syncthreads();
    // Q and dO are dead from here on: each wave owns a [16 rows][64] bf16 tile (144-byte rows) in their LDS space through
    // which its dK / dV tile -- and, after phase B, its dQ tile -- is turned from the accumulator layout (8 bytes per lane,
    // 16 rows apart) into 16-byte row-contiguous stores: 2 store instructions per tile instead of 4, whole 128-byte rows
    // (the 8-byte form kept the CU's store path at ~8 B/clk: 10.7 of the launch's 46 us).
    char* xt = Qs + wave * (16 * XROW);            // one tile per wave, reused: a wave's LDS operations execute in order
    auto tile_out = [&](const bf16x4 (&t)[4], char* buf, bf16* gbase, int row0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<bf16x4*>(buf + i16 * XROW + dt * 32 + g * 8) = t[dt];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = p * 8 + (lane >> 3), c = lane & 7;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(buf + r * XROW + c * 16);
            if (!F8MX) {
                if (row0 + r < S && !(dbg & 1)) *reinterpret_cast<bf16x8*>(gbase + (size_t)(row0 + r) * ld + c * 8) = v;
            } else {
                // lanes c = 0..3 / 4..7 of a row hold one 32-column block each: its maximum over the 4 lanes, the E8M0 exponent
                // e = ceil(log2(amax / 448)) from the float's bits, codes = e4m3(v * 2^-e) (|.| <= 448 by construction)
                float f[8], amax = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
#pragma unroll
                for (int e = 0; e < 8; e += 2) amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(f[e]), __builtin_fabsf(f[e + 1])));
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                // (1 + 2^-20) / 448: the rounded quotient can only err upwards, so |v| * 2^-e <= 448 exactly and no clamp is needed
                const unsigned xb = __float_as_uint(amax * (1.00000095f / 448.0f));
                int ex = (int)((xb >> 23) & 0xffu) - 127 + ((xb & 0x7fffffu) ? 1 : 0);
                ex = amax > 0.f ? max(-126, min(126, ex)) : -126;
                const float inv = __uint_as_float((unsigned)(127 - ex) << 23);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] *= inv;
                int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
                int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
                if (row0 + r < S && !(dbg & 1)) {
                    // gbase addresses the 16-bit layout [rows, 3H] in elements: the same element offsets address the byte layout
                    const size_t off = (size_t)(gbase - dqkv) + (size_t)(row0 + r) * ld + c * 8;
                    *reinterpret_cast<int2*>(dq8 + off) = int2{lo, hi};
                    if ((c & 3) == 0) dqsc[off >> 5] = (uint8_t)(ex + 127);
                }
            }
        }
    };
    if (have_kv) {
        tile_out(dk16, xt, dq_base + H, wave * 16);
        tile_out(dv16, xt, dq_base + 2 * H, wave * 16);
    }
    if (more) prefetch2(pair + gridDim.x);

    // ---------------- phase B: this wave's query tile ----------------
    {
        const int qt = wave;
        if (qt * 16 < S && !(dbg & 4)) {
        const char* panel = Pn + qt * S_pad * 32;
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks * 32 >= S) break;
            // dS operand: lane (g, i16 = query) needs keys 32 ks + 4g .. +3 and 32 ks + 16 + 4g .. +3 of its query: a
            // transposed read of the 4 x 16 blocks of the panel (32-byte rows, 16 queries wide)
            const int i = lane & 15;
            auto tr = [&](int r0) {
                const int r = r0 + (i >> 2);
                const char* pp = panel + r * 32 + (((i & 3) ^ ((r >> 2) & 3)) << 3);
                return fd_ds_read_tr16(pp);
            };
            const bf16x4 a4 = tr(ks * 32 + 4 * g), b4 = tr(ks * 32 + 16 + 4 * g);
            const bf16x8 dsb = bf16x8{a4[0], a4[1], a4[2], a4[3], b4[0], b4[1], b4[2], b4[3]};
            const int r0a = ks * 32 + 4 * g, r0b = ks * 32 + 16 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16x32(tr_frag8(Ks, r0a, r0b, dt * 16, lane), dsb, dq[dt]);
        }
        bf16x4 dq16[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq16[dt] = cvt4(dq[dt] * f32x4{0.125f, 0.125f, 0.125f, 0.125f});
        tile_out(dq16, xt, dq_base, qt * 16);
        }
    }
    __syncthreads();        // every wave is done with this pair's LDS before the next pair's operands land in it
  }
}

template <typename K>
int set_lds(K kern, int bytes) {
    return fd_set_max_lds((const void*)kern, bytes) == FEDDAT_OK ? 0 : 1;
}

// hardware-semantics probe for tests: returns, per lane, tr_frag8(rows 4g.., rows 16+4g.., col block 16) of a
// 64x64 bf16 matrix staged exactly like the attention operands.
__global__ __launch_bounds__(256) void probe_tr16_kernel(const bf16* __restrict__ in, bf16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    load_head(in, 64, 64, 64, smem, threadIdx.x);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, g = lane >> 4;
        const bf16x8 v = tr_frag8(smem, 4 * g, 16 + 4 * g, 16, lane);
        *reinterpret_cast<bf16x8*>(out + lane * 8) = v;
    }
}

}  // namespace

extern "C" int feddat_attn_fwd(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int B, int S,
                               int heads, hipStream_t stream) {
    FD_CHECK_ARG(qkv && ctx && B > 0 && S > 0 && S <= 320 && heads > 0);
    const int nks = (S + 31) / 32;
    const int lds = nks * 32 * ROWB * 2 + nks * 32 * 4;
#define NKS_MAX_LDS_F(N) ((N) * 32 * ROWB * 2 + (N) * 32 * 4)
#define ATTN_FWD(N)                                                                                           \
    case N:                                                                                                   \
        if (set_lds(attn_fwd_kernel<N>, NKS_MAX_LDS_F(N))) return FEDDAT_ELAUNCH;                                        \
        hipLaunchKernelGGL(attn_fwd_kernel<N>, dim3(B * heads), dim3(256), lds, stream, (const bf16*)qkv,     \
                           key_mask, (bf16*)ctx, lse, S, heads);                                              \
        break;
    switch (nks) {
        ATTN_FWD(1) ATTN_FWD(2) ATTN_FWD(3) ATTN_FWD(4) ATTN_FWD(5) ATTN_FWD(6) ATTN_FWD(7) ATTN_FWD(8) ATTN_FWD(9)
        ATTN_FWD(10)
        default: return FEDDAT_EINVAL;
    }
#undef ATTN_FWD
    FD_LAUNCH_RET();
}

extern "C" int feddat_attn_bwd(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse,
                               const void* dctx, void* dqkv, int B, int S, int heads, hipStream_t stream) {
    FD_CHECK_ARG(qkv && ctx && lse && dctx && dqkv && B > 0 && S > 0 && S <= 320 && heads > 0);
    const int nks = (S + 31) / 32;
    if (nks <= 6 && !(fd_debug_flags() & 2)) {        // one block per (sample, head): operands and probabilities once
        const int sp = nks * 32;
        const int ldsf = 3 * sp * ROWB + sp * sp * 2 + 3 * sp * 4;
        // one block per CU (the LDS footprint allows no more), each walking (sample, head) pairs with the next pair's
        // operands prefetched into registers; debug bit 23: one block per pair (the r02 launch, nothing to prefetch)
        int n_cu = 0;
        if (fd_device_cus(&n_cu) != FEDDAT_OK) return FEDDAT_ELAUNCH;
        const int grid = (fd_debug_flags() & (1 << 23)) || B * heads < n_cu ? B * heads : n_cu;
#define ATTN_BWD_F(N)                                                                                          \
    case N:                                                                                                    \
        if (set_lds(attn_bwd_fused_kernel<N>, 3 * (N) * 32 * ROWB + (N) * 32 * (N) * 32 * 2 + 3 * (N) * 32 * 4))  \
            return FEDDAT_ELAUNCH;                                                                             \
        hipLaunchKernelGGL(attn_bwd_fused_kernel<N>, dim3(grid), dim3((N) * 128), ldsf, stream,                \
                           (const bf16*)qkv, key_mask, (const bf16*)ctx, lse, (const bf16*)dctx, (bf16*)dqkv, S, heads, \
                           B * heads, FD_ABL((fd_debug_flags() >> 20) & 7));                                          \
        break;
        switch (nks) {
            ATTN_BWD_F(1) ATTN_BWD_F(2) ATTN_BWD_F(3) ATTN_BWD_F(4) ATTN_BWD_F(5) ATTN_BWD_F(6)
            default: return FEDDAT_EINVAL;
        }
#undef ATTN_BWD_F
        FD_LAUNCH_RET();
    }
    const int lds = nks * 32 * ROWB * 2 + nks * 32 * 4 * 3;
#define NKS_MAX_LDS_B(N) ((N) * 32 * ROWB * 2 + (N) * 32 * 4 * 3)
#define ATTN_BWD(N)                                                                                           \
    case N:                                                                                                   \
        if (set_lds(attn_bwd_kernel<N>, NKS_MAX_LDS_B(N))) return FEDDAT_ELAUNCH;                                        \
        hipLaunchKernelGGL(attn_bwd_kernel<N>, dim3(B * heads, 2), dim3(256), lds, stream, (const bf16*)qkv,     \
                           key_mask, (const bf16*)ctx, lse, (const bf16*)dctx, (bf16*)dqkv, S, heads);        \
        break;
    switch (nks) {
        ATTN_BWD(1) ATTN_BWD(2) ATTN_BWD(3) ATTN_BWD(4) ATTN_BWD(5) ATTN_BWD(6) ATTN_BWD(7) ATTN_BWD(8) ATTN_BWD(9)
        ATTN_BWD(10)
        default: return FEDDAT_EINVAL;
    }
#undef ATTN_BWD
    FD_LAUNCH_RET();
}

// configs[4]: the same backward with dq | dk | dv written as MX-scaled e4m3 (attn_bwd_fused_kernel<N, true>): dq8 [B S, 3 H] bytes
// in the layout of the 16-bit dqkv, dq_scale [B S, 3 H / 32] E8M0 bytes.  Sequences of up to 192 tokens (the fused kernel).
extern "C" int feddat_attn_bwd_fp8mx(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse,
                                     const void* dctx, uint8_t* dq8, uint8_t* dq_scale, int B, int S, int heads,
                                     hipStream_t stream) {
    FD_CHECK_ARG(qkv && ctx && lse && dctx && dq8 && dq_scale && B > 0 && S > 0 && S <= 192 && heads > 0);
    FD_CHECK_ARG(((uintptr_t)dq8 & 7) == 0);
    const int nks = (S + 31) / 32;
    const int sp = nks * 32;
    const int ldsf = 3 * sp * ROWB + sp * sp * 2 + 3 * sp * 4;
    int n_cu = 0;
    if (fd_device_cus(&n_cu) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    const int grid = B * heads < n_cu ? B * heads : n_cu;
#define ATTN_BWD_MX(N)                                                                                               \
    case N:                                                                                                          \
        if (set_lds(attn_bwd_fused_kernel<N, true>, ldsf)) return FEDDAT_ELAUNCH;                                    \
        hipLaunchKernelGGL((attn_bwd_fused_kernel<N, true>), dim3(grid), dim3((N) * 128), ldsf, stream,              \
                           (const bf16*)qkv, key_mask, (const bf16*)ctx, lse, (const bf16*)dctx, (bf16*)dq8, S, heads, \
                           B * heads, 0, dq8, dq_scale);                                                             \
        break;
    switch (nks) {
        ATTN_BWD_MX(1) ATTN_BWD_MX(2) ATTN_BWD_MX(3) ATTN_BWD_MX(4) ATTN_BWD_MX(5) ATTN_BWD_MX(6)
        default: return FEDDAT_EINVAL;
    }
#undef ATTN_BWD_MX
    FD_LAUNCH_RET();
}

int fd_prepare_attn_kernels() {
#define PREP(N)                                                                          \
    if (set_lds(attn_fwd_kernel<N>, (N) * 32 * ROWB * 2 + (N) * 32 * 4)) return FEDDAT_ELAUNCH; \
    if (set_lds(attn_bwd_kernel<N>, (N) * 32 * ROWB * 2 + (N) * 32 * 4 * 3)) return FEDDAT_ELAUNCH;
    PREP(1) PREP(2) PREP(3) PREP(4) PREP(5) PREP(6) PREP(7) PREP(8) PREP(9) PREP(10)
#undef PREP
#define PREPF(N)                                                                                                      \
    if (set_lds(attn_bwd_fused_kernel<N>, 3 * (N) * 32 * ROWB + (N) * 32 * (N) * 32 * 2 + 3 * (N) * 32 * 4)) return FEDDAT_ELAUNCH;
    PREPF(1) PREPF(2) PREPF(3) PREPF(4) PREPF(5) PREPF(6)
#undef PREPF
#define PREPM(N)                                                                                                      \
    if (set_lds(attn_bwd_fused_kernel<N, true>, 3 * (N) * 32 * ROWB + (N) * 32 * (N) * 32 * 2 + 3 * (N) * 32 * 4)) return FEDDAT_ELAUNCH;
    PREPM(1) PREPM(2) PREPM(3) PREPM(4) PREPM(5) PREPM(6)
#undef PREPM
    return FEDDAT_OK;
}

extern "C" int feddat_probe_tr16(const void* in_bf16_64x64, void* out_bf16_64x8, hipStream_t stream) {
    FD_CHECK_ARG(in_bf16_64x64 && out_bf16_64x8);
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(256), 64 * ROWB, stream, (const bf16*)in_bf16_64x64,
                       (bf16*)out_bf16_64x8);
    FD_LAUNCH_RET();
}
