// K4: fused dual Pfeiffer adapter (DAT module), forward and backward.
// Reference: src/modeling/models/adapter.py:124-163 (single: 125-131; gating: 133-146, get_agg_out 118-122),
// called as adapter(h, h) from Adaptered_ViltOutput.forward (src/modeling/adaptered_output.py:77).
//
// One block (4 waves) owns 16 tokens; wave w owns features / output columns [192 w, 192 w + 192) of them.
//   * HBM <-> registers is ROW-CONTIGUOUS: a wave moves its [16 x 192] fp32 slice with 12 x 16-byte accesses per lane in
//     which consecutive lanes touch consecutive addresses (runs of 768 B).  The MFMA operand / accumulator layout wants
//     lane & 15 = token, i.e. 16 different rows per 16-lane group -- loading or storing in THAT layout makes every
//     wave instruction 64 separate 16-byte L1 accesses (measured: TCP_TOTAL_CACHE_ACCESSES 26-32 per VMEM instruction,
//     the L1 tag pipeline busy for the whole kernel), so the slice is transposed through a per-wave LDS tile instead
//     (800-byte rows: fragment reads conflict-free).
//   * The [16 x 48] bottleneck never reaches HBM: the down-projection is computed as Z^T[r, tok] so that its
//     accumulator layout (lane: token = lane & 15, four consecutive r) is already the operand layout of the
//     up-projection; each wave contracts its quarter of the features, partials are summed through LDS in a fixed order.
//   * The up-projection is computed as Y^T[c, tok]: every lane ends with 4 consecutive output columns of one token, the
//     residual comes from the fragment registers kept from the down-projection (x is read from HBM once).
//   * No store is issued before the last load: loads and stores share vmcnt and retire out of order with respect to
//     each other, so a store in flight turns every later wait into vmcnt(0) = an HBM write round trip.
// Adapter weights (72 KiB per matrix, bf16, fragment-major copies) are read through L1/L2.
// HBM-bound: 2 x T x 768 x 4 B algorithmic bytes per forward call (+ T x 768 x 2 B with the fused LayerNorm).
#include "common.hip.h"

namespace {

constexpr int H = 768, R = 48, NT = 3, KS = H / 32, CT = H / 16;
constexpr int WCOLS = H / 4;                  // columns per wave
constexpr int WCH = WCOLS / 4;                // 16-byte chunks per row of a wave's slice (48)
constexpr int NLD = 16 * WCH / 64;            // row-contiguous accesses per lane and slice (12)
constexpr int ROWF = 200;                     // padded LDS row (floats): fragment reads hit 64 distinct banks
constexpr int STG_WAVE = 16 * ROWF;           // floats of LDS per wave (12.5 KiB)

struct AdapterLaunch {
    feddat_adapter_seg seg[2];
    int nseg;
    int tiles0;  // number of 16-token tiles of segment 0
};

#define FD_COMPILER_FENCE() asm volatile("" ::: "memory")

// ---- row-contiguous slice <-> registers -------------------------------------------------------------------------
// access j of lane l covers chunk f = 64 j + l of the slice in row-major chunk order: row f / 48, chunk f % 48
__device__ __forceinline__ void slice_load(const float* __restrict__ src, int nvalid, int lane, f32x4 (&v)[NLD]) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
        const int rr = r < nvalid ? r : nvalid - 1;          // rows past the segment end re-read its last row
        v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)rr * H + c * 4);
    }
}
__device__ __forceinline__ void slice_to_lds(float* stg, int lane, const f32x4 (&v)[NLD]) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
        *reinterpret_cast<f32x4*>(stg + r * ROWF + c * 4) = v[j];
    }
}
// MFMA operand fragments of the slice: in k-step k lane group g supplies features 32k + 4g + (0..3) and
// 32k + 16 + 4g + (0..3) of token lane & 15 -- exactly the two float4 this lane needs again for the residual add of
// output tiles 2k and 2k+1 (accumulator layout: 4 consecutive columns 16 ct + 4g).  The weight operand uses the same slot
// permutation (adapter_pack).
__device__ __forceinline__ void frags_from_lds(const float* stg, int lane, f32x4 (&keep)[KS / 4][2]) {
    const float* p = stg + (lane & 15) * ROWF + 4 * (lane >> 4);
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) {
        keep[k][0] = *reinterpret_cast<const f32x4*>(p + k * 32);
        keep[k][1] = *reinterpret_cast<const f32x4*>(p + k * 32 + 16);
    }
}
__device__ __forceinline__ void frags_to_lds(float* stg, int lane, const f32x4 (&keep)[KS / 4][2]) {
    float* p = stg + (lane & 15) * ROWF + 4 * (lane >> 4);
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) {
        *reinterpret_cast<f32x4*>(p + k * 32) = keep[k][0];
        *reinterpret_cast<f32x4*>(p + k * 32 + 16) = keep[k][1];
    }
}

// Z^T[a][nt] += W[a] (rows r) x X^T (cols tok), contraction over the wave's quarter of the 768 features (k-steps
// ks0 .. ks0+5 of 32 features).  `w` is the slot-permuted fragment-major bf16 copy written by adapter_pack
// ([48][768], position 32q + 8g + 4*half + j): one contiguous 1 KiB load per wave and MFMA.
template <int NA>
__device__ __forceinline__ void down_proj(const bf16* const* w, int lane, int ks0, f32x4 (&z)[2][NT],
                                          const f32x4 (&keep)[KS / 4][2]) {
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) {
        const int q = ks0 + k;
        const bf16x8 xf = cvt8(keep[k][0], keep[k][1]);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(w[a] + ((nt * KS + q) * 64 + lane) * 8);
                z[a][nt] = mfma16x32(wf, xf, z[a][nt]);
            }
    }
}

// weight operand for the contraction over r = 48 with slots (g, j<4) -> r = 4g + j, (g, j>=4) -> r = 16 + 4g + j - 4
// (first MFMA) and (g, j<4) -> r = 32 + 4g + j, (g, j>=4) -> zero padding (second MFMA).  w is [rows, 48] bf16.
// Both products use the K=32 instruction: chaining v_mfma_f32_16x16x32_bf16 -> v_mfma_f32_16x16x16_bf16 on one
// accumulator returned stale values in accumulator registers 0-1 on gfx950 / ROCm 7.2 (measured), so the K=16
// form is not used anywhere.
__device__ __forceinline__ void load_w48(const bf16* w, int ct, int lane, bf16x8& w01, bf16x8& w2) {
    // fragment-major copy written by adapter_pack: [ct][lane][8] (r = 4g+j, 16+4g+j) then [ct][lane][4] (r = 32+4g+j)
    w01 = *reinterpret_cast<const bf16x8*>(w + (ct * 64 + lane) * 8);
    const bf16x4 c = *reinterpret_cast<const bf16x4*>(w + CT * 64 * 8 + (ct * 64 + lane) * 4);
    w2 = bf16x8{c[0], c[1], c[2], c[3], 0, 0, 0, 0};
}
__device__ __forceinline__ bf16x8 pad8(const f32x4 a) {
    return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], 0, 0, 0, 0};
}

// K-split partial bottleneck tiles: wave w parks its partials at the start of its OWN staging region (its slice has
// been consumed by then), slot s of NS at [s][lane] (f32x4); after the block barrier every wave sums the four regions
// in a fixed order.
template <int NS>
__device__ __forceinline__ void ksplit_store(float* stg_all, int wave, int lane, int s, const f32x4 v) {
    reinterpret_cast<f32x4*>(stg_all + wave * STG_WAVE)[s * 64 + lane] = v;
}
template <int NS>
__device__ __forceinline__ f32x4 ksplit_sum(const float* stg_all, int lane, int s) {
    f32x4 t = reinterpret_cast<const f32x4*>(stg_all)[s * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) t = t + reinterpret_cast<const f32x4*>(stg_all + w * STG_WAVE)[s * 64 + lane];
    return t;
}

struct LnFuse {            // optional LayerNorm of the adapter output (the next layer's layernorm_before), fused
    const float* gamma;    // null = off
    const float* beta;
    bf16* y16;             // [T, H] bf16: LN(out)
    float* stats;          // [T, 2]: mean, rstd (for the LN backward)
    float eps;
};

template <int NA>
__device__ __forceinline__ void fwd_body(const float* __restrict__ x, float* __restrict__ out,
                                         const feddat_adapter_seg& sg, int row0, float* stg_all, const LnFuse& ln,
                                         float* lnred, float* __restrict__ z_save) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int nvalid = (sg.row_end - row0) < 16 ? (sg.row_end - row0) : 16;
    float* stg = stg_all + wave * STG_WAVE;

    // 1. this wave's [16 x 192] slice of x: HBM -> registers (row-contiguous) -> LDS -> MFMA fragments
    f32x4 xk[KS / 4][2];
    {
        f32x4 v[NLD];
        slice_load(x + (size_t)(row0 + sg.x_row_delta) * H + wave * WCOLS, nvalid, lane, v);
        FD_COMPILER_FENCE();
        slice_to_lds(stg, lane, v);
    }
    frags_from_lds(stg, lane, xk);

    // 2. down-projection of this wave's feature quarter, K-split reduce
    const bf16* wd[2] = {(const bf16*)sg.wd[0], (const bf16*)sg.wd[NA - 1]};
    f32x4 z[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) z[a][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    down_proj<NA>(wd, lane, wave * (KS / 4), z, xk);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ksplit_store<NA * NT>(stg_all, wave, lane, a * NT + nt, z[a][nt]);
    __syncthreads();
    bf16x8 zb01[NA], zb2[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 t = ksplit_sum<NA * NT>(stg_all, lane, a * NT + nt);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sg.bd[a] + nt * 16 + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) z[a][nt][e] = fmaxf(t[e] + b4[e], 0.f);
            // relu(Wd x + bd), fp32 [T][2][48], saved for the backward (which then neither re-reads x nor repeats the
            // down-projection).  Wave nt stores r-tile nt right here: the one early store of the kernel costs the later
            // weight waits one L2 write acknowledgement, keeping 6 tiles alive to the end cost spills
            if (z_save && nt == wave && i16 < nvalid)
                *reinterpret_cast<f32x4*>(z_save + (size_t)(row0 + i16) * (2 * R) + a * R + nt * 16 + 4 * g) = z[a][nt];
        }
        zb01[a] = cvt8(z[a][0], z[a][1]);
        zb2[a] = pad8(z[a][2]);
    }

    // 3. up-projection of this wave's 12 column tiles; the results replace the kept x fragments
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) {
        const int ct = wave * (CT / 4) + k;
        const int c = ct * 16 + 4 * g;
        f32x4 o = xk[k >> 1][k & 1];           // x[row][c .. c+3]
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            bf16x8 w01, w2;
            load_w48((const bf16*)sg.wu[a], ct, lane, w01, w2);
            f32x4 y = mfma16x32(w01, zb01[a], f32x4{0.f, 0.f, 0.f, 0.f});
            y = mfma16x32(w2, zb2[a], y);
            const f32x4 bu4 = *reinterpret_cast<const f32x4*>(sg.bu[a] + c);
            const float sc = sg.scale[a];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += sc * (y[e] + bu4[e]);
        }
        xk[k >> 1][k & 1] = o;
    }

    // 4. LayerNorm statistics over the 768 outputs of each token: 48 per lane -> 4 lane groups (shuffles) -> 4 waves
    // (LDS, fixed order); two passes (mean, then centred second moment) like the stand-alone LN kernel
    float mean = 0.f, rstd = 0.f;
    f32x4 gam[3], bet[3];                      // read-back path: lane l handles chunks (l + 16 j) % 48, j = 0..2
    if (ln.gamma) {                            // uniform over the launch
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = (lane + 16 * j) % WCH;
            gam[j] = *reinterpret_cast<const f32x4*>(ln.gamma + wave * WCOLS + c * 4);
            bet[j] = *reinterpret_cast<const f32x4*>(ln.beta + wave * WCOLS + c * 4);
        }
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < CT / 4; ++k) {
            const f32x4 o = xk[k >> 1][k & 1];
            s1 += (o[0] + o[1]) + (o[2] + o[3]);
        }
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if (g == 0) lnred[wave * 16 + i16] = s1;
        __syncthreads();
        mean = ((lnred[i16] + lnred[16 + i16]) + (lnred[32 + i16] + lnred[48 + i16])) * (1.0f / H);
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CT / 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = xk[k >> 1][k & 1][e] - mean;
                s2 += d * d;
            }
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (g == 0) lnred[64 + wave * 16 + i16] = s2;
        __syncthreads();
        const float var = ((lnred[64 + i16] + lnred[80 + i16]) + (lnred[96 + i16] + lnred[112 + i16])) * (1.0f / H);
        rstd = rsqrtf(var + ln.eps);
        // per-wave copy of the row statistics for the read-back path (rows are indexed by f / 48 there, not by lane & 15)
        if (g == 0) {
            lnred[128 + wave * 32 + 2 * i16] = mean;
            lnred[128 + wave * 32 + 2 * i16 + 1] = rstd;
        }
    } else {
        __syncthreads();                       // every wave is done with the partials parked in the staging regions
    }

    // 5. outputs: fragment layout -> LDS -> row-contiguous stores (fp32 out; bf16 LN(out) computed on the way)
    frags_to_lds(stg, lane, xk);
    float* orow = out + (size_t)row0 * H + wave * WCOLS;
    if (!ln.gamma) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
            const f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * ROWF + c * 4);
            if (r < nvalid) *reinterpret_cast<f32x4*>(orow + (size_t)r * H + c * 4) = v;
        }
        return;
    }
    bf16* yrow = ln.y16 + (size_t)row0 * H + wave * WCOLS;
    f32x4 v[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
        v[j] = *reinterpret_cast<const f32x4*>(stg + r * ROWF + c * 4);
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
        if (r < nvalid) *reinterpret_cast<f32x4*>(orow + (size_t)r * H + c * 4) = v[j];
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
        const float m = lnred[128 + wave * 32 + 2 * r], rs = lnred[128 + wave * 32 + 2 * r + 1];
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (v[j][e] - m) * rs * gam[j % 3][e] + bet[j % 3][e];
        if (r < nvalid) *reinterpret_cast<bf16x4*>(yrow + (size_t)r * H + c * 4) = cvt4(y);
    }
    if (wave == 0 && g == 0 && i16 < nvalid && ln.stats) {
        ln.stats[2 * (size_t)(row0 + i16)] = mean;
        ln.stats[2 * (size_t)(row0 + i16) + 1] = rstd;
    }
}

__global__ __launch_bounds__(256, 3) void adapter_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             AdapterLaunch L, LnFuse ln, float* __restrict__ z_save) {
    __shared__ __attribute__((aligned(16))) float stg[4 * STG_WAVE];
    __shared__ float lnred[128 + 4 * 32];
    const int tile = blockIdx.x;
    const int s = tile < L.tiles0 ? 0 : 1;
    const int t = s ? tile - L.tiles0 : tile;
    const feddat_adapter_seg& sg = L.seg[s];
    const int row0 = sg.row_begin + t * 16;
    if (sg.n_adapters == 2) fwd_body<2>(x, out, sg, row0, stg, ln, lnred, z_save);
    else fwd_body<1>(x, out, sg, row0, stg, ln, lnred, z_save);
}

// ZS: the forward saved z = relu(Wd x + bd) for every adapter of the segment (feddat_adapter_fwd*'s z_save): the backward
// then reads dy and 192-384 B of z per token instead of dy and x, and runs one K=768 product per adapter instead of two.
template <int NA, bool ZS>
__device__ __forceinline__ void bwd_body(const float* __restrict__ x, const float* __restrict__ dy,
                                         float* __restrict__ dx, bf16* __restrict__ dx16, float* __restrict__ z_out,
                                         float* __restrict__ dz_out, const feddat_adapter_seg& sg, int row0,
                                         float* stg_all, const float* __restrict__ z_saved) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int nvalid = (sg.row_end - row0) < 16 ? (sg.row_end - row0) : 16;
    const bool valid = i16 < nvalid;
    const int row = row0 + i16;
    float* stg = stg_all + wave * STG_WAVE;

    const bf16* wd[2] = {(const bf16*)sg.wd[0], (const bf16*)sg.wd[NA - 1]};
    const bf16* wuT[2] = {(const bf16*)sg.wuT[0], (const bf16*)sg.wuT[NA - 1]};
    f32x4 z[2][NT], gr[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            z[a][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gr[a][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x4 dyk[KS / 4][2];
    f32x4 vd[NLD];
    if (ZS) {
        // 1. the dy slice leaves HBM row-contiguous (12 x 16 B per lane in flight), next to it the saved z of the 16 tokens
        slice_load(dy + (size_t)row0 * H + wave * WCOLS, nvalid, lane, vd);
        const float* zrow = z_saved + (size_t)(valid ? row : row0 + nvalid - 1) * (2 * R) + 4 * g;
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) z[a][nt] = *reinterpret_cast<const f32x4*>(zrow + a * R + nt * 16);
        FD_COMPILER_FENCE();
    } else {
        // 1. the x slice leaves HBM first (row-contiguous, 12 x 16 B per lane in flight); the dy slice is requested as
        // soon as x has been handed to LDS, so its latency hides behind the first down-projection
        f32x4 vx[NLD];
        slice_load(x + (size_t)(row0 + sg.x_row_delta) * H + wave * WCOLS, nvalid, lane, vx);
        FD_COMPILER_FENCE();
        // 2. recompute z = relu(Wd x + bd) (K-split over the waves)
        f32x4 xk[KS / 4][2];
        slice_to_lds(stg, lane, vx);
        slice_load(dy + (size_t)row0 * H + wave * WCOLS, nvalid, lane, vd);
        FD_COMPILER_FENCE();
        frags_from_lds(stg, lane, xk);
        down_proj<NA>(wd, lane, wave * (KS / 4), z, xk);
    }
    // 3. g = Wu^T dy (weight operand = WuT [48, 768]), K-split over the waves; the dy fragments are kept for the residual
    slice_to_lds(stg, lane, vd);
    frags_from_lds(stg, lane, dyk);
    down_proj<NA>(wuT, lane, wave * (KS / 4), gr, dyk);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (!ZS) ksplit_store<2 * NA * NT>(stg_all, wave, lane, a * NT + nt, z[a][nt]);
            ksplit_store<2 * NA * NT>(stg_all, wave, lane, NA * NT + a * NT + nt, gr[a][nt]);
        }
    __syncthreads();

    // 4. dz = scale * g * (z > 0); z and dz of the trainable slot are exported for the weight gradients -- at the END of
    // the kernel: no store may precede a load
    bf16x8 dzb01[NA], dzb2[NA];
    f32x4 z_keep = f32x4{0.f, 0.f, 0.f, 0.f}, dz_keep = z_keep;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const float sc = sg.scale[a];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 gs = ksplit_sum<2 * NA * NT>(stg_all, lane, NA * NT + a * NT + nt);
            f32x4 zz, dz;
            if (ZS) {
                zz = z[a][nt];
            } else {
                const f32x4 zs = ksplit_sum<2 * NA * NT>(stg_all, lane, a * NT + nt);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sg.bd[a] + nt * 16 + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) zz[e] = fmaxf(zs[e] + b4[e], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] = zz[e] > 0.f ? sc * gs[e] : 0.f;
            gr[a][nt] = dz;
            if (a == sg.train_slot && nt == wave) {
                z_keep = zz;
                dz_keep = dz;
            }
        }
        dzb01[a] = cvt8(gr[a][0], gr[a][1]);
        dzb2[a] = pad8(gr[a][2]);
    }
    const bool export_z = sg.train_slot >= 0 && wave < NT && valid && z_out;

    // 5. dx = dy + sum_a Wd[a]^T dz[a]   (weight operand = WdT [768, 48]); this wave's quarter of the columns; results
    // accumulate in place of the kept dy fragments
    if (dx) {
#pragma unroll
        for (int k = 0; k < CT / 4; ++k) {
            const int ct = wave * (CT / 4) + k;
            f32x4 o = dyk[k >> 1][k & 1];          // dy[row][c .. c+3]
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                bf16x8 w01, w2;
                load_w48((const bf16*)sg.wdT[a], ct, lane, w01, w2);
                f32x4 y = mfma16x32(w01, dzb01[a], f32x4{0.f, 0.f, 0.f, 0.f});
                y = mfma16x32(w2, dzb2[a], y);
                o = o + y;
            }
            dyk[k >> 1][k & 1] = o;
        }
        __syncthreads();                           // every wave is done with the partials parked in the staging regions
        frags_to_lds(stg, lane, dyk);
    }
    if (export_z) {
        *reinterpret_cast<f32x4*>(z_out + (size_t)row * R + wave * 16 + 4 * g) = z_keep;
        *reinterpret_cast<f32x4*>(dz_out + (size_t)row * R + wave * 16 + 4 * g) = dz_keep;
    }
    if (!dx) return;
    // 6. fragment layout -> LDS -> row-contiguous stores (fp32 dx and its bf16 copy for the next GEMM)
    float* orow = dx + (size_t)row0 * H + wave * WCOLS;
    f32x4 v[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
        v[j] = *reinterpret_cast<const f32x4*>(stg + r * ROWF + c * 4);
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
        if (r < nvalid) *reinterpret_cast<f32x4*>(orow + (size_t)r * H + c * 4) = v[j];
    }
    if (dx16) {
        bf16* brow = dx16 + (size_t)row0 * H + wave * WCOLS;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int f = j * 64 + lane, r = f / WCH, c = f - r * WCH;
            if (r < nvalid) *reinterpret_cast<bf16x4*>(brow + (size_t)r * H + c * 4) = cvt4(v[j]);
        }
    }
}

template <bool ZS>
__global__ __launch_bounds__(256, 3) void adapter_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dx, bf16* __restrict__ dx16,
                                                          float* __restrict__ z_out, float* __restrict__ dz_out,
                                                          AdapterLaunch L, const float* __restrict__ z_saved) {
    __shared__ __attribute__((aligned(16))) float stg[4 * STG_WAVE];
    const int tile = blockIdx.x;
    const int s = tile < L.tiles0 ? 0 : 1;
    const int t = s ? tile - L.tiles0 : tile;
    const feddat_adapter_seg& sg = L.seg[s];
    const int row0 = sg.row_begin + t * 16;
    if (sg.n_adapters == 2) bwd_body<2, ZS>(x, dy, dx, dx16, z_out, dz_out, sg, row0, stg, z_saved);
    else bwd_body<1, ZS>(x, dy, dx, dx16, z_out, dz_out, sg, row0, stg, z_saved);
}

__global__ __launch_bounds__(256) void adapter_pack_kernel(const float* __restrict__ wd, const float* __restrict__ wu,
                                                           bf16* __restrict__ wd16, bf16* __restrict__ wdT16,
                                                           bf16* __restrict__ wu16, bf16* __restrict__ wuT16,
                                                           long stride32, long stride16) {
    // blockIdx.y = adapter module (layer) of a strided batch
    wd += blockIdx.y * stride32; wu += blockIdx.y * stride32;
    wd16 += blockIdx.y * stride16; wdT16 += blockIdx.y * stride16;
    wu16 += blockIdx.y * stride16; wuT16 += blockIdx.y * stride16;
    // wd [R,H] -> wd16 [R,H] slot-permuted along H, wdT16 [H,R];  wu [H,R] -> wu16 [H,R], wuT16 [R,H] slot-permuted.
    // permutation of a feature index c (see down_proj): q = c / 32, half = (c % 32) / 16, g = (c % 16) / 4, j = c % 4
    //   -> position 32 q + 8 g + 4 half + j
    // Both operand copies are FRAGMENT-MAJOR: the 64 lanes of a wave read 64 consecutive 16-byte (8-byte) pieces, so
    // every weight load of the adapter kernels is one contiguous 1 KiB (512 B) burst.
    //   "down-type" copy of a [R][H] matrix (wd, wuT): piece ((nt * KS + q) * 64 + g * 16 + i16) = row nt * 16 + i16,
    //       permuted positions 32 q + 8 g + (0..7)
    //   "up-type" copy of a [H][R] matrix (wu, wdT): piece (ct * 64 + g * 16 + i16) of 8 = row ct * 16 + i16, columns
    //       4g..4g+3, 16+4g..16+4g+3; after CT * 64 of those, pieces of 4 = columns 32+4g..32+4g+3
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R * H) return;
    auto down_idx = [](int r, int c) {
        const int pc = (c & ~31) + (((c & 15) >> 2) << 3) + (((c >> 4) & 1) << 2) + (c & 3);
        const int q = pc >> 5, g = (pc & 31) >> 3, j = pc & 7;
        return ((((r >> 4) * KS + q) * 64 + g * 16 + (r & 15)) << 3) + j;
    };
    auto up_idx = [](int c, int r) {
        const int ct = c >> 4, i16 = c & 15, seg = r >> 4, g = (r & 15) >> 2, e = r & 3;
        const int lane = g * 16 + i16;
        return seg < 2 ? ((ct * 64 + lane) << 3) + seg * 4 + e : CT * 64 * 8 + ((ct * 64 + lane) << 2) + e;
    };
    {
        const int r = i / H, c = i - r * H;
        const bf16 v = (bf16)wd[i];
        wd16[down_idx(r, c)] = v;
        wdT16[up_idx(c, r)] = v;
    }
    {
        const int c = i / R, r = i - c * R;
        const bf16 v = (bf16)wu[i];
        wu16[up_idx(c, r)] = v;
        wuT16[down_idx(r, c)] = v;
    }
}

int prep_launch(const feddat_adapter_seg* segs, int nseg, int T, AdapterLaunch& L, int& tiles, bool bwd) {
    if (!segs || nseg < 1 || nseg > 2) return FEDDAT_EINVAL;
    L.nseg = nseg;
    tiles = 0;
    for (int s = 0; s < nseg; ++s) {
        const feddat_adapter_seg& sg = segs[s];
        if (sg.row_begin < 0 || sg.row_end > T || sg.row_end < sg.row_begin) return FEDDAT_EINVAL;
        if (sg.row_begin + sg.x_row_delta < 0 || sg.row_end + sg.x_row_delta > T) return FEDDAT_EINVAL;
        if (sg.n_adapters != 1 && sg.n_adapters != 2) return FEDDAT_EINVAL;
        for (int a = 0; a < sg.n_adapters; ++a) {
            if (!sg.wd[a] || !sg.wu[a] || !sg.bd[a] || !sg.bu[a]) return FEDDAT_EINVAL;
            if (bwd && (!sg.wdT[a] || !sg.wuT[a])) return FEDDAT_EINVAL;
        }
        L.seg[s] = sg;
        const int t = (sg.row_end - sg.row_begin + 15) / 16;
        if (s == 0) L.tiles0 = t;
        tiles += t;
    }
    if (nseg == 1) L.seg[1] = L.seg[0];
    return FEDDAT_OK;
}

// tools/ ablation: bits 16..23 of the debug flags = extra dynamic LDS in KiB (caps the resident blocks per CU)
int dbg_extra_lds() { return ((fd_debug_flags() >> 16) & 0xff) * 1024; }

}  // namespace

extern "C" int feddat_adapter_fwd(const float* x, float* out, int T, int Hd, int r, const feddat_adapter_seg* segs,
                                  int nseg, float* z_save, hipStream_t stream) {
    FD_CHECK_ARG(x && out && T > 0 && Hd == H && r == R);
    AdapterLaunch L;
    int tiles;
    const int rc = prep_launch(segs, nseg, T, L, tiles, false);
    if (rc) return rc;
    if (tiles == 0) return FEDDAT_OK;
    hipLaunchKernelGGL(adapter_fwd_kernel, dim3(tiles), dim3(256), dbg_extra_lds(), stream, x, out, L,
                       LnFuse{nullptr, nullptr, nullptr, nullptr, 0.f}, z_save);
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_fwd_ln(const float* x, float* out, int T, int Hd, int r, const feddat_adapter_seg* segs,
                                     int nseg, const float* ln_gamma, const float* ln_beta, float eps, void* y_bf16,
                                     float* stats, float* z_save, hipStream_t stream) {
    FD_CHECK_ARG(x && out && T > 0 && Hd == H && r == R && ln_gamma && ln_beta && y_bf16);
    AdapterLaunch L;
    int tiles;
    const int rc = prep_launch(segs, nseg, T, L, tiles, false);
    if (rc) return rc;
    if (tiles == 0) return FEDDAT_OK;
    hipLaunchKernelGGL(adapter_fwd_kernel, dim3(tiles), dim3(256), dbg_extra_lds(), stream, x, out, L,
                       LnFuse{ln_gamma, ln_beta, (bf16*)y_bf16, stats, eps}, z_save);
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_bwd(const float* x, const float* z_saved, const float* dy, float* dx, void* dx_bf16,
                                  float* z_out, float* dz_out, int T, int Hd, int r, const feddat_adapter_seg* segs,
                                  int nseg, hipStream_t stream) {
    FD_CHECK_ARG((x || z_saved) && dy && (dx || z_out) && T > 0 && Hd == H && r == R);
    FD_CHECK_ARG((z_out == nullptr) == (dz_out == nullptr));
    AdapterLaunch L;
    int tiles;
    const int rc = prep_launch(segs, nseg, T, L, tiles, true);
    if (rc) return rc;
    if (tiles == 0) return FEDDAT_OK;
    if (z_saved)
        hipLaunchKernelGGL(adapter_bwd_kernel<true>, dim3(tiles), dim3(256), dbg_extra_lds(), stream, x, dy, dx,
                           (bf16*)dx_bf16, z_out, dz_out, L, z_saved);
    else
        hipLaunchKernelGGL(adapter_bwd_kernel<false>, dim3(tiles), dim3(256), dbg_extra_lds(), stream, x, dy, dx,
                           (bf16*)dx_bf16, z_out, dz_out, L, z_saved);
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_pack(const float* wd, const float* wu, void* wd_bf16, void* wdT_bf16, void* wu_bf16,
                                   void* wuT_bf16, int Hd, int r, hipStream_t stream) {
    FD_CHECK_ARG(wd && wu && wd_bf16 && wdT_bf16 && wu_bf16 && wuT_bf16 && Hd == H && r == R);
    hipLaunchKernelGGL(adapter_pack_kernel, dim3((R * H + 255) / 256), dim3(256), 0, stream, wd, wu, (bf16*)wd_bf16,
                       (bf16*)wdT_bf16, (bf16*)wu_bf16, (bf16*)wuT_bf16, 0L, 0L);
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_pack_strided(const float* wd, const float* wu, long stride_f32, void* wd_bf16,
                                           void* wdT_bf16, void* wu_bf16, void* wuT_bf16, long stride_bf16, int n,
                                           int Hd, int r, hipStream_t stream) {
    FD_CHECK_ARG(wd && wu && wd_bf16 && wdT_bf16 && wu_bf16 && wuT_bf16 && Hd == H && r == R);
    FD_CHECK_ARG(n > 0 && n <= 65535 && stride_f32 >= 0 && stride_bf16 >= 0);
    hipLaunchKernelGGL(adapter_pack_kernel, dim3((R * H + 255) / 256, n), dim3(256), 0, stream, wd, wu, (bf16*)wd_bf16,
                       (bf16*)wdT_bf16, (bf16*)wu_bf16, (bf16*)wuT_bf16, stride_f32, stride_bf16);
    FD_LAUNCH_RET();
}
