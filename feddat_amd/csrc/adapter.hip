// K4: fused dual Pfeiffer adapter (DAT module), forward and backward.
// Reference: src/modeling/models/adapter.py:124-163 (single: 125-131; gating: 133-146, get_agg_out 118-122),
// called as adapter(h, h) from Adaptered_ViltOutput.forward (src/modeling/adaptered_output.py:77).
//
// One block (4 waves) owns 16 tokens.  The token rows stream straight from HBM into MFMA operand registers
// (fp32 -> bf16 in flight), the [16 x 48] bottleneck never reaches HBM: the down-projection is
// computed as Z^T[r, tok] so that its accumulator layout (lane: token = lane & 15, four consecutive r)
// is already the operand layout of the up-projection (contraction slots are paired (g, j) <-> (g, j),
// so the slot -> r permutation only has to be applied to the weight operand).  The up-projection is
// computed as Y^T[c, tok]: every lane ends with 4 consecutive output columns of one token ->
// 16-byte residual loads and stores.  Adapter weights (72 KiB per matrix, bf16) are read through L1/L2.
// HBM-bound: 2 x T x 768 x 4 B algorithmic bytes per call.
#include "common.hip.h"

namespace {

constexpr int H = 768, R = 48, NT = 3, KS = H / 32, CT = H / 16;

struct AdapterLaunch {
    feddat_adapter_seg seg[2];
    int nseg;
    int tiles0;  // number of 16-token tiles of segment 0
};

__device__ __forceinline__ bf16x8 load_x8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    return cvt8(a, b);
}

// Z^T[a][nt] += W[a] (rows r) x X^T (cols tok), contraction over the wave's quarter of the 768 features
// (k-steps ks0 .. ks0+5 of 32 features).  Contraction slots are permuted: in k-step q lane group g supplies features
// 32q + 4g + (0..3) and 32q + 16 + 4g + (0..3) -- exactly the two float4 this lane needs again for the residual add of
// output tiles 2q and 2q+1 (accumulator layout: 4 consecutive columns 16 ct + 4g), so the fp32 row values are kept in
// registers (`keep`) and x is read from HBM once.  The weight operand uses the same permutation: `w` is the
// slot-permuted bf16 copy written by adapter_pack ([48][768], position 32q + 8g + 4*half + j), one 16-byte load.
// All of a wave's HBM reads are issued up front (12 x 16-byte loads per lane = 12 KiB per wave in flight), ahead of any
// weight load or MFMA: the row data is what comes from HBM, everything else from L2, and a shallow load window would
// turn the kernel into a chain of HBM round trips.  The compiler fence keeps the loads from being sunk to their uses.
__device__ __forceinline__ void load_rows(const float* __restrict__ xrow, int lane, int ks0, f32x4 (&keep)[KS / 4][2]) {
    const int g = lane >> 4;
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) {
        const int q = ks0 + k;
        keep[k][0] = *reinterpret_cast<const f32x4*>(xrow + q * 32 + 4 * g);
        keep[k][1] = *reinterpret_cast<const f32x4*>(xrow + q * 32 + 16 + 4 * g);
    }
}
#define FD_COMPILER_FENCE() asm volatile("" ::: "memory")

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

// cross-wave sum of the K-split partial bottleneck tiles: part[w][slot][lane] (f32x4), slot = a * NT + nt
template <int NA>
__device__ __forceinline__ void ksplit_reduce(f32x4* part, int wave, int lane, f32x4 (&z)[2][NT]) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) part[(wave * (2 * NT) + a * NT + nt) * 64 + lane] = z[a][nt];
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 s = part[(0 * (2 * NT) + a * NT + nt) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) s = s + part[(w * (2 * NT) + a * NT + nt) * 64 + lane];
            z[a][nt] = s;
        }
}

// One block (4 waves) = 16 tokens.  Each wave contracts a quarter of the 768 features in the down-projection
// (partials summed through LDS, fixed order) and then owns a quarter of the 768 output columns of the
// up-projection: 4x shorter dependent chain per wave and 4x more waves in flight than one-wave-per-tile.
struct LnFuse {            // optional LayerNorm of the adapter output (the next layer's layernorm_before), fused
    const float* gamma;    // null = off
    const float* beta;
    bf16* y16;             // [T, H] bf16: LN(out)
    float* stats;          // [T, 2]: mean, rstd (for the LN backward)
    float eps;
};

template <int NA>
__device__ __forceinline__ void fwd_body(const float* __restrict__ x, float* __restrict__ out,
                                         const feddat_adapter_seg& sg, int row0, f32x4* part, const LnFuse& ln,
                                         float* lnred, const float* lngb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int row = row0 + i16;
    const bool valid = row < sg.row_end;
    const float* xrow = x + (size_t)((valid ? row : sg.row_end - 1) + sg.x_row_delta) * H;

    const bf16* wd[2] = {(const bf16*)sg.wd[0], (const bf16*)sg.wd[NA - 1]};
    f32x4 z[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) z[a][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 xk[KS / 4][2];
    load_rows(xrow, lane, wave * (KS / 4), xk);
    FD_COMPILER_FENCE();
    down_proj<NA>(wd, lane, wave * (KS / 4), z, xk);
    ksplit_reduce<NA>(part, wave, lane, z);

    bf16x8 zb01[NA], zb2[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sg.bd[a] + nt * 16 + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) z[a][nt][e] = fmaxf(z[a][nt][e] + b4[e], 0.f);
        }
        zb01[a] = cvt8(z[a][0], z[a][1]);
        zb2[a] = pad8(z[a][2]);
    }
    // Up-projection of this wave's 12 column tiles.  NO store is issued before the last load of the kernel: with a store
    // in flight every s_waitcnt in front of an MFMA has to be vmcnt(0) (loads and stores share the counter and retire
    // out of order with respect to each other), i.e. one full HBM write round trip per column tile -- that serial chain,
    // not bandwidth, was 2/3 of this kernel's time.  All outputs stay in registers (48 floats per lane) until the end.
    f32x4 oo[CT / 4];
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) {
        const int ct = wave * (CT / 4) + k;
        const int c = ct * 16 + 4 * g;
        f32x4 o = xk[k >> 1][k & 1];           // x[row][c .. c+3], kept from the down-projection
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
        oo[k] = o;
    }
    float* orow = out + (size_t)row * H + wave * (CT / 4) * 16 + 4 * g;
    if (!ln.gamma) {                            // uniform over the launch
        if (valid) {
#pragma unroll
            for (int k = 0; k < CT / 4; ++k) *reinterpret_cast<f32x4*>(orow + k * 16) = oo[k];
        }
        return;
    }
    // LayerNorm over the 768 outputs of each token: 48 per lane -> 4 lane groups (shuffles) -> 4 waves (LDS, fixed
    // order); two passes (mean, then centred second moment) like the stand-alone LN kernel
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) s1 += (oo[k][0] + oo[k][1]) + (oo[k][2] + oo[k][3]);
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    if (g == 0) lnred[wave * 16 + i16] = s1;
    __syncthreads();
    const float mean = ((lnred[i16] + lnred[16 + i16]) + (lnred[32 + i16] + lnred[48 + i16])) * (1.0f / H);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < CT / 4; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = oo[k][e] - mean;
            s2 += d * d;
        }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (g == 0) lnred[64 + wave * 16 + i16] = s2;
    __syncthreads();
    const float var = ((lnred[64 + i16] + lnred[80 + i16]) + (lnred[96 + i16] + lnred[112 + i16])) * (1.0f / H);
    const float rstd = rsqrtf(var + ln.eps);
    bf16x4 y16v[CT / 4];                        // gamma / beta come from LDS (staged at kernel start): no VMEM round trips
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) {
        const int c = (wave * (CT / 4) + k) * 16 + 4 * g;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(lngb + c);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lngb + H + c);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = (oo[k][e] - mean) * rstd * g4[e] + b4[e];
        y16v[k] = cvt4(y);
    }
    if (!valid) return;
    bf16* yrow = ln.y16 + (size_t)row * H + wave * (CT / 4) * 16 + 4 * g;
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) *reinterpret_cast<f32x4*>(orow + k * 16) = oo[k];
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) *reinterpret_cast<bf16x4*>(yrow + k * 16) = y16v[k];
    if (wave == 0 && g == 0 && ln.stats) {
        ln.stats[2 * (size_t)row] = mean;
        ln.stats[2 * (size_t)row + 1] = rstd;
    }
}

__global__ __launch_bounds__(256, 3) void adapter_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          AdapterLaunch L, LnFuse ln) {
    __shared__ __attribute__((aligned(16))) f32x4 part[4 * 2 * NT * 64];
    __shared__ float lnred[128];
    __shared__ __attribute__((aligned(16))) float lngb[2 * H];    // LN gamma | beta (visible after the k-split barrier)
    if (ln.gamma) {
        for (int i = threadIdx.x; i < H; i += 256) {
            lngb[i] = ln.gamma[i];
            lngb[H + i] = ln.beta[i];
        }
    }
    const int tile = blockIdx.x;
    const int s = tile < L.tiles0 ? 0 : 1;
    const int t = s ? tile - L.tiles0 : tile;
    const feddat_adapter_seg& sg = L.seg[s];
    const int row0 = sg.row_begin + t * 16;
    if (sg.n_adapters == 2) fwd_body<2>(x, out, sg, row0, part, ln, lnred, lngb);
    else fwd_body<1>(x, out, sg, row0, part, ln, lnred, lngb);
}

template <int NA>
__device__ __forceinline__ void bwd_body(const float* __restrict__ x, const float* __restrict__ dy,
                                         float* __restrict__ dx, bf16* __restrict__ dx16, float* __restrict__ z_out,
                                         float* __restrict__ dz_out, const feddat_adapter_seg& sg, int row0,
                                         f32x4* part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int row = row0 + i16;
    const bool valid = row < sg.row_end;
    const size_t rclamp = (size_t)(valid ? row : sg.row_end - 1);
    const float* xrow = x + (rclamp + sg.x_row_delta) * H;
    const float* dyrow = dy + rclamp * H;

    // 1. recompute z = relu(Wd x + bd);  2. g = Wu^T dy (weight operand = WuT [48, 768]); both K-split over the waves
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
    f32x4 xk[KS / 4][2], dyk[KS / 4][2];
    load_rows(xrow, lane, wave * (KS / 4), xk);
    load_rows(dyrow, lane, wave * (KS / 4), dyk);
    FD_COMPILER_FENCE();
    down_proj<NA>(wd, lane, wave * (KS / 4), z, xk);
    down_proj<NA>(wuT, lane, wave * (KS / 4), gr, dyk);
    ksplit_reduce<NA>(part, wave, lane, z);
    ksplit_reduce<NA>(part + 4 * 2 * NT * 64, wave, lane, gr);

    // 3. dz = scale * g * (z > 0); z and dz of the trainable slot are exported for the weight gradients -- at the END of
    // the kernel: no store may precede a load (see fwd_body)
    bf16x8 dzb01[NA], dzb2[NA];
    f32x4 z_keep = f32x4{0.f, 0.f, 0.f, 0.f}, dz_keep = z_keep;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const float sc = sg.scale[a];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(sg.bd[a] + nt * 16 + 4 * g);
            f32x4 zz, dz;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                zz[e] = fmaxf(z[a][nt][e] + b4[e], 0.f);
                dz[e] = zz[e] > 0.f ? sc * gr[a][nt][e] : 0.f;
            }
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

    // 4. dx = dy + sum_a Wd[a]^T dz[a]   (weight operand = WdT [768, 48]); this wave's quarter of the columns; results
    // accumulate in place of the kept dy values, all stores after the last weight load
    if (dx) {
#pragma unroll
        for (int k = 0; k < CT / 4; ++k) {
            const int ct = wave * (CT / 4) + k;
            f32x4 o = dyk[k >> 1][k & 1];          // dy[row][c .. c+3], kept from the Wu^T dy product
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
    }
    if (export_z) {
        *reinterpret_cast<f32x4*>(z_out + (size_t)row * R + wave * 16 + 4 * g) = z_keep;
        *reinterpret_cast<f32x4*>(dz_out + (size_t)row * R + wave * 16 + 4 * g) = dz_keep;
    }
    if (!dx || !valid) return;
    const size_t off = (size_t)row * H + wave * (CT / 4) * 16 + 4 * g;
#pragma unroll
    for (int k = 0; k < CT / 4; ++k) *reinterpret_cast<f32x4*>(dx + off + k * 16) = dyk[k >> 1][k & 1];
    if (dx16) {
#pragma unroll
        for (int k = 0; k < CT / 4; ++k) *reinterpret_cast<bf16x4*>(dx16 + off + k * 16) = cvt4(dyk[k >> 1][k & 1]);
    }
}

__global__ __launch_bounds__(256, 2) void adapter_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dx, bf16* __restrict__ dx16,
                                                          float* __restrict__ z_out, float* __restrict__ dz_out,
                                                          AdapterLaunch L) {
    __shared__ __attribute__((aligned(16))) f32x4 part[2 * 4 * 2 * NT * 64];
    const int tile = blockIdx.x;
    const int s = tile < L.tiles0 ? 0 : 1;
    const int t = s ? tile - L.tiles0 : tile;
    const feddat_adapter_seg& sg = L.seg[s];
    const int row0 = sg.row_begin + t * 16;
    if (sg.n_adapters == 2) bwd_body<2>(x, dy, dx, dx16, z_out, dz_out, sg, row0, part);
    else bwd_body<1>(x, dy, dx, dx16, z_out, dz_out, sg, row0, part);
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

}  // namespace

extern "C" int feddat_adapter_fwd(const float* x, float* out, int T, int Hd, int r, const feddat_adapter_seg* segs,
                                  int nseg, hipStream_t stream) {
    FD_CHECK_ARG(x && out && T > 0 && Hd == H && r == R);
    AdapterLaunch L;
    int tiles;
    const int rc = prep_launch(segs, nseg, T, L, tiles, false);
    if (rc) return rc;
    if (tiles == 0) return FEDDAT_OK;
    hipLaunchKernelGGL(adapter_fwd_kernel, dim3(tiles), dim3(256), 0, stream, x, out, L,
                       LnFuse{nullptr, nullptr, nullptr, nullptr, 0.f});
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_fwd_ln(const float* x, float* out, int T, int Hd, int r, const feddat_adapter_seg* segs,
                                     int nseg, const float* ln_gamma, const float* ln_beta, float eps, void* y_bf16,
                                     float* stats, hipStream_t stream) {
    FD_CHECK_ARG(x && out && T > 0 && Hd == H && r == R && ln_gamma && ln_beta && y_bf16);
    AdapterLaunch L;
    int tiles;
    const int rc = prep_launch(segs, nseg, T, L, tiles, false);
    if (rc) return rc;
    if (tiles == 0) return FEDDAT_OK;
    hipLaunchKernelGGL(adapter_fwd_kernel, dim3(tiles), dim3(256), 0, stream, x, out, L,
                       LnFuse{ln_gamma, ln_beta, (bf16*)y_bf16, stats, eps});
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_bwd(const float* x, const float* dy, float* dx, void* dx_bf16, float* z_out,
                                  float* dz_out, int T, int Hd, int r, const feddat_adapter_seg* segs, int nseg,
                                  hipStream_t stream) {
    FD_CHECK_ARG(x && dy && (dx || z_out) && T > 0 && Hd == H && r == R);
    FD_CHECK_ARG((z_out == nullptr) == (dz_out == nullptr));
    AdapterLaunch L;
    int tiles;
    const int rc = prep_launch(segs, nseg, T, L, tiles, true);
    if (rc) return rc;
    if (tiles == 0) return FEDDAT_OK;
    hipLaunchKernelGGL(adapter_bwd_kernel, dim3(tiles), dim3(256), 0, stream, x, dy, dx, (bf16*)dx_bf16, z_out, dz_out,
                       L);
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
