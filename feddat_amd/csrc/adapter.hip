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
#include <type_traits>

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
__global__ __launch_bounds__(256, 2) void adapter_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
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

// =====================================================================================================================
// Weight-stationary persistent forms (the ones the entry points launch for every shape).
//
// What bounded the one-tile-per-block kernels above (r02, PMC): 740 blocks in a single generation, each a serial chain
// load -> ~1 us of compute fed by 216-288 KB of weight fragments through L1/L2 -> store, with HBM idle during the compute
// of all of them: 2.7 TB/s.  Here ONE block of 4 waves (one per SIMD, 512 registers each) stays on every CU:
//   * wave w keeps its quarter of every weight matrix of its segment IN REGISTERS for the whole launch: the K-quarter
//     of the down-type matrix (6 k-steps x 3 r-tiles x 4 dwords = 72 per adapter) and the 12 column tiles of the up-type
//     matrix (12 x 6 = 72 per adapter): 288 dwords per lane for the gated segment; after the prologue the compute phase
//     touches no memory but LDS;
//   * 16-token tiles stream through two LDS buffers filled by LDS-DMA (global_load_lds, 1 KiB row pieces, no VGPR round
//     trip): tile t+1 is in flight while tile t is computed and stored;
//   * rows keep the 8-float padding (776-float rows) that makes the lane & 15 = token fragment reads conflict-free, and
//     every HBM access stays row-contiguous.
// Loads and stores share vmcnt and retire out of order with respect to each other, so the wait for tile t+1's DMA at the
// top of an iteration is vmcnt(0) and also covers the stores of tile t; with 128 KB of traffic per tile and CU (6.5 us at
// the CU's share of HBM) against ~1.2 us of compute that wait is where an HBM-bound kernel should be waiting.
// =====================================================================================================================
constexpr int ROWT = H + 8;                   // floats per LDS tile row
constexpr int TILE_F = 16 * ROWT;             // one 16-token tile: 49 664 B
constexpr int KSPL_F = 4 * 6 * 64 * 4;        // K-split partials: 4 waves x 6 slots x 64 lanes x f32x4 = 24 KiB
constexpr int ZT_F = 16 * 2 * R;              // saved-z tile of the backward: 6 KiB

struct WsLaunch {
    AdapterLaunch a;
    int g0;        // blocks [0, g0) walk segment 0, blocks [g0, gridDim.x) segment 1
    int dbg;       // tools/ ablations (feddat_set_debug_flags bits 24..26; 0 in production): 1 no global stores, 2 no compute,
                   // 4 no DMA beyond the first tile -- timing only, results are wrong
};

#define FD_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// After [tile 0 pieces][weight loads][tile 1 pieces] have been issued: tile 0 has landed once at most as many requests are
// outstanding as were issued BEHIND it.  The count is derived from the loop bounds of ws_load_weights (one vector load per
// fragment: NA x (NT*KS/4 down-type + CT/4 + 2*CT/8 up-type) = 42 per adapter) plus the 12 pieces of tile 1, capped at the
// counter's 63; it is only valid when tile 1 exists -- a block with a single tile has just the weight loads behind tile 0 and
// must wait for everything (ws_wait_tile0 below; block-uniform condition).
constexpr int WS_WEIGHT_LOADS_PER_ADAPTER = NT * (KS / 4) + CT / 4 + 2 * (CT / 8);
static_assert(WS_WEIGHT_LOADS_PER_ADAPTER == 42, "ws_load_weights changed: re-derive the counted wait");
template <int NA>
__device__ __forceinline__ void ws_wait_tile0(bool has_tile1) {
    constexpr int behind = NA * WS_WEIGHT_LOADS_PER_ADAPTER + 12;
    constexpr int cnt = behind < 63 ? behind : 63;
    if (has_tile1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(cnt) : "memory");
    else FD_WAIT_VM0();
}

// LDS-DMA piece: the 64 lanes' 16-byte loads land at LDS bytes [m0, m0 + 1024).  Issued as inline asm ON PURPOSE: for the
// builtin, hipcc's waitcnt pass assumes every later ds_read may alias the DMA's destination and puts vmcnt(0) in front of
// each of them -- the prefetch of the NEXT tile would be waited for by the first fragment read of THIS one.  The kernels
// wait by hand (FD_WAIT_VM0 + barrier before a tile buffer is read).  s_nop: SALU write of M0 -> LDS-DMA needs 1 wait state.
__device__ __forceinline__ void ws_dma16(const float* g, const float* lds_dst) {
    const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_dst));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(a) : "memory");
}
// one 16-token tile, HBM -> LDS by DMA: 48 row pieces of 1 KiB, 12 per wave (wave w: rows 4w .. 4w+3)
__device__ __forceinline__ void ws_tile_dma(const float* __restrict__ src, int nvalid, float* buf, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int r = wave * 4 + j / 3, part = j % 3;
        const int rr = r < nvalid ? r : nvalid - 1;       // rows past the segment end re-read its last row
        ws_dma16(src + (size_t)rr * H + part * 256 + lane * 4, buf + r * ROWT + part * 256);
    }
}
__device__ __forceinline__ void ws_frags_from_lds(const float* buf, int wave, int lane, f32x4 (&keep)[KS / 4][2]) {
    const float* p = buf + (lane & 15) * ROWT + wave * WCOLS + 4 * (lane >> 4);
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) {
        keep[k][0] = *reinterpret_cast<const f32x4*>(p + k * 32);
        keep[k][1] = *reinterpret_cast<const f32x4*>(p + k * 32 + 16);
    }
}
__device__ __forceinline__ void ws_frags_to_lds(float* buf, int wave, int lane, const f32x4 (&keep)[KS / 4][2]) {
    float* p = buf + (lane & 15) * ROWT + wave * WCOLS + 4 * (lane >> 4);
#pragma unroll
    for (int k = 0; k < KS / 4; ++k) {
        *reinterpret_cast<f32x4*>(p + k * 32) = keep[k][0];
        *reinterpret_cast<f32x4*>(p + k * 32 + 16) = keep[k][1];
    }
}

// register-resident weights of one wave: NA adapters x (K-quarter of the down-type matrix, 12 column tiles of the up-type one)
template <int NA>
struct WsWeights {
    bf16x8 dn[NA][NT][KS / 4];
    bf16x8 up01[NA][CT / 4];
    bf16x8 up2[NA][CT / 8];      // r = 32..47 of column tiles 2p (slots 0-3) and 2p+1 (slots 4-7)
};
template <int NA>
__device__ __forceinline__ void ws_load_weights(WsWeights<NA>& W, const void* const* dn, const void* const* up, int wave,
                                                int lane) {
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const bf16* d = (const bf16*)dn[a];
        const bf16* u = (const bf16*)up[a];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < KS / 4; ++k)
                W.dn[a][nt][k] = *reinterpret_cast<const bf16x8*>(d + ((nt * KS + wave * (KS / 4) + k) * 64 + lane) * 8);
#pragma unroll
        for (int k = 0; k < CT / 4; ++k) {
            const int ct = wave * (CT / 4) + k;
            W.up01[a][k] = *reinterpret_cast<const bf16x8*>(u + (ct * 64 + lane) * 8);
        }
#pragma unroll
        for (int p = 0; p < CT / 8; ++p) {
            const int ct = wave * (CT / 4) + 2 * p;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(u + CT * 64 * 8 + (ct * 64 + lane) * 4);
            const bf16x4 hi = *reinterpret_cast<const bf16x4*>(u + CT * 64 * 8 + ((ct + 1) * 64 + lane) * 4);
            W.up2[a][p] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    }
}
// the r = 32..47 remainder of the bottleneck as the B operand of a K = 32 MFMA whose A operand carries the remainders of
// TWO column tiles: [z | 0] selects the even tile's weights (slots 0-3), [0 | z] the odd tile's (slots 4-7)
__device__ __forceinline__ bf16x8 pad8hi(const f32x4 a) {
    return bf16x8{0, 0, 0, 0, (bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3]};
}

// block -> (segment, first tile, tile stride)
__device__ __forceinline__ void ws_walk(const WsLaunch& L, int& s, int& t0, int& tstep, int& ntiles) {
    const int b = blockIdx.x;
    s = b < L.g0 ? 0 : 1;
    t0 = s ? b - L.g0 : b;
    tstep = s ? (int)gridDim.x - L.g0 : L.g0;
    const feddat_adapter_seg& sg = L.a.seg[s];
    ntiles = (sg.row_end - sg.row_begin + 15) / 16;
}

template <int NA>
__device__ __forceinline__ void ws_fwd_body(const float* __restrict__ x, float* __restrict__ out,
                                            const feddat_adapter_seg& sg, int t0, int tstep, int ntiles, float* smem,
                                            const LnFuse& ln, float* __restrict__ z_save, const int dbg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    float* kspl = smem + 2 * TILE_F;
    float* lnred = kspl + KSPL_F;              // 256 floats
    float* sbu = lnred + 256;                  // [2][H]  up-projection biases
    float* sbd = sbu + 2 * H;                  // [2][64] down-projection biases
    float* sgam = sbd + 128;                   // [H] gamma, [H] beta of the fused LayerNorm
    float* sbet = sgam + H;
    if (t0 >= ntiles) return;

    // ---- prologue: small tables -> LDS, then (in this order in the memory pipe) tile 0, the weights -> registers, tile 1.
    // Only tile 0 is waited for by hand (ws_wait_tile0: everything older than the last 63 requests has landed, and more than 63
    // are younger than tile 0's pieces); the weight registers are waited for by the compiler where they are first used -- the
    // down-type matrix ahead of the down-projection, the up-type one ahead of the up-projection -- so the first tile's
    // down-projection runs while the second half of the 295 KB of weights is still on its way (the whole prologue used to
    // stand behind one vmcnt(0): 8.8 us of a 30 us launch).
    auto tile_dma = [&](int p) {
        const int t = t0 + p * tstep;
        if (t < ntiles && !(p && (dbg & 4))) {
            const int row0 = sg.row_begin + t * 16;
            const int nvalid = (sg.row_end - row0) < 16 ? (sg.row_end - row0) : 16;
            ws_tile_dma(x + (size_t)(row0 + sg.x_row_delta) * H, nvalid, smem + p * TILE_F, wave, lane);
        }
    };
    for (int i = threadIdx.x; i < NA * H; i += 256) sbu[i] = sg.bu[i / H][i % H];
    if (threadIdx.x < NA * R) sbd[(threadIdx.x / R) * 64 + threadIdx.x % R] = sg.bd[threadIdx.x / R][threadIdx.x % R];
    if (ln.gamma)
        for (int i = threadIdx.x; i < H; i += 256) {
            sgam[i] = ln.gamma[i];
            sbet[i] = ln.beta[i];
        }
    float sc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) sc[a] = sg.scale[a];
    tile_dma(0);
    WsWeights<NA> W;
    ws_load_weights<NA>(W, sg.wd, sg.wu, wave, lane);
    tile_dma(1);
    ws_wait_tile0<NA>(!(dbg & 4) && t0 + tstep < ntiles);
    __syncthreads();

    // Steady state, iteration t (its tile is in LDS): compute -> vmcnt(0) [tile t+1 landed; the stores of tile t-1, issued
    // a whole compute phase ago, are out] -> stores of tile t -> barrier -> DMA of tile t+2 into the buffer just freed.
    // The stores of tile t and the DMA of tile t+2 are in flight during the compute phase of tile t+1.
    int cur = 0;
    for (int t = t0; t < ntiles; t += tstep, cur ^= 1) {
        const int row0 = sg.row_begin + t * 16;
        const int nvalid = (sg.row_end - row0) < 16 ? (sg.row_end - row0) : 16;
        float* buf = smem + cur * TILE_F;      // (one base pointer + offset: keeps the accesses typed as LDS, not flat)

        f32x4 z_keep[NA];
        float mean = 0.f, rstd = 0.f;
#pragma unroll
        for (int a = 0; a < NA; ++a) z_keep[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(dbg & 2)) {
        // 1. + 2. down-projection of this wave's feature quarter (fragments straight from the tile buffer; the kept copy
        // of x for the residual is re-read from LDS in step 3 instead of living in 48 registers), K-split reduce
        const float* frag = buf + i16 * ROWT + wave * WCOLS + 4 * g;
        f32x4 z[NA][NT];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) z[a][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS / 4; ++k) {
            const bf16x8 xf = cvt8(*reinterpret_cast<const f32x4*>(frag + k * 32),
                                   *reinterpret_cast<const f32x4*>(frag + k * 32 + 16));
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) z[a][nt] = mfma16x32(W.dn[a][nt][k], xf, z[a][nt]);
        }
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                reinterpret_cast<f32x4*>(kspl)[(wave * 6 + a * NT + nt) * 64 + lane] = z[a][nt];
        __syncthreads();
        bf16x8 zb01[NA], zb2[NA][2];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 tsum = reinterpret_cast<const f32x4*>(kspl)[(a * NT + nt) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) tsum = tsum + reinterpret_cast<const f32x4*>(kspl)[(w * 6 + a * NT + nt) * 64 + lane];
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbd + a * 64 + nt * 16 + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[a][nt][e] = fmaxf(tsum[e] + b4[e], 0.f);
                if (nt == wave) z_keep[a] = z[a][nt];      // wave nt stores r-tile nt (with the other stores, below)
            }
            zb01[a] = cvt8(z[a][0], z[a][1]);
            zb2[a][0] = pad8(z[a][2]);
            zb2[a][1] = pad8hi(z[a][2]);
        }

        // 3. up-projection of this wave's 12 column tiles on top of the residual x (fragment k of the tile buffer)
        f32x4 xk[KS / 4][2];
#pragma unroll
        for (int k = 0; k < CT / 4; ++k) {
            const int c = (wave * (CT / 4) + k) * 16 + 4 * g;
            f32x4 o = *reinterpret_cast<const f32x4*>(frag + k * 16);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                f32x4 y = mfma16x32(W.up01[a][k], zb01[a], f32x4{0.f, 0.f, 0.f, 0.f});
                y = mfma16x32(W.up2[a][k >> 1], zb2[a][k & 1], y);
                const f32x4 bu4 = *reinterpret_cast<const f32x4*>(sbu + a * H + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += sc[a] * (y[e] + bu4[e]);
            }
            xk[k >> 1][k & 1] = o;
        }

        // 4. LayerNorm statistics with ONE block barrier: every wave reduces its own 192 columns of each token exactly (mean,
        // then centred second moment, both in registers + two shuffles), the four (mean_w, M2_w) pairs are combined with the
        // pairwise formula M2 = sum M2_w + 192 sum (mean_w - mean)^2 -- as robust as the two-pass form over all 768 columns
        if (ln.gamma) {
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < CT / 4; ++k) {
                const f32x4 o = xk[k >> 1][k & 1];
                s1 += (o[0] + o[1]) + (o[2] + o[3]);
            }
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            const float mw = s1 * (1.0f / WCOLS);
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < CT / 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = xk[k >> 1][k & 1][e] - mw;
                    s2 += d * d;
                }
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (g == 0) {
                lnred[wave * 32 + 2 * i16] = mw;
                lnred[wave * 32 + 2 * i16 + 1] = s2;
            }
            __syncthreads();
            const float m0 = lnred[2 * i16], m1 = lnred[32 + 2 * i16], m2 = lnred[64 + 2 * i16], m3 = lnred[96 + 2 * i16];
            mean = ((m0 + m1) + (m2 + m3)) * 0.25f;
            const float q0 = m0 - mean, q1 = m1 - mean, q2 = m2 - mean, q3 = m3 - mean;
            const float M2 = ((lnred[2 * i16 + 1] + lnred[32 + 2 * i16 + 1]) + (lnred[64 + 2 * i16 + 1] + lnred[96 + 2 * i16 + 1])) +
                             (float)WCOLS * ((q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3));
            rstd = rsqrtf(M2 * (1.0f / H) + ln.eps);
            if (g == 0) {
                lnred[128 + wave * 32 + 2 * i16] = mean;
                lnred[128 + wave * 32 + 2 * i16 + 1] = rstd;
            }
        }

        // 5. outputs: fragment layout -> LDS (this wave's own columns of the tile buffer) -> row-contiguous stores
        ws_frags_to_lds(buf, wave, lane, xk);
        }
        float* orow = out + (size_t)row0 * H + wave * WCOLS;
        const float* srow = buf + wave * WCOLS;
        f32x4 v[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int r = 4 * (j / 3) + g, c = 16 * (j % 3) + i16;
            v[j] = *reinterpret_cast<const f32x4*>(srow + r * ROWT + c * 4);
        }
        FD_WAIT_VM0();                         // next tile's DMA has landed; no store is issued before this point
        const bool st_on = !(dbg & 1);
        // full tiles (all of them when the segment is a multiple of 16 rows) store without per-access predicates: the exec-mask
        // branch around every guarded store keeps the scheduler from batching the 24 + stores of a wave
        auto emit = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            if (z_save && wave < NT && (FULL || i16 < nvalid)) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
                    *reinterpret_cast<f32x4*>(z_save + (size_t)(row0 + i16) * (2 * R) + a * R + wave * 16 + 4 * g) = z_keep[a];
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int r = 4 * (j / 3) + g, c = 16 * (j % 3) + i16;
                if (FULL || r < nvalid) *reinterpret_cast<f32x4*>(orow + (size_t)r * H + c * 4) = v[j];
            }
            if (ln.gamma) {
                bf16* yrow = ln.y16 + (size_t)row0 * H + wave * WCOLS;
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int r = 4 * (j / 3) + g, c = 16 * (j % 3) + i16;
                    const float m = lnred[128 + wave * 32 + 2 * r], rs = lnred[128 + wave * 32 + 2 * r + 1];
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(sgam + wave * WCOLS + c * 4);
                    const f32x4 bt = *reinterpret_cast<const f32x4*>(sbet + wave * WCOLS + c * 4);
                    f32x4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = (v[j][e] - m) * rs * gm[e] + bt[e];
                    if (FULL || r < nvalid) *reinterpret_cast<bf16x4*>(yrow + (size_t)r * H + c * 4) = cvt4(y);
                }
                if (wave == 0 && g == 0 && (FULL || i16 < nvalid) && ln.stats) {
                    ln.stats[2 * (size_t)(row0 + i16)] = mean;
                    ln.stats[2 * (size_t)(row0 + i16) + 1] = rstd;
                }
            }
        };
        if (st_on) {
            if (nvalid == 16) emit(std::true_type{});
            else emit(std::false_type{});
        }
        __syncthreads();                       // every wave: next tile visible, this buffer consumed
        if (t + 2 * tstep < ntiles && !(dbg & 4)) {
            const int rn = sg.row_begin + (t + 2 * tstep) * 16;
            const int nv = (sg.row_end - rn) < 16 ? (sg.row_end - rn) : 16;
            ws_tile_dma(x + (size_t)(rn + sg.x_row_delta) * H, nv, buf, wave, lane);
        }
    }
}

constexpr int WS_FWD_LDS = (2 * TILE_F + KSPL_F + 256 + 2 * H + 128 + 2 * H) * 4;

__global__ __launch_bounds__(256, 1) void adapter_fwd_ws_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                WsLaunch L, LnFuse ln, float* __restrict__ z_save) {
    extern __shared__ __attribute__((aligned(16))) float ws_smem[];
    int s, t0, tstep, ntiles;
    ws_walk(L, s, t0, tstep, ntiles);
    const feddat_adapter_seg& sg = L.a.seg[s];
    if (sg.n_adapters == 2) ws_fwd_body<2>(x, out, sg, t0, tstep, ntiles, ws_smem, ln, z_save, FD_ABL(L.dbg));
    else ws_fwd_body<1>(x, out, sg, t0, tstep, ntiles, ws_smem, ln, z_save, FD_ABL(L.dbg));
}


// saved-z tile of the backward: 16 tokens x [2][48] fp32 = 6 KiB, contiguous in HBM -> 6 DMA pieces
__device__ __forceinline__ void ws_ztile_dma(const float* __restrict__ zs, int nvalid, float* dst, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wave + 4 * j;
        if (piece < 6) {
            const int fo = piece * 256 + lane * 4, r = fo / (2 * R), c = fo - r * (2 * R);
            const int rr = r < nvalid ? r : nvalid - 1;
            ws_dma16(zs + (size_t)rr * (2 * R) + c, dst + piece * 256);
        }
    }
}

// Backward, z saved by the forward.  Register-resident: WuT (down-type: g = Wu^T dy) and WdT (up-type: dx = dy + Wd^T dz).
// Q8 (configs[4] only, its own instantiation so that the default kernel's register plan is untouched): dx also as e4m3 rows.
template <int NA, bool Q8>
__device__ __forceinline__ void ws_bwd_body(const float* __restrict__ dy, float* __restrict__ dx, bf16* __restrict__ dx16,
                                            float* __restrict__ z_out, float* __restrict__ dz_out,
                                            const feddat_adapter_seg& sg, int t0, int tstep, int ntiles, float* smem,
                                            const float* __restrict__ z_saved, const int dbg,
                                            unsigned char* __restrict__ dx8, float* __restrict__ dx8_scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    float* kspl = smem + 2 * TILE_F;
    float* ztile = kspl + KSPL_F;              // [2][ZT_F]
    float* amred = ztile + 2 * ZT_F;           // [4 waves][16 rows]: row maxima of |dx| (fp8 copy only)
    if (t0 >= ntiles) return;
    auto tile_dma = [&](int p) {       // prologue order and waits as in ws_fwd_body
        const int t = t0 + p * tstep;
        if (t < ntiles && !(p && (dbg & 4))) {
            const int row0 = sg.row_begin + t * 16;
            const int nvalid = (sg.row_end - row0) < 16 ? (sg.row_end - row0) : 16;
            ws_tile_dma(dy + (size_t)row0 * H, nvalid, smem + p * TILE_F, wave, lane);
            ws_ztile_dma(z_saved + (size_t)row0 * (2 * R), nvalid, ztile + p * ZT_F, wave, lane);
        }
    };
    if (Q8) {      // (the fp8 instantiation keeps the one-wait prologue: any other form costs it 7 spilled registers)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int t = t0 + p * tstep;
            if (t < ntiles && !(p && (dbg & 4))) {
                const int row0 = sg.row_begin + t * 16;
                const int nvalid = (sg.row_end - row0) < 16 ? (sg.row_end - row0) : 16;
                ws_tile_dma(dy + (size_t)row0 * H, nvalid, smem + p * TILE_F, wave, lane);
                ws_ztile_dma(z_saved + (size_t)row0 * (2 * R), nvalid, ztile + p * ZT_F, wave, lane);
            }
        }
    } else {
        tile_dma(0);
    }
    WsWeights<NA> W;
    ws_load_weights<NA>(W, sg.wuT, sg.wdT, wave, lane);
    if (!Q8) tile_dma(1);
    float sc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) sc[a] = sg.scale[a];
    const int train_slot = sg.train_slot;
    const bool exp_z = train_slot >= 0 && wave < NT && z_out;
    ws_wait_tile0<NA>(!Q8 && !(dbg & 4) && t0 + tstep < ntiles);
    __syncthreads();

    int cur = 0;      // pipeline order as in ws_fwd_body
    for (int t = t0; t < ntiles; t += tstep, cur ^= 1) {
        const int row0 = sg.row_begin + t * 16;
        const int nvalid = (sg.row_end - row0) < 16 ? (sg.row_end - row0) : 16;
        float* buf = smem + cur * TILE_F;
        float* zt = ztile + cur * ZT_F;

        f32x4 z_keep = f32x4{0.f, 0.f, 0.f, 0.f}, dz_keep = z_keep;
        if (!(dbg & 2)) {
        // g = Wu^T dy over this wave's feature quarter, K-split reduce
        const float* frag = buf + i16 * ROWT + wave * WCOLS + 4 * g;
        f32x4 gr[NA][NT];
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) gr[a][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KS / 4; ++k) {
            const bf16x8 df = cvt8(*reinterpret_cast<const f32x4*>(frag + k * 32),
                                   *reinterpret_cast<const f32x4*>(frag + k * 32 + 16));
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) gr[a][nt] = mfma16x32(W.dn[a][nt][k], df, gr[a][nt]);
        }
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                reinterpret_cast<f32x4*>(kspl)[(wave * 6 + a * NT + nt) * 64 + lane] = gr[a][nt];
        __syncthreads();

        // dz = scale * g * (z > 0); z and dz of the trainable slot go out for the weight gradients
        bf16x8 dzb01[NA], dzb2[NA][2];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 gs = reinterpret_cast<const f32x4*>(kspl)[(a * NT + nt) * 64 + lane];
#pragma unroll
                for (int w = 1; w < 4; ++w) gs = gs + reinterpret_cast<const f32x4*>(kspl)[(w * 6 + a * NT + nt) * 64 + lane];
                const f32x4 zz = *reinterpret_cast<const f32x4*>(zt + i16 * (2 * R) + a * R + nt * 16 + 4 * g);
                f32x4 dz;
#pragma unroll
                for (int e = 0; e < 4; ++e) dz[e] = zz[e] > 0.f ? sc[a] * gs[e] : 0.f;
                gr[a][nt] = dz;
                if (a == train_slot && nt == wave) {
                    z_keep = zz;
                    dz_keep = dz;
                }
            }
            dzb01[a] = cvt8(gr[a][0], gr[a][1]);
            dzb2[a][0] = pad8(gr[a][2]);
            dzb2[a][1] = pad8hi(gr[a][2]);
        }
        if (dx) {
            // dx = dy + sum_a Wd[a]^T dz[a] on this wave's 12 column tiles, written back over the dy fragments
#pragma unroll
            for (int k = 0; k < CT / 4; ++k) {
                f32x4 o = *reinterpret_cast<const f32x4*>(frag + k * 16);
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    f32x4 y = mfma16x32(W.up01[a][k], dzb01[a], f32x4{0.f, 0.f, 0.f, 0.f});
                    y = mfma16x32(W.up2[a][k >> 1], dzb2[a][k & 1], y);
                    o = o + y;
                }
                *reinterpret_cast<f32x4*>(buf + i16 * ROWT + wave * WCOLS + 4 * g + k * 16) = o;
            }
        }
        }
        f32x4 v[NLD];
        if (dx) {
            const float* srow = buf + wave * WCOLS;
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int r = 4 * (j / 3) + g, c = 16 * (j % 3) + i16;
                v[j] = *reinterpret_cast<const f32x4*>(srow + r * ROWT + c * 4);
            }
        }
        // configs[4]: dx also leaves as e4m3 rows + per-row scale amax / 448 (the A operand of the fp8 FFN2^T product): row
        // maxima over this wave's 192 columns (16 lanes per row: four shuffles), over the four waves through LDS
        float sc8[4] = {1.f, 1.f, 1.f, 1.f};
        if (Q8 && dx && dx8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float am = 0.f;
#pragma unroll
                for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) am = fmaxf(am, fabsf(v[3 * q + jj][e]));
                am = fmaxf(am, __shfl_xor(am, 1, 64));
                am = fmaxf(am, __shfl_xor(am, 2, 64));
                am = fmaxf(am, __shfl_xor(am, 4, 64));
                am = fmaxf(am, __shfl_xor(am, 8, 64));
                if (i16 == 0) amred[wave * 16 + 4 * q + g] = am;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * q + g;
                const float am = fmaxf(fmaxf(amred[r], amred[16 + r]), fmaxf(amred[32 + r], amred[48 + r]));
                sc8[q] = am > 0.f ? am * (1.0f / 448.0f) : 1.0f;
            }
        }
        FD_WAIT_VM0();                         // next tile's DMA has landed; no store is issued before this point
        const bool st_on = !(dbg & 1);
        auto emit = [&](auto full_tag) {        // full tiles: no per-access predicates (see ws_fwd_body)
            constexpr bool FULL = decltype(full_tag)::value;
            if (exp_z && (FULL || i16 < nvalid)) {
                *reinterpret_cast<f32x4*>(z_out + (size_t)(row0 + i16) * R + wave * 16 + 4 * g) = z_keep;
                *reinterpret_cast<f32x4*>(dz_out + (size_t)(row0 + i16) * R + wave * 16 + 4 * g) = dz_keep;
            }
            if (dx) {
                float* orow = dx + (size_t)row0 * H + wave * WCOLS;
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int r = 4 * (j / 3) + g, c = 16 * (j % 3) + i16;
                    if (FULL || r < nvalid) *reinterpret_cast<f32x4*>(orow + (size_t)r * H + c * 4) = v[j];
                }
                if (dx16) {
                    bf16* brow = dx16 + (size_t)row0 * H + wave * WCOLS;
#pragma unroll
                    for (int j = 0; j < NLD; ++j) {
                        const int r = 4 * (j / 3) + g, c = 16 * (j % 3) + i16;
                        if (FULL || r < nvalid) *reinterpret_cast<bf16x4*>(brow + (size_t)r * H + c * 4) = cvt4(v[j]);
                    }
                }
                if (Q8 && dx8) {
                    unsigned char* qrow = dx8 + (size_t)row0 * H + wave * WCOLS;
#pragma unroll
                    for (int j = 0; j < NLD; ++j) {
                        const int r = 4 * (j / 3) + g, c = 16 * (j % 3) + i16;
                        const float inv = 1.0f / sc8[j / 3];
                        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[j][0] * inv, v[j][1] * inv, 0, false);
                        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[j][2] * inv, v[j][3] * inv, pk, true);
                        if (FULL || r < nvalid) *reinterpret_cast<int*>(qrow + (size_t)r * H + c * 4) = pk;
                    }
                    if (wave == 0 && i16 == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (FULL || 4 * q + g < nvalid) dx8_scale[row0 + 4 * q + g] = sc8[q];
                    }
                }
            }
        };
        if (st_on) {
            if (nvalid == 16) emit(std::true_type{});
            else emit(std::false_type{});
        }
        __syncthreads();                       // every wave: next tile visible, this buffer consumed
        if (t + 2 * tstep < ntiles && !(dbg & 4)) {
            const int rn = sg.row_begin + (t + 2 * tstep) * 16;
            const int nv = (sg.row_end - rn) < 16 ? (sg.row_end - rn) : 16;
            ws_tile_dma(dy + (size_t)rn * H, nv, buf, wave, lane);
            ws_ztile_dma(z_saved + (size_t)rn * (2 * R), nv, zt, wave, lane);
        }
    }
}

constexpr int WS_BWD_LDS = (2 * TILE_F + KSPL_F + 2 * ZT_F + 64) * 4;

template <bool Q8>
__global__ __launch_bounds__(256, 1) void adapter_bwd_ws_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                bf16* __restrict__ dx16, float* __restrict__ z_out,
                                                                float* __restrict__ dz_out, WsLaunch L,
                                                                const float* __restrict__ z_saved,
                                                                unsigned char* __restrict__ dx8, float* __restrict__ dx8_scale) {
    extern __shared__ __attribute__((aligned(16))) float ws_smem[];
    int s, t0, tstep, ntiles;
    ws_walk(L, s, t0, tstep, ntiles);
    const feddat_adapter_seg& sg = L.a.seg[s];
    if (sg.n_adapters == 2) ws_bwd_body<2, Q8>(dy, dx, dx16, z_out, dz_out, sg, t0, tstep, ntiles, ws_smem, z_saved, FD_ABL(L.dbg), dx8,
                                               dx8_scale);
    else ws_bwd_body<1, Q8>(dy, dx, dx16, z_out, dz_out, sg, t0, tstep, ntiles, ws_smem, z_saved, FD_ABL(L.dbg), dx8, dx8_scale);
}

// host: blocks per segment, proportional to the tiles (at least one block per non-empty segment, one block per CU in all)
int ws_plan(const AdapterLaunch& A, int tiles, WsLaunch& L, int& grid) {
    int n_cu = 0;
    const int rc = fd_device_cus(&n_cu);
    if (rc) return rc;
    L.a = A;
    L.dbg = FD_ABL((fd_debug_flags() >> 24) & 7);
    const int t0 = A.tiles0, t1 = tiles - A.tiles0;
    grid = tiles < n_cu ? tiles : n_cu;
    if (t1 == 0) L.g0 = grid;
    else if (t0 == 0) L.g0 = 0;
    else {
        int g0 = (int)(((long)grid * t0 + tiles / 2) / tiles);
        g0 = g0 < 1 ? 1 : (g0 > grid - 1 ? grid - 1 : g0);
        L.g0 = g0;
    }
    return FEDDAT_OK;
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
int dbg_extra_lds() { return FD_ABL(((fd_debug_flags() >> 16) & 0xff) * 1024); }

int ws_launch_fwd(const float* x, float* out, const AdapterLaunch& A, int tiles, const LnFuse& ln, float* z_save,
                  hipStream_t stream) {
    WsLaunch W;
    int grid;
    int rc = ws_plan(A, tiles, W, grid);
    if (rc) return rc;
    rc = fd_set_max_lds((const void*)adapter_fwd_ws_kernel, WS_FWD_LDS);
    if (rc) return rc;
    hipLaunchKernelGGL(adapter_fwd_ws_kernel, dim3(grid), dim3(256), WS_FWD_LDS, stream, x, out, W, ln, z_save);
    FD_LAUNCH_RET();
}

}  // namespace

extern "C" int feddat_adapter_fwd(const float* x, float* out, int T, int Hd, int r, const feddat_adapter_seg* segs,
                                  int nseg, float* z_save, hipStream_t stream) {
    FD_CHECK_ARG(x && out && T > 0 && Hd == H && r == R);
    AdapterLaunch L;
    int tiles;
    const int rc = prep_launch(segs, nseg, T, L, tiles, false);
    if (rc) return rc;
    if (tiles == 0) return FEDDAT_OK;
    return ws_launch_fwd(x, out, L, tiles, LnFuse{nullptr, nullptr, nullptr, nullptr, 0.f}, z_save, stream);
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
    return ws_launch_fwd(x, out, L, tiles, LnFuse{ln_gamma, ln_beta, (bf16*)y_bf16, stats, eps}, z_save, stream);
}

static int adapter_bwd_launch(const float* x, const float* z_saved, const float* dy, float* dx, void* dx_bf16, void* dx_fp8,
                              float* dx_scale, float* z_out, float* dz_out, int T, int Hd, int r,
                              const feddat_adapter_seg* segs, int nseg, hipStream_t stream) {
    FD_CHECK_ARG((x || z_saved) && dy && (dx || z_out) && T > 0 && Hd == H && r == R);
    FD_CHECK_ARG((z_out == nullptr) == (dz_out == nullptr));
    FD_CHECK_ARG((dx_fp8 == nullptr) == (dx_scale == nullptr) && (!dx_fp8 || (z_saved && dx)));
    AdapterLaunch L;
    int tiles;
    const int rc = prep_launch(segs, nseg, T, L, tiles, true);
    if (rc) return rc;
    if (tiles == 0) return FEDDAT_OK;
    if (z_saved) {
        WsLaunch W;
        int grid;
        const int rc2 = ws_plan(L, tiles, W, grid);
        if (rc2) return rc2;
        if (dx_fp8) {
            const int rc3 = fd_set_max_lds((const void*)adapter_bwd_ws_kernel<true>, WS_BWD_LDS);
            if (rc3) return rc3;
            hipLaunchKernelGGL(adapter_bwd_ws_kernel<true>, dim3(grid), dim3(256), WS_BWD_LDS, stream, dy, dx, (bf16*)dx_bf16,
                               z_out, dz_out, W, z_saved, (unsigned char*)dx_fp8, dx_scale);
        } else {
            const int rc3 = fd_set_max_lds((const void*)adapter_bwd_ws_kernel<false>, WS_BWD_LDS);
            if (rc3) return rc3;
            hipLaunchKernelGGL(adapter_bwd_ws_kernel<false>, dim3(grid), dim3(256), WS_BWD_LDS, stream, dy, dx, (bf16*)dx_bf16,
                               z_out, dz_out, W, z_saved, (unsigned char*)nullptr, (float*)nullptr);
        }
    } else {
        hipLaunchKernelGGL(adapter_bwd_kernel<false>, dim3(tiles), dim3(256), dbg_extra_lds(), stream, x, dy, dx,
                           (bf16*)dx_bf16, z_out, dz_out, L, z_saved);
    }
    FD_LAUNCH_RET();
}

extern "C" int feddat_adapter_bwd(const float* x, const float* z_saved, const float* dy, float* dx, void* dx_bf16,
                                  float* z_out, float* dz_out, int T, int Hd, int r, const feddat_adapter_seg* segs,
                                  int nseg, hipStream_t stream) {
    return adapter_bwd_launch(x, z_saved, dy, dx, dx_bf16, nullptr, nullptr, z_out, dz_out, T, Hd, r, segs, nseg, stream);
}

extern "C" int feddat_adapter_bwd_fp8(const float* z_saved, const float* dy, float* dx, void* dx_fp8, float* dx_scale,
                                      float* z_out, float* dz_out, int T, int Hd, int r, const feddat_adapter_seg* segs,
                                      int nseg, hipStream_t stream) {
    FD_CHECK_ARG(z_saved && dx && dx_fp8 && dx_scale);
    return adapter_bwd_launch(nullptr, z_saved, dy, dx, nullptr, dx_fp8, dx_scale, z_out, dz_out, T, Hd, r, segs, nseg, stream);
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
