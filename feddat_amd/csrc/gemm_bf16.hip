// K1: bf16 MFMA GEMM, C[M,N] = A[M,K] * B[N,K]^T with fused epilogues, for the frozen ViLT linears
// (QKV / attention-out / FFN1 / FFN2 and their dX-only backward products).
//
// Layout: A and B are both K-contiguous ("NT").  nn.Linear weights [out,in] are used as B for the
// forward; the backward dX = dY * W uses the pre-transposed copy W^T [in,out] as B.
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 tiles; tiles stream
// HBM -> LDS with global_load_lds (16 B / lane, no VGPR round trip), double-buffered; the LDS image is
// XOR-swizzled on the SOURCE address (rule 21 of the CDNA guide) so the ds_read_b128 fragment reads
// are conflict-free.  MFMA operands are swapped (B-rows as the A operand) so every lane ends up
// holding 4 consecutive output columns of one row -> 8/16-byte epilogue loads and stores.
#include <stdlib.h>

#include <type_traits>

#include "common.hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmArgs {
    const bf16* A;
    const bf16* B;
    const float* bias;
    const float* resid;
    const bf16* aux;
    float* out_f32;
    bf16* out_bf16;
    bf16* out2_bf16;
    int M, N, K;
    int lda, ldb, ldr, ldaux, ldo32, ldo16, ldo2;
    int epi;
    int nostore;   // tools/ ablation (debug flag 16): the bf16 epilogues do everything but their global stores
    const float* sa;   // fp8 operands: per-row scale of A [M] and per-row scale of B [N] (acc * sa[m] * sw[n]); else null
    const float* sw;
    const uint8_t* amx;   // MXA kernels: E8M0 block scales of A, one byte per (row, 32 consecutive k): [M, ld_mx], 2^(byte - 127)
    int ld_mx;
};

__device__ __forceinline__ void stage_tile(const bf16* __restrict__ src, int ld, int row0, int rows_max, int k0,
                                           char* lds_tile, int wave, int lane) {
    // 16 wave-instructions of 8 rows x 128 B; wave w issues instructions 4w..4w+3.
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = (wave * 4 + i) * 8;
        const int r = rbase + (lane >> 3);
        const int pchunk = lane & 7;
        const int lchunk = pchunk ^ (r & 7);
        int gr = row0 + r;
        gr = gr < rows_max ? gr : rows_max - 1;
        const bf16* g = src + (size_t)gr * ld + k0 + lchunk * 8;
        __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(lds_tile + rbase * 128), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int lchunk) {
    const int p = lchunk ^ (row & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + p * 16);
}

// Whole-tile epilogue for one wave: NI x NJ accumulator tiles, lane holds C[m = mbase + 16 i + (lane & 15)]
// [n = nbase + 16 j + 4 (lane >> 4) + 0..3].  EPI is a template parameter and the switch sits OUTSIDE the loops, so
// all residual / aux loads of the tile are issued back-to-back instead of load-wait-store per element.
template <int EPI, int NI, int NJ>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& g, f32x4 (&acc)[NI][NJ], int mbase, int nbase, int m_end,
                                              int lane) {
    const int frow = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nbase + j * 16 + fg * 4;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
        f32x4 rr[NI];
        bf16x4 uu[NI];
        if (EPI == FEDDAT_EPI_RESID_F32) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int m = min(mbase + i * 16 + frow, m_end - 1);
                rr[i] = *reinterpret_cast<const f32x4*>(g.resid + (size_t)m * g.ldr + n);
            }
        }
        if (EPI == FEDDAT_EPI_MUL_DGELU) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int m = min(mbase + i * 16 + frow, m_end - 1);
                uu[i] = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)m * g.ldaux + n);
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int m = mbase + i * 16 + frow;
            if (m >= m_end) continue;
            const f32x4 v = acc[i][j] + bias4;
            if (EPI == FEDDAT_EPI_BF16) {
                *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(v);
            } else if (EPI == FEDDAT_EPI_RESID_F32) {
                *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) = v + rr[i];
            } else if (EPI == FEDDAT_EPI_GELU) {
                if (g.out2_bf16) *reinterpret_cast<bf16x4*>(g.out2_bf16 + (size_t)m * g.ldo2 + n) = cvt4(v);
                f32x4 a;
                a = gelu4_pk(v);
                *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(a);
            } else if (EPI == FEDDAT_EPI_MUL_DGELU) {
                f32x4 a;
                a = v * gelu_grad4_pk(f32x4{(float)uu[i][0], (float)uu[i][1], (float)uu[i][2], (float)uu[i][3]});
                *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(a);
            } else {
                *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) = v;
            }
        }
    }
}

template <int NI, int NJ>
__device__ __forceinline__ void tile_epilogue_dispatch(const GemmArgs& g, f32x4 (&acc)[NI][NJ], int mbase, int nbase,
                                                       int m_end, int lane) {
    switch (g.epi) {
        case FEDDAT_EPI_BF16: tile_epilogue<FEDDAT_EPI_BF16, NI, NJ>(g, acc, mbase, nbase, m_end, lane); break;
        case FEDDAT_EPI_RESID_F32: tile_epilogue<FEDDAT_EPI_RESID_F32, NI, NJ>(g, acc, mbase, nbase, m_end, lane); break;
        case FEDDAT_EPI_GELU: tile_epilogue<FEDDAT_EPI_GELU, NI, NJ>(g, acc, mbase, nbase, m_end, lane); break;
        case FEDDAT_EPI_MUL_DGELU: tile_epilogue<FEDDAT_EPI_MUL_DGELU, NI, NJ>(g, acc, mbase, nbase, m_end, lane); break;
        default: tile_epilogue<FEDDAT_EPI_F32, NI, NJ>(g, acc, mbase, nbase, m_end, lane); break;
    }
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (A,B) x 16 KiB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective remap: each XCD (bid % 8) walks a contiguous chunk of tiles so that the
    // blocks sharing an A row-panel hit the same L2.
    const int tiles_n = g.N / BN;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    stage_tile(g.A, g.lda, m0, g.M, 0, smem, wave, lane);
    stage_tile(g.B, g.ldb, n0, g.N, 0, smem + TILE_BYTES, wave, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    const int frow = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * TILE_BYTES;
        if (kt + 1 < nk) {
            char* nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
            stage_tile(g.A, g.lda, m0, g.M, (kt + 1) * BK, nxt, wave, lane);
            stage_tile(g.B, g.ldb, n0, g.N, (kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
        const char* ta = cur;
        const char* tb = cur + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = read_frag(ta, wm * 64 + i * 16 + frow, ks * 4 + fg);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = read_frag(tb, wn * 64 + j * 16 + frow, ks * 4 + fg);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x32(fb[j], fa[i], acc[i][j]);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
    }

    tile_epilogue_dispatch<4, 4>(g, acc, m0 + wm * 64, n0 + wn * 64, g.M, lane);
}


// ------------------------------------------------------------------------------------------------------------
// "mid": the products with few rows (M < 1024: the 800-row text streams and 128-row answer streams of the ALBEF path,
// its LM head).  With so few tiles a block's k-loop is a chain of dependent L2 / HBM round trips, so the kernel is built
// for LATENCY: 64 x 64 tiles (4x the blocks of the 128 x 128 kernel, whose 2-stage loop took one full round trip per
// 64-deep k-tile) and R k-tiles in flight per block IN REGISTERS (global_load_dwordx4 -> ds_write_b128 into a 2-slot LDS
// image; an LDS-DMA ring was tried first: the CU retires only ~1 KB of global_load_lds per 60-100 cycles, 0.46 us per
// k-tile with one block per CU and 1.2 us with two).  4 waves (2 x 2), wave tile 32 x 32; same swizzled LDS image,
// fragment layout and epilogues as gemm_nt_kernel.  The loop body is unrolled R times so the register sets are static;
// k-tiles past the end are clamped re-loads whose MFMAs are skipped.
constexpr int MID_R = 6;
constexpr int MID_LDS = 2 * (64 * 128 + 64 * 128);

__global__ __launch_bounds__(256, 2) void gemm_nt_mid_kernel(GemmArgs g) {
    constexpr int A_BYTES = 64 * 128, SLOT = 2 * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 slots x (A 8 KiB, B 8 KiB)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = g.N / 64;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    // every XCD owns a contiguous range of work items, i.e. whole rows of one operand's tiles and ALL tiles of the other, which
    // it pulls through its own L2: let that be the SMALLER operand (few rows against a 768..3072-row weight: an XCD owns a
    // range of N tiles and every M tile, so the weight matrix is fetched once per launch, not once per XCD)
    const int tiles_m = (g.M + 63) / 64;
    int tm, tn;
    if (g.M < g.N) {
        tn = wg / tiles_m;
        tm = wg - tn * tiles_m;
    } else {
        tm = wg / tiles_n;
        tn = wg - tm * tiles_n;
    }
    const int m0 = tm * 64, n0 = tn * 64;

    // this thread's four 16-byte pieces of a k-tile: rows r0, r0 + 32 of A and of B, logical chunk c
    const int r0 = tid >> 3, c = tid & 7;
    const bf16* pa[2];
    const bf16* pb[2];
    int lofs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = r0 + 32 * i;
        pa[i] = g.A + (size_t)min(m0 + r, g.M - 1) * g.lda + c * 8;
        pb[i] = g.B + (size_t)(n0 + r) * g.ldb + c * 8;
        lofs[i] = r * 128 + ((c ^ (r & 7)) << 4);
    }
    const int nk = g.K / BK;
    u32x4 regs[MID_R][4];
    auto load = [&](int set, int kt) {
        const int k0 = (kt < nk ? kt : nk - 1) * BK;
        regs[set][0] = *reinterpret_cast<const u32x4*>(pa[0] + k0);
        regs[set][1] = *reinterpret_cast<const u32x4*>(pa[1] + k0);
        regs[set][2] = *reinterpret_cast<const u32x4*>(pb[0] + k0);
        regs[set][3] = *reinterpret_cast<const u32x4*>(pb[1] + k0);
    };
    auto to_lds = [&](int set, int slot) {
        char* base = smem + slot * SLOT;
        *reinterpret_cast<u32x4*>(base + lofs[0]) = regs[set][0];
        *reinterpret_cast<u32x4*>(base + lofs[1]) = regs[set][1];
        *reinterpret_cast<u32x4*>(base + A_BYTES + lofs[0]) = regs[set][2];
        *reinterpret_cast<u32x4*>(base + A_BYTES + lofs[1]) = regs[set][3];
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int r = 0; r < MID_R; ++r) load(r, r);
    to_lds(0, 0);
    load(0, MID_R);
    __syncthreads();

    const int frow = lane & 15, fg = lane >> 4;
    for (int kt0 = 0; kt0 < nk; kt0 += MID_R) {
#pragma unroll
        for (int r = 0; r < MID_R; ++r) {
            const int kt = kt0 + r;
            // k-tile kt + 1 (register set (r + 1) % R) goes into the slot read one iteration ago, its set is refilled
            to_lds((r + 1) % MID_R, (kt + 1) & 1);
            load((r + 1) % MID_R, kt + 1 + MID_R);
            if (kt < nk) {
                const char* ta = smem + (kt & 1) * SLOT;
                const char* tb = ta + A_BYTES;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 fa[2], fb[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = read_frag(ta, wm * 32 + i * 16 + frow, ks * 4 + fg);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[j] = read_frag(tb, wn * 32 + j * 16 + frow, ks * 4 + fg);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16x32(fb[j], fa[i], acc[i][j]);
                }
            }
            __syncthreads();
        }
    }
    tile_epilogue_dispatch<2, 2>(g, acc, m0 + wm * 32, n0 + wn * 32, g.M, lane);
}


// ------------------------------------------------------------------------------------------------------------
// v2: persistent ping-pong kernel.  192 x 192 x 64 tile, 8 waves (4 x 2, 48 x 96 per wave), one block per CU.
//   * M-tile height is a runtime value <= 192 (rows beyond it are clamped on load, masked on store): with
//     bm = 185 = one sample's tokens the 11840-row activations of configs[1] split into exactly 64 x (N / 192) tiles
//     = 1.0 / 3.0 / 4.0 waves of the 256 CUs for N = 768 / 2304 / 3072 -- no tile-quantisation tail.
//   * Block b walks tiles b, b + grid, ...; the k-tile stream is continuous across tiles (the next tile's first
//     k-tiles are already in flight during the epilogue).
//   * The 8 waves form two groups (waves 0-3 / 4-7: one of each per SIMD) that run ONE PHASE APART.  Every k-tile is
//     an L phase (18 ds_read_b128 fragment reads + staging of a later k-tile) and a C phase (36 MFMAs) separated by
//     CU-wide barriers; group 1 executes one extra barrier up front, so on every SIMD one wave is in its MFMA phase
//     while its partner is in its load phase.  In lockstep, barrier + LDS + staging + MFMA time simply add up
//     (measured 62 us for 11840 x 768 x 3072 against 22 us of MFMA time).
//   * Operand staging is global_load_dwordx4 -> registers -> ds_write_b128 (one register set + two LDS stages: k-tile
//     it+2 in flight, it+1 in LDS, it being consumed), NOT LDS-DMA: a global_load_lds issue costs the wave ~90-150
//     cycles, six of them plus the 18 fragment reads made the L phase ~1000 cycles against 576 cycles of MFMA (the
//     phase length is max(L, C)); the register path issues in ~20 cycles per piece.  The staging loads stay plain
//     compiler-visible loads: hiding them in inline asm to keep two sets in flight let hipcc copy the destination
//     registers before the data had landed (rare garbage tiles).
//   * Epilogue staged through 5 KiB of LDS per wave: accumulators go down as [16 rows][48 cols] fp32 chunks and come
//     back row-contiguous, so residual / aux loads and the output stores are 96..192-byte runs per row instead of the
//     32-byte segments of the accumulator layout.
// The kernel is templated on WM = 16-row MFMA tiles per wave along M: WM = 3 -> 192 x 192 block tile (48 x 96 per wave),
// WM = 4 -> 256 x 192 (64 x 96 per wave: 48 MFMAs per 20 fragment reads instead of 36 per 18, for launches whose tile
// count then still fills whole rounds of the 256 CUs, i.e. N = 3072 at M = 11 840: 768 tiles = 3 rounds).
constexpr int V2_BN = 192;
constexpr int V2_TILE_B = V2_BN * BK * 2;     // 24 KiB
constexpr int V2_EPI_WAVE = 5120;             // staging bytes per wave
constexpr int V2_EPI_LD = 208;                // bytes per staged row (192 + 16 pad: conflict-free ds_write_b128)
template <int WM> struct V2Cfg {
    static constexpr int BM = 64 * WM;                       // 4 wave rows x 16 WM
    static constexpr int TILE_A = BM * BK * 2;               // 24 / 32 KiB
    static constexpr int STAGE = TILE_A + V2_TILE_B;         // 48 / 56 KiB
    static constexpr int EPI_OFF = 2 * STAGE;
    static constexpr int LDS = EPI_OFF + 8 * V2_EPI_WAVE;    // 136 / 152 KiB
    static constexpr int NP = WM + 3;                        // staging pieces per wave and k-tile (A: WM, B: 3)
};


struct GemmArgsV2 {
    GemmArgs g;
    int bm;        // rows per M tile (<= 64 WM)
    int tiles_m;
    int nx, tm_per, tn_per;   // XCD grid (8 / nx) x nx, tiles per XCD along M / N (see v2_tile_coords)
    int dbg;       // FEDDAT_GEMM_DEBUG ablation flags: 8 = skip epilogue; 32 / 64 = force WM 3 / 4; bits 8.. = block cap
};

// Tile id -> tile.  Consecutive workgroups land on consecutive XCDs (8 private 4 MiB L2s), so tile_id & 7 is the XCD.
//   nx == 1: every XCD owns a contiguous range of M tiles with all their N tiles (A read once per launch, the whole B
//            once per XCD and round -- fine while B fits the L2 or the launch is a single round);
//   nx == 2: the XCDs form a 4 (M) x 2 (N) grid; an XCD owns tm_per M tiles x tn_per N tiles, walked row-major, so its
//            32 concurrent tiles are 32 / tn_per M tiles x its half of B.  For N = 3072, K = 768 (B = 4.7 MB > L2,
//            4 rounds) this cut the measured L2 miss traffic of a launch from 161 MB to the ~55 MB the shape needs.
__device__ __forceinline__ void v2_tile_coords(const GemmArgsV2& a, int tile_id, int total, int& m0, int& n0,
                                               int& m_last) {
    const int tiles_n = a.g.N / V2_BN;
    int tm, tn;
    if (a.nx == 1) {
        const int q = total >> 3, r8 = total & 7, xcd = tile_id & 7;
        const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (tile_id >> 3);
        tm = wg / tiles_n;
        tn = wg - tm * tiles_n;
    } else {
        const int xcd = tile_id & 7, li = tile_id >> 3;
        const int xm = xcd / a.nx, xn = xcd - xm * a.nx;
        const int lm = li / a.tn_per;
        tm = xm * a.tm_per + lm;
        tn = xn * a.tn_per + (li - lm * a.tn_per);
    }
    m0 = tm * a.bm;
    n0 = tn * V2_BN;
    m_last = min(m0 + a.bm, a.g.M) - 1;
}

// per-lane source pointers (k = 0) of this wave's 6 staging pieces of a tile: A pieces 3w..3w+2, B pieces 3w..3w+2;
// a piece = 8 rows x 128 B, lane l -> row l >> 3, 16-byte chunk l & 7 (full 128-byte lines per row)
template <int WM>
__device__ __forceinline__ void v2_piece_ptrs(const GemmArgs& g, int m0, int m_last, int n0, int wave, int lane,
                                              const bf16* (&pa)[WM], const bf16* (&pb)[3]) {
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int r = (wave * WM + i) * 8 + (lane >> 3);
        pa[i] = g.A + (size_t)min(m0 + r, m_last) * g.lda + (lane & 7) * 8;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = (wave * 3 + i) * 8 + (lane >> 3);
        pb[i] = g.B + (size_t)min(n0 + r, g.N - 1) * g.ldb + (lane & 7) * 8;
    }
}

// bf16-output epilogues (plain, GELU + u, . gelu'(aux)) of one wave tile ([16 WM] x 96): every global access is 16 bytes
// per lane.  All arithmetic happens in the accumulator layout (bias, GELU); a [16 rows][96 cols] slab goes to the wave's
// LDS buffer as bf16 and comes back as 192-byte row runs (12 lanes x 8 bf16), i.e. 3 x global_store_dwordx4 per slab
// instead of 6 x dwordx2 (the 8-byte form was store-issue-bound at ~9 B/clk/CU: 9.5 us of every 27 us FFN1 tile).
// The aux operand of MUL_DGELU takes the opposite way: 16-byte row-contiguous loads -> LDS -> accumulator layout.
constexpr int V2_EPI_LD16 = 208;              // bytes per staged bf16 row (96 x 2 + 16 pad)
constexpr int V2_EPI_LD8 = 112;               // bytes per staged row of 8-bit gelu' codes (96 + 16 pad); 22 rows are addressable
// SINK (v3: one wave per SIMD, nothing else to hide a stall): stores of rows outside the tile go to a dummy line
// instead of being predicated, so the whole epilogue is one basic block the scheduler can interleave.
// Streaming hints on the heavy epilogues' HBM streams (the 16-bit / code outputs that nothing in this launch reads back, the
// code input that is read once): -DFD_EPI_NT builds them as non-temporal accesses so that they do not displace the B half an
// XCD's L2 holds for the whole launch (tools/ A/B: scripts/ab_r05_nt.sh; results in DESIGN.md section 7d).
#ifdef FD_EPI_NT
#define FD_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define FD_STREAM_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define FD_STREAM_STORE(ptr, val) (*(ptr) = (val))
#define FD_STREAM_LOAD(ptr) (*(ptr))
#endif
__device__ uint4 fd_epi_sink[64];
template <int EPI, int WM, bool SINK = false>
__device__ __forceinline__ void v2_epilogue_bf16(const GemmArgs& g, f32x4 (&acc)[WM][6], char* stg, int mbase, int nbase,
                                                 int m_end, int lane) {
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fg = lane >> 4;
    constexpr bool G8OUT_ = EPI == FEDDAT_EPI_GELU_G8 || EPI == FEDDAT_EPI_MUL_G8 || EPI == FEDDAT_EPI_GELU_G8_F8 ||
                            EPI == FEDDAT_EPI_MUL_G8_F8;      // column constants in LDS
    // GELU_G8 (two polynomials per element) / MUL_G8 (its codes in flight) have no registers for per-column constants at
    // 256-row tiles (57 / 26 spilled, and scratch traffic in the k-loop breaks its counted vmcnt waits): their bias / fp8
    // channel scales wait in the wave's staging buffer behind the slab and are re-read per row group
    constexpr int COLC_OFF = 16 * V2_EPI_LD16;
    static_assert(COLC_OFF + 2 * 384 <= V2_EPI_WAVE, "column constants fit behind the staged slab");
    f32x4 bias4[G8OUT_ ? 1 : 6];
    if (G8OUT_) {
        if (lane < 24) {
            *reinterpret_cast<f32x4*>(stg + COLC_OFF + lane * 16) =
                g.bias ? *reinterpret_cast<const f32x4*>(g.bias + nbase + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (g.sw) *reinterpret_cast<f32x4*>(stg + COLC_OFF + 384 + lane * 16) = *reinterpret_cast<const f32x4*>(g.sw + nbase + lane * 4);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 6; ++j)
            bias4[j] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + nbase + j * 16 + fg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int srow[3], sc8[3];                       // row-contiguous slots of this lane: (row, 8-column group) = divmod(64 p + lane, 12)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int idx = p * 64 + lane;
        srow[p] = idx / 12;
        sc8[p] = idx - srow[p] * 12;
    }
    // 8-bit gelu' codes (GELU_G8 / MUL_G8): a slab is [16 rows][96 bytes] = 96 x 16-byte chunks, one per lane + 32
    constexpr bool G8OUT = EPI == FEDDAT_EPI_GELU_G8 || EPI == FEDDAT_EPI_GELU_G8_F8;
    constexpr bool G8IN = EPI == FEDDAT_EPI_MUL_G8 || EPI == FEDDAT_EPI_MUL_G8_F8;
    constexpr bool F8OUT = EPI == FEDDAT_EPI_GELU_G8_F8 || EPI == FEDDAT_EPI_MUL_G8_F8;      // the main output leaves as e4m3
    int r8[2], c8[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = p * 64 + lane;          // idx >= 96 (p = 1, lanes 32..63): staged beyond the slab, never used
        r8[p] = idx / 6;
        c8[p] = idx - r8[p] * 6;
    }
    char* wr8 = stg + frow * V2_EPI_LD8 + fg * 4;           // accumulator-layout position of 4 codes: row frow, cols 16 j + 4 fg
    u32x4 ux8[G8IN ? WM : 1][2];
    auto aux8_load = [&](int i) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const unsigned mc = (unsigned)min(mbase + i * 16 + r8[p], m_end - 1);
            const unsigned off = mc * (unsigned)g.ldaux + (unsigned)(nbase + c8[p] * 16);
            ux8[i][p] = FD_STREAM_LOAD(reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(g.aux) + off));
        }
    };
    if (G8IN) {
#pragma unroll
        for (int i = 0; i < WM; ++i) aux8_load(i);
    }
    bf16x8 ux[G8IN ? 1 : WM][3];
    auto aux_load = [&](int i) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned mc = (unsigned)min(mbase + i * 16 + srow[p], m_end - 1);
            const unsigned off = (mc * (unsigned)g.ldaux + (unsigned)(nbase + sc8[p] * 8)) * 2u;
            ux[i][p] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(g.aux) + off);
        }
    };
    constexpr int AUX_AHEAD = WM < 4 ? WM : 4; // row groups of aux in flight: 12 loads per lane (all of them for WM <= 4)
    if (EPI == FEDDAT_EPI_MUL_DGELU) {
#pragma unroll
        for (int i = 0; i < AUX_AHEAD; ++i) aux_load(i);
    }
    char* wr = stg + frow * V2_EPI_LD16 + fg * 8;          // accumulator-layout position: row frow, cols 16 j + 4 fg
    auto put = [&](bf16* dst, int ld, int i, const f32x4 (&val)[6]) {
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<bf16x4*>(wr + j * 32) = cvt4(val[j]);
        bf16x8 v[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) v[p] = *reinterpret_cast<const bf16x8*>(stg + srow[p] * V2_EPI_LD16 + sc8[p] * 16);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int m = mbase + i * 16 + srow[p];
            if (SINK) {
                bf16* o = m < m_end ? dst + (size_t)m * ld + nbase + sc8[p] * 8 : reinterpret_cast<bf16*>(fd_epi_sink) + lane * 8;
                *reinterpret_cast<bf16x8*>(o) = v[p];
            } else if (m < m_end && !FD_ABL(g.nostore & 1)) {
                FD_STREAM_STORE(reinterpret_cast<bf16x8*>(dst + (size_t)m * ld + nbase + sc8[p] * 8), v[p]);
            }
        }
    };
    // a staged [16][96] slab of bytes (codes or e4m3) -> 16-byte row-contiguous stores
    auto slab8_out = [&](uint8_t* o8, int ld, int i) {
        u32x4 cv[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) cv[p] = *reinterpret_cast<const u32x4*>(stg + r8[p] * V2_EPI_LD8 + c8[p] * 16);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int m = mbase + i * 16 + r8[p];
            if (m < m_end && (p == 0 || lane < 32) && !FD_ABL(g.nostore & 1))
                FD_STREAM_STORE(reinterpret_cast<u32x4*>(o8 + (size_t)m * ld + nbase + c8[p] * 16), cv[p]);
        }
    };
    f32x4 sw4[G8OUT_ ? 1 : 6];
    if (g.sw && !G8OUT_) {
#pragma unroll
        for (int j = 0; j < 6; ++j) sw4[j] = *reinterpret_cast<const f32x4*>(g.sw + nbase + j * 16 + fg * 4);
    }
    auto colc = [&](int which, int j) { return *reinterpret_cast<const f32x4*>(stg + COLC_OFF + which * 384 + j * 64 + fg * 16); };
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        f32x4 val[6];
        if (g.sw) {          // fp8 operands: dequantise the accumulators (per-row scale of A x per-channel scale of B)
            const float sa = g.sa ? g.sa[min(mbase + i * 16 + frow, m_end - 1)] : 1.0f;      // (MXA: the block scales are in the MFMA)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (EPI == FEDDAT_EPI_MUL_G8_F8)      // the output keeps the input's row scale (x FEDDAT_F8_GRAD_HEADROOM): no sa
                    val[j] = acc[i][j] * (colc(1, j) * f32x4{1.0f / FEDDAT_F8_GRAD_HEADROOM, 1.0f / FEDDAT_F8_GRAD_HEADROOM,
                                                              1.0f / FEDDAT_F8_GRAD_HEADROOM, 1.0f / FEDDAT_F8_GRAD_HEADROOM});
                else
                    val[j] = acc[i][j] * ((G8OUT_ ? colc(1, j) : sw4[j]) * f32x4{sa, sa, sa, sa}) + (G8OUT_ ? colc(0, j) : bias4[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) val[j] = acc[i][j] + (G8OUT_ ? colc(0, j) : bias4[j]);
        }
        if (EPI == FEDDAT_EPI_MUL_DGELU) {
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(stg + srow[p] * V2_EPI_LD16 + sc8[p] * 16) = ux[i][p];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const bf16x4 u = *reinterpret_cast<const bf16x4*>(wr + j * 32);
                if (!FD_ABL(g.nostore & 2)) val[j] = val[j] * gelu_grad4_pk(f32x4{(float)u[0], (float)u[1], (float)u[2], (float)u[3]});
                else val[j] = val[j] * f32x4{(float)u[0], (float)u[1], (float)u[2], (float)u[3]};
            }
            if (i + AUX_AHEAD < WM) aux_load(i + AUX_AHEAD);
        }
        if (EPI == FEDDAT_EPI_GELU) {
            if (g.out2_bf16) put(g.out2_bf16, g.ldo2, i, val);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (!FD_ABL(g.nostore & 2)) val[j] = gelu4_pk(val[j]);
        }
        if (G8IN) {          // codes: row-contiguous 16-byte chunks -> LDS -> one dword (4 codes) per accumulator
#pragma unroll
            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(stg + r8[p] * V2_EPI_LD8 + c8[p] * 16) = ux8[i][p];
#pragma unroll
            for (int j = 0; j < 6; ++j) val[j] = val[j] * fd_g8_decode4(*reinterpret_cast<const unsigned*>(wr8 + j * 16));
        }
        if (G8OUT) {         // gelu'(u) from the fp32 u, as codes, and gelu(u): one pass, shared erf polynomial
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x4 fj, gj;
                gelu_and_grad4_pk(val[j], fj, gj);
                // (debug flag 4, timing only: codes without the gelu' arithmetic)
                *reinterpret_cast<unsigned*>(wr8 + j * 16) = fd_g8_encode4(FD_ABL(g.nostore & 2) ? val[j] : gj);
                val[j] = fj;
            }
            slab8_out(reinterpret_cast<uint8_t*>(g.out2_bf16), g.ldo2, i);
        }
        if (F8OUT) {         // e4m3 with a scale the consumer knows (header): saturate, convert, same 8-bit slab route
            constexpr float S = EPI == FEDDAT_EPI_GELU_G8_F8 ? 1.0f / FEDDAT_F8_ACT_SCALE : 1.0f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x4 x = val[j] * f32x4{S, S, S, S};
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = __builtin_amdgcn_fmed3f(x[e], -448.0f, 448.0f);
                int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
                pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], pk, true);
                *reinterpret_cast<int*>(wr8 + j * 16) = pk;
            }
            slab8_out(reinterpret_cast<uint8_t*>(g.out_bf16), g.ldo16, i);
        } else {
            put(g.out_bf16, g.ldo16, i, val);
        }
    }
}

// Epilogue of one wave tile (48 x 96) through its private LDS staging buffer.  The accumulators (+ bias, added in
// the accumulator layout from registers) go down as [16 rows][48 cols] fp32 chunks and come back row-contiguous
// (192-byte runs per row); the residual / aux operands of chunk c+1 are requested before chunk c is stored, so the
// six chunks do not serialise on HBM latency.
template <int EPI, int WM, bool DEQ = false>
__device__ __forceinline__ void v2_epilogue(const GemmArgs& g, f32x4 (&acc)[WM][6], char* stg, int mbase, int nbase,
                                            int m_end, int lane) {
    // keep the per-lane index math of the (several, inlined) epilogue sites out of the main loop's live ranges:
    // an opaque copy of the lane id cannot be hoisted across the k-loop
    asm volatile("" : "+v"(lane));
    const int frow = lane & 15, fg = lane >> 4;
    f32x4 bias4[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
        bias4[j] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + nbase + j * 16 + fg * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sw4[DEQ ? 6 : 1];            // DEQ = fp8 operands: per-channel scale of B (x per-row scale of A below)
    if (DEQ) {
#pragma unroll
        for (int j = 0; j < 6; ++j) sw4[j] = *reinterpret_cast<const f32x4*>(g.sw + nbase + j * 16 + fg * 4);
    }
    // read-back slots of this lane: 3 per chunk, (row, 4-column group) = divmod(p * 64 + lane, 12)
    int srow[3], sc4[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int idx = p * 64 + lane;
        srow[p] = idx / 12;
        sc4[p] = idx - srow[p] * 12;
    }
    // residual / aux operands of the whole wave tile are requested up front (18 loads per lane in flight: one exposed
    // HBM latency per tile instead of one per chunk); the fragment registers of the k-loop are dead here
    // residual / aux operands: three of the six chunks are in flight at any time (uniform base + 32-bit per-lane byte
    // offsets; FD_CHECK_ARG bounds the operand below 4 GiB)
    constexpr int NC = 2 * WM;              // chunks of [16 rows][48 cols]
    f32x4 rr[NC][3];
    bf16x4 uu[NC][3];
    const char* rbase = reinterpret_cast<const char*>(EPI == FEDDAT_EPI_RESID_F32 ? (const void*)(g.resid + nbase)
                                                                                  : (const void*)(g.aux + nbase));
    const unsigned esz = EPI == FEDDAT_EPI_RESID_F32 ? 4u : 2u;
    const unsigned ld_b = (EPI == FEDDAT_EPI_RESID_F32 ? (unsigned)g.ldr : (unsigned)g.ldaux) * esz;
    auto prefetch = [&](int c) {
        const int i = c >> 1, half = c & 1;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned mc = (unsigned)min(mbase + i * 16 + srow[p], m_end - 1);
            const unsigned off = mc * ld_b + ((unsigned)sc4[p] * 4u + half * 48) * esz;
            if (EPI == FEDDAT_EPI_RESID_F32) rr[c][p] = *reinterpret_cast<const f32x4*>(rbase + off);
            if (EPI == FEDDAT_EPI_MUL_DGELU) uu[c][p] = *reinterpret_cast<const bf16x4*>(rbase + off);
        }
    };
    if (EPI == FEDDAT_EPI_RESID_F32 || EPI == FEDDAT_EPI_MUL_DGELU) {
        prefetch(0); prefetch(1); prefetch(2);
    }
    auto chunk = [&](int c, f32x4 (&r)[3], bf16x4 (&u)[3]) {
        const int i = c >> 1, half = c & 1;
        if (DEQ) {
            const float sa = g.sa[min(mbase + i * 16 + frow, m_end - 1)];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                *reinterpret_cast<f32x4*>(stg + frow * V2_EPI_LD + (jj * 16 + fg * 4) * 4) =
                    acc[i][half * 3 + jj] * (sw4[DEQ ? half * 3 + jj : 0] * f32x4{sa, sa, sa, sa}) + bias4[half * 3 + jj];
        } else {
#pragma unroll
            for (int jj = 0; jj < 3; ++jj)
                *reinterpret_cast<f32x4*>(stg + frow * V2_EPI_LD + (jj * 16 + fg * 4) * 4) =
                    acc[i][half * 3 + jj] + bias4[half * 3 + jj];
        }
        f32x4 v[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) v[p] = *reinterpret_cast<const f32x4*>(stg + srow[p] * V2_EPI_LD + sc4[p] * 16);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int m = mbase + i * 16 + srow[p];
            const int n = nbase + half * 48 + sc4[p] * 4;
            const bool ok = m < m_end;
            const f32x4 x = v[p];
            if (EPI == FEDDAT_EPI_BF16) {
                if (ok) *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(x);
            } else if (EPI == FEDDAT_EPI_RESID_F32) {
                if (ok) *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) = x + r[p];
            } else if (EPI == FEDDAT_EPI_GELU) {
                const f32x4 a = gelu4_pk(x);
                if (ok) {
                    if (g.out2_bf16) *reinterpret_cast<bf16x4*>(g.out2_bf16 + (size_t)m * g.ldo2 + n) = cvt4(x);
                    *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(a);
                }
            } else if (EPI == FEDDAT_EPI_MUL_DGELU) {
                const f32x4 a = x * gelu_grad4_pk(f32x4{(float)u[p][0], (float)u[p][1], (float)u[p][2], (float)u[p][3]});
                if (ok) *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(a);
            } else {
                if (ok) *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) = x;
            }
        }
    };
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        chunk(c, rr[c], uu[c]);
        if ((EPI == FEDDAT_EPI_RESID_F32 || EPI == FEDDAT_EPI_MUL_DGELU) && c + 3 < NC) prefetch(c + 3);
    }
}

template <int WM>
struct V2State {
    const bf16* pa[WM];
    const bf16* pb[3];
    const uint8_t* psc;   // MXA: this lane's row of A block scales (k-tile 0)
    int l_tile, l_kt;     // tile index / k-tile of the next staging load
};

// MXA (fp8, K = 128 MFMA only): A carries true MX block scales -- one E8M0 byte per (row, 32 consecutive k) in g.amx -- which the
// block-scaled MFMA applies itself.  Measured semantics of v_mfma_scale_f32_16x16x128_f8f6f4 (tools/mx_scale_probe.py): lane
// (row r, group g) holds k = 16 g .. 16 g + 15 in its first four operand registers and k = 64 + 16 g .. in the last four --
// exactly this kernel's fragment pair (16-byte chunks g and 4 + g of the 128-byte row) -- and the scale of block kb = k / 32 of
// row r is taken from the scale register of lane (r, g = kb): the lane does NOT scale its own 32 values, it supplies the scale of
// the kb-th 32-block of its row.  So the memory order of A is the hardware's k order, and lane (r, g) passes amx[r][4 kt + g].
// The scales of a k-tile (4 bytes per row) ride the staging pipeline as one more piece per wave: a dword per row -> LDS
// [rows][4] per stage; the consumer reads its byte next to its fragments.
template <int EPI, int WM, bool FP8 = false, bool FP8_K32 = false, bool MXA = false>
__global__ __launch_bounds__(512, 1) void gemm_nt_v2_kernel(GemmArgsV2 a) {
    using Cfg = V2Cfg<WM>;
    constexpr int NP = Cfg::NP;
    static_assert(!MXA || (FP8 && !FP8_K32), "MX block scales exist on the K = 128 fp8 path only");
    constexpr int SC_OFF = Cfg::LDS;              // MXA: 2 stages x [64 WM rows][4 bytes] behind everything else
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef FEDDAT_ABLATE
    // tools/gemm_dephase.py (timing probe): blocks start (bid / 8) % 4 x q x 2 us apart, so that the CUs' epilogues (HBM write
    // bursts) stop coinciding
    if (const int q = (a.dbg >> 24) & 7) {
        const long t0 = wall_clock64(), d = (long)((blockIdx.x >> 3) & 3) * q * 200;
        while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const GemmArgs& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int wm = wave >> 1, wn = wave & 1;
    const int total = a.tiles_m * (g.N / V2_BN);
    const int grid = gridDim.x, bid = blockIdx.x;
    const int my_tiles = (total - bid + grid - 1) / grid;
    const int nk = g.K / BK;
    const int total_it = my_tiles * nk;

    f32x4 acc[WM][6];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int m0, n0, m_last;                 // tile being computed
    v2_tile_coords(a, bid, total, m0, n0, m_last);
    V2State<WM> L;
    v2_piece_ptrs<WM>(g, m0, m_last, n0, wave, lane, L.pa, L.pb);
    // MXA: the wave stages the block scales of the A rows it stages (8 WM rows: lanes beyond repeat the last one)
    auto sc_ptr = [&](int tm0, int tml) {
        const int r = wave * (8 * WM) + min(lane, 8 * WM - 1);
        return g.amx + (size_t)min(tm0 + r, tml) * g.ld_mx;
    };
    L.psc = MXA ? sc_ptr(m0, m_last) : nullptr;
    L.l_tile = 0;
    L.l_kt = 0;
    int issued = 0;                     // k-tiles whose staging loads have been issued (may run past total_it)
    // LDS byte offset of this lane inside a piece: row (lane >> 3), chunk (lane & 7) ^ (row & 7); piece p at p * 1024
    const int lds_lane = (lane >> 3) * 128 + ((((lane & 7) ^ (lane >> 3)) & 7) << 4);

    // Staging registers (one set): the k-tile this wave writes to LDS next.  The stream position (pointers, l_kt)
    // always addresses k-tile min(issued, total_it - 1), so loads past the end of the stream re-read the last k-tile
    // and the k-loop needs no branches around its staging instructions.
    u32x4 rs[NP];           // pieces 0..WM-1 = A rows, WM..WM+2 = B rows
    int written = 0;        // k-tiles of the stream this wave has written to LDS
    auto gload_piece = [&](int p) {
        const int koff = L.l_kt * BK;
        rs[p] = *reinterpret_cast<const u32x4*>((p < WM ? L.pa[p < WM ? p : 0] : L.pb[p < WM ? 0 : p - WM]) + koff);
    };
    auto stream_advance = [&]() {
        if (issued + 1 < total_it) {
            if (++L.l_kt == nk) {
                L.l_kt = 0;
                ++L.l_tile;
                int lm0, ln0, lml;
                v2_tile_coords(a, bid + L.l_tile * grid, total, lm0, ln0, lml);
                v2_piece_ptrs<WM>(g, lm0, lml, ln0, wave, lane, L.pa, L.pb);
                if (MXA) L.psc = sc_ptr(lm0, lml);
            }
        }
        ++issued;
    };
    auto lwrite_piece = [&](int p, int stage) {
        char* sb = smem + stage * Cfg::STAGE + lds_lane +
                   (p < WM ? (wave * WM + p) * 1024 : Cfg::TILE_A + (wave * 3 + p - WM) * 1024);
        *reinterpret_cast<u32x4*>(sb) = rs[p];
    };
    unsigned rsc = 0;       // MXA: the scale piece (4 block scales of this lane's row for the k-tile being staged)
    auto gload_sc = [&]() {
        if (MXA) rsc = *reinterpret_cast<const unsigned*>(L.psc + L.l_kt * 4);
    };
    auto lwrite_sc = [&](int stage) {
        if (MXA && lane < 8 * WM) *reinterpret_cast<unsigned*>(smem + SC_OFF + stage * (256 * WM) + (wave * (8 * WM) + lane) * 4) = rsc;
    };
    auto gload = [&]() {
#pragma unroll
        for (int p = 0; p < NP; ++p) gload_piece(p);
        gload_sc();
        stream_advance();
    };
    auto lwrite = [&](int stage) {
#pragma unroll
        for (int p = 0; p < NP; ++p) lwrite_piece(p, stage);
        lwrite_sc(stage);
        ++written;
    };

    const int frow = lane & 15, fg = lane >> 4;
    // fragment read of row r (a multiple of 16 + frow), k-half ks: 16-byte chunk (4 ks + fg) ^ (frow & 7) of the row
    const int frag_off[2] = {frow * 128 + (((0 + fg) ^ (frow & 7)) << 4), frow * 128 + (((4 + fg) ^ (frow & 7)) << 4)};
    char* stg = smem + Cfg::EPI_OFF + wave * V2_EPI_WAVE;
    int kt = 0, c_tile = 0, st = 0;
    bool pend = false;                  // a finished tile whose epilogue has not run yet
    int pm0 = 0, pn0 = 0, pml = 0;
    auto run_epilogue = [&](int em0, int en0, int eml) {
        const int mb = em0 + wm * (16 * WM), nb = en0 + wn * 96, me = eml + 1;
        if (!FD_ABL(a.dbg & 8)) {
            if (EPI == FEDDAT_EPI_BF16 || EPI == FEDDAT_EPI_GELU || EPI == FEDDAT_EPI_MUL_DGELU || EPI == FEDDAT_EPI_GELU_G8 ||
                EPI == FEDDAT_EPI_MUL_G8 || EPI == FEDDAT_EPI_GELU_G8_F8 || EPI == FEDDAT_EPI_MUL_G8_F8)
                v2_epilogue_bf16<EPI, WM>(g, acc, stg, mb, nb, me, lane);
            else
                v2_epilogue<EPI, WM, FP8>(g, acc, stg, mb, nb, me, lane);
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // Staging runs in the shadow of the MFMAs: in its C phase of k-tile j a wave of group 0 writes its pieces of
    // k-tile j+1 (read by group 0 in the very next barrier slot), a wave of group 1 -- one slot behind -- its pieces
    // of k-tile j+2; each piece's register is refilled from global memory right after it has been written.
    //   slot:      2j        2j+1       2j+2       2j+3       2j+4
    //   group 0:   L(j)      C(j) w j+1 L(j+1)     C(j+1) w j+2  L(j+2)
    //   group 1:   C(j-1) w j+1   L(j)  C(j) w j+2 L(j+1)     C(j+1) w j+3
    // A stage is rewritten only after both groups' reads of it have retired (lgkmcnt(0) before a barrier they passed).
    // prologue: k-tile 0 (group 1: also k-tile 1) -> LDS, the next one in flight to registers
    gload();
    lwrite(0);
    gload();
    if (grp == 1) {
        lwrite(1);
        gload();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // stagger: group 1 runs one phase behind group 0
    __builtin_amdgcn_sched_barrier(0);

    for (int it = 0; it < total_it; ++it) {
        // ------------------------------ L phase ------------------------------
        bf16x8 fa[2][WM], fb[2][6];
        int asc[MXA ? WM : 1];      // MXA: E8M0 scale of this lane's 32 k-values of A row group i
        auto read_frags = [&]() {
            if (MXA) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    asc[i] = *reinterpret_cast<const uint8_t*>(smem + SC_OFF + st * (256 * WM) + (wm * (16 * WM) + i * 16 + frow) * 4 + fg);
            }
            // address = per-lane swizzled offset (loop invariant, one per k-half) + wave-uniform tile base + immediate
            const int a_base = st * Cfg::STAGE + wm * (16 * WM * 128);
            const int b_base = st * Cfg::STAGE + Cfg::TILE_A + wn * (96 * 128);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* pb_ = smem + (frag_off[ks] + b_base);
                const char* pa_ = smem + (frag_off[ks] + a_base);
#pragma unroll
                for (int j = 0; j < 6; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(pb_ + j * 2048);
#pragma unroll
                for (int i = 0; i < WM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(pa_ + i * 2048);
                __builtin_amdgcn_sched_barrier(0);      // keep the k-half-0 reads first in the LDS queue
            }
        };
        // Tile boundary: both wave groups run the epilogue of the finished tile in the SAME barrier slot (8 waves
        // hide each other's LDS / store latency; staggered, each group's epilogue was exposed on its own).
        //   group 0:  C(last) | staging of L(next 0) | EPILOGUE, fragment reads of L(next 0) | C(next 0) | ...
        //   group 1:  L(last) | C(last)              | EPILOGUE                              | L(next 0) | ...
        // No fragment registers are live while the epilogue runs.
        const bool boundary = pend;
        if (boundary && grp == 0) {      // group 0 idles one slot (group 1 finishes its last C phase)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (boundary) {
            run_epilogue(pm0, pn0, pml);
            pend = false;
            if (grp == 1) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        read_frags();
        // Group 1 overwrites the stage it has just read (k-tile j+2 -> stage j & 1) in the very next slot, so the
        // fragment reads are retired BEFORE the barrier (a restage one phase after the last read needs the reads
        // retired by an lgkmcnt in front of the reading phase's barrier; group 0 restages two phases later and would
        // not need it -- measured cost of waiting in both groups: none).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ------------------------------ C phase ------------------------------
        // 2 WM groups of 6 MFMAs; after group p (< NP): ds_write of staging piece p, then its global reload
        const int wstage = written & 1;
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8 && !FP8_K32) {
            // CDNA4 form: ONE block-scaled K = 128 instruction per output tile and k-tile (both k-halves' fragments = the 32
            // bytes per lane it takes) at twice the bf16 rate: WM groups of 6 MFMAs, two staging pieces behind each
#pragma unroll
            for (int i = 0; i < WM; ++i) {
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = MXA ? mfma16x128_fp8_mx_sb(fb[0][j], fb[1][j], fa[0][i], fa[1][i], acc[i][j], asc[MXA ? i : 0])
                                    : mfma16x128_fp8_mx(fb[0][j], fb[1][j], fa[0][i], fa[1][i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 2 * i; p < 2 * i + 2; ++p)
                    if (p < NP) {
                        lwrite_piece(p, wstage);
                        gload_piece(p);
                    }
                if (MXA && i == WM - 1) {
                    lwrite_sc(wstage);
                    gload_sc();
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < WM; ++i) {
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = FP8 ? mfma16x64_fp8(fb[ks][j], fa[ks][i], acc[i][j]) : mfma16x32(fb[ks][j], fa[ks][i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);     // MFMAs first: the write's lgkmcnt must not gate them
                if (ks * WM + i < NP) {                // compile-time: straight-line code, so the compiler's vmcnt /
                    lwrite_piece(ks * WM + i, wstage); // lgkmcnt counts stay exact (a runtime branch here degrades
                    gload_piece(ks * WM + i);          // them to waits for 0)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        ++written;
        stream_advance();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this phase's ds_writes, before the barrier below
        st ^= 1;
        if (++kt == nk) {
            kt = 0;
            pend = true; pm0 = m0; pn0 = n0; pml = m_last;      // epilogue: next iteration's boundary slot / after the loop
            if (++c_tile < my_tiles) v2_tile_coords(a, bid + c_tile * grid, total, m0, n0, m_last);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier
    __builtin_amdgcn_sched_barrier(0);
    if (pend) run_epilogue(pm0, pn0, pml);        // last tile, both groups concurrently (A/B: not slower than staggered)
}


// ------------------------------------------------------------------------------------------------------------------
// v3 (the default for the plain and residual epilogues; debug flag 1 keeps everything on v2, flag 2 forces v3): ONE wave per SIMD.  192 x 192 x 64 tile, 4 waves (2 x 2, 96 x 96 per wave = 6 x 6
// MFMA tiles: 72 MFMAs per 24 fragment reads), persistent blocks and the continuous k-tile stream of v2.  The
// accumulators are pinned to AGPRs by issuing the MFMA as inline asm ("+a"): with the builtin hipcc parked part of the
// 144 accumulator registers in other registers and moved them around every MFMA.  Every MFMA is followed by exactly one
// other instruction (fragment read of the next k-half, staging write, staging load), which issues while the MFMA runs.
//   half 0 of k-tile j (stage s):  MFMA(j, 0) | read frags(j, 1) from s | ds_write k-tile j+1 -> s^1 | global loads of j+2
//   lgkmcnt(0), barrier
//   half 1:                        MFMA(j, 1) | read frags(j+1, 0) from s^1



__device__ __forceinline__ void mfma_agpr(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile(FD_MFMA_16X16X32_ASM " %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// first product of a tile: C = 0 as an inline constant, the accumulator needs no zeroing
__device__ __forceinline__ void mfma_agpr_first(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile(FD_MFMA_16X16X32_ASM " %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

// the same with the accumulators in ordinary VGPRs (DUAL: 256 registers per wave, which hipcc splits 128 / 128 as soon as one
// operand is constrained to an AGPR -- with "v" the whole budget is one file)
__device__ __forceinline__ void mfma_vgpr(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile(FD_MFMA_16X16X32_ASM " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vgpr_first(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm volatile(FD_MFMA_16X16X32_ASM " %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}

// (A variant that DEFERRED a tile's epilogue into the next tile's k-loop was built on this kernel -- accumulators copied
// aside at the boundary, one twelfth of the stores per k-tile between the MFMAs -- and was bit-exact, but slower: on this ISA
// loads and stores share vmcnt and complete out of order with respect to each other, so with one store in flight every
// wait for a staging load becomes vmcnt(0).  The epilogue therefore stays at the tile boundary, as in v2.)
// RT = 16-row MFMA tiles per wave along M: 6 -> 192 x 192 block tile (96 x 96 per wave), 8 -> 256 x 192 (128 x 96 per
// wave) for launches whose tile count then still fills whole rounds (N = 3072 at M = 11 840: 3 rounds instead of 4).
template <int RT> struct V3Cfg {
    static constexpr int BM = 32 * RT;
    static constexpr int TILE_A = BM * BK * 2;
    static constexpr int STAGE = TILE_A + V2_TILE_B;
    static constexpr int EPI_OFF = 2 * STAGE;
    static constexpr int LDS = EPI_OFF + 4 * V2_EPI_WAVE;      // 116 / 132 KiB
    static constexpr int NP = RT + 6;                         // staging pieces per wave and k-tile (A: RT, B: 6)
};

// FAKE = 1 (tools/ timing probe, debug flag 512; results are WRONG): no epilogue at the tile boundary; instead every k-tile
// issues, between the MFMAs of its second half, the share of an epilogue a DEFERRED form would issue there -- 3 of the 36
// sub-tiles' conversions / activation math and their 8-byte row-per-lane stores (inline asm: invisible to hipcc's vmcnt
// bookkeeping, so the staging loads keep their counted waits) plus the aux / residual loads.  Answers one question before
// the real thing is built: does the k-loop absorb the epilogue's issue slots and stores?
// DUAL (round 6, "v4"): the same kernel as TWO independent 4-wave workgroups per CU (RT = 4: 128 x 192 tiles, 64 x 96 per wave,
// 256 registers per wave, 2 x 80 KiB of LDS -- the epilogue's staging buffer aliases the k-tile stage the tile has just
// consumed).  The two workgroups share nothing and drift apart by themselves: whenever one is in its epilogue (VALU / LDS /
// stores) the other has the matrix pipe to itself, so the epilogues -- 25-30 % of the code-epilogue launches, which nothing
// overlapped -- run under the neighbour's k-loop.  Same k order as every other tile height: bit-identical results.
template <int EPI, int RT, int FAKE = 0, bool DUAL = false>
__global__ __launch_bounds__(256, DUAL ? 2 : 1) void gemm_nt_v3_kernel(GemmArgsV2 a) {
    using Cfg = V3Cfg<RT>;
    static_assert(!DUAL || (RT == 4 && 4 * V2_EPI_WAVE <= Cfg::STAGE), "the dual form: 128-row tiles, staging inside one stage");
    constexpr int NP = Cfg::NP, NM = 6 * RT;                  // MFMAs per k-half
    // one filler per MFMA.  Half 0 has NM slots: the NP fragment reads of half 1, then every piece's write into the other
    // stage (all NP of them: the stage must be complete at the barrier) and as many reloads as still fit (LH0); the
    // remaining NP - LH0 reloads (RT = 5: 3 of 11) follow behind the first MFMAs of half 1
    constexpr int LH0 = (NM - 2 * NP) < NP ? (NM - 2 * NP) : NP;
    static_assert(LH0 >= 0 && NP - LH0 <= 12, "writes fit half 0, deferred reloads fit ahead of half 1's fragment reads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef FEDDAT_ABLATE
    // tools/gemm_dephase.py (timing probe): blocks start (bid / 8) % 4 x q x 2 us apart, so that the CUs' epilogues (HBM write
    // bursts) stop coinciding
    if (const int q = (a.dbg >> 24) & 7) {
        const long t0 = wall_clock64(), d = (long)((blockIdx.x >> 3) & 3) * q * 200;
        while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const GemmArgs& g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int total = a.tiles_m * (g.N / V2_BN);
    const int grid = gridDim.x, bid = blockIdx.x;
    const int my_tiles = (total - bid + grid - 1) / grid;
    const int nk = g.K / BK;
    const int total_it = my_tiles * nk;

    f32x4 acc[RT][6];
    int m0, n0, m_last;
    v2_tile_coords(a, bid, total, m0, n0, m_last);
    unsigned pp[NP];                     // byte offsets of this lane's staging pieces from A / B (operands < 4 GiB: checked)
    auto piece_ptrs = [&](int tm0, int tml, int tn0) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int r = (wave * RT + i) * 8 + (lane >> 3);
            pp[i] = ((unsigned)min(tm0 + r, tml) * (unsigned)g.lda + (lane & 7) * 8) * 2u;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int r = (wave * 6 + i) * 8 + (lane >> 3);
            pp[RT + i] = ((unsigned)min(tn0 + r, g.N - 1) * (unsigned)g.ldb + (lane & 7) * 8) * 2u;
        }
    };
    piece_ptrs(m0, m_last, n0);
    int l_tile = 0, l_kt = 0, issued = 0;
    const int lds_lane = (lane >> 3) * 128 + ((((lane & 7) ^ (lane >> 3)) & 7) << 4);
    u32x4 rs[NP];
    auto gload_piece = [&](int p) {
        // scalar base (operand + k offset of the stream position) + this lane's 32-bit piece offset: no vector address math
        const char* base = reinterpret_cast<const char*>(p < RT ? g.A : g.B) + (size_t)l_kt * (BK * 2);
        rs[p] = *reinterpret_cast<const u32x4*>(base + pp[p]);
    };
    auto stream_advance = [&]() {
        if (issued + 1 < total_it) {
            if (++l_kt == nk) {
                l_kt = 0;
                ++l_tile;
                int lm0, ln0, lml;
                v2_tile_coords(a, bid + l_tile * grid, total, lm0, ln0, lml);
                piece_ptrs(lm0, lml, ln0);
            }
        }
        ++issued;
    };
    auto lwrite_piece = [&](int p, int stage) {
        char* sb = smem + stage * Cfg::STAGE + lds_lane +
                   (p < RT ? (wave * RT + p) * 1024 : Cfg::TILE_A + (wave * 6 + p - RT) * 1024);
        *reinterpret_cast<u32x4*>(sb) = rs[p];
    };

    const int frow = lane & 15, fg = lane >> 4;
    const int frag_off[2] = {frow * 128 + (((0 + fg) ^ (frow & 7)) << 4), frow * 128 + (((4 + fg) ^ (frow & 7)) << 4)};
    bf16x8 fa[2][RT], fb[2][6];
    auto read_a = [&](int stage, int ks, int i) {
        fa[ks][i] = *reinterpret_cast<const bf16x8*>(smem + stage * Cfg::STAGE + frag_off[ks] + (wm * (16 * RT) + i * 16) * 128);
    };
    auto read_b = [&](int stage, int ks, int j) {
        fb[ks][j] = *reinterpret_cast<const bf16x8*>(smem + stage * Cfg::STAGE + Cfg::TILE_A + frag_off[ks] + (wn * 96 + j * 16) * 128);
    };
    auto read_frag = [&](int stage, int ks, int q) {           // q < 6: B fragment q (every MFMA row needs them), then A
        if (q < 6) read_b(stage, ks, q);
        else read_a(stage, ks, q - 6);
    };

#pragma unroll
    for (int p = 0; p < NP; ++p) gload_piece(p);
    stream_advance();
#pragma unroll
    for (int p = 0; p < NP; ++p) lwrite_piece(p, 0);
#pragma unroll
    for (int p = 0; p < NP; ++p) gload_piece(p);
    stream_advance();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!DUAL) {
#pragma unroll
        for (int q = 0; q < NP; ++q) read_frag(0, 0, q);
    }

    auto fake_slice = [&](const int sidx, const int kt) {       // FAKE only: sub-tile (2 sidx, sidx) stands in for slice kt
        const int row = min(m0 + wm * (16 * RT) + 32 * sidx + frow, g.M - 1);
        const int col = n0 + wn * 96 + 16 * (kt % 6) + 4 * fg;
        f32x4 v = acc[2 * sidx][sidx];
        if (EPI == FEDDAT_EPI_RESID_F32 || EPI == FEDDAT_EPI_F32) {
            if (EPI == FEDDAT_EPI_RESID_F32) v = v + *reinterpret_cast<const f32x4*>(g.resid + (size_t)row * g.ldr + col);
            const float* dst = g.out_f32 + (size_t)row * g.ldo32 + col;
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
            return;
        }
        if (EPI == FEDDAT_EPI_MUL_DGELU) {
            const bf16x4 u4 = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)row * g.ldaux + col);
            v = v * gelu_grad4_pk(f32x4{(float)u4[0], (float)u4[1], (float)u4[2], (float)u4[3]});
        }
        if (EPI == FEDDAT_EPI_GELU) {
            const bf16x4 u16 = cvt4(v);
            const bf16* dst2 = g.out2_bf16 + (size_t)row * g.ldo2 + col;
            asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(dst2), "v"(u16) : "memory");
            v = gelu4_pk(v);
        }
        const bf16x4 o16 = cvt4(v);
        const bf16* dst = g.out_bf16 + (size_t)row * g.ldo16 + col;
        asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(o16) : "memory");
    };
    // one k-tile; FIRST: the tile's first k-tile, whose half 0 starts the accumulators from the constant 0.
    // MODE 0: the continuous k-tile stream across tiles (stream_advance).  DUAL, where every tile is self-contained (prologue,
    // k-loop, epilogue: nothing but the accumulators is live across the epilogue, which therefore has the wave's whole VGPR half):
    // MODE 1 = a k-tile with both successors (writes k-tile kt + 1, loads kt + 2), 2 = the penultimate (no loads), 3 = the last
    // (MFMAs and its own half-1 fragments only).
    auto k_tile = [&](auto first_tag, const int st, const int kt, auto mode_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool WR = MODE != 3, LD = MODE < 2, NXT = MODE != 3;
        if (MODE != 0) l_kt = kt + 2;
        // ---- half 0: MFMA (i, j) then filler #(6 i + j): NP fragment reads, then per piece its write and its reload
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (FIRST) mfma_agpr_first(acc[i][j], fb[0][j], fa[0][i]);
                else mfma_agpr(acc[i][j], fb[0][j], fa[0][i]);
                const int f = 6 * i + j;
                __builtin_amdgcn_sched_barrier(0);
                if (f < NP) {                              // fragments of k-half 1
                    read_frag(st, 1, f);
                } else if (f < NP + 2 * LH0) {             // k-tile j+1 -> the other stage, piece by piece, each register
                    const int q = f - NP;                  // refilled with k-tile j+2 right behind its write (all writes in a
                    if ((q & 1) == 0) { if (WR) lwrite_piece(q >> 1, st ^ 1); }   // row, then the loads, with the barrier pulled forward:
                    else { if (LD) gload_piece(q >> 1); }  // 10 % slower -- the four waves' writes collide)
                } else if (f < 2 * NP + LH0) {             // (RT = 5) the last pieces' writes; their reloads wait for half 1
                    if (WR) lwrite_piece(f - NP - LH0, st ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if (LH0 == NP && MODE == 0) stream_advance();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- half 1: the fragments of the next k-tile's half 0 behind the first two MFMA rows
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                mfma_agpr(acc[i][j], fb[1][j], fa[1][i]);
                const int f = 6 * i + j;
                __builtin_amdgcn_sched_barrier(0);
                if (LH0 < NP && f < NP - LH0) {            // deferred reloads (same stream position as half 0's), then advance
                    if (LD) gload_piece(LH0 + f);
                    if (f == NP - LH0 - 1 && MODE == 0) stream_advance();
                }
                if (NXT && f >= 12 && f < 12 + NP) read_frag(st ^ 1, 0, f - 12);
                if (FAKE && RT == 6 && (f == 28 || f == 30 || f == 32) && kt < 12) fake_slice((f - 28) / 2, kt);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto run_epilogue = [&](char* stg) {
        // The compiler cannot see that the asm blocks are MFMAs: it would read their results right behind them.  The wait
        // states are attached to the accumulators themselves (in / out operands), row by row.
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            if (DUAL)
                asm volatile("s_nop 15\n\ts_nop 15"
                             : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]), "+v"(acc[i][4]), "+v"(acc[i][5]));
            else
                asm volatile("s_nop 15\n\ts_nop 15"
                             : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]));
        }
        if (!FD_ABL(a.dbg & 8) && !FAKE) {
            const int mb = m0 + wm * (16 * RT), nb = n0 + wn * 96, me = m_last + 1;
            if (EPI == FEDDAT_EPI_BF16 || EPI == FEDDAT_EPI_GELU || EPI == FEDDAT_EPI_MUL_DGELU || EPI == FEDDAT_EPI_GELU_G8 ||
                EPI == FEDDAT_EPI_MUL_G8)
                v2_epilogue_bf16<EPI, RT, true>(g, acc, stg, mb, nb, me, lane);
            else
                v2_epilogue<EPI, RT>(g, acc, stg, mb, nb, me, lane);
        }
    };

    if constexpr (DUAL) {
        // Every tile on its own: prologue (k-tile 0 -> stage 0, k-tile 1 in flight, fragments of half 0), k-loop, epilogue; nothing
        // but the accumulators is live across the epilogue.  ONE fragment set (128 VGPRs per wave next to 128 AGPRs: two sets +
        // the staging registers do not fit): half 0's MFMAs carry the whole staging of the k-tile as fillers (10 writes, 10
        // loads in 24 slots), then the half-1 fragments are read into the same registers ahead of the barrier, half 1's MFMAs
        // run bare, and the next k-tile's half-0 fragments follow them -- the two fragment waits per k-tile are covered by the
        // OTHER workgroup's wave on the same SIMD, not by this wave's own MFMAs.  The staging buffer of the epilogue aliases the
        // stage of k-tile nk - 2 (dead since that k-tile's barrier; nothing is written into it afterwards); the barrier behind
        // the epilogue releases both stages for the next tile's prologue.
        bf16x8 ga[RT], gb[6];
        auto rd = [&](int stage, int ks) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
                gb[j] = *reinterpret_cast<const bf16x8*>(smem + stage * Cfg::STAGE + Cfg::TILE_A + frag_off[ks] + (wn * 96 + j * 16) * 128);
#pragma unroll
            for (int i = 0; i < RT; ++i)
                ga[i] = *reinterpret_cast<const bf16x8*>(smem + stage * Cfg::STAGE + frag_off[ks] + (wm * (16 * RT) + i * 16) * 128);
        };
        auto k_tile_d = [&](auto first_tag, const int st, const int kt, auto mode_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int MODE = decltype(mode_tag)::value;       // 1: writes k-tile kt + 1, loads kt + 2; 2: no loads; 3: the last
            constexpr bool WR = MODE != 3, LD = MODE < 2;
            l_kt = kt + 2;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    if (FIRST) mfma_vgpr_first(acc[i][j], gb[j], ga[i]);
                    else mfma_vgpr(acc[i][j], gb[j], ga[i]);
                    const int f = 6 * i + j;
                    __builtin_amdgcn_sched_barrier(0);
                    if (f < 2 * NP) {
                        if ((f & 1) == 0) { if (WR) lwrite_piece(f >> 1, st ^ 1); }
                        else { if (LD) gload_piece(f >> 1); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            rd(st, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    mfma_vgpr(acc[i][j], gb[j], ga[i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (WR) rd(st ^ 1, 0);
        };
        static_assert(2 * NP <= NM, "the whole staging of a k-tile rides behind half 0's MFMAs");
        for (int tile = 0; tile < my_tiles; ++tile) {
            if (tile) {
                v2_tile_coords(a, bid + tile * grid, total, m0, n0, m_last);
                piece_ptrs(m0, m_last, n0);
                l_kt = 0;
#pragma unroll
                for (int p = 0; p < NP; ++p) gload_piece(p);
#pragma unroll
                for (int p = 0; p < NP; ++p) lwrite_piece(p, 0);
                l_kt = 1;
#pragma unroll
                for (int p = 0; p < NP; ++p) gload_piece(p);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            rd(0, 0);
            k_tile_d(std::true_type{}, 0, 0, std::integral_constant<int, 1>{});
            for (int kt = 1; kt < nk - 2; ++kt) k_tile_d(std::false_type{}, kt & 1, kt, std::integral_constant<int, 1>{});
            k_tile_d(std::false_type{}, (nk - 2) & 1, nk - 2, std::integral_constant<int, 2>{});
            k_tile_d(std::false_type{}, (nk - 1) & 1, nk - 1, std::integral_constant<int, 3>{});
            run_epilogue(smem + (nk & 1) * Cfg::STAGE + wave * V2_EPI_WAVE);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        char* stg = smem + Cfg::EPI_OFF + wave * V2_EPI_WAVE;
        int it = 0;
        for (int tile = 0; tile < my_tiles; ++tile) {
            k_tile(std::true_type{}, it & 1, 0, std::integral_constant<int, 0>{});
            ++it;
            for (int kt = 1; kt < nk; ++kt, ++it) k_tile(std::false_type{}, it & 1, kt, std::integral_constant<int, 0>{});
            run_epilogue(stg);
            if (tile + 1 < my_tiles) v2_tile_coords(a, bid + (tile + 1) * grid, total, m0, n0, m_last);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Skinny GEMM (M <= 64): the top ViLT layer runs everything behind its attention on the 2B token-0 rows only
// (engine._top_layer_fwd / _bwd).  With so few rows the product is a stream over the weight matrix, so it is split over
// (N / 64) x ksplit blocks -- enough blocks to pull B through every CU -- into fp32 partials [ksplit][M][N]; a second
// kernel sums them in a fixed order and applies the epilogue.  The 128 x 128 kernel put 6..24 blocks on the chip for
// these shapes (16..47 us per launch).
// block = 4 waves; wave w owns weight rows n0 + 16 w .. +15 (A operand) x all 64 (padded) activation rows.
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g, float* __restrict__ part, int kslice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fg = lane >> 4, i16 = lane & 15;
    const int n0 = blockIdx.x * 64, ks = blockIdx.y;
    const int k0 = ks * kslice;
    const bf16* wrow = g.B + (size_t)(n0 + wave * 16 + i16) * g.ldb + k0 + fg * 8;
    const bf16* arow[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) arow[mt] = g.A + (size_t)min(mt * 16 + i16, g.M - 1) * g.lda + k0 + fg * 8;
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < kslice; k += 64) {
        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wrow + k);
        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wrow + k + 32);
        bf16x8 a0[4], a1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            a0[mt] = *reinterpret_cast<const bf16x8*>(arow[mt] + k);
            a1[mt] = *reinterpret_cast<const bf16x8*>(arow[mt] + k + 32);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            acc[mt] = mfma16x32(w0, a0[mt], acc[mt]);
            acc[mt] = mfma16x32(w1, a1[mt], acc[mt]);
        }
    }
    // lane: n = n0 + 16 wave + 4 fg + (0..3), m = 16 mt + i16
    float* p = part + (size_t)ks * g.M * g.N + n0 + wave * 16 + fg * 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = mt * 16 + i16;
        if (m < g.M) *reinterpret_cast<f32x4*>(p + (size_t)m * g.N) = acc[mt];
    }
}

__global__ __launch_bounds__(256) void gemm_skinny_epilogue_kernel(GemmArgs g, const float* __restrict__ part,
                                                                   int ksplit) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nq = g.N >> 2;
    if (i >= g.M * nq) return;
    const int m = i / nq, n = (i - m * nq) * 4;
    // fixed summation order 0, 1, 2, ...; the loads of 6 partials are issued together (ksplit is a multiple of 6 or
    // small for every shape the engine uses; the tail loop covers the rest)
    const float* pp = part + (size_t)m * g.N + n;
    const size_t ps = (size_t)g.M * g.N;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int s0 = 0;
    for (; s0 + 6 <= ksplit; s0 += 6) {
        f32x4 t[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) t[u] = *reinterpret_cast<const f32x4*>(pp + (s0 + u) * ps);
#pragma unroll
        for (int u = 0; u < 6; ++u) v = v + t[u];
    }
    for (; s0 < ksplit; ++s0) v = v + *reinterpret_cast<const f32x4*>(pp + s0 * ps);
    if (g.bias) v = v + *reinterpret_cast<const f32x4*>(g.bias + n);
    switch (g.epi) {
        case FEDDAT_EPI_BF16:
            *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(v);
            break;
        case FEDDAT_EPI_RESID_F32:
            *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) =
                v + *reinterpret_cast<const f32x4*>(g.resid + (size_t)m * g.ldr + n);
            break;
        case FEDDAT_EPI_GELU:
            if (g.out2_bf16) *reinterpret_cast<bf16x4*>(g.out2_bf16 + (size_t)m * g.ldo2 + n) = cvt4(v);
            *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(gelu4_pk(v));
            break;
        case FEDDAT_EPI_MUL_DGELU: {
            const bf16x4 u = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)m * g.ldaux + n);
            *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) =
                cvt4(v * gelu_grad4_pk(f32x4{(float)u[0], (float)u[1], (float)u[2], (float)u[3]}));
            break;
        }
        default:
            *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) = v;
    }
}

// The same products in ONE launch (round 4): a block = 16 output columns x all (<= 64) rows, its up to 16 waves split K in
// 64-deep slices and are summed through LDS in wave order, wave 0 applies the epilogue -- no fp32 partials in HBM, no
// second kernel (the split-K pair cost 13-15 us per product in the step, 2 graph nodes; this form 6-8 us, 1 node).
// N / 16 blocks (48 for N = 768, 192 for N = 3072) keep the weight stream spread over the chip.
constexpr int SKF_NW = 16, SKF_MAXSL = 3;
// (round 6) a block = 16 output columns x ONE 16-row tile (blockIdx.y): with all (<= 64) rows per block a launch had N / 16 = 48
// workgroups at N = 768, each streaming the whole 64 x K A panel (393 KB at K = 3072) through one CU: 20 us for the two
// K = 3072 products of the top layer.  Four times the workgroups, a quarter of the A bytes each; the per-element summation
// order (slices per wave, then waves in order) is unchanged: bit-identical results.
__global__ __launch_bounds__(SKF_NW * 64) void gemm_skinny_fused_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float red[SKF_NW - 1][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fg = lane >> 4, i16 = lane & 15;
    const int n0 = blockIdx.x * 16, mt = blockIdx.y;
    const int nslice = g.K / 64;                       // 64-deep k slices, dealt round-robin to the waves
    const bf16* wrow = g.B + (size_t)(n0 + i16) * g.ldb + fg * 8;
    const bf16* arow = g.A + (size_t)min(mt * 16 + i16, g.M - 1) * g.lda + fg * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // a wave's slices (<= SKF_MAXSL for K <= 3072; the loop covers longer K): ALL their loads are issued before the first MFMA
    // -- these products are one chain of dependent load batches, so the batches must overlap, not follow each other
    for (int sl0 = wave; sl0 < nslice; sl0 += SKF_NW * SKF_MAXSL) {
        bf16x8 w0[SKF_MAXSL], w1[SKF_MAXSL], a0[SKF_MAXSL], a1[SKF_MAXSL];
#pragma unroll
        for (int u = 0; u < SKF_MAXSL; ++u) {
            const int k = min(sl0 + u * SKF_NW, nslice - 1) * 64;           // (clamped: the surplus loads are not used)
            w0[u] = *reinterpret_cast<const bf16x8*>(wrow + k);
            w1[u] = *reinterpret_cast<const bf16x8*>(wrow + k + 32);
            a0[u] = *reinterpret_cast<const bf16x8*>(arow + k);
            a1[u] = *reinterpret_cast<const bf16x8*>(arow + k + 32);
        }
#pragma unroll
        for (int u = 0; u < SKF_MAXSL; ++u) {
            if (sl0 + u * SKF_NW >= nslice) break;
            acc = mfma16x32(w0[u], a0[u], acc);
            acc = mfma16x32(w1[u], a1[u], acc);
        }
    }
    if (wave > 0) *reinterpret_cast<f32x4*>(&red[wave - 1][lane][0]) = acc;
    __syncthreads();
    if (wave != 0) return;
    const int nw = min(SKF_NW, nslice);
#pragma unroll 1
    for (int w = 1; w < nw; ++w) acc = acc + *reinterpret_cast<const f32x4*>(&red[w - 1][lane][0]);
    // lane: n = n0 + 4 fg + (0..3), m = 16 mt + i16
    const int n = n0 + 4 * fg;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
    const int m = mt * 16 + i16;
    if (m >= g.M) return;
    const f32x4 v = acc + b4;
    switch (g.epi) {
        case FEDDAT_EPI_BF16:
            *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(v);
            break;
        case FEDDAT_EPI_RESID_F32:
            *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) =
                v + *reinterpret_cast<const f32x4*>(g.resid + (size_t)m * g.ldr + n);
            break;
        case FEDDAT_EPI_GELU:
            if (g.out2_bf16) *reinterpret_cast<bf16x4*>(g.out2_bf16 + (size_t)m * g.ldo2 + n) = cvt4(v);
            *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(gelu4_pk(v));
            break;
        case FEDDAT_EPI_MUL_DGELU: {
            const bf16x4 u = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)m * g.ldaux + n);
            *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) =
                cvt4(v * gelu_grad4_pk(f32x4{(float)u[0], (float)u[1], (float)u[2], (float)u[3]}));
            break;
        }
        default:
            *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) = v;
    }
}

// split factor: a multiple-of-64 K slice and about two blocks per CU
int skinny_ksplit(int N, int K) {
    const int nb = N / 64, kt = K / 64;
    int best = 1;
    for (int s = 1; s <= kt; ++s)
        if (kt % s == 0 && nb * s <= 640) best = s;
    return best;
}

}  // namespace

extern "C" long feddat_gemm_skinny_workspace_elems(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % 64) return 0;
    return (long)skinny_ksplit(N, K) * M * N;
}

extern "C" int feddat_gemm_bf16_nt_skinny(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi,
                                          const float* bias, const float* resid, int ldr, const void* aux, int ldaux,
                                          float* out_f32, int ldo32, void* out_bf16, int ldo16, void* out2_bf16,
                                          int ldo2, float* workspace, long workspace_elems, hipStream_t stream) {
    FD_CHECK_ARG(A && B && workspace && M > 0 && M <= 64 && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0);
    FD_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K);
    switch (epi) {
        case FEDDAT_EPI_BF16: FD_CHECK_ARG(out_bf16 && ldo16 % 4 == 0); break;
        case FEDDAT_EPI_RESID_F32: FD_CHECK_ARG(out_f32 && resid && ldr % 4 == 0 && ldo32 % 4 == 0); break;
        case FEDDAT_EPI_GELU: FD_CHECK_ARG(out_bf16 && ldo16 % 4 == 0 && (!out2_bf16 || ldo2 % 4 == 0)); break;
        case FEDDAT_EPI_MUL_DGELU: FD_CHECK_ARG(out_bf16 && aux && ldaux % 4 == 0 && ldo16 % 4 == 0); break;
        case FEDDAT_EPI_F32: FD_CHECK_ARG(out_f32 && ldo32 % 4 == 0); break;
        default: return FEDDAT_EINVAL;
    }
    const int ksplit = skinny_ksplit(N, K);
    FD_CHECK_ARG(workspace_elems >= (long)ksplit * M * N);
    GemmArgs g{};
    g.A = (const bf16*)A; g.B = (const bf16*)B; g.bias = bias; g.resid = resid; g.aux = (const bf16*)aux;
    g.out_f32 = out_f32; g.out_bf16 = (bf16*)out_bf16; g.out2_bf16 = (bf16*)out2_bf16;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldr = ldr; g.ldaux = ldaux;
    g.ldo32 = ldo32; g.ldo16 = ldo16; g.ldo2 = ldo2; g.epi = epi;
    if (!(fd_debug_flags() & 128)) {       // one launch: K split over the waves of a block (debug flag 128: the split-K pair)
        hipLaunchKernelGGL(gemm_skinny_fused_kernel, dim3(N / 16, (M + 15) / 16), dim3(SKF_NW * 64), 0, stream, g);
        FD_LAUNCH_RET();
    }
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3(N / 64, ksplit), dim3(256), 0, stream, g, workspace, K / ksplit);
    hipLaunchKernelGGL(gemm_skinny_epilogue_kernel, dim3((M * (N / 4) + 255) / 256), dim3(256), 0, stream, g,
                       (const float*)workspace, ksplit);
    FD_LAUNCH_RET();
}

namespace {
}  // namespace

using V2Kernel = void (*)(GemmArgsV2);
static const V2Kernel (*v2_kernel_table())[7] {
    static const V2Kernel kernels[2][7] = {
        {gemm_nt_v2_kernel<FEDDAT_EPI_BF16, 3>, gemm_nt_v2_kernel<FEDDAT_EPI_RESID_F32, 3>,
         gemm_nt_v2_kernel<FEDDAT_EPI_GELU, 3>, gemm_nt_v2_kernel<FEDDAT_EPI_MUL_DGELU, 3>,
         gemm_nt_v2_kernel<FEDDAT_EPI_F32, 3>, gemm_nt_v2_kernel<FEDDAT_EPI_GELU_G8, 3>,
         gemm_nt_v2_kernel<FEDDAT_EPI_MUL_G8, 3>},
        {gemm_nt_v2_kernel<FEDDAT_EPI_BF16, 4>, gemm_nt_v2_kernel<FEDDAT_EPI_RESID_F32, 4>,
         gemm_nt_v2_kernel<FEDDAT_EPI_GELU, 4>, gemm_nt_v2_kernel<FEDDAT_EPI_MUL_DGELU, 4>,
         gemm_nt_v2_kernel<FEDDAT_EPI_F32, 4>, gemm_nt_v2_kernel<FEDDAT_EPI_GELU_G8, 4>,
         gemm_nt_v2_kernel<FEDDAT_EPI_MUL_G8, 4>}};
    return kernels;
}

static const V2Kernel (*v3_kernel_table())[5] {        // [0]: 192-row tiles (RT = 6), [1]: 256 (RT = 8), [2]: 160 (RT = 5), [3]: 224 (RT = 7)
    static const V2Kernel kernels[4][5] = {
        {gemm_nt_v3_kernel<FEDDAT_EPI_BF16, 6>, gemm_nt_v3_kernel<FEDDAT_EPI_RESID_F32, 6>, gemm_nt_v3_kernel<FEDDAT_EPI_GELU, 6>,
         gemm_nt_v3_kernel<FEDDAT_EPI_MUL_DGELU, 6>, gemm_nt_v3_kernel<FEDDAT_EPI_F32, 6>},
        {gemm_nt_v3_kernel<FEDDAT_EPI_BF16, 8>, gemm_nt_v3_kernel<FEDDAT_EPI_RESID_F32, 8>, gemm_nt_v3_kernel<FEDDAT_EPI_GELU, 8>,
         gemm_nt_v3_kernel<FEDDAT_EPI_MUL_DGELU, 8>, gemm_nt_v3_kernel<FEDDAT_EPI_F32, 8>},
        {gemm_nt_v3_kernel<FEDDAT_EPI_BF16, 5>, gemm_nt_v3_kernel<FEDDAT_EPI_RESID_F32, 5>, gemm_nt_v3_kernel<FEDDAT_EPI_GELU, 5>,
         gemm_nt_v3_kernel<FEDDAT_EPI_MUL_DGELU, 5>, gemm_nt_v3_kernel<FEDDAT_EPI_F32, 5>},
        {gemm_nt_v3_kernel<FEDDAT_EPI_BF16, 7>, gemm_nt_v3_kernel<FEDDAT_EPI_RESID_F32, 7>, gemm_nt_v3_kernel<FEDDAT_EPI_GELU, 7>,
         gemm_nt_v3_kernel<FEDDAT_EPI_MUL_DGELU, 7>, gemm_nt_v3_kernel<FEDDAT_EPI_F32, 7>}};
    return kernels;
}
// "v4": two independent 128-row workgroups per CU (gemm_nt_v3_kernel<E, 4, 0, true>), every epilogue incl. the 8-bit codes
static const V2Kernel* v4_kernel_table() {
    static const V2Kernel kernels[7] = {
        gemm_nt_v3_kernel<FEDDAT_EPI_BF16, 4, 0, true>, gemm_nt_v3_kernel<FEDDAT_EPI_RESID_F32, 4, 0, true>,
        gemm_nt_v3_kernel<FEDDAT_EPI_GELU, 4, 0, true>, gemm_nt_v3_kernel<FEDDAT_EPI_MUL_DGELU, 4, 0, true>,
        gemm_nt_v3_kernel<FEDDAT_EPI_F32, 4, 0, true>, gemm_nt_v3_kernel<FEDDAT_EPI_GELU_G8, 4, 0, true>,
        gemm_nt_v3_kernel<FEDDAT_EPI_MUL_G8, 4, 0, true>};
    return kernels;
}
constexpr int V4_LDS = 2 * V3Cfg<4>::STAGE;      // 80 KiB: two workgroups fill the CU's 160 KiB exactly
static int v3_lds(int which) {
    return which == 1 ? V3Cfg<8>::LDS : which == 2 ? V3Cfg<5>::LDS : which == 3 ? V3Cfg<7>::LDS : V3Cfg<6>::LDS;
}

// diagnostics: resident workgroups per CU the runtime grants the dual form (2 = the design point; 1 = it degenerates into a
// one-wave-per-SIMD kernel with 128-row tiles)
extern "C" int feddat_gemm_dual_blocks_per_cu(int* out) {
    FD_CHECK_ARG(out);
    const V2Kernel k = v4_kernel_table()[FEDDAT_EPI_BF16];
    if (fd_set_max_lds((const void*)k, V4_LDS) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)k, 256, V4_LDS) != hipSuccess) return FEDDAT_ELAUNCH;
    *out = n;
    return FEDDAT_OK;
}

int fd_prepare_gemm_kernels() {
    for (int w = 0; w < 4; ++w)
        for (int e = 0; e < 5; ++e)
            if (fd_set_max_lds((const void*)v3_kernel_table()[w][e], v3_lds(w)) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    for (int w = 0; w < 2; ++w)
        for (int e = 0; e < 7; ++e)
            if (fd_set_max_lds((const void*)v2_kernel_table()[w][e], w ? V2Cfg<4>::LDS : V2Cfg<3>::LDS) != FEDDAT_OK)
                return FEDDAT_ELAUNCH;
    for (int e = 0; e < 7; ++e)
        if (fd_set_max_lds((const void*)v4_kernel_table()[e], V4_LDS) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    if (fd_set_max_lds((const void*)gemm_nt_mid_kernel, MID_LDS) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    return fd_set_max_lds((const void*)gemm_nt_kernel, 4 * TILE_BYTES);
}

// fp8 (e4m3) operands on the same persistent kernel: byte for byte the data movement of a bf16 product with K / 2
// "elements" (128 fp8 per 128-byte LDS row); only the MFMA -- ONE block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 per output
// tile and k-tile, unit block scales, twice the bf16 rate (debug flag 256: the K = 32 fp8 instruction it replaced, which
// issues at the bf16 rate) -- and the dequantising epilogue differ.
// shared launch of the fp8 persistent kernel; EPI_RESID_F32 / EPI_F32 through feddat_gemm_fp8_nt_f32
static int fp8_launch(GemmArgsV2& a2, int M, int N, int K, int epi, hipStream_t stream) {
    GemmArgs& g = a2.g;
    a2.dbg = FD_ABL(fd_debug_flags() & 8);       // -DFEDDAT_ABLATE build: 8 = skip the epilogue (k-loop timing)
    int n_cu = 0;
    if (fd_device_cus(&n_cu) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    const int tiles_n = N / V2_BN;
    auto plan = [&](int BMx, GemmArgsV2& o) {
        int nmt = (M + BMx - 1) / BMx;
        o.nx = 1;
        if ((size_t)N * K > (3u << 20) && tiles_n % 2 == 0 && nmt * tiles_n > n_cu) {
            o.nx = 2;
            nmt = (nmt + 3) & ~3;
        }
        const int bm = (M + nmt - 1) / nmt;
        o.bm = bm;
        o.tiles_m = o.nx == 1 ? (M + bm - 1) / bm : nmt;
        o.tm_per = o.tiles_m / (8 / o.nx);
        o.tn_per = tiles_n / o.nx;
        return (o.tiles_m * tiles_n + n_cu - 1) / n_cu;
    };
    GemmArgsV2 a3 = a2, a4 = a2;
    const int rounds3 = plan(192, a3), rounds4 = plan(256, a4);
    // . gelu'(bf16 u) and the dequantising + residual epilogue stay on 192-row tiles (their 256-row instantiations spill)
    const bool wm4 = rounds4 * 12 < rounds3 * 10 && epi != FEDDAT_EPI_MUL_DGELU && epi != FEDDAT_EPI_RESID_F32;
    a2 = wm4 ? a4 : a3;
    (void)g;
    using KernelFn = void (*)(GemmArgsV2);
    KernelFn kern = nullptr;
#define FD_FP8_PICK(E) kern = wm4 ? gemm_nt_v2_kernel<E, 4, true> : gemm_nt_v2_kernel<E, 3, true>
    switch (epi) {
        case FEDDAT_EPI_MUL_DGELU: kern = gemm_nt_v2_kernel<FEDDAT_EPI_MUL_DGELU, 3, true>; break;
        case FEDDAT_EPI_MUL_G8: FD_FP8_PICK(FEDDAT_EPI_MUL_G8); break;
        case FEDDAT_EPI_GELU_G8: FD_FP8_PICK(FEDDAT_EPI_GELU_G8); break;
        case FEDDAT_EPI_MUL_G8_F8: FD_FP8_PICK(FEDDAT_EPI_MUL_G8_F8); break;
        case FEDDAT_EPI_GELU_G8_F8: FD_FP8_PICK(FEDDAT_EPI_GELU_G8_F8); break;
        case FEDDAT_EPI_RESID_F32: kern = gemm_nt_v2_kernel<FEDDAT_EPI_RESID_F32, 3, true>; break;
        case FEDDAT_EPI_F32: FD_FP8_PICK(FEDDAT_EPI_F32); break;
        case FEDDAT_EPI_BF16:
            if (g.amx) kern = wm4 ? gemm_nt_v2_kernel<FEDDAT_EPI_BF16, 4, true, false, true> : gemm_nt_v2_kernel<FEDDAT_EPI_BF16, 3, true, false, true>;
            else if (fd_debug_flags() & 256) kern = wm4 ? gemm_nt_v2_kernel<FEDDAT_EPI_BF16, 4, true, true> : gemm_nt_v2_kernel<FEDDAT_EPI_BF16, 3, true, true>;
            else FD_FP8_PICK(FEDDAT_EPI_BF16);
            break;
        case FEDDAT_EPI_GELU:      // debug flag 256, tools/ A/B only: the CDNA3-style K = 32 fp8 instruction (bf16 issue rate)
            if (fd_debug_flags() & 256) kern = wm4 ? gemm_nt_v2_kernel<FEDDAT_EPI_GELU, 4, true, true> : gemm_nt_v2_kernel<FEDDAT_EPI_GELU, 3, true, true>;
            else FD_FP8_PICK(FEDDAT_EPI_GELU);
            break;
        default: return FEDDAT_EINVAL;
    }
#undef FD_FP8_PICK
    const int lds_bytes = (wm4 ? V2Cfg<4>::LDS : V2Cfg<3>::LDS) + (g.amx ? 2 * 256 * (wm4 ? 4 : 3) : 0);   // MXA: + the scale stages
    if (fd_set_max_lds((const void*)kern, lds_bytes) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    const int total = a2.tiles_m * tiles_n;
    hipLaunchKernelGGL(kern, dim3(total < n_cu ? total : n_cu), dim3(512), lds_bytes, stream, a2);
    FD_LAUNCH_RET();
}

extern "C" int feddat_gemm_fp8mx_nt(const void* A8, int lda, const uint8_t* a_mx, int ld_mx, const void* B8, int ldb,
                                    const float* b_scale, int M, int N, int K, const float* bias, void* out_bf16, int ldo16,
                                    hipStream_t stream) {
    FD_CHECK_ARG(A8 && a_mx && B8 && b_scale && out_bf16 && M >= 1024 && N > 0 && N % V2_BN == 0 && K > 0 && K % 128 == 0);
    FD_CHECK_ARG(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K && ((uintptr_t)out_bf16 & 15) == 0 && ldo16 % 8 == 0);
    FD_CHECK_ARG(ld_mx % 4 == 0 && ld_mx >= K / 32 && ((uintptr_t)a_mx & 3) == 0);
    GemmArgsV2 a2;
    GemmArgs& g = a2.g;
    g = GemmArgs{};
    g.A = (const bf16*)A8; g.B = (const bf16*)B8; g.bias = bias; g.sa = nullptr; g.sw = b_scale; g.amx = a_mx; g.ld_mx = ld_mx;
    g.out_bf16 = (bf16*)out_bf16;
    g.M = M; g.N = N; g.K = K / 2; g.lda = lda / 2; g.ldb = ldb / 2; g.ldo16 = ldo16; g.epi = FEDDAT_EPI_BF16;
    return fp8_launch(a2, M, N, K, FEDDAT_EPI_BF16, stream);
}

extern "C" int feddat_gemm_fp8_nt(const void* A8, int lda, const float* a_scale, const void* B8, int ldb,
                                  const float* b_scale, int M, int N, int K, int epi, const float* bias, const void* aux,
                                  int ldaux, void* out_bf16, int ldo16, void* out2_bf16, int ldo2, hipStream_t stream) {
    FD_CHECK_ARG(A8 && B8 && a_scale && b_scale && out_bf16 && M >= 1024 && N > 0 && N % V2_BN == 0 && K > 0 && K % 128 == 0);
    const bool g8in = epi == FEDDAT_EPI_MUL_G8 || epi == FEDDAT_EPI_MUL_G8_F8;
    const bool g8out = epi == FEDDAT_EPI_GELU_G8 || epi == FEDDAT_EPI_GELU_G8_F8;
    const bool f8out = epi == FEDDAT_EPI_GELU_G8_F8 || epi == FEDDAT_EPI_MUL_G8_F8;
    FD_CHECK_ARG(epi == FEDDAT_EPI_BF16 || epi == FEDDAT_EPI_GELU || epi == FEDDAT_EPI_MUL_DGELU || g8in || g8out);
    FD_CHECK_ARG(epi != FEDDAT_EPI_MUL_DGELU || (aux && ldaux % 8 == 0 && ((uintptr_t)aux & 15) == 0 &&
                                                (size_t)M * ldaux * 2 < (1ull << 32)));
    FD_CHECK_ARG(!g8in || (aux && ldaux % 16 == 0 && ldaux >= N && ((uintptr_t)aux & 15) == 0 && (size_t)M * ldaux < (1ull << 32)));
    FD_CHECK_ARG(!g8out || (out2_bf16 && ldo2 % 16 == 0 && ldo2 >= N));
    FD_CHECK_ARG(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K && ((uintptr_t)out_bf16 & 15) == 0);
    FD_CHECK_ARG(f8out ? (ldo16 % 16 == 0 && ldo16 >= N) : ldo16 % 8 == 0);
    FD_CHECK_ARG(!out2_bf16 || (ldo2 % 8 == 0 && ((uintptr_t)out2_bf16 & 15) == 0));
    GemmArgsV2 a2;
    GemmArgs& g = a2.g;
    g = GemmArgs{};
    g.A = (const bf16*)A8; g.B = (const bf16*)B8; g.bias = bias; g.sa = a_scale; g.sw = b_scale;
    g.aux = (const bf16*)aux; g.ldaux = ldaux;
    g.out_bf16 = (bf16*)out_bf16; g.out2_bf16 = (bf16*)out2_bf16;
    g.M = M; g.N = N; g.K = K / 2; g.lda = lda / 2; g.ldb = ldb / 2; g.ldo16 = ldo16; g.ldo2 = ldo2; g.epi = epi;
    return fp8_launch(a2, M, N, K, epi, stream);
}

extern "C" int feddat_gemm_fp8_nt_f32(const void* A8, int lda, const float* a_scale, const void* B8, int ldb,
                                      const float* b_scale, int M, int N, int K, const float* bias, const float* resid,
                                      int ldr, float* out_f32, int ldo32, hipStream_t stream) {
    FD_CHECK_ARG(A8 && B8 && a_scale && b_scale && out_f32 && M >= 1024 && N > 0 && N % V2_BN == 0 && K > 0 && K % 128 == 0);
    FD_CHECK_ARG(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K && ldo32 % 4 == 0 && ldo32 >= N);
    FD_CHECK_ARG(!resid || (ldr % 4 == 0 && ldr >= N && (size_t)M * ldr * 4 < (1ull << 32)));
    GemmArgsV2 a2;
    GemmArgs& g = a2.g;
    g = GemmArgs{};
    g.A = (const bf16*)A8; g.B = (const bf16*)B8; g.bias = bias; g.sa = a_scale; g.sw = b_scale;
    g.resid = resid; g.ldr = ldr; g.out_f32 = out_f32; g.ldo32 = ldo32;
    const int epi = resid ? FEDDAT_EPI_RESID_F32 : FEDDAT_EPI_F32;
    g.M = M; g.N = N; g.K = K / 2; g.lda = lda / 2; g.ldb = ldb / 2; g.epi = epi;
    return fp8_launch(a2, M, N, K, epi, stream);
}

extern "C" int feddat_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi,
                                   const float* bias, const float* resid, int ldr, const void* aux, int ldaux,
                                   float* out_f32, int ldo32, void* out_bf16, int ldo16, void* out2_bf16, int ldo2,
                                   hipStream_t stream) {
    FD_CHECK_ARG(A && B && M > 0 && N > 0 && K > 0);
    bool use_v2 = (N % V2_BN == 0) && (K % BK == 0) && (M >= 1024);
    // Medium M (ALBEF's stacked text streams: 2 x 800 rows): the persistent kernels would put 1600 x 768 on 9 x 4 = 36 tiles, i.e.
    // 36 of the 256 CUs; the small-tile kernel fills the chip with 64 x 64 tiles (same k order: bit-identical results).  Taken
    // when a launch has fewer 192-row tiles than 0.6 x the CUs; not for the gelu' code epilogues (persistent kernels only);
    // debug flag 1 (everything on the two-group persistent kernel) keeps the old routing (A/B: tools/albef_stack_ab.py).
    bool small_grid = false;
    if (use_v2 && M < 4096 && epi != FEDDAT_EPI_GELU_G8 && epi != FEDDAT_EPI_MUL_G8 && !(fd_debug_flags() & 1)) {
        int n_cu = 0;
        if (fd_device_cus(&n_cu) != FEDDAT_OK) return FEDDAT_ELAUNCH;
        // (debug flag 128 = "no small-tile kernel": such launches stay on the persistent kernel instead of falling through to the
        //  128 x 128 kernel, which needs N % 128 == 0 -- N = 192 x odd would silently lose its tail columns there)
        small_grid = ((M + 191) / 192) * (N / V2_BN) * 10 < n_cu * 6 && !(fd_debug_flags() & 128);
        if (small_grid) use_v2 = false;
    }
    FD_CHECK_ARG((N % BN == 0 || use_v2 || ((M < 1024 || small_grid) && N % 64 == 0)) && K % BK == 0);
    FD_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K);
    switch (epi) {
        case FEDDAT_EPI_BF16: FD_CHECK_ARG(out_bf16 && ldo16 % 4 == 0); break;
        case FEDDAT_EPI_RESID_F32:
            FD_CHECK_ARG(out_f32 && resid && ldr % 4 == 0 && ldo32 % 4 == 0 && (size_t)M * ldr * 4 < (1ull << 32));
            break;
        case FEDDAT_EPI_GELU: FD_CHECK_ARG(out_bf16 && ldo16 % 4 == 0 && (!out2_bf16 || ldo2 % 4 == 0)); break;
        case FEDDAT_EPI_MUL_DGELU:
            FD_CHECK_ARG(out_bf16 && aux && ldaux % 4 == 0 && ldo16 % 4 == 0 && (size_t)M * ldaux * 2 < (1ull << 32));
            break;
        case FEDDAT_EPI_F32: FD_CHECK_ARG(out_f32 && ldo32 % 4 == 0); break;
        case FEDDAT_EPI_GELU_G8:
            FD_CHECK_ARG(use_v2 && out_bf16 && out2_bf16 && ldo2 % 16 == 0 && ldo2 >= N && ((uintptr_t)out2_bf16 & 15) == 0);
            break;
        case FEDDAT_EPI_MUL_G8:
            FD_CHECK_ARG(use_v2 && out_bf16 && aux && ldaux % 16 == 0 && ldaux >= N && ((uintptr_t)aux & 15) == 0 &&
                         (size_t)M * ldaux < (1ull << 32));
            break;
        default: return FEDDAT_EINVAL;
    }
    const bool g8 = epi == FEDDAT_EPI_GELU_G8 || epi == FEDDAT_EPI_MUL_G8;
    if (use_v2) {      // 16-byte bf16 stores / aux loads of the persistent kernel's epilogue
        if (epi == FEDDAT_EPI_BF16 || epi == FEDDAT_EPI_GELU || epi == FEDDAT_EPI_MUL_DGELU || g8)
            FD_CHECK_ARG(ldo16 % 8 == 0 && ((uintptr_t)out_bf16 & 15) == 0);
        if (epi == FEDDAT_EPI_GELU && out2_bf16) FD_CHECK_ARG(ldo2 % 8 == 0 && ((uintptr_t)out2_bf16 & 15) == 0);
        if (epi == FEDDAT_EPI_MUL_DGELU) FD_CHECK_ARG(ldaux % 8 == 0 && ((uintptr_t)aux & 15) == 0);
    }
    GemmArgs g{};
    g.A = (const bf16*)A; g.B = (const bf16*)B; g.bias = bias; g.resid = resid; g.aux = (const bf16*)aux;
    g.out_f32 = out_f32; g.out_bf16 = (bf16*)out_bf16; g.out2_bf16 = (bf16*)out2_bf16;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldr = ldr; g.ldaux = ldaux;
    g.ldo32 = ldo32; g.ldo16 = ldo16; g.ldo2 = ldo2; g.epi = epi;
    g.nostore = FD_ABL(((fd_debug_flags() & 16) ? 1 : 0) | ((fd_debug_flags() & 4) ? 2 : 0));   // 4: ablate the GELU math
    if (use_v2) {
        GemmArgsV2 a2;
        a2.g = g;
        int dbg = fd_debug_flags();            // tools/ ablations only (feddat_set_debug_flags); 0 in production
        a2.dbg = dbg;
        int n_cu = 0;
        if (fd_device_cus(&n_cu) != FEDDAT_OK) return FEDDAT_ELAUNCH;
        // tools/overlap_probe.py: bits 28..31 of the debug flags cap the persistent grid at 16 x value workgroups, so that a
        // launch on a side stream leaves compute units to the kernels of the main stream
        if (const int cap16 = (dbg >> 28) & 0xf) n_cu = n_cu < cap16 * 16 ? n_cu : cap16 * 16;
        const int tiles_n = N / V2_BN;
        // balanced M tiles of <= BM rows; XCD-aware tile order: split the XCDs over N as well when B (N x K bf16) would
        // not stay in a 4 MiB L2 and the launch takes more than one round of tiles
        auto plan = [&](int BMx, GemmArgsV2& o) {
            int nmt = (M + BMx - 1) / BMx;
            o.nx = 1;
            if ((size_t)N * K * 2 > (3u << 20) && tiles_n % 2 == 0 && nmt * tiles_n > n_cu) {
                o.nx = 2;
                nmt = (nmt + 3) & ~3;                      // 4 M groups of equal size
            }
            const int bm = (M + nmt - 1) / nmt;
            o.bm = bm;
            o.tiles_m = o.nx == 1 ? (M + bm - 1) / bm : nmt;
            o.tm_per = o.tiles_m / (8 / o.nx);
            o.tn_per = tiles_n / o.nx;
            return (o.tiles_m * tiles_n + n_cu - 1) / n_cu;     // rounds of the persistent grid
        };
        // The DUAL form (selection flags 1 | 2 together: every persistent launch; 1 | 2 | 64: only launches of at least two full
        // rounds of the doubled grid, e.g. N = 3072 at M = 11 840: 96 x 16 tiles = 3.0 rounds of 512): two independent 128-row
        // workgroups per CU.  Measured, not the default: profiles/r06_gemm_dual_ab.txt, DESIGN.md section 7e.
        if ((dbg & 3) == 3 && K / BK >= 3) {      // (its self-contained tiles need a first, a penultimate and a last k-tile)
            GemmArgsV2 a1 = a2;
            const int cu1 = n_cu;
            n_cu *= 2;
            plan(128, a1);
            n_cu = cu1;
            const int total1 = a1.tiles_m * tiles_n;
            if (!(dbg & 64) || total1 >= 4 * n_cu) {
                const V2Kernel k4 = v4_kernel_table()[epi];
                if (fd_set_max_lds((const void*)k4, V4_LDS) != FEDDAT_OK) return FEDDAT_ELAUNCH;
                hipLaunchKernelGGL(k4, dim3(total1 < 2 * n_cu ? total1 : 2 * n_cu), dim3(256), V4_LDS, stream, a1);
                FD_LAUNCH_RET();
            }
        }
        if ((dbg & 3) == 3) {       // not taken: the production routing below
            dbg &= ~3;
            a2.dbg = dbg;
        }
        GemmArgsV2 a3 = a2, a4 = a2, a5 = a2, a7 = a2;
        const int rounds3 = plan(192, a3), rounds4 = plan(256, a4), rounds5 = plan(160, a5), rounds7 = plan(224, a7);
        // a 256-row tile costs about 1.2x a 192-row tile (48 vs 36 MFMAs per k-tile and wave, L phase 20 vs 18 reads)
        bool wm4 = rounds4 * 12 < rounds3 * 10;
        if (epi == FEDDAT_EPI_MUL_DGELU) wm4 = false;      // its 256-row instantiation spills (180 B of scratch per lane and tile)
        if (dbg & 32) wm4 = false;
        if (dbg & 64) wm4 = true;
        a2 = wm4 ? a4 : a3;
        // v3 (one wave per SIMD) has the faster k-loop (1.1-1.28 PF/s against 0.96-1.15) but only four waves to run an
        // epilogue: it takes every launch except the two heavy epilogues (GELU with two outputs; . gelu'(aux) with its cold
        // aux operand) -- in isolation v3 is level or ahead on those too (84 against 98 us for . gelu'), in the step, with
        // nothing cache-warm, it is behind (86 / 89 us against 82 / 81: tools/step_breakdown.py --detail); debug flag 1
        // keeps everything on v2, flag 2 forces v3
        const bool v3_pick = epi != FEDDAT_EPI_GELU && epi != FEDDAT_EPI_MUL_DGELU && !g8;
        if (g8 && (dbg & (2 | 512))) return FEDDAT_EINVAL;      // the code epilogues exist on the two-group (and the dual) kernel only
#ifdef FEDDAT_ABLATE
        if (dbg & 512) {        // tools/gemm_defer_probe.py: the deferred-epilogue timing probe (RT = 6; wrong results)
            static const V2Kernel fk[5] = {gemm_nt_v3_kernel<FEDDAT_EPI_BF16, 6, 1>, gemm_nt_v3_kernel<FEDDAT_EPI_RESID_F32, 6, 1>,
                                           gemm_nt_v3_kernel<FEDDAT_EPI_GELU, 6, 1>, gemm_nt_v3_kernel<FEDDAT_EPI_MUL_DGELU, 6, 1>,
                                           gemm_nt_v3_kernel<FEDDAT_EPI_F32, 6, 1>};
            a2 = a3;
            if (fd_set_max_lds((const void*)fk[epi], V3Cfg<6>::LDS) != FEDDAT_OK) return FEDDAT_ELAUNCH;
            const int total3 = a2.tiles_m * (N / V2_BN);
            hipLaunchKernelGGL(fk[epi], dim3(total3 < n_cu ? total3 : n_cu), dim3(256), V3Cfg<6>::LDS, stream, a2);
            FD_LAUNCH_RET();
        }
#endif
        if (((dbg & 2) || v3_pick) && !(dbg & 1)) {
            const bool rt8 = (dbg & 64) ? true : (dbg & 32) ? false : wm4;
            // 160-row tiles (RT = 5, ~0.87 of a 192-row tile's time) where they fill the rounds better: 18 464 rows x N = 768
            // (ALBEF's ViT) = 388 tiles of 192 rows = 1.52 rounds of the 256 CUs, paid as 2; 464 tiles of 160 rows = 1.81 rounds,
            // paid as 2 x 0.87.  configs[1]'s 11 840 rows (64 x 185) keep their exact rounds of 192-row tiles.
            // 224-row tiles (RT = 7, ~1.1 of a 192-row tile's time) likewise: 18 464 rows x N = 2304 = 4 rounds of either 256- or
            // 224-row tiles
            const int cost68 = rt8 ? rounds4 * 120 : rounds3 * 100;
            const bool odd_ok = !(dbg & (32 | 64 | (1 << 27)));
            const bool rt5 = odd_ok && rounds5 * 87 < cost68 && rounds5 * 87 <= rounds7 * 110;
            const bool rt7 = odd_ok && !rt5 && rounds7 * 110 < cost68;
            const int which = rt5 ? 2 : rt7 ? 3 : rt8 ? 1 : 0;
            const V2Kernel k3 = v3_kernel_table()[which][epi];
            a2 = rt5 ? a5 : rt7 ? a7 : rt8 ? a4 : a3;
            const int lds3 = v3_lds(which);
            if (fd_set_max_lds((const void*)k3, lds3) != FEDDAT_OK) return FEDDAT_ELAUNCH;
            const int total3 = a2.tiles_m * (N / V2_BN);
            hipLaunchKernelGGL(k3, dim3(total3 < n_cu ? total3 : n_cu), dim3(256), lds3, stream, a2);
            FD_LAUNCH_RET();
        }
        const V2Kernel kern = v2_kernel_table()[wm4 ? 1 : 0][epi];
        const int lds_bytes = wm4 ? V2Cfg<4>::LDS : V2Cfg<3>::LDS;
        if (fd_set_max_lds((const void*)kern, lds_bytes) != FEDDAT_OK) return FEDDAT_ELAUNCH;
        const int total = a2.tiles_m * (N / V2_BN);
        int grid = total < n_cu ? total : n_cu;
        if (FD_ABL((dbg >> 8) & 0xfff) > 0 && FD_ABL((dbg >> 8) & 0xfff) < grid) grid = (dbg >> 8) & 0xfff;      // ablation: cap the number of persistent blocks
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, stream, a2);
        FD_LAUNCH_RET();
    }
    // few rows and too few 128 x 128 tiles to fill the chip: the latency-oriented small-tile kernel
    const bool v1_ok = N % BN == 0;
    if ((M < 1024 || small_grid) && N % 64 == 0 && (!v1_ok || ((M + BM - 1) / BM) * (N / BN) < 150) && !(fd_debug_flags() & 128)) {
        const int tm = (M + 63) / 64;
        if (fd_set_max_lds((const void*)gemm_nt_mid_kernel, MID_LDS) != FEDDAT_OK) return FEDDAT_ELAUNCH;
        hipLaunchKernelGGL(gemm_nt_mid_kernel, dim3(tm * (N / 64)), dim3(256), MID_LDS, stream, g);
        FD_LAUNCH_RET();
    }
    FD_CHECK_ARG(v1_ok);      // the 128 x 128 kernel has no column tail
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    if (fd_set_max_lds((const void*)gemm_nt_kernel, 4 * TILE_BYTES) != FEDDAT_OK) return FEDDAT_ELAUNCH;
    hipLaunchKernelGGL(gemm_nt_kernel, dim3(tiles), dim3(256), 4 * TILE_BYTES, stream, g);
    FD_LAUNCH_RET();
}
