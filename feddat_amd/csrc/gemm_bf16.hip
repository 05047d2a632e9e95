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
#include "common.hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
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

    // epilogue: lane holds C[m = .. + (lane & 15)][n = .. + 4 * (lane >> 4) + 0..3]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fg * 4;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bias4 = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + frow;
            if (m >= g.M) continue;
            f32x4 v = acc[i][j] + bias4;
            switch (g.epi) {
                case FEDDAT_EPI_BF16: {
                    *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(v);
                } break;
                case FEDDAT_EPI_RESID_F32: {
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(g.resid + (size_t)m * g.ldr + n);
                    *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) = v + rr;
                } break;
                case FEDDAT_EPI_GELU: {
                    if (g.out2_bf16) *reinterpret_cast<bf16x4*>(g.out2_bf16 + (size_t)m * g.ldo2 + n) = cvt4(v);
                    f32x4 a;
                    a[0] = gelu_f(v[0]); a[1] = gelu_f(v[1]); a[2] = gelu_f(v[2]); a[3] = gelu_f(v[3]);
                    *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(a);
                } break;
                case FEDDAT_EPI_MUL_DGELU: {
                    const bf16x4 u = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)m * g.ldaux + n);
                    f32x4 a;
                    a[0] = v[0] * gelu_grad_f((float)u[0]); a[1] = v[1] * gelu_grad_f((float)u[1]);
                    a[2] = v[2] * gelu_grad_f((float)u[2]); a[3] = v[3] * gelu_grad_f((float)u[3]);
                    *reinterpret_cast<bf16x4*>(g.out_bf16 + (size_t)m * g.ldo16 + n) = cvt4(a);
                } break;
                case FEDDAT_EPI_F32: {
                    *reinterpret_cast<f32x4*>(g.out_f32 + (size_t)m * g.ldo32 + n) = v;
                } break;
                default: break;
            }
        }
    }
}

}  // namespace

extern "C" int feddat_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, int M, int N, int K, int epi,
                                   const float* bias, const float* resid, int ldr, const void* aux, int ldaux,
                                   float* out_f32, int ldo32, void* out_bf16, int ldo16, void* out2_bf16, int ldo2,
                                   hipStream_t stream) {
    FD_CHECK_ARG(A && B && M > 0 && N > 0 && K > 0);
    FD_CHECK_ARG(N % BN == 0 && K % BK == 0);
    FD_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K);
    switch (epi) {
        case FEDDAT_EPI_BF16: FD_CHECK_ARG(out_bf16 && ldo16 % 4 == 0); break;
        case FEDDAT_EPI_RESID_F32: FD_CHECK_ARG(out_f32 && resid && ldr % 4 == 0 && ldo32 % 4 == 0); break;
        case FEDDAT_EPI_GELU: FD_CHECK_ARG(out_bf16 && ldo16 % 4 == 0 && (!out2_bf16 || ldo2 % 4 == 0)); break;
        case FEDDAT_EPI_MUL_DGELU: FD_CHECK_ARG(out_bf16 && aux && ldaux % 4 == 0 && ldo16 % 4 == 0); break;
        case FEDDAT_EPI_F32: FD_CHECK_ARG(out_f32 && ldo32 % 4 == 0); break;
        default: return FEDDAT_EINVAL;
    }
    GemmArgs g;
    g.A = (const bf16*)A; g.B = (const bf16*)B; g.bias = bias; g.resid = resid; g.aux = (const bf16*)aux;
    g.out_f32 = out_f32; g.out_bf16 = (bf16*)out_bf16; g.out2_bf16 = (bf16*)out2_bf16;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldr = ldr; g.ldaux = ldaux;
    g.ldo32 = ldo32; g.ldo16 = ldo16; g.ldo2 = ldo2; g.epi = epi;
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_nt_kernel, dim3(tiles), dim3(256), 4 * TILE_BYTES, stream, g);
    FD_LAUNCH_RET();
}
