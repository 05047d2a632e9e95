// K2b: general fused attention (any S_q, S_kv; separate Q / K / V operands; key-padding mask; causal), forward and
// backward -- the ALBEF path: ViT-B/16 self-attention over 577 tokens (src/modeling/models/vit.py:60-76), BERT self- and
// cross-attention of the text encoder / decoder (src/modeling/models/xbert.py BertSelfAttention: S_q <= 40,
// S_kv = 577 image tokens or <= 40 text tokens; decoder causal).  The ViLT kernels (attention.hip) keep a whole head in
// LDS and stop at S = 320; here K / V (forward, dQ) or Q / dO (dK, dV) stream through LDS in 64-row chunks with an
// online softmax, so S is unbounded.  head_dim = 64, scores scaled by 1/8.
//
// One workgroup (4 waves) = one (sample, head, 64-row block); wave w owns rows 16 w .. 16 w + 15 of the block.  All
// HBM accesses are row-contiguous 16-byte pieces (operands are staged through LDS cooperatively, results leave through
// LDS): the MFMA lane layout (lane & 15 = row) would make every VMEM instruction 64 separate L1 accesses.
// Operand plumbing as in attention.hip: score tiles are produced in the orientation whose accumulator layout is the
// operand layout of the next product.
#include "common.hip.h"

#include <type_traits>

namespace {

constexpr int D = 64;
constexpr int ROWB = D * 2;            // bytes per LDS row
constexpr int BLK = 64;                // rows per block and per streamed chunk
constexpr float LOG2E = 1.4426950408889634f;
constexpr float SC = 0.125f * LOG2E;   // scores in the log2 domain

__device__ __forceinline__ int sw_off(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ bf16x8 row_frag(const char* m, int row, int chunk) {
    return *reinterpret_cast<const bf16x8*>(m + sw_off(row, chunk));
}
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
// cooperative, row-contiguous load of rows [r0, r0 + ROWS) of one head's [S][64] bf16 slice into swizzled LDS (zero beyond S)
template <int ROWS>
__device__ __forceinline__ void load_rows(const bf16* __restrict__ src, long ld, int r0, int S, char* dst, int tid) {
#pragma unroll
    for (int idx = tid; idx < ROWS * 8; idx += 256) {
        const int row = idx >> 3, chunk = idx & 7;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 + row < S) v = *reinterpret_cast<const bf16x8*>(src + (size_t)(r0 + row) * ld + chunk * 8);
        *reinterpret_cast<bf16x8*>(dst + sw_off(row, chunk)) = v;
    }
}
// the same for one streamed 64-row chunk, split in two: the loads of chunk j + 1 are issued (into registers) before the
// products on chunk j and land in LDS after them
__device__ __forceinline__ void chunk_fetch(const bf16* __restrict__ src, long ld, int r0, int S, int tid, bf16x8 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 32 * i, chunk = tid & 7;
        v[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 + row < S) v[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)(r0 + row) * ld + chunk * 8);
    }
}
__device__ __forceinline__ void chunk_commit(char* dst, int tid, const bf16x8 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 32 * i, chunk = tid & 7;
        *reinterpret_cast<bf16x8*>(dst + sw_off(row, chunk)) = v[i];
    }
}
// The streamed side of a kernel: two operands (K and V, or Q and dO) of one head, 64-row chunks, two 16-byte pieces per
// lane and operand.  The lane's row pointers ADVANCE chunk by chunk (no per-chunk address arithmetic), and only a chunk that
// crosses the operand's last row takes the bounds-checked path (block-uniform branch): the loop around the products is
// VALU-issue bound (three waves per SIMD share one VALU port), so every instruction outside the softmax counts.
struct Stream2 {
    const bf16 *pa, *pb;          // this lane's piece 0 of the NEXT chunk to fetch (piece 1: 32 rows further)
    long step_a, step_b;          // elements per 32 rows
    int next_row;                 // first row of that chunk
    __device__ __forceinline__ void init(const bf16* A, long lda, const bf16* B, long ldb, int r0, int tid) {
        const int row = tid >> 3, chunk = tid & 7;
        pa = A + (size_t)(r0 + row) * lda + chunk * 8;
        pb = B + (size_t)(r0 + row) * ldb + chunk * 8;
        step_a = 32 * lda;
        step_b = 32 * ldb;
        next_row = r0;
    }
    __device__ __forceinline__ void fetch(int S, int tid, bf16x8 (&va)[2], bf16x8 (&vb)[2]) {
        if (next_row + BLK <= S) {
            va[0] = *reinterpret_cast<const bf16x8*>(pa);
            va[1] = *reinterpret_cast<const bf16x8*>(pa + step_a);
            vb[0] = *reinterpret_cast<const bf16x8*>(pb);
            vb[1] = *reinterpret_cast<const bf16x8*>(pb + step_b);
        } else {
            const int row = next_row + (tid >> 3);
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            va[0] = va[1] = vb[0] = vb[1] = z;
            if (row < S) {
                va[0] = *reinterpret_cast<const bf16x8*>(pa);
                vb[0] = *reinterpret_cast<const bf16x8*>(pb);
            }
            if (row + 32 < S) {
                va[1] = *reinterpret_cast<const bf16x8*>(pa + step_a);
                vb[1] = *reinterpret_cast<const bf16x8*>(pb + step_b);
            }
        }
        pa += 2 * step_a;
        pb += 2 * step_b;
        next_row += BLK;
    }
};
// LDS side: rows tid / 8 and tid / 8 + 32 of the chunk tile ((row + 32) & 7 == row & 7: one swizzled offset + 32 rows)
__device__ __forceinline__ void chunk_commit2(char* da, char* db, int tid, const bf16x8 (&va)[2], const bf16x8 (&vb)[2]) {
    const int off = sw_off(tid >> 3, tid & 7);
    *reinterpret_cast<bf16x8*>(da + off) = va[0];
    *reinterpret_cast<bf16x8*>(da + off + 32 * ROWB) = va[1];
    *reinterpret_cast<bf16x8*>(db + off) = vb[0];
    *reinterpret_cast<bf16x8*>(db + off + 32 * ROWB) = vb[1];
}
// a wave's [16][64] result tile in accumulator layout (row rowbase + i16, cols 16 dt + 4 g ..) goes to the staging tile,
// which then leaves LDS row-contiguously
__device__ __forceinline__ void put_acc(char* stg, int rowbase, int lane, const f32x4 (&o)[4], float mul) {
    const int g = lane >> 4, i16 = lane & 15;
    const int row = rowbase + i16;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int col = dt * 16 + 4 * g;
        *reinterpret_cast<bf16x4*>(stg + sw_off(row, col >> 3) + ((col & 7) << 1)) =
            cvt4(o[dt] * f32x4{mul, mul, mul, mul});
    }
}
template <int ROWS>
__device__ __forceinline__ void store_rows(bf16* __restrict__ dst, long ld, int r0, int S, const char* stg, int tid) {
#pragma unroll
    for (int idx = tid; idx < ROWS * 8; idx += 256) {
        const int row = idx >> 3, chunk = idx & 7;
        if (r0 + row < S)
            *reinterpret_cast<bf16x8*>(dst + (size_t)(r0 + row) * ld + chunk * 8) =
                *reinterpret_cast<const bf16x8*>(stg + sw_off(row, chunk));
    }
}

// XCD-aware work order.  The blocks of one (sample, head) -- its 64 QT-row blocks, 5 at 577 tokens -- all stream the SAME
// K / V (or Q / dO) through LDS, and consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with a private L2:
// in launch order the 5 blocks of a head sat on 5 different L2s and the streamed operand was fetched from HBM / MALL five
// times.  Workgroup i (XCD i % 8, that XCD's (i / 8)-th workgroup) therefore takes work item start(i % 8) + i / 8, so every
// XCD owns a contiguous range of (sample, head, block) items and the blocks of a head run back to back on one L2.
__device__ __forceinline__ void xcd_work_item(int& x, int& y, int& z) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy * (int)gridDim.z;
    const int i = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int c = i & 7, q = total >> 3, r = total & 7;
    const int w = c * q + (c < r ? c : r) + (i >> 3);
    x = w % gx;
    const int t = w / gx;
    y = t % gy;
    z = t / gy;
}

struct Attn2Args {
    const bf16 *q, *k, *v;
    long ldq, ldk, ldv;          // row strides (elements); head h at column 64 h
    long sq_b, skv_b;            // rows per sample of the q-side / kv-side operands (batch strides = rows * ld)
    const uint8_t* kmask;        // [B, Skv] 1 = attend, or null
    int Sq, Skv, heads, causal;
    bf16* o;  long ldo;          // ctx [B*Sq, ldo]
    float* lse;                  // [B, heads, Sq]
    // backward
    const bf16* dout;  long lddo;
    float* dsum;                 // [B, heads, Sq]: rowsum(dO * O), written by the dQ kernel, read by the dK/dV kernel
    bf16 *dq, *dk, *dv;  long lddq, lddk, lddv;
    // dropout on the attention probabilities (BertSelfAttention, xbert.py:333): element ((b * heads + h) * Sq + q) * Skv + key
    // of the [B, heads, Sq, Skv] probability tensor is kept iff fd_drop_keep says so; p = 0: off (the DROP = false kernels)
    float drop_p;  uint32_t dkey0, dkey1;  const int* dstep;
};

// All three kernels are templated on QT = 16-row tiles per wave on the block's own side (block = 64 QT rows): every
// fragment of the streamed side read from LDS feeds QT MFMAs.  With QT = 1 a chunk costs the CU's one LDS pipe ~2x the
// cycles its four MFMA pipes need (4 waves x (8 ds_read_b128 + 16 ds_read_b64_tr) against 16 MFMAs per wave); QT = 2 is
// used for long sequences (577 image tokens), QT = 1 for the <= 64-row text streams.
template <int QT, bool CAUSAL, bool DROP, bool MASK = true>
__global__ __launch_bounds__(256, 3) void attn2_fwd_kernel(Attn2Args a) {
    static_assert(MASK || (!CAUSAL && !DROP), "the mask-free form is the plain one");
    constexpr int QB = BLK * QT;
    __shared__ __attribute__((aligned(16))) char Qs[QB * ROWB], Ks[BLK * ROWB], Vs[BLK * ROWB];
    __shared__ float mask_add[BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    int qb, h, b;
    xcd_work_item(qb, h, b);
    const int q0 = qb * QB;
    const bf16* Q = a.q + (size_t)b * a.sq_b * a.ldq + h * D;
    const bf16* K = a.k + (size_t)b * a.skv_b * a.ldk + h * D;
    const bf16* V = a.v + (size_t)b * a.skv_b * a.ldv + h * D;
    const int kend = CAUSAL ? min(a.Skv, q0 + QB) : a.Skv;
    const FdDrop drop = DROP ? fd_drop_make(a.drop_p, a.dkey0, a.dkey1, a.dstep) : FdDrop{};
    bf16x8 kpre[2], vpre[2];
    Stream2 kv;
    kv.init(K, a.ldk, V, a.ldv, 0, tid);
    kv.fetch(a.Skv, tid, kpre, vpre);
    load_rows<QB>(Q, a.ldq, q0, a.Sq, Qs, tid);
    __syncthreads();
    bf16x8 qf0[QT], qf1[QT];
    int qi[QT];                                    // this lane's queries
    float m[QT], l[QT];
    f32x4 o[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qrow = (wave * QT + t) * 16 + i16;
        qf0[t] = row_frag(Qs, qrow, g);
        qf1[t] = row_frag(Qs, qrow, 4 + g);
        qi[t] = q0 + qrow;
        m[t] = -INFINITY;
        l[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // One streamed chunk.  NKT = 16-key tiles of the chunk that hold keys at all (4; fewer on the LAST chunk: 577 image tokens
    // = 9 chunks + 1 key, whose chunk costs a quarter), MASK = some key of the chunk may be hidden (key-padding mask, causal,
    // or beyond S_kv).  Scores stay raw until the exponent: p = exp2(x * SC - m) is one FMA per element (packed pairs), the
    // running maximum is taken on the raw scores (SC > 0) and scaled once per row.
    auto do_chunk = [&](auto nkt_tag, auto mask_tag, const int k0) {
        constexpr int NKT = decltype(nkt_tag)::value;
        constexpr bool CMASK = decltype(mask_tag)::value;        // this chunk applies mask_add
        constexpr int NST = (NKT + 1) / 2;
        f32x4 s[QT][NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const bf16x8 kf0 = row_frag(Ks, kt * 16 + i16, g), kf1 = row_frag(Ks, kt * 16 + i16, 4 + g);
            f32x4 ma = {0.f, 0.f, 0.f, 0.f};
            if (CMASK) ma = *reinterpret_cast<const f32x4*>(mask_add + kt * 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x4 x = {0.f, 0.f, 0.f, 0.f};
                x = mfma16x32(kf0, qf0[t], x);            // S^T: rows = keys kt*16 + 4g + e, col = query i16
                x = mfma16x32(kf1, qf1[t], x);
                if (CMASK) {
                    x = x + ma;                            // 0 or -inf
                    if (CAUSAL) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (k0 + kt * 16 + 4 * g + e > qi[t]) x[e] = -INFINITY;
                    }
                }
                s[t][kt] = x;
            }
        }
        bf16x8 pb[QT][NST];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float cmx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) cmx = fmaxf(cmx, s[t][kt][e]);
            cmx = fmaxf(cmx, __shfl_xor(cmx, 16, 64));
            cmx = fmaxf(cmx, __shfl_xor(cmx, 32, 64));
            cmx *= SC;                                                      // log2 domain, like m
            // lazy running maximum: the reference point m only moves when the chunk's maximum exceeds it by more than 2^8
            // (probabilities stay <= 256, exact in fp32 / same relative rounding in bf16), so the rescale of l and of the 16
            // accumulator registers is skipped (wave-uniformly) in most chunks; o / l and the LSE do not depend on m
            const bool move = cmx > m[t] + 8.0f;                            // m = -inf: any finite score moves it
            if (__builtin_amdgcn_ballot_w64(move) != 0) {
                const float m_upd = move ? cmx : m[t];
                const float alpha = m_upd == m[t] ? 1.0f : __builtin_amdgcn_exp2f(m[t] - m_upd);   // m = -inf -> 0
                l[t] *= alpha;
                m[t] = m_upd;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[t][dt] = o[t][dt] * f32x4{alpha, alpha, alpha, alpha};
            }
            const float nm = m[t] == -INFINITY ? 0.f : -m[t];             // a fully masked prefix contributes nothing
            const f32x4 nm4 = {nm, nm, nm, nm}, sc4 = {SC, SC, SC, SC};
            f32x4 cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const f32x4 y = s[t][kt] * sc4 + nm4;
#pragma unroll
                for (int e = 0; e < 4; ++e) s[t][kt][e] = __builtin_amdgcn_exp2f(y[e]);
                cs = cs + s[t][kt];
            }
            float csum = (cs[0] + cs[1]) + (cs[2] + cs[3]);
            csum += __shfl_xor(csum, 16, 64);
            csum += __shfl_xor(csum, 32, 64);
            l[t] += csum;
            if (DROP) {      // softmax normalises over ALL keys (l above); the dropped, rescaled probabilities meet V
                const uint32_t base = (uint32_t)(((size_t)(b * a.heads + h) * a.Sq + qi[t]) * a.Skv + k0 + 4 * g);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[t][kt][e] = fd_drop_keep(drop, base + kt * 16 + e) ? s[t][kt][e] * drop.scale : 0.f;
            }
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            pb[t][0] = cvt8(s[t][0], NKT > 1 ? s[t][NKT > 1 ? 1 : 0] : z4);
            if (NST == 2) pb[t][NST - 1] = cvt8(s[t][NKT > 2 ? 2 : 0], NKT > 3 ? s[t][NKT > 3 ? 3 : 0] : z4);
        }
#pragma unroll
        for (int st = 0; st < NST; ++st)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {                             // O^T[d][q] += V^T[d][key] P^T[key][q]
                const bf16x8 vt = tr_frag8(Vs, st * 32 + 4 * g, st * 32 + 16 + 4 * g, dt * 16, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][dt] = mfma16x32(vt, pb[t][st], o[t][dt]);
            }
    };
    // MASK = false (launcher: no key-padding mask, not causal, S_kv % 64 in [0, 16]): full chunks run without any masking and
    // only the short last chunk looks at mask_add.  Two chunk bodies per kernel (more instantiations cost registers).
    for (int k0 = 0; k0 < kend; k0 += BLK) {
        __syncthreads();
        chunk_commit2(Ks, Vs, tid, kpre, vpre);
        const int rem = kend - k0;                 // keys of this chunk (block-uniform)
        if ((MASK || rem < BLK) && tid < BLK) {
            const int kk = k0 + tid;
            mask_add[tid] = (kk < a.Skv && (!a.kmask || a.kmask[(size_t)b * a.Skv + kk])) ? 0.f : -INFINITY;
        }
        if (k0 + BLK < kend) kv.fetch(a.Skv, tid, kpre, vpre);
        __syncthreads();
        using std::integral_constant;
        if (rem <= 16) do_chunk(integral_constant<int, 1>{}, std::true_type{}, k0);
        else do_chunk(integral_constant<int, 4>{}, integral_constant<bool, MASK>{}, k0);
    }
    __syncthreads();                                                     // Qs is reused as the output staging tile
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        put_acc(Qs, (wave * QT + t) * 16, lane, o[t], l[t] > 0.f ? 1.0f / l[t] : 0.f);
        if (g == 0 && qi[t] < a.Sq && a.lse)
            a.lse[((size_t)b * a.heads + h) * a.Sq + qi[t]] =
                l[t] > 0.f ? m[t] * (1.0f / LOG2E) + __logf(l[t]) : -INFINITY;
    }
    __syncthreads();
    store_rows<QB>(a.o + (size_t)b * a.sq_b * a.ldo + h * D, a.ldo, q0, a.Sq, Qs, tid);
}

// dQ (and D = rowsum(dO * O)) of one 64 QT-query block: K / V stream through LDS.
template <int QT, bool CAUSAL, bool DROP, bool MASK = true>
__global__ __launch_bounds__(256, 3) void attn2_bwd_dq_kernel(Attn2Args a) {
    static_assert(MASK || (!CAUSAL && !DROP), "the mask-free form is the plain one");
    constexpr int QB = BLK * QT;
    __shared__ __attribute__((aligned(16))) char Qs[QB * ROWB], Gs[QB * ROWB], Ks[BLK * ROWB], Vs[BLK * ROWB];
    __shared__ float Dv[QB], kvalid[BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    int qb, h, b;
    xcd_work_item(qb, h, b);
    const int q0 = qb * QB;
    const bf16* Q = a.q + (size_t)b * a.sq_b * a.ldq + h * D;
    const bf16* K = a.k + (size_t)b * a.skv_b * a.ldk + h * D;
    const bf16* V = a.v + (size_t)b * a.skv_b * a.ldv + h * D;
    const bf16* O = a.o + (size_t)b * a.sq_b * a.ldo + h * D;
    const bf16* G = a.dout + (size_t)b * a.sq_b * a.lddo + h * D;
    const int kend = CAUSAL ? min(a.Skv, q0 + QB) : a.Skv;
    const FdDrop drop = DROP ? fd_drop_make(a.drop_p, a.dkey0, a.dkey1, a.dstep) : FdDrop{};
    bf16x8 kpre[2], vpre[2];
    Stream2 kv;
    kv.init(K, a.ldk, V, a.ldv, 0, tid);
    kv.fetch(a.Skv, tid, kpre, vpre);
    load_rows<QB>(Q, a.ldq, q0, a.Sq, Qs, tid);
#pragma unroll
    for (int idx = tid; idx < QB * 8; idx += 256) {          // dO -> LDS, D[q] = sum_d dO[q][d] O[q][d]
        const int row = idx >> 3, chunk = idx & 7;
        bf16x8 gv = {0, 0, 0, 0, 0, 0, 0, 0};
        float part = 0.f;
        if (q0 + row < a.Sq) {
            gv = *reinterpret_cast<const bf16x8*>(G + (size_t)(q0 + row) * a.lddo + chunk * 8);
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(O + (size_t)(q0 + row) * a.ldo + chunk * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) part += (float)gv[e] * (float)ov[e];
        }
        *reinterpret_cast<bf16x8*>(Gs + sw_off(row, chunk)) = gv;
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (chunk == 0) {
            Dv[row] = part;
            if (q0 + row < a.Sq) a.dsum[((size_t)b * a.heads + h) * a.Sq + q0 + row] = part;
        }
    }
    __syncthreads();
    bf16x8 qf0[QT], qf1[QT], gf0[QT], gf1[QT];
    int qi[QT];
    float dq_[QT], lq[QT];
    f32x4 dq[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int qrow = (wave * QT + t) * 16 + i16;
        qi[t] = q0 + qrow;
        qf0[t] = row_frag(Qs, qrow, g);
        qf1[t] = row_frag(Qs, qrow, 4 + g);
        gf0[t] = row_frag(Gs, qrow, g);
        gf1[t] = row_frag(Gs, qrow, 4 + g);
        dq_[t] = Dv[qrow];
        lq[t] = qi[t] < a.Sq ? a.lse[((size_t)b * a.heads + h) * a.Sq + qi[t]] * LOG2E : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // MASK = false (no key-padding mask, not causal): no validity factor at all -- keys beyond S_kv are zero rows of the K
    // tile, so whatever dS they get meets a zero row in dQ += dS K.  A last chunk of <= 32 keys (577 = 9 x 64 + 1) runs one
    // of its two 32-key halves.
    for (int k0 = 0; k0 < kend; k0 += BLK) {
        __syncthreads();
        chunk_commit2(Ks, Vs, tid, kpre, vpre);
        if (MASK && tid < BLK) {
            const int kk = k0 + tid;
            kvalid[tid] = (kk < a.Skv && (!a.kmask || a.kmask[(size_t)b * a.Skv + kk])) ? 1.f : 0.f;
        }
        if (k0 + BLK < kend) kv.fetch(a.Skv, tid, kpre, vpre);
        __syncthreads();
        const int nks = kend - k0 <= 32 ? 1 : 2;
        for (int ks = 0; ks < nks; ++ks) {
            f32x4 ds[QT][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int krow = (2 * ks + tt) * 16;
                const bf16x8 kf0 = row_frag(Ks, krow + i16, g), kf1 = row_frag(Ks, krow + i16, 4 + g);
                const bf16x8 vf0 = row_frag(Vs, krow + i16, g), vf1 = row_frag(Vs, krow + i16, 4 + g);
                f32x4 kv4 = {1.f, 1.f, 1.f, 1.f};
                if (MASK) kv4 = *reinterpret_cast<const f32x4*>(kvalid + krow + 4 * g);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sc = mfma16x32(kf0, qf0[t], sc);
                    sc = mfma16x32(kf1, qf1[t], sc);
                    dp = mfma16x32(vf0, gf0[t], dp);
                    dp = mfma16x32(vf1, gf1[t], dp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pe = __builtin_amdgcn_exp2f(sc[e] * SC - lq[t]);
                        if (MASK) pe *= kv4[e];
                        if (CAUSAL && k0 + krow + 4 * g + e > qi[t]) pe = 0.f;
                        float dpe = dp[e];
                        if (DROP)      // dP = mask / (1 - p) . (dO V^T); D = rowsum(dO . O) already holds the dropped O
                            dpe = fd_drop_keep(drop, (uint32_t)(((size_t)(b * a.heads + h) * a.Sq + qi[t]) * a.Skv + k0 + krow +
                                                                4 * g + e)) ? dpe * drop.scale : 0.f;
                        ds[t][tt][e] = pe * (dpe - dq_[t]);                // the 1/8 of dS is applied at the end
                    }
                }
            }
            bf16x8 dsb[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) dsb[t] = cvt8(ds[t][0], ds[t][1]);
            const int r0a = (2 * ks) * 16 + 4 * g, r0b = (2 * ks + 1) * 16 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kt = tr_frag8(Ks, r0a, r0b, dt * 16, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) dq[t][dt] = mfma16x32(kt, dsb[t], dq[t][dt]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < QT; ++t) put_acc(Qs, (wave * QT + t) * 16, lane, dq[t], 0.125f);
    __syncthreads();
    store_rows<QB>(a.dq + (size_t)b * a.sq_b * a.lddq + h * D, a.lddq, q0, a.Sq, Qs, tid);
}

// dK, dV of one 64 QT-key block: Q / dO (and their LSE / D) stream through LDS.
template <int QT, bool CAUSAL, bool DROP, bool MASK = true>
__global__ __launch_bounds__(256, 2) void attn2_bwd_dkv_kernel(Attn2Args a) {
    static_assert(MASK || (!CAUSAL && !DROP), "the mask-free form is the plain one");
    constexpr int KB = BLK * QT;
    __shared__ __attribute__((aligned(16))) char Ks[KB * ROWB], Vs[KB * ROWB], Qs[BLK * ROWB], Gs[BLK * ROWB];
    __shared__ float Ls[BLK], Dv[BLK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    int kb, h, b;
    xcd_work_item(kb, h, b);
    const int k0 = kb * KB;
    const bf16* Q = a.q + (size_t)b * a.sq_b * a.ldq + h * D;
    const bf16* K = a.k + (size_t)b * a.skv_b * a.ldk + h * D;
    const bf16* V = a.v + (size_t)b * a.skv_b * a.ldv + h * D;
    const bf16* G = a.dout + (size_t)b * a.sq_b * a.lddo + h * D;
    const int qstart = CAUSAL ? k0 : 0;                    // queries before the block's first key see none of it
    const FdDrop drop = DROP ? fd_drop_make(a.drop_p, a.dkey0, a.dkey1, a.dstep) : FdDrop{};
    bf16x8 qpre[2], gpre[2];
    Stream2 qg;
    qg.init(Q, a.ldq, G, a.lddo, qstart, tid);
    qg.fetch(a.Sq, tid, qpre, gpre);
    load_rows<KB>(K, a.ldk, k0, a.Skv, Ks, tid);
    load_rows<KB>(V, a.ldv, k0, a.Skv, Vs, tid);
    __syncthreads();
    bf16x8 kf0[QT], kf1[QT], vf0[QT], vf1[QT];
    int key[QT];
    float kv[QT];
    f32x4 dv[QT][4], dk[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int krow = (wave * QT + t) * 16 + i16;
        key[t] = k0 + krow;
        kf0[t] = row_frag(Ks, krow, g);
        kf1[t] = row_frag(Ks, krow, 4 + g);
        vf0[t] = row_frag(Vs, krow, g);
        vf1[t] = row_frag(Vs, krow, 4 + g);
        kv[t] = (key[t] < a.Skv && (!a.kmask || a.kmask[(size_t)b * a.Skv + key[t]])) ? 1.f : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dk[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int q0 = qstart; q0 < a.Sq; q0 += BLK) {
        __syncthreads();
        chunk_commit2(Qs, Gs, tid, qpre, gpre);
        if (tid < BLK) {
            const int qq = q0 + tid;
            const size_t si = ((size_t)b * a.heads + h) * a.Sq + qq;
            Ls[tid] = qq < a.Sq ? a.lse[si] * LOG2E : INFINITY;          // rows past Sq: p = exp2(-inf) = 0
            Dv[tid] = qq < a.Sq ? a.dsum[si] : 0.f;
        }
        if (q0 + BLK < a.Sq) qg.fetch(a.Sq, tid, qpre, gpre);
        __syncthreads();
        const int nqs = a.Sq - q0 <= 32 ? 1 : 2;                          // a last chunk of <= 32 queries: its second half is empty
        for (int qs = 0; qs < nqs; ++qs) {
            f32x4 p[QT][2], ds[QT][2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int qrow = (2 * qs + tt) * 16;
                const bf16x8 qa = row_frag(Qs, qrow + i16, g), qb2 = row_frag(Qs, qrow + i16, 4 + g);
                const bf16x8 ga = row_frag(Gs, qrow + i16, g), gb2 = row_frag(Gs, qrow + i16, 4 + g);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qrow + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(Dv + qrow + 4 * g);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sc = mfma16x32(qa, kf0[t], sc);                     // S: rows = queries qrow + 4g + e, col = key i16
                    sc = mfma16x32(qb2, kf1[t], sc);
                    dp = mfma16x32(ga, vf0[t], dp);
                    dp = mfma16x32(gb2, vf1[t], dp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pe = __builtin_amdgcn_exp2f(sc[e] * SC - l4[e]);
                        if (MASK) pe *= kv[t];                           // (without a mask: rows beyond S_kv are never stored)
                        if (CAUSAL && key[t] > q0 + qrow + 4 * g + e) pe = 0.f;
                        float mk = 1.0f;
                        if (DROP)
                            mk = fd_drop_keep(drop, (uint32_t)(((size_t)(b * a.heads + h) * a.Sq + q0 + qrow + 4 * g + e) * a.Skv +
                                                               key[t])) ? drop.scale : 0.f;
                        p[t][tt][e] = pe * mk;                           // dV = (P . mask / (1 - p))^T dO
                        ds[t][tt][e] = pe * (dp[e] * mk - d4[e]);
                    }
                }
            }
            bf16x8 pb[QT], dsb[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                pb[t] = cvt8(p[t][0], p[t][1]);
                dsb[t] = cvt8(ds[t][0], ds[t][1]);
            }
            const int r0a = (2 * qs) * 16 + 4 * g, r0b = (2 * qs + 1) * 16 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 gt = tr_frag8(Gs, r0a, r0b, dt * 16, lane);
                const bf16x8 qt = tr_frag8(Qs, r0a, r0b, dt * 16, lane);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    dv[t][dt] = mfma16x32(gt, pb[t], dv[t][dt]);
                    dk[t][dt] = mfma16x32(qt, dsb[t], dk[t][dt]);
                }
            }
        }
    }
    __syncthreads();                                         // K / V fragments live in registers: their tiles are the staging
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        put_acc(Ks, (wave * QT + t) * 16, lane, dk[t], 0.125f);
        put_acc(Vs, (wave * QT + t) * 16, lane, dv[t], 1.0f);
    }
    __syncthreads();
    store_rows<KB>(a.dk + (size_t)b * a.skv_b * a.lddk + h * D, a.lddk, k0, a.Skv, Ks, tid);
    store_rows<KB>(a.dv + (size_t)b * a.skv_b * a.lddv + h * D, a.lddv, k0, a.Skv, Vs, tid);
}

int check(const Attn2Args& a, int B) {
    if (!a.q || !a.k || !a.v || !a.o || B <= 0 || a.Sq <= 0 || a.Skv <= 0 || a.heads <= 0) return FEDDAT_EINVAL;
    if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 8) return FEDDAT_EINVAL;
    if (a.sq_b < a.Sq || a.skv_b < a.Skv) return FEDDAT_EINVAL;
    return FEDDAT_OK;
}

}  // namespace

#define FD_ATTN2_LAUNCH(KERN, QT_, GRID)                                                                              \
    do {                                                                                                           \
        if (a.drop_p > 0.f) {                                                                                      \
            if (a.causal) hipLaunchKernelGGL((KERN<QT_, true, true>), GRID, dim3(256), 0, stream, a);              \
            else hipLaunchKernelGGL((KERN<QT_, false, true>), GRID, dim3(256), 0, stream, a);                      \
        } else {                                                                                                   \
            if (a.causal) hipLaunchKernelGGL((KERN<QT_, true, false>), GRID, dim3(256), 0, stream, a);             \
            else hipLaunchKernelGGL((KERN<QT_, false, false>), GRID, dim3(256), 0, stream, a);                     \
        }                                                                                                          \
    } while (0)

static int attn2_fwd_launch(Attn2Args& a, int B, hipStream_t stream) {
    const int rc = check(a, B);
    if (rc) return rc;
    FD_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f && (size_t)B * a.heads * a.Sq * a.Skv < (1ull << 32));
    const dim3 grid2((a.Sq + 2 * BLK - 1) / (2 * BLK), a.heads, B), grid1((a.Sq + BLK - 1) / BLK, a.heads, B);
    if (!a.kmask && !a.causal && !(a.drop_p > 0.f) && a.Skv % BLK <= 16) {      // the ViT's case: the mask-free instantiations
        if (a.Sq > BLK) hipLaunchKernelGGL((attn2_fwd_kernel<2, false, false, false>), grid2, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((attn2_fwd_kernel<1, false, false, false>), grid1, dim3(256), 0, stream, a);
        FD_LAUNCH_RET();
    }
    if (a.Sq > BLK) FD_ATTN2_LAUNCH(attn2_fwd_kernel, 2, grid2);
    else FD_ATTN2_LAUNCH(attn2_fwd_kernel, 1, grid1);
    FD_LAUNCH_RET();
}

static int attn2_bwd_launch(Attn2Args& a, int B, hipStream_t stream) {
    const int rc = check(a, B);
    if (rc) return rc;
    FD_CHECK_ARG(a.lse && a.dout && a.dsum && a.dq && a.dk && a.dv && a.lddo % 8 == 0 && a.lddq % 8 == 0 && a.lddk % 8 == 0 &&
                 a.lddv % 8 == 0);
    FD_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f && (size_t)B * a.heads * a.Sq * a.Skv < (1ull << 32));
    const int Sq = a.Sq, Skv = a.Skv, heads = a.heads;
    const dim3 gq2((Sq + 2 * BLK - 1) / (2 * BLK), heads, B), gk2((Skv + 2 * BLK - 1) / (2 * BLK), heads, B),
        g1((Sq + BLK - 1) / BLK, heads, B), g1k((Skv + BLK - 1) / BLK, heads, B);
    const bool plain = !a.kmask && !a.causal && !(a.drop_p > 0.f);      // the ViT's case: the mask-free instantiations
    if (plain) {
        if (Sq > BLK) hipLaunchKernelGGL((attn2_bwd_dq_kernel<2, false, false, false>), gq2, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((attn2_bwd_dq_kernel<1, false, false, false>), g1, dim3(256), 0, stream, a);
        if (Skv > BLK) hipLaunchKernelGGL((attn2_bwd_dkv_kernel<2, false, false, false>), gk2, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((attn2_bwd_dkv_kernel<1, false, false, false>), g1k, dim3(256), 0, stream, a);
        FD_LAUNCH_RET();
    }
    if (Sq > BLK) FD_ATTN2_LAUNCH(attn2_bwd_dq_kernel, 2, gq2);
    else FD_ATTN2_LAUNCH(attn2_bwd_dq_kernel, 1, g1);
    if (Skv > BLK) FD_ATTN2_LAUNCH(attn2_bwd_dkv_kernel, 2, gk2);
    else FD_ATTN2_LAUNCH(attn2_bwd_dkv_kernel, 1, g1k);
    FD_LAUNCH_RET();
}

static Attn2Args attn2_args(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const uint8_t* key_mask,
                            int causal, const void* ctx, long ldo, const float* lse, int Sq, int Skv, long q_rows_per_sample,
                            long kv_rows_per_sample, int heads) {
    Attn2Args a{};
    a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.sq_b = q_rows_per_sample; a.skv_b = kv_rows_per_sample;
    a.kmask = key_mask; a.Sq = Sq; a.Skv = Skv; a.heads = heads; a.causal = causal;
    a.o = (bf16*)const_cast<void*>(ctx); a.ldo = ldo; a.lse = const_cast<float*>(lse);
    return a;
}

extern "C" int feddat_attn2_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                const uint8_t* key_mask, int causal, void* ctx, long ldo, float* lse, int B, int Sq,
                                int Skv, long q_rows_per_sample, long kv_rows_per_sample, int heads, hipStream_t stream) {
    Attn2Args a = attn2_args(q, ldq, k, ldk, v, ldv, key_mask, causal, ctx, ldo, lse, Sq, Skv, q_rows_per_sample,
                             kv_rows_per_sample, heads);
    return attn2_fwd_launch(a, B, stream);
}

extern "C" int feddat_attn2_fwd_dropout(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                        const uint8_t* key_mask, int causal, void* ctx, long ldo, float* lse, int B, int Sq,
                                        int Skv, long q_rows_per_sample, long kv_rows_per_sample, int heads, float p,
                                        unsigned key0, unsigned key1, const int* step_ctr, hipStream_t stream) {
    Attn2Args a = attn2_args(q, ldq, k, ldk, v, ldv, key_mask, causal, ctx, ldo, lse, Sq, Skv, q_rows_per_sample,
                             kv_rows_per_sample, heads);
    a.drop_p = p; a.dkey0 = key0; a.dkey1 = key1; a.dstep = step_ctr;
    return attn2_fwd_launch(a, B, stream);
}

static void attn2_bwd_args(Attn2Args& a, const void* dctx, long lddo, float* dsum_ws, void* dq, long lddq, void* dk, long lddk,
                           void* dv, long lddv) {
    a.dout = (const bf16*)dctx; a.lddo = lddo; a.dsum = dsum_ws;
    a.dq = (bf16*)dq; a.dk = (bf16*)dk; a.dv = (bf16*)dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
}

extern "C" int feddat_attn2_bwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                const uint8_t* key_mask, int causal, const void* ctx, long ldo, const float* lse,
                                const void* dctx, long lddo, float* dsum_ws, void* dq, long lddq, void* dk, long lddk,
                                void* dv, long lddv, int B, int Sq, int Skv, long q_rows_per_sample,
                                long kv_rows_per_sample, int heads, hipStream_t stream) {
    Attn2Args a = attn2_args(q, ldq, k, ldk, v, ldv, key_mask, causal, ctx, ldo, lse, Sq, Skv, q_rows_per_sample,
                             kv_rows_per_sample, heads);
    attn2_bwd_args(a, dctx, lddo, dsum_ws, dq, lddq, dk, lddk, dv, lddv);
    return attn2_bwd_launch(a, B, stream);
}

extern "C" int feddat_attn2_bwd_dropout(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv,
                                        const uint8_t* key_mask, int causal, const void* ctx, long ldo, const float* lse,
                                        const void* dctx, long lddo, float* dsum_ws, void* dq, long lddq, void* dk, long lddk,
                                        void* dv, long lddv, int B, int Sq, int Skv, long q_rows_per_sample,
                                        long kv_rows_per_sample, int heads, float p, unsigned key0, unsigned key1,
                                        const int* step_ctr, hipStream_t stream) {
    Attn2Args a = attn2_args(q, ldq, k, ldk, v, ldv, key_mask, causal, ctx, ldo, lse, Sq, Skv, q_rows_per_sample,
                             kv_rows_per_sample, heads);
    attn2_bwd_args(a, dctx, lddo, dsum_ws, dq, lddq, dk, lddk, dv, lddv);
    a.drop_p = p; a.dkey0 = key0; a.dkey1 = key1; a.dstep = step_ctr;
    return attn2_bwd_launch(a, B, stream);
}
