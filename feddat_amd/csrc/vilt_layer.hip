// Composite entry points: one HF ViltLayer with the reference's Adaptered_ViltOutput, forward and dX / adapter-gradient
// backward, as ONE C-ABI call each -- the sequencing feddat_amd/engine.py does per layer, available to callers that have
// no Python (feddat_vilt_layer_fwd / feddat_vilt_layer_bwd).
//   forward  (transformers ViltLayer.forward as called from src/modeling/vilt.py:127, with
//             src/modeling/adaptered_output.py:73-78 in place of ViltOutput):
//     x = LN1(h_in); qkv = x Wqkv^T + b; ctx = softmax(q k^T / 8 + mask) v; h2 = ctx Wo^T + bo + h_in;
//     y = LN2(h2); u = y W1^T + b1; h3 = gelu(u) W2^T + b2 + h2; h_out = adapter(h3, h3)  [+ next layer's LN1]
//   backward (autograd through the frozen weights, src/train/visionlanguage_tasks/task_trainer.py:302,323): d h_out -> d h_in,
//     plus the trainable adapter's weight gradients.
#include "common.hip.h"

struct feddat_ctx;

// The composites launch on the calling thread's CURRENT device with buffers the caller allocated there: a context that was
// created for another device is a caller bug (its per-device kernel attributes were prepared elsewhere), reported, not
// silently launched.  Shapes are the ViLT-B ones the kernels underneath are built for (H = 768, r = 48, I = 4 H).
static int check_ctx_and_shape(const feddat_ctx* ctx, int heads) {
    int cdev = -1, cur = -2;
    if (feddat_ctx_device(ctx, &cdev, nullptr) != FEDDAT_OK) return FEDDAT_EINVAL;
    if (hipGetDevice(&cur) != hipSuccess) return FEDDAT_ELAUNCH;
    if (cdev != cur) return FEDDAT_EINVAL;
    return heads * 64 == 768 ? FEDDAT_OK : FEDDAT_EINVAL;
}

extern "C" int feddat_vilt_layer_fwd(feddat_ctx* ctx, const feddat_vilt_layer_weights* W, const feddat_vilt_layer_acts* A,
                                     int nb, int S, int heads, const uint8_t* key_mask, int ln1_done,
                                     const feddat_adapter_seg* segs, int nseg, const float* next_ln_g,
                                     const float* next_ln_b, hipStream_t stream) {
    FD_CHECK_ARG(ctx && W && A && nb > 0 && S > 0 && heads > 0 && segs);
    int rc = check_ctx_and_shape(ctx, heads);
    if (rc != FEDDAT_OK) return rc;
    const int rows = nb * S, H = heads * 64, I = 4 * H;
    FD_CHECK_ARG(A->h_in && A->qkv && A->ctx && A->lse && A->h2 && A->st2 && A->u && A->h3 && A->h_out && A->x16 && A->f16);
    FD_CHECK_ARG(W->wqkv && W->wo && W->w1 && W->w2 && W->bqkv && W->bo && W->b1 && W->b2 && W->ln2_g && W->ln2_b);
    FD_CHECK_ARG(ln1_done || (A->st1 && W->ln1_g && W->ln1_b));       // LN1 runs here: its weights and statistics buffer
    FD_CHECK_ARG(!next_ln_g || (next_ln_b && A->st1_next));           // fused next-layer LN: beta and its statistics buffer
#define FD_TRY(call) do { rc = (call); if (rc != FEDDAT_OK) return rc; } while (0)
    if (!ln1_done)
        FD_TRY(feddat_layernorm_fwd(A->h_in, H, W->ln1_g, W->ln1_b, W->ln_eps, rows, H, A->x16, nullptr, A->st1, stream));
    FD_TRY(feddat_gemm_bf16_nt(A->x16, H, W->wqkv, H, rows, 3 * H, H, FEDDAT_EPI_BF16, W->bqkv, nullptr, 0, nullptr, 0,
                               nullptr, 0, A->qkv, 3 * H, nullptr, 0, stream));
    FD_TRY(feddat_attn_fwd(A->qkv, key_mask, A->ctx, A->lse, nb, S, heads, stream));
    FD_TRY(feddat_gemm_bf16_nt(A->ctx, H, W->wo, H, rows, H, H, FEDDAT_EPI_RESID_F32, W->bo, A->h_in, H, nullptr, 0, A->h2, H,
                               nullptr, 0, nullptr, 0, stream));
    FD_TRY(feddat_layernorm_fwd(A->h2, H, W->ln2_g, W->ln2_b, W->ln_eps, rows, H, A->x16, nullptr, A->st2, stream));
    // A->u: what FFN2^T will need of the pre-GELU u -- 8-bit gelu'(u) codes where the persistent GEMM runs, else u in bf16
    FD_TRY(feddat_gemm_bf16_nt(A->x16, H, W->w1, H, rows, I, H, rows >= 1024 ? FEDDAT_EPI_GELU_G8 : FEDDAT_EPI_GELU, W->b1,
                               nullptr, 0, nullptr, 0, nullptr, 0, A->f16, I, A->u, I, stream));
    FD_TRY(feddat_gemm_bf16_nt(A->f16, I, W->w2, I, rows, H, I, FEDDAT_EPI_RESID_F32, W->b2, A->h2, H, nullptr, 0, A->h3, H,
                               nullptr, 0, nullptr, 0, stream));
    if (next_ln_g)
        FD_TRY(feddat_adapter_fwd_ln(A->h3, A->h_out, rows, H, 48, segs, nseg, next_ln_g, next_ln_b, W->ln_eps, A->x16,
                                     A->st1_next, A->z_save, stream));
    else
        FD_TRY(feddat_adapter_fwd(A->h3, A->h_out, rows, H, 48, segs, nseg, A->z_save, stream));
    return FEDDAT_OK;
}

extern "C" int feddat_vilt_layer_bwd(feddat_ctx* ctx, const feddat_vilt_layer_weights* W, const feddat_vilt_layer_acts* A,
                                     const feddat_vilt_layer_grads* G, int nb, int S, int heads, const uint8_t* key_mask,
                                     const feddat_adapter_seg* segs, int nseg, const feddat_wgrad_seg* wsegs, int nwseg,
                                     float* wgrad_partials, long wgrad_partials_elems, int wgrad_reduce_now,
                                     hipStream_t stream) {
    FD_CHECK_ARG(ctx && W && A && G && nb > 0 && S > 0 && heads > 0 && segs);
    FD_CHECK_ARG(G->dh_out && G->dh_in && G->dh3 && G->dh16 && G->dU && G->dx16 && G->dctx && G->dqkv && G->z && G->dz);
    int rc = check_ctx_and_shape(ctx, heads);
    if (rc != FEDDAT_OK) return rc;
    FD_CHECK_ARG(A->h_in && A->st1 && A->qkv && A->ctx && A->lse && A->h2 && A->st2 && A->u && (A->z_save || A->h3));
    FD_CHECK_ARG(W->w2T && W->w1T && W->woT && W->wqkvT && W->ln1_g && W->ln2_g);
    const int rows = nb * S, H = heads * 64, I = 4 * H;
    // adapter: d h3 (fp32 + bf16 copy), z / dz for the weight gradients of the trainable adapter(s)
    FD_TRY(feddat_adapter_bwd(A->z_save ? nullptr : A->h3, A->z_save, G->dh_out, G->dh3, G->dh16, G->z, G->dz, rows, H, 48,
                              segs, nseg, stream));
    if (wsegs && nwseg > 0) {      // wgrad_reduce_now = 0: the caller folds all layers' partials with one feddat_adapter_wgrad_reduce
        if (wgrad_reduce_now) FD_TRY(feddat_adapter_wgrad(wsegs, nwseg, wgrad_partials, wgrad_partials_elems, H, 48, stream));
        else FD_TRY(feddat_adapter_wgrad_partial(wsegs, nwseg, wgrad_partials, wgrad_partials_elems, H, 48, stream));
    }
    // FFN2^T (. gelu'), FFN1^T, LN2 backward (+ residual)
    FD_TRY(feddat_gemm_bf16_nt(G->dh16, H, W->w2T, H, rows, I, H, rows >= 1024 ? FEDDAT_EPI_MUL_G8 : FEDDAT_EPI_MUL_DGELU, nullptr,
                               nullptr, 0, A->u, I, nullptr, 0, G->dU, I, nullptr, 0, stream));
    FD_TRY(feddat_gemm_bf16_nt(G->dU, I, W->w1T, I, rows, H, I, FEDDAT_EPI_BF16, nullptr, nullptr, 0, nullptr, 0, nullptr, 0,
                               G->dx16, H, nullptr, 0, stream));
    FD_TRY(feddat_layernorm_bwd_dx(G->dx16, nullptr, H, A->h2, H, A->st2, W->ln2_g, G->dh3, H, rows, H, G->dh_out, H, G->dh16,
                                   stream));          // d h2 -> dh_out's buffer (its content is dead), bf16 copy -> dh16
    // attention-out^T, attention backward, QKV^T, LN1 backward (+ residual)
    FD_TRY(feddat_gemm_bf16_nt(G->dh16, H, W->woT, H, rows, H, H, FEDDAT_EPI_BF16, nullptr, nullptr, 0, nullptr, 0, nullptr, 0,
                               G->dctx, H, nullptr, 0, stream));
    FD_TRY(feddat_attn_bwd(A->qkv, key_mask, A->ctx, A->lse, G->dctx, G->dqkv, nb, S, heads, stream));
    FD_TRY(feddat_gemm_bf16_nt(G->dqkv, 3 * H, W->wqkvT, 3 * H, rows, H, 3 * H, FEDDAT_EPI_BF16, nullptr, nullptr, 0, nullptr,
                               0, nullptr, 0, G->dx16, H, nullptr, 0, stream));
    FD_TRY(feddat_layernorm_bwd_dx(G->dx16, nullptr, H, A->h_in, H, A->st1, W->ln1_g, G->dh_out, H, rows, H, G->dh_in, H,
                                   nullptr, stream));
#undef FD_TRY
    return FEDDAT_OK;
}
