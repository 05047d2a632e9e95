"""ALBEF train-mode dropout on the device (xbert.py:216,333,360,440 with p = 0.1,
src/configs/model_configs.py:44-46): the counter-based masks of libfeddat_hip.so against the oracle's restatement of the
same function, the fused attention with dropped probabilities against an fp32 restatement, and the engine against the
REFERENCE's own run under model.train() with these masks (tests/golden/g12_albef_dropout.npz)."""
import numpy as np
import pytest
import torch

from oracle import albef_oracle as A
from tests.golden_util import load

pytestmark = pytest.mark.gpu
DEV = "cuda"
SMALL = dict(vit_depth=2, enc_layers=3, fusion_layer=1, dec_layers=2, image=64, vocab=3072, max_pos=64)


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd import lib
    lib.load()
    return lib


def _dev(b):
    return {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}


@pytest.mark.parametrize("n,p,step", [(75 * 768, 0.1, 0), (800 * 768, 0.1, 7), (4 * 768, 0.5, 3)])
def test_dropout_kernel_is_the_oracle_mask_bit_for_bit(L, n, p, step):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    res = torch.randn(n, generator=g)
    k0, k1 = L.dropout_keys(77, 2, 41)
    assert (k0, k1) == A.dropout_keys(77, 2, 41)
    ctr = torch.tensor([step, 0], dtype=torch.int32, device=DEV)
    keep = A.dropout_keep(n, p, k0, k1, step)
    scale = torch.tensor(1.0) / (torch.tensor(1.0) - torch.tensor(p, dtype=torch.float32))
    want = x * (keep.float() * scale)
    out = torch.empty(n, device=DEV)
    o16 = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    L.dropout(x.to(DEV), (p, k0, k1, ctr), out_f32=out, out_bf16=o16)
    assert torch.equal(out.cpu(), want)
    assert torch.equal(o16.cpu(), want.to(torch.bfloat16))
    L.dropout(x.to(DEV), (p, k0, k1, ctr), resid=res.to(DEV), out_f32=out)
    assert torch.equal(out.cpu(), want + res)
    L.dropout(x.to(torch.bfloat16).to(DEV), (p, k0, k1, None), out_f32=out)           # bf16 input, no counter = step 0
    keep0 = A.dropout_keep(n, p, k0, k1, 0)
    assert torch.equal(out.cpu(), x.to(torch.bfloat16).float() * (keep0.float() * scale))
    assert abs(float(keep.float().mean()) - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5


@pytest.mark.parametrize("B,Sq,Skv,heads,causal,masked", [
    (3, 25, 25, 12, False, True),        # text self-attention, padded questions
    (5, 7, 7, 12, True, True),           # decoder: causal + padded answers
    (5, 7, 25, 12, False, True),         # decoder -> question cross-attention
    (2, 25, 577, 12, False, False),      # text encoder -> 577 image tokens (streamed K / V, 128-key dK/dV blocks)
    (2, 130, 70, 2, True, False),        # 128-query blocks (QT = 2 on the query side)
])
def test_attn2_dropout_fwd_bwd_vs_fp32_reference(L, B, Sq, Skv, heads, causal, masked):
    p, step = 0.1, 3
    k0, k1 = L.dropout_keys(5, 1, 9)
    ctr = torch.tensor([step, 0], dtype=torch.int32, device=DEV)
    drop = (p, k0, k1, ctr)
    g = torch.Generator().manual_seed(Sq * 1000 + Skv)
    H = heads * 64
    q = torch.randn(B * Sq, H, generator=g).to(torch.bfloat16).to(DEV)
    kv = torch.randn(B * Skv, 2 * H, generator=g).to(torch.bfloat16).to(DEV)
    k, v = kv[:, :H], kv[:, H:]
    do = torch.randn(B * Sq, H, generator=g).to(torch.bfloat16).to(DEV)
    km = None
    if masked:
        km = torch.ones(B, Skv, dtype=torch.uint8)
        for b in range(B):
            km[b, max(1, Skv - 1 - 2 * b):] = 0
        km = km.to(DEV)
    ctx = torch.zeros(B * Sq, H, dtype=torch.bfloat16, device=DEV)
    ctx0 = torch.zeros_like(ctx)
    lse = torch.zeros(B, heads, Sq, device=DEV)
    L.attn2_fwd(q, k, v, ctx, lse, B, Sq, Skv, heads, key_mask=km, causal=causal, drop=drop)
    L.attn2_fwd(q, k, v, ctx0, lse, B, Sq, Skv, heads, key_mask=km, causal=causal)
    assert (ctx.float() - ctx0.float()).abs().max() > 1e-2                  # the mask really is applied
    qr = q.float().view(B, Sq, heads, 64).transpose(1, 2).requires_grad_(True)
    kr = k.float().reshape(B, Skv, heads, 64).transpose(1, 2).requires_grad_(True)
    vr = v.float().reshape(B, Skv, heads, 64).transpose(1, 2).requires_grad_(True)
    sc = qr @ kr.transpose(-1, -2) / 8.0
    if km is not None:
        sc = sc + (1.0 - km.float())[:, None, None, :] * -10000.0
    if causal:
        sc = sc + torch.triu(torch.full((Sq, Skv), -10000.0, device=DEV), diagonal=1)
    keep = A.dropout_keep(B * heads * Sq * Skv, p, k0, k1, step).view(B, heads, Sq, Skv).to(DEV)
    pr = sc.softmax(-1) * (keep.float() / (1.0 - p))                         # nn.Dropout on attention_probs, xbert.py:333
    ref = (pr @ vr).transpose(1, 2).reshape(B * Sq, H)
    assert (ctx.float() - ref).abs().max() < 2e-2
    ref.backward(do.float())
    dq = torch.zeros_like(q)
    dkv = torch.zeros_like(kv)
    ws = torch.empty(B, heads, Sq, device=DEV)
    L.attn2_bwd(q, k, v, ctx, lse, do, ws, dq, dkv[:, :H], dkv[:, H:], B, Sq, Skv, heads, key_mask=km, causal=causal, drop=drop)
    for got, want in ((dq, qr.grad.transpose(1, 2).reshape(B * Sq, H)),
                      (dkv[:, :H], kr.grad.transpose(1, 2).reshape(B * Skv, H)),
                      (dkv[:, H:], vr.grad.transpose(1, 2).reshape(B * Skv, H))):
        err = float((got.float() - want).abs().max() / (want.abs().max() + 1e-12))
        assert err < 2e-2, err


def test_engine_with_dropout_vs_the_reference_in_train_mode(golden_dir):
    """G12 = the reference's own modules under model.train() with p = 0.1, their nn.Dropout masks replaced by the shared
    counter-based function: 3 train_steps of the small configuration (every code path: ragged, k = [2, 1, 3]).  The engine
    (eager for 2 steps, then a captured hipGraph whose masks come from the device-side step counter) must reproduce the
    losses and the update of every adapter tensor within the bf16 path's usual bounds."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd import albef_engine
    g = load(golden_dir, "g12_albef_dropout.npz")
    steps, p, seed = int(g["steps"]), float(g["p"]), int(g["seed"])
    d = A.AlbefDims(**SMALL)
    P = A.make_params(d)
    P0 = {k: v.clone() for k, v in P.items()}
    eng = albef_engine.AlbefDatEngine(P, DEV, batch=3, n_answers=6, q_len=12, a_len=5, vit_depth=SMALL["vit_depth"],
                                      enc_layers=SMALL["enc_layers"], fusion_layer=SMALL["fusion_layer"],
                                      dec_layers=SMALL["dec_layers"], image=SMALL["image"], vocab=SMALL["vocab"],
                                      dropout=p, seed=seed)
    eng.begin_local_update(steps_per_epoch=steps, num_epochs=1)
    for s in range(steps):
        b = A.synthetic_batch(3, d, 900 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)
        out = eng.train_step(_dev(b), use_graph=(s >= 2))
        torch.cuda.synchronize()
        ref = float(g["losses"][s])
        assert abs(float(out[0]) - ref) < 3e-3 * ref, (s, float(out[0]), ref)
    assert int(eng.drop_ctr[0]) == steps
    sd = eng.state_dict()
    worst_max, worst_ratio = 0.0, 0.0
    for k in [k.split("::", 1)[1] for k in g if k.startswith("dsamp::")]:
        dw = (sd[k].cpu() - P0[k]).flatten()
        idx = torch.linspace(0, dw.numel() - 1, min(512, dw.numel())).long()
        err = (dw[idx] - torch.from_numpy(g["dsamp::" + k])).abs()
        move = float(g["dmean::" + k])
        assert float(err.max()) < 1e-3 and float(err.mean()) <= 0.1 * move, (k, float(err.max()), float(err.mean()), move)
        worst_max, worst_ratio = max(worst_max, float(err.max())), max(worst_ratio, float(err.mean()) / max(move, 1e-12))
    print(f"ALBEF small with dropout 0.1, 3 steps vs the reference: worst max |ddW| {worst_max:.2e}, mean ratio {worst_ratio:.3f}")


def test_dropout_zero_is_the_deterministic_engine(golden_dir):
    """dropout = 0 takes the fused single-gated-pass schedule: bit-identical with an engine built without the argument."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd import albef_engine
    d = A.AlbefDims(**SMALL)
    outs = []
    for kw in ({}, {"dropout": 0.0, "seed": 5}):
        P = A.make_params(d)
        eng = albef_engine.AlbefDatEngine(P, DEV, batch=3, n_answers=6, q_len=12, a_len=5, vit_depth=SMALL["vit_depth"],
                                          enc_layers=SMALL["enc_layers"], fusion_layer=SMALL["fusion_layer"],
                                          dec_layers=SMALL["dec_layers"], image=SMALL["image"], vocab=SMALL["vocab"], **kw)
        eng.begin_local_update(steps_per_epoch=2, num_epochs=1)
        for s in range(2):
            eng.train_step(_dev(A.synthetic_batch(3, d, 900 + s, q_len=12, a_len=5, k=[2, 1, 3], ragged=True)))
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in eng.state_dict().items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
