"""The fp16-operand build of the same kernels (libfeddat_hip_f16.so; include/feddat_hip.h "Conventions"): every 16-bit MFMA
operand is IEEE half instead of bf16 -- the reference's own GPU arithmetic (fp16 autocast, src/accelerate_config.yaml:8) --
and the backward carries a power-of-two loss scale (what its GradScaler does, task_trainer.py:302,323).

  * ops through the C ABI inside lib.operands("f16"), against fp32 restatements, at tolerances 4-8x tighter than the bf16
    build's (10 instead of 7 mantissa bits);
  * the ViLT engine with operands="f16" against the reference's fixtures (G3) and the oracle;
  * the loss scale is removed exactly: bit-identical training for any power-of-two scale in the bf16 build (whose operand
    range cannot under- or overflow), and equal-to-rounding in the fp16 build.
The round-length test at the metric's own configuration is tests/test_round_b32_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

from oracle import feddat_oracle as O
from tests.golden_util import assert_update_parity, load
from tests.test_engine_gpu import _golden_update_parity
from tests.test_ops_gpu import _attn_ref, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _needs_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from feddat_amd import lib
    with lib.operands("f16"):
        lib.load()
    return lib


def h(x):
    return x.to(torch.float16)


def _dev(b):
    return {k: v.to(DEV) for k, v in b.items()}


def test_the_two_libraries_are_different_builds_of_one_abi(L):
    a = L.load()
    with L.operands("f16"):
        b = L.load()
        assert b.feddat_operand_format() == L.OPERANDS_FP16 and b.feddat_abi_version() == L.ABI_VERSION
    assert a is not b and a.feddat_operand_format() == L.OPERANDS_BF16


@pytest.mark.parametrize("M,N,K", [(300, 256, 192), (800, 3072, 768), (1200, 256, 192), (5920, 768, 3072),
                                   (11840, 2304, 768), (11849, 3072, 768), (64, 768, 3072)])
def test_gemm_epilogues_fp16_operands(L, M, N, K):
    """Same products and epilogues as test_ops_gpu.test_gemm_epilogues on fp16 operands.  fp16 INPUTS are exact in fp32, so the
    fp32-output epilogues are exact up to the accumulation order; 16-bit outputs carry one fp16 rounding (2^-11 relative)."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = h(torch.randn(M, K, generator=g)).to(DEV)
    B = h(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    aux = h(torch.randn(M, N, generator=g)).to(DEV)
    ref = A.float() @ B.float().t()
    tol = 2e-3
    skinny = torch.empty(max(1, L.gemm_skinny_workspace_elems(M, N, K)), device=DEV) if M <= 64 else None
    with L.operands("f16"):
        o16 = torch.empty(M, N, dtype=torch.float16, device=DEV)
        L.gemm_bf16_nt(A, B, L.EPI_BF16, bias=bias, out_bf16=o16, skinny_workspace=skinny)
        assert rel_err(o16, ref + bias) < tol
        o32 = torch.empty(M, N, device=DEV)
        L.gemm_bf16_nt(A, B, L.EPI_RESID_F32, bias=bias, resid=resid, out_f32=o32, skinny_workspace=skinny)
        assert rel_err(o32, ref + bias + resid) < 2e-4
        u16 = torch.empty(M, N, dtype=torch.float16, device=DEV)
        L.gemm_bf16_nt(A, B, L.EPI_GELU, bias=bias, out_bf16=o16, out2_bf16=u16, skinny_workspace=skinny)
        assert rel_err(u16, ref + bias) < tol
        assert rel_err(o16, F.gelu(ref + bias)) < tol
        L.gemm_bf16_nt(A, B, L.EPI_MUL_DGELU, aux=aux, out_bf16=o16, skinny_workspace=skinny)
        a32 = aux.float().requires_grad_(True)
        F.gelu(a32).sum().backward()
        assert rel_err(o16, ref * a32.grad) < tol
        if M >= 1024 and N % 192 == 0:      # the 8-bit gelu' code epilogues of the persistent kernels
            codes = torch.empty(M, N, dtype=torch.uint8, device=DEV)
            L.gemm_bf16_nt(A, B, L.EPI_GELU_G8, bias=bias, out_bf16=o16, out2_bf16=codes)
            assert rel_err(o16, F.gelu(ref + bias)) < tol
            u32 = (ref + bias).requires_grad_(True)
            F.gelu(u32).sum().backward()
            assert (codes.float() * L.G8_STEP + L.G8_LO - u32.grad).abs().max() < 0.5 * L.G8_STEP + 1e-3
            L.gemm_bf16_nt(A, B, L.EPI_MUL_G8, aux=codes, out_bf16=o16)
            assert rel_err(o16, ref * (codes.float() * L.G8_STEP + L.G8_LO)) < tol
        torch.cuda.synchronize()


def test_fp16_products_are_finer_than_bf16_on_the_same_fp32_data(L):
    """What the format buys: the same fp32 A, W through each library's own conversion + product: error vs fp64 ~8x smaller."""
    g = torch.Generator().manual_seed(5)
    A32 = torch.randn(2048, 768, generator=g).to(DEV)
    W32 = (torch.randn(768, 768, generator=g) * 0.02).to(DEV)
    ref = (A32.double() @ W32.double().t()).float()
    errs = {}
    for fmt in ("bf16", "f16"):
        with L.operands(fmt):
            dt = L.OPERAND_DTYPE[fmt]
            A, W = torch.empty(2048, 768, dtype=dt, device=DEV), torch.empty(768, 768, dtype=dt, device=DEV)
            L.cvt_f32_bf16(A32, A)
            L.cvt_f32_bf16(W32, W)
            assert torch.equal(A, A32.to(dt)) and torch.equal(W, W32.to(dt))      # round-to-nearest-even of the build's format
            out = torch.empty(2048, 768, device=DEV)
            L.gemm_bf16_nt(A, W, L.EPI_F32, out_f32=out)
            errs[fmt] = float((out - ref).abs().mean())
    print("mean |error| of a K = 768 product:", errs)
    assert errs["f16"] < errs["bf16"] / 5


@pytest.mark.parametrize("B,S,heads,masked", [(2, 185, 12, False), (3, 90, 12, True), (64, 185, 12, False), (2, 281, 12, True)])
def test_attention_fwd_bwd_fp16_operands(L, B, S, heads, masked):
    g = torch.Generator().manual_seed(S)
    H = heads * 64
    qkv = h(torch.randn(B * S, 3 * H, generator=g)).to(DEV)
    mask = None
    if masked:
        mask = torch.ones(B, S, dtype=torch.uint8)
        for b in range(B):
            mask[b, 20 + 3 * b: 20 + 3 * b + 7] = 0
        mask = mask.to(DEV)
    ctx = torch.empty(B * S, H, dtype=torch.float16, device=DEV)
    lse = torch.empty(B, heads, S, device=DEV)
    qr = qkv.float().requires_grad_(True)
    cref, lref = _attn_ref(qr, B, S, heads, mask)
    dctx = h(torch.randn(B * S, H, generator=g)).to(DEV)
    cref.backward(dctx.float())
    dqkv = torch.full((B * S, 3 * H), float("nan"), dtype=torch.float16, device=DEV)
    with L.operands("f16"):
        L.attn_fwd(qkv, ctx, lse, B, S, heads, key_mask=mask)
        L.attn_bwd(qkv, ctx, lse, dctx, dqkv, B, S, heads, key_mask=mask)
        if S <= 192:      # the last layer's token-0-only form
            ctx0 = torch.zeros_like(ctx)
            lse0 = torch.zeros_like(lse)
            L.attn_cls_fwd(qkv, ctx0, lse0, B, S, heads, key_mask=mask)
            assert rel_err(ctx0.view(B, S, H)[:, 0], cref.view(B, S, H)[:, 0]) < 3e-3
    assert rel_err(ctx, cref) < 3e-3
    assert (lse - lref).abs().max() < 1e-3
    for part, name in enumerate("QKV"):
        a, r = dqkv[:, part * H:(part + 1) * H], qr.grad[:, part * H:(part + 1) * H]
        assert torch.isfinite(a.float()).all(), name
        assert rel_err(a, r) < 6e-3, (name, rel_err(a, r))


def test_layernorm_and_adapter_fp16_operands(L, golden_dir):
    """LayerNorm's 16-bit output and the fused adapter (G1: the reference's own Adapter module, adapter.py:124-163) in the fp16
    build; and feddat_wgrad_seg.grad_unscale: gradients scaled by 2^10 going in leave unscaled, bit-identically."""
    x = torch.randn(5920, 768, device=DEV) * 2 + 0.3
    gma, bta = torch.rand(768, device=DEV) + 0.5, torch.randn(768, device=DEV) * 0.1
    ref = F.layer_norm(x, (768,), gma, bta, 1e-12)
    g = load(golden_dir, "g1_adapter.npz")
    xg = torch.from_numpy(g["x"]).reshape(-1, 768).to(DEV)
    dy = torch.from_numpy(g["dy"]).reshape(-1, 768).to(DEV)
    T = xg.shape[0]
    hT = T // 2
    with L.operands("f16"):
        y = torch.empty(5920, 768, dtype=torch.float16, device=DEV)
        st = torch.empty(5920, 2, device=DEV)
        L.layernorm_fwd(x, gma, bta, 1e-12, 5920, 768, y_bf16=y, stats=st)
        assert rel_err(y, ref) < 1e-3
        par = {}
        for a in range(3):
            wd = torch.from_numpy(g[f"p.adapter_{a}_down.weight"]).to(DEV)
            wu = torch.from_numpy(g[f"p.adapter_{a}_up.weight"]).to(DEV)
            w = [torch.empty(48, 768, dtype=torch.float16, device=DEV), torch.empty(768, 48, dtype=torch.float16, device=DEV),
                 torch.empty(768, 48, dtype=torch.float16, device=DEV), torch.empty(48, 768, dtype=torch.float16, device=DEV)]
            L.adapter_pack(wd, wu, *w)
            par[a] = dict(wd=w[0], wdT=w[1], wu=w[2], wuT=w[3], bd=torch.from_numpy(g[f"p.adapter_{a}_down.bias"]).to(DEV),
                          bu=torch.from_numpy(g[f"p.adapter_{a}_up.bias"]).to(DEV))
        segs = L.make_segs([dict(row_begin=0, row_end=hT, train_slot=0, adapters=[dict(par[0], scale=0.5), dict(par[2], scale=0.5)]),
                            dict(row_begin=hT, row_end=T, train_slot=0, adapters=[dict(par[1], scale=1.0)])])
        out = torch.zeros_like(xg)
        L.adapter_fwd(xg, out, segs, T)
        yg = torch.from_numpy(g["gating.y"]).reshape(-1, 768).to(DEV)
        ys = torch.from_numpy(g["adapter_1.y"]).reshape(-1, 768).to(DEV)
        assert (out[:hT] - yg[:hT]).abs().max() < 2.5e-3       # bf16 build: 1.2e-2 on the same data
        assert (out[hT:] - ys[hT:]).abs().max() < 2.5e-3
        dx = torch.zeros_like(xg)
        dx16 = torch.zeros(T, 768, dtype=torch.float16, device=DEV)
        z, dz = torch.zeros(T, 48, device=DEV), torch.zeros(T, 48, device=DEV)
        L.adapter_bwd(xg, dy, dx, segs, T, dx_bf16=dx16, z_out=z, dz_out=dz)
        assert rel_err(dx16, dx) < 1.5e-3
        n = 48 * 768 + 48 + 768 * 48 + 768
        part = torch.empty(L.adapter_wgrad_workspace_elems(1), device=DEV)
        g1, g2 = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
        L.adapter_wgrad(L.make_wgrad_segs([dict(x=xg, dy=dy, z=z, dz=dz, grad=g1, rows=hT, scale=0.5)]), part)
        L.adapter_wgrad(L.make_wgrad_segs([dict(x=xg, dy=dy * 1024, z=z, dz=dz * 1024, grad=g2, rows=hT, scale=0.5,
                                                grad_unscale=1.0 / 1024)]), part)
        assert torch.equal(g1, g2) and float(g1.abs().max()) > 0


@pytest.mark.parametrize("use_graph", [False, True])
def test_train_steps_two_layers_fp16_vs_reference_golden(golden_dir, use_graph):
    """G3 (B = 4, 224 x 224): losses and every trainable tensor after 1, 2, 5 train_steps of ViltDatEngine(operands='f16')
    against the reference's own run and the oracle -- the bounds of the bf16 engine's test, which the fp16 engine meets with
    a several times smaller mean ratio (printed)."""
    from feddat_amd import engine
    g = load(golden_dir, "g3_vilt2_224.npz")
    d = O.ViltDims(layers=2)
    P = O.make_params(d, ["art", "gqa"], bias_std=0.02)
    P0 = {k: v.clone() for k, v in P.items()}
    eng = engine.ViltDatEngine(P, ["art", "gqa"], DEV, batch=4, res=224, layers=2, operands="f16")
    assert eng.x16.dtype == torch.float16 and eng.layers[0]["wqkv"].dtype == torch.float16 and eng.loss_scale == 16384.0
    batches = [O.synthetic_batch(4, 224, 1234 + s) for s in range(5)]
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=5)
    eng.begin_local_update("art", steps_per_epoch=5)
    names = O.trainable_names(P, "art", 0) + [n for n in O.trainable_names(P, "art", 1) if "adapter_1" in n]
    for s, b in enumerate(batches):
        ref_loss = float(client.train_step(b)[0])
        out = eng.train_step(_dev(b), use_graph=use_graph)
        loss = float(out[0])
        assert abs(loss - ref_loss) < 5e-4 * abs(ref_loss) + 5e-4, (s, loss, ref_loss)
        assert abs(loss - float(g["losses"][s])) < 5e-4 * abs(ref_loss) + 5e-4
        worst = assert_update_parity(names, eng.state_dict(), P, P0, 1e-3, 0.05, f"oracle step {s + 1}")
        if s + 1 in (1, 2, 5):
            worst_g = _golden_update_parity(g, f"after{s+1}.", eng.state_dict(), P0)
    print("fp16 operands, after 5 steps: worst (max |ddW|, mean ratio) vs oracle", worst, "vs reference golden", worst_g)
    assert worst[1] < 0.03
    for mode in ("gating", "adapter_1"):       # forward logits in the fp16 build: 4x inside the bf16 engine's 3e-2
        Pf = O.make_params(d, ["art", "gqa"], bias_std=0.02)
        e2 = engine.ViltDatEngine(Pf, ["art", "gqa"], DEV, batch=4, res=224, layers=2, operands="f16")
        pooled, logits = e2.forward(_dev(batches[0]), mode, "art")
        assert (logits.cpu() - torch.from_numpy(g[f"fwd.{mode}.logits"])).abs().max() < 8e-3
        assert (pooled.cpu() - torch.from_numpy(g[f"fwd.{mode}.pooled"])).abs().max() < 8e-3


def _run(engine, operands, loss_scale, steps=3, layers=3, **kw):
    d = O.ViltDims(layers=layers)
    P = O.make_params(d, ["art"], bias_std=0.02)
    eng = engine.ViltDatEngine(P, ["art"], DEV, batch=3, res=224, layers=layers, operands=operands, loss_scale=loss_scale, **kw)
    eng.begin_local_update("art", steps_per_epoch=steps)
    for s in range(steps):
        eng.train_step(_dev(O.synthetic_batch(3, 224, 300 + s)))
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in eng.state_dict().items()}


def test_the_loss_scale_leaves_exactly():
    """Every kernel of the backbone's backward is linear in the gradient and a power of two commutes with every rounding, so
    in the bf16 build (fp32's exponent range on every operand) training with loss scales 1, 2^8 and 2^14 is BIT-IDENTICAL;
    in the fp16 build scales 2^10 and 2^14 agree to the rounding of the few gradient operands that sit in fp16's subnormal
    range at the smaller scale.  The engine rejects scales that are not powers of two."""
    from feddat_amd import engine, lib
    base = _run(engine, "bf16", 1.0)
    for sc in (256.0, 16384.0):
        got = _run(engine, "bf16", sc)
        for k in base:
            assert torch.equal(base[k], got[k]), (sc, k)
    # (AdamW turns an element whose gradient is ~0 into +-lr steps, so a last-bit difference of such a gradient is worth up to
    #  2 x sum(lr) = 1.3e-4 over these three warm-up steps on that element; the bulk agrees to 1e-7)
    a, b = _run(engine, "f16", 1024.0), _run(engine, "f16", 16384.0)
    for k in a:
        d = (a[k] - b[k]).abs()
        assert float(d.max()) < 1.5e-4 and float(d.mean()) < 2e-7, (k, float(d.max()), float(d.mean()))
    with pytest.raises(lib.FeddatHipError):
        engine.ViltDatEngine(O.make_params(O.ViltDims(layers=1), ["art"]), ["art"], DEV, batch=1, res=224, layers=1,
                             loss_scale=1000.0)


def test_both_operand_formats_in_one_process_interleaved():
    """A bf16 and an fp16 engine stepping alternately (each binds its own library per call): same results as alone."""
    from feddat_amd import engine
    alone = {f: _run(engine, f, None, steps=2, layers=2) for f in ("bf16", "f16")}
    d = O.ViltDims(layers=2)
    engs = {}
    for f in ("bf16", "f16"):
        P = O.make_params(d, ["art"], bias_std=0.02)
        engs[f] = engine.ViltDatEngine(P, ["art"], DEV, batch=3, res=224, layers=2, operands=f)
        engs[f].begin_local_update("art", steps_per_epoch=2)
    for s in range(2):
        for f in ("bf16", "f16"):
            engs[f].train_step(_dev(O.synthetic_batch(3, 224, 300 + s)), use_graph=True)
    for f in ("bf16", "f16"):
        sd = engs[f].state_dict()
        for k in sd:
            assert torch.equal(sd[k], alone[f][k]), (f, k)


def test_an_overflowing_loss_scale_is_reported_not_trained_through():
    """With a STATIC scale (dynamic_loss_scale=False; the default is the device-side GradScaler of tests/test_dynscale_gpu.py) no
    overflowed step is skipped: a scale that pushes the gradient operands out of fp16's range must surface as an error at the
    end of the local update, not as silently corrupted adapters.  2^36 x the ~1e-3 gradients of this model overflows 65 504;
    the default 2^14 does not."""
    from feddat_amd import engine, lib
    d = O.ViltDims(layers=2)
    for scale, ok in ((16384.0, True), (2.0 ** 36, False)):
        P = O.make_params(d, ["art"], bias_std=0.02)
        eng = engine.ViltDatEngine(P, ["art"], DEV, batch=3, res=224, layers=2, operands="f16", loss_scale=scale,
                                   dynamic_loss_scale=False)
        eng.begin_local_update("art", steps_per_epoch=2)
        for s in range(2):
            eng.train_step(_dev(O.synthetic_batch(3, 224, 300 + s)))
        if ok:
            eng.assert_finite()
        else:
            with pytest.raises(lib.FeddatHipError, match="loss scale"):
                eng.assert_finite()
