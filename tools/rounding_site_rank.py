#!/usr/bin/env python
"""Which bf16 rounding site of the HIP engine carries the max |ddW| at round length?  

CPU-only.  The fp32 oracle (oracle/feddat_oracle.py, pinned to the reference) runs an N-step round next to EMULATIONS of
the engine's arithmetic: the same oracle code with bf16 round-to-nearest-even applied at the places where the engine
stores or feeds a bf16 value (forward: frozen weights, LN outputs, qkv, softmax probabilities, ctx, gelu(u), what is saved
of the pre-GELU u -- 8-bit gelu' codes since round 3 --, the adapter's operand copies; backward: every dY / dX that is a bf16 GEMM operand or a bf16 store).  Sites are
switched individually:

    all          every site on  (should land where the real engine lands: tools/round_length_probe.py)
    only:<s>     only site s on
    all-<s>      every site but s

For each configuration: worst max |ddW| and worst mean ratio over all trainable tensors against the fp32 run after
N steps (B = 4, 384 x 384, 12 layers, the schedule of an N-step round).

    python tools/rounding_site_rank.py [steps=80] [configs...]     # configs default: a standard sweep
"""
import os
import sys
import math
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import feddat_oracle as O  # noqa: E402

FWD_SITES = ["W", "x1", "qkv", "p", "ctx", "x2", "f", "u", "ad_x", "ad_w", "ad_z", "patch"]
BWD_SITES = ["WT", "dh3", "dU", "dx2", "dh2", "dctx", "dqkv", "dx1", "ad_dy", "ad_dz"]
ALL_SITES = FWD_SITES + BWD_SITES
ON = set()


MANT = int(os.environ.get("ROUND_MANT_BITS", "7"))   # explicit mantissa bits kept: 7 = bf16, 10 = fp16 / tf32, 15 = bf16 hi+lo
MANT_B = int(os.environ.get("ROUND_MANT_BITS_BWD", str(MANT)))   # the same for the backward sites (WT, dY / dX operands)
BATCH = int(os.environ.get("ROUND_BATCH", "4"))      # 32 = configs[1]'s batch: the non-chaotic case


def bf(t, mant=None):
    """Round to nearest even at MANT explicit mantissa bits (MANT = 7 is exactly the fp32 -> bf16 -> fp32 round trip)."""
    mant = MANT if mant is None else mant
    if mant == 7:
        return t.to(torch.bfloat16).to(torch.float32)
    drop = 23 - mant
    xi = t.contiguous().view(torch.int32)
    xi = xi + ((1 << (drop - 1)) - 1) + ((xi >> drop) & 1)
    xi = xi & ~((1 << drop) - 1)
    return xi.view(torch.float32)


class _RF(torch.autograd.Function):        # round the VALUE, pass the gradient through
    @staticmethod
    def forward(ctx, x):
        return bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RB(torch.autograd.Function):        # pass the value through, round the GRADIENT
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g, MANT_B)


def rf(site, x):
    return _RF.apply(x) if site in ON else x


def rb(site, x):
    return _RB.apply(x) if (site in ON and x.requires_grad) else x


class _LinW(torch.autograd.Function):
    """y = x W^T + b with a frozen W: forward uses W (maybe bf16-rounded: site W), backward dX = dY W uses the
    transposed copy (site WT) -- the engine holds both as separate bf16 tensors (same values)."""
    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(W)
        Wf = bf(W) if "W" in ON else W
        return F.linear(x, Wf, b)

    @staticmethod
    def backward(ctx, g):
        (W,) = ctx.saved_tensors
        Wb = bf(W, MANT_B) if "WT" in ON else W
        return g @ Wb, None, None


def lin(x, W, b):
    return _LinW.apply(x, W, b)


G8_LO, G8_STEP = -0.135, 0.005            # FEDDAT_G8_LO / FEDDAT_G8_STEP (include/feddat_hip.h)
U_AS_CODES = os.environ.get("ROUND_U_CODES", "1") != "0"


class _Gelu(torch.autograd.Function):
    """f = gelu(u) from the fp32 accumulator; site u = what the engine keeps of u for the backward: since round 3 the 8-bit
    code of gelu'(u) computed from the fp32 u (FEDDAT_EPI_GELU_G8; ROUND_U_CODES=0: the bf16 u of rounds 1-2)."""
    @staticmethod
    def forward(ctx, u):
        if "u" in ON and U_AS_CODES:
            cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2)))
            gp = cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
            ctx.codes = True
            ctx.save_for_backward(G8_LO + G8_STEP * torch.round((gp - G8_LO) / G8_STEP).clamp(0, 255))
        else:
            ctx.codes = False
            ctx.save_for_backward(bf(u) if "u" in ON else u)
        return F.gelu(u)

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        if ctx.codes:
            return g * u                       # the saved tensor IS the decoded gelu'
        cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2)))
        pdf = torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
        return g * (cdf + u * pdf)


def layer_body(P, d, i, h, kmask=None):
    L = O.LAYER.format(i=i)
    B, S, H = h.shape
    x = F.layer_norm(h, (H,), P[L + "layernorm_before.weight"], P[L + "layernorm_before.bias"], d.ln_eps)
    x = rb("dx1", rf("x1", x))
    q = lin(x, P[L + "attention.attention.query.weight"], P[L + "attention.attention.query.bias"])
    k = lin(x, P[L + "attention.attention.key.weight"], P[L + "attention.attention.key.bias"])
    v = lin(x, P[L + "attention.attention.value.weight"], P[L + "attention.attention.value.bias"])
    sh = (B, S, d.heads, d.head_dim)
    q, k, v = (rb("dqkv", rf("qkv", t)).view(sh).transpose(1, 2) for t in (q, k, v))
    sc = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d.head_dim)
    if kmask is not None:
        sc = sc.masked_fill(~kmask[:, None, None, :], torch.finfo(sc.dtype).min)
    p = torch.softmax(sc, dim=-1)
    p = rf("p", p)
    ctx = torch.matmul(p, v).permute(0, 2, 1, 3).reshape(B, S, H)
    ctx = rb("dctx", rf("ctx", ctx))
    o = rb("dh2", lin(ctx, P[L + "attention.output.dense.weight"], P[L + "attention.output.dense.bias"]))
    h2 = o + h
    x2 = F.layer_norm(h2, (H,), P[L + "layernorm_after.weight"], P[L + "layernorm_after.bias"], d.ln_eps)
    x2 = rb("dx2", rf("x2", x2))
    u = rb("dU", lin(x2, P[L + "intermediate.dense.weight"], P[L + "intermediate.dense.bias"]))
    f = rf("f", _Gelu.apply(u) if u.requires_grad else F.gelu(u))
    y = rb("dh3", lin(f, P[L + "output.layer.dense.weight"], P[L + "output.layer.dense.bias"]))
    return y + h2


class _Adapter(torch.autograd.Function):
    """up(relu(down(h))) with the engine's operand roundings: x and W_down as bf16 operands (fp32 accumulate), z kept
    fp32 but fed to the up-projection as bf16, W_up bf16.  Backward: dy as a bf16 operand for dz = dy W_up, dz as a bf16
    operand for dx = dz W_down; the WEIGHT gradients use split operands (fp32-exact) on the unrounded x, z, dy, dz."""
    @staticmethod
    def forward(ctx, h, Wd, bd, Wu, bu):
        xo = bf(h) if "ad_x" in ON else h
        wd = bf(Wd) if "ad_w" in ON else Wd
        wu = bf(Wu) if "ad_w" in ON else Wu
        z = F.relu(F.linear(xo, wd, bd))
        ctx.save_for_backward(h, z, wd, wu)
        return F.linear(bf(z) if "ad_z" in ON else z, wu, bu)

    @staticmethod
    def backward(ctx, dy):
        h, z, wd, wu = ctx.saved_tensors
        dyo = bf(dy, MANT_B) if "ad_dy" in ON else dy
        dz = (dyo @ wu) * (z > 0)
        dzo = bf(dz, MANT_B) if "ad_dz" in ON else dz
        dx = dzo @ wd
        dy2, z2, dz2, h2 = (t.reshape(-1, t.shape[-1]) for t in (dy, z, dz, h))
        return dx, dz2.t() @ h2, dz2.sum(0), dy2.t() @ z2, dy2.sum(0)


def _ad(h, Wd, bd, Wu, bu):
    return _Adapter.apply(h, Wd, bd, Wu, bu)


def adapter_single(h, inp, Wd, bd, Wu, bu):
    return inp + _ad(h, Wd, bd, Wu, bu)


def adapter_gated(h, inp, A, B):
    return inp + (0.5 * _ad(h, *A) + 0.5 * _ad(h, *B))


def vilt_embed(P, d, batch):
    e = O.ENC + "embeddings."
    ids, tt = batch["input_ids"], batch["token_type_ids"]
    B, Lt = ids.shape
    te = P[e + "text_embeddings.word_embeddings.weight"][ids]
    te = te + P[e + "text_embeddings.token_type_embeddings.weight"][tt]
    te = te + P[e + "text_embeddings.position_embeddings.weight"][:Lt][None]
    te = F.layer_norm(te, (d.hidden,), P[e + "text_embeddings.LayerNorm.weight"],
                      P[e + "text_embeddings.LayerNorm.bias"], d.ln_eps)
    px = batch["pixel_values"]
    w = P[e + "patch_embeddings.projection.weight"]
    if "patch" in ON:
        px, w = bf(px), bf(w)
    x = F.conv2d(px, w, P[e + "patch_embeddings.projection.bias"], stride=d.patch)
    gh, gw = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2) + O.interp_pos_embed(P, d, gh, gw)
    cls = P[e + "cls_token"].expand(B, -1, -1) + P[e + "position_embeddings"][:, :1, :]
    ie = torch.cat([cls, x], dim=1)
    tok = P[e + "token_type_embeddings.weight"]
    return torch.cat([te + tok[0], ie + tok[1]], dim=1)


ORIG = dict(vilt_layer_body=O.vilt_layer_body, adapter_single=O.adapter_single, adapter_gated=O.adapter_gated,
            vilt_embed=O.vilt_embed)
EMU = dict(vilt_layer_body=layer_body, adapter_single=adapter_single, adapter_gated=adapter_gated,
           vilt_embed=vilt_embed)


def patch(on: bool):
    for k, v in (EMU if on else ORIG).items():
        setattr(O, k, v)


def parse(cfg):
    """cfg[@F[/B]]: site set, optionally with the explicit mantissa bits of the forward / backward sites (all@10/7 = an
    fp16 forward next to a bf16 backward)."""
    global MANT, MANT_B
    if "@" in cfg:
        cfg, m = cfg.split("@")
        f, _, b = m.partition("/")
        MANT, MANT_B = int(f), int(b or f)
    if cfg == "all":
        return set(ALL_SITES)
    if cfg == "none":
        return set()
    if cfg.startswith("only:"):
        return set(cfg[5:].split(","))
    if cfg.startswith("all-"):
        return set(ALL_SITES) - set(cfg[4:].split(","))
    raise SystemExit("bad config " + cfg)


def run(steps, cfg, B=None, ref=None, P0=None, every=20):
    B = BATCH if B is None else B
    global ON
    d = O.ViltDims(layers=12)
    P = O.make_params(d, ["art"], bias_std=0.02)
    ON = parse(cfg)
    patch(cfg.split("@")[0] != "none")
    client = O.DatClient(P, d, "art", lr=1e-4, steps_per_epoch=steps)
    names = [k for k in P if ("adapter_0" in k or "adapter_1" in k or "task_layer" in k)]
    out = {}
    for s in range(steps):
        client.train_step(O.synthetic_batch(B, 384, 8000 + s))
        if (s + 1) % every == 0 or s + 1 == steps:
            out[s + 1] = {k: P[k].detach().clone() for k in names}
    patch(False)
    return out


def compare(ref, got, P0):
    wmax, wratio, wk = 0.0, 0.0, ""
    for k in ref:
        dref, dgot = ref[k] - P0[k], got[k] - P0[k]
        if float(dref.abs().max()) == 0:
            continue
        e = (dgot - dref).abs()
        if float(e.max()) > wmax:
            wmax, wk = float(e.max()), k
        wratio = max(wratio, float(e.mean()) / float(dref.abs().mean()))
    return wmax, wratio, wk


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    cfgs = sys.argv[2:] or (["all"] + ["all-" + s for s in ("W,WT", "x1,x2", "qkv", "p", "ctx", "f", "u", "ad_x,ad_w,ad_z",
                                                              "dh3,dh2", "dU", "dx1,dx2", "dctx,dqkv", "ad_dy,ad_dz")])
    torch.set_num_threads(int(os.environ.get("ROUND_THREADS", os.cpu_count())))
    d = O.ViltDims(layers=12)
    P0 = O.make_params(d, ["art"], bias_std=0.02)
    t0 = time.time()
    cache = os.environ.get("ROUND_REF_CACHE")             # the fp32 run of a (steps, batch) pair, kept between invocations
    if cache and os.path.exists(cache):
        ref = torch.load(cache)
    else:
        ref = run(steps, "none")
        if cache:
            torch.save(ref, cache)
    print(f"fp32 reference run (B = {BATCH}): {time.time() - t0:.0f} s", flush=True)
    for cfg in cfgs:
        t0 = time.time()
        got = run(steps, cfg)
        line = f"{cfg:28s}"
        for n in sorted(ref):
            mx, ratio, k = compare(ref[n], got[n], P0)
            line += f" | {n:3d}: max {mx:.2e} ratio {ratio:.3f}"
        print(line + f" | worst {k.split('layer.')[-1]} ({time.time() - t0:.0f} s)", flush=True)
