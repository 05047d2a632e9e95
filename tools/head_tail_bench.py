"""Isolated timing of the fused-tail launches (csrc/head_tail.hip) at configs[1]'s sizes (B = 32): each launch repeated
back to back, HIP events around the batch.  python tools/head_tail_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L

dev = "cuda"
H, B, C, S = 768, 32, 100, 185
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
hl, gam, bet = r(2 * B, S * H), r(H), r(H)
Wp, bp = r(H, H) * 0.05, r(H)
W0, b0 = r(2 * H, H) * 0.05, r(2 * H)
W1, b1 = r(C, 2 * H) * 0.05, r(C)
pooled, st = torch.empty(2 * B, H, device=dev), torch.empty(2 * B, 2, device=dev)
a0, n0, g0 = (torch.empty(2 * B, 2 * H, device=dev) for _ in range(3))
st0 = torch.empty(2 * B, 2, device=dev)
logits = torch.empty(2 * B, C, device=dev)
dl, dn0, da0 = r(B, C), torch.empty(B, 2 * H, device=dev), r(B, 2 * H)
dW1, db1, dW0, db0 = torch.empty(C, 2 * H, device=dev), torch.empty(C, device=dev), torch.empty(2 * H, H, device=dev), torch.empty(2 * H, device=dev)
dpooled, dcls = r(2 * B, H), torch.empty(2 * B, H, device=dev)
tgt = torch.rand(B, C, generator=g).to(dev)
sc = torch.empty(4 + 2 * B, device=dev)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)       # flush: the step's activations evict the head's weights


def bench(name, fn, reps=20, cold=True):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if cold:
            big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{name:44s} {'cold' if cold else 'warm'} median {ts[len(ts) // 2]:7.1f} us   min {ts[0]:7.1f}")


J = L.ht_job
jobs = {
    "pool: LN + dense + tanh (64x768x768)": lambda: L.head_gemm(J(hl, S * H, 1, Wp, 1, H, 2 * B, H, H, pooled, bias_j=bp, pro=L.HT_PRO_LN, pro_a=gam, pro_b=bet, pro_eps=1e-12, stats_out=st, epi=L.HT_EPI_TANH)),
    "fc0 both (64x1536x768)": lambda: L.head_gemm(J(pooled, H, 1, W0, 1, H, 2 * B, 2 * H, H, a0, bias_j=b0)),
    "ln_gelu (64x1536)": lambda: L.head_ln_gelu(a0, b0, b0, 1e-5, n0, st0, g0),
    "fc1 both (64x100x1536)": lambda: L.head_gemm(J(g0, 2 * H, 1, W1, 1, 2 * H, 2 * B, C, 2 * H, logits, bias_j=b1)),
    "loss single": lambda: L.dat_loss_fwd_bwd_single(logits[:B], logits[B:], tgt, dl, sc),
    "loss two launches": lambda: L.dat_loss_fwd_bwd(logits[:B], logits[B:], tgt, dl, sc),
    "bwd A: dW_fc1 | dn0": lambda: L.head_gemm(J(dl, 1, C, g0, 2 * H, 1, C, 2 * H, B, dW1, mode=1, colsum=db1), J(dl, C, 1, W1, 2 * H, 1, B, 2 * H, C, dn0, epi=L.HT_EPI_MUL_DGELU, aux=n0, ld_aux=2 * H)),
    "  dW_fc1 alone": lambda: L.head_gemm(J(dl, 1, C, g0, 2 * H, 1, C, 2 * H, B, dW1, mode=1, colsum=db1)),
    "  dn0 alone": lambda: L.head_gemm(J(dl, C, 1, W1, 2 * H, 1, B, 2 * H, C, dn0, epi=L.HT_EPI_MUL_DGELU, aux=n0, ld_aux=2 * H)),
    "ln_bwd_full (32x1536)": lambda: L.head_ln_bwd_full(dn0, a0, st0, b0, da0, db0, db0),
    "bwd C: dW_fc0 | dpooled": lambda: L.head_gemm(J(da0, 1, 2 * H, pooled, H, 1, 2 * H, H, B, dW0, mode=1, colsum=db0), J(da0, 2 * H, 1, W0, H, 1, B, H, 2 * H, dpooled)),
    "  dW_fc0 alone": lambda: L.head_gemm(J(da0, 1, 2 * H, pooled, H, 1, 2 * H, H, B, dW0, mode=1, colsum=db0)),
    "  dpooled alone": lambda: L.head_gemm(J(da0, 2 * H, 1, W0, H, 1, B, H, 2 * H, dpooled)),
    "dcls: tanh' prologue (64x768x768)": lambda: L.head_gemm(J(dpooled, H, 1, Wp, H, 1, 2 * B, H, H, dcls, pro=L.HT_PRO_TANH_BWD, pro_a=pooled)),
    "old: sgemm fc0 ksplit 4 + reduce": None,
}
part = torch.empty(16 * 2 * B * 2 * H, device=dev)


def old_fc0():
    L.sgemm_f32(pooled, H, 1, W0, 1, H, 2 * B, 2 * H, H, part, ksplit=4, bias_j=b0, out_split_stride=2 * B * 2 * H)
    L.reduce_partials(part, 2 * B * 2 * H, 4, 2 * B * 2 * H, a0)


jobs["old: sgemm fc0 ksplit 4 + reduce"] = old_fc0
for cold in (True, False):
    for n, f in jobs.items():
        bench(n, f, cold=cold)
