"""Token-0-only attention of the last layer against the dense kernels, isolated (configs[1]: 2B = 64 samples, S = 185)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L
B, S, heads, H = 64, 185, 12, 768
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, 3 * H, generator=g) * 0.7).to(torch.bfloat16).cuda()
ctx = torch.zeros(B * S, H, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, heads, S, device="cuda")
d0 = torch.randn(B, H, generator=g).cuda(); dctx = torch.zeros(B * S, H, dtype=torch.bfloat16, device="cuda")
dqkv = torch.empty_like(qkv)
big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
def bench(name, fn, reps=20):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); print(f"{name:28s} median {ts[len(ts)//2]:7.1f} us  min {ts[0]:7.1f}")
bench("attn_cls_fwd", lambda: L.attn_cls_fwd(qkv, ctx, lse, B, S, heads))
bench("attn_fwd (dense)", lambda: L.attn_fwd(qkv, ctx, lse, B, S, heads))
bench("attn_cls_bwd", lambda: L.attn_cls_bwd(qkv, ctx, lse, d0, dqkv, B, S, heads))
bench("attn_bwd (dense)", lambda: L.attn_bwd(qkv, ctx, lse, dctx, dqkv, B, S, heads))
