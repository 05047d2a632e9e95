import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L
dev = "cuda"
def t(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K, epi) in [(11840, 2304, 768, 0), (11840, 3072, 768, 2), (23680, 2304, 768, 0), (23680, 3072, 768, 2)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; bias = torch.randn(N, device=dev)
    A16, W16 = A.to(torch.bfloat16), W.to(torch.bfloat16)
    A8, sa = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
    W8, sw = torch.empty(N, K, dtype=torch.uint8, device=dev), torch.empty(N, device=dev)
    L.quant_rows_fp8(A, A8, sa); L.quant_rows_fp8(W, W8, sw)
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev); u = torch.empty_like(o)
    kw = dict(out_bf16=o, out2_bf16=u) if epi == 2 else dict(out_bf16=o)
    tb = t(lambda: L.gemm_bf16_nt(A16, W16, epi, bias=bias, **kw))
    t8 = t(lambda: L.gemm_fp8_nt(A8, sa, W8, sw, epi, bias=bias, **kw))
    f = 2.0 * M * N * K
    print(f"({M},{N},{K}) epi {epi}: bf16 {tb:.1f} us ({f / tb / 1e6:.0f} TF/s)   fp8 {t8:.1f} us ({f / t8 / 1e6:.0f} TF/s)", flush=True)
