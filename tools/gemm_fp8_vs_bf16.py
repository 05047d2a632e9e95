import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L
L.use_ablation_build()      # timing-only probes live in libfeddat_hip_ablate.so (python -m feddat_amd.build --ablate)
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
    o_mx = o.clone()
    L.set_debug_flags(256)            # the K = 32 fp8 instruction (bf16 issue rate) the MX-scaled K = 128 one replaced
    t8k32 = t(lambda: L.gemm_fp8_nt(A8, sa, W8, sw, epi, bias=bias, **kw))
    L.set_debug_flags(256 | 8)        # ... and both without their epilogues: the k-loops alone
    k32 = t(lambda: L.gemm_fp8_nt(A8, sa, W8, sw, epi, bias=bias, **kw))
    L.set_debug_flags(8)
    kmx = t(lambda: L.gemm_fp8_nt(A8, sa, W8, sw, epi, bias=bias, **kw))
    kb = t(lambda: L.gemm_bf16_nt(A16, W16, epi, bias=bias, **kw))
    L.set_debug_flags(0)
    f = 2.0 * M * N * K
    print(f"({M},{N},{K}) epi {epi}: bf16 {tb:.1f} us ({f / tb / 1e6:.0f} TF/s)   fp8 MX K=128 {t8:.1f} us ({f / t8 / 1e6:.0f} TF/s)   "
          f"fp8 K=32 {t8k32:.1f} us   | k-loop only: bf16 {kb:.1f} us ({f / kb / 1e6:.0f} TF/s), fp8 MX {kmx:.1f} us "
          f"({f / kmx / 1e6:.0f} TF/s), fp8 K=32 {k32:.1f} us ({f / k32 / 1e6:.0f} TF/s)   max |MX - K32| {float((o_mx.float() - o.float()).abs().max()):.3g}", flush=True)
