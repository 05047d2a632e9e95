import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feddat_amd import lib as L
L.use_ablation_build()      # timing-only probes live in libfeddat_hip_ablate.so (python -m feddat_amd.build --ablate)
dev = "cuda"
def run(M, N, K, epi, iters=30):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    kw = {}
    if epi in (0, 2, 3, 5, 6): kw["out_bf16"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epi == 2: kw["out2_bf16"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epi == 3: kw["aux"] = torch.randn(M, N, device=dev).to(torch.bfloat16)
    if epi == 5: kw["out2_bf16"] = torch.empty(M, N, dtype=torch.uint8, device=dev)
    if epi == 6: kw["aux"] = torch.randint(0, 255, (M, N), dtype=torch.uint8, device=dev)
    if epi == 1: kw["resid"] = torch.randn(M, N, device=dev); kw["out_f32"] = torch.empty(M, N, device=dev)
    if epi not in (3, 6): kw["bias"] = torch.randn(N, device=dev)
    for _ in range(10): L.gemm_bf16_nt(A, B, epi, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.gemm_bf16_nt(A, B, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K, epi) in [(11840, 3072, 768, 2), (11840, 3072, 768, 5), (11840, 3072, 768, 3), (11840, 3072, 768, 6), (11840, 3072, 768, 0), (11840, 768, 3072, 1), (11840, 768, 768, 1)]:
    res = {}
    for name, d in (("full", 0), ("no-gelu-math", 4), ("no-stores", 16), ("no-epilogue", 8)):
        L.set_debug_flags(d)
        res[name] = round(run(M, N, K, epi), 1)
    L.set_debug_flags(0)
    f = 2.0 * M * N * K
    print((M, N, K, epi), res, "TF/s full %.0f, k-loop only %.0f" % (f / res["full"] / 1e6, f / res["no-epilogue"] / 1e6), flush=True)
