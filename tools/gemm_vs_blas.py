import sys, torch
sys.path.insert(0, '.')
from feddat_amd import lib as L
dev = 'cuda'
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, K) in [(11840, 768, 768), (11840, 2304, 768), (11840, 3072, 768), (11840, 768, 3072), (11840, 768, 2304), (5920, 2304, 768), (8192, 8192, 8192)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    us_mine = t(lambda: L.gemm_bf16_nt(A, B, 0, out_bf16=out))
    us_blas = t(lambda: torch.mm(A, B.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"{M}x{N}x{K}: mine {us_mine:7.1f} us {fl/us_mine/1e6:7.0f} TF/s | hipblaslt {us_blas:7.1f} us {fl/us_blas/1e6:7.0f} TF/s")
