import sys, torch
sys.path.insert(0, "/root/repo")
from feddat_amd import lib as L
dev = "cuda:0"
M = 11840
for (N, K, epi) in [(3072, 768, 3), (3072, 768, 2), (3072, 768, 0), (2304, 768, 0), (768, 3072, 1), (768, 768, 1), (768, 3072, 0), (768, 2304, 0), (768, 768, 0)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    o = torch.empty(M, N, dtype=torch.bfloat16, device=dev); o2 = torch.empty_like(o)
    aux = torch.randn(M, N, device=dev).bfloat16(); resid = torch.randn(M, N, device=dev); o32 = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    def call():
        if epi == 0: L.gemm_bf16_nt(A, B, 0, bias=bias, out_bf16=o)
        elif epi == 1: L.gemm_bf16_nt(A, B, 1, bias=bias, resid=resid, out_f32=o32)
        elif epi == 2: L.gemm_bf16_nt(A, B, 2, bias=bias, out_bf16=o, out2_bf16=o2)
        elif epi == 3: L.gemm_bf16_nt(A, B, 3, aux=aux, out_bf16=o)
    res = []
    for flag in (1 | 32, 1 | 64, 2 | 32, 2 | 64, 2 | 32 | 8, 2 | 64 | 8):      # 1: v2 only, 2: force v3; 32 / 64: 192- / 256-row tiles
        L.set_debug_flags(flag)
        call(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20): call()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    L.set_debug_flags(0)
    print(f"N={N} K={K} epi={epi}: v2 192 {res[0]:.1f} / 256 {res[1]:.1f} us | v3 192 {res[2]:.1f} / 256 {res[3]:.1f} us | v3 k-loop only 192 {res[4]:.1f} ({2*M*N*K/res[4]/1e6:.0f} TF/s) / 256 {res[5]:.1f} ({2*M*N*K/res[5]/1e6:.0f} TF/s)")
