import sys, torch
sys.path.insert(0, "/root/repo")
from feddat_amd import lib as L
dev = "cuda:0"
torch.manual_seed(0)
for (M, N, K) in [(11840, 768, 768), (11849, 3072, 768), (5920, 2304, 768), (1200, 192, 256), (11840, 2304, 768), (11840, 768, 3072), (18464, 768, 2304), (1030, 192, 64)]:
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev); aux = torch.randn(M, N, device=dev).bfloat16()
    outs = {}
    for flag in (1, int(sys.argv[1]) if len(sys.argv) > 1 else 2):     # 1 = v2 only
        L.set_debug_flags(flag)
        o = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); o2 = torch.zeros_like(o); o3 = torch.zeros_like(o)
        o32 = torch.zeros(M, N, device=dev); o4 = torch.zeros_like(o); o5 = torch.zeros(M, N, device=dev)
        for rep in range(2):
            L.gemm_bf16_nt(A, B, 0, bias=bias, out_bf16=o)
            L.gemm_bf16_nt(A, B, 1, bias=bias, resid=resid, out_f32=o32)
            L.gemm_bf16_nt(A, B, 2, bias=bias, out_bf16=o2, out2_bf16=o3)
            L.gemm_bf16_nt(A, B, 3, aux=aux, out_bf16=o4)
            L.gemm_bf16_nt(A, B, 4, bias=bias, out_f32=o5)
        torch.cuda.synchronize()
        outs[flag] = [o, o32, o2, o3, o4, o5]
    L.set_debug_flags(0)
    k1 = [k for k in outs if k != 1][0]
    d = [float((x.float() - y.float()).abs().max()) for x, y in zip(outs[1], outs[k1])]
    print(M, N, K, "max abs diff v2 vs v3 per epilogue:", ["%.3g" % v for v in d])
