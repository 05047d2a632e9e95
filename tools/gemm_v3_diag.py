import sys, torch
sys.path.insert(0, "/root/repo")
from feddat_amd import lib as L
dev = "cuda:0"
torch.manual_seed(0)
M, N, K = 11840, 2304, 768
A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
flag = int(sys.argv[1])
o0 = torch.zeros(M, N, dtype=torch.bfloat16, device=dev); o1 = torch.full((M, N), 777.0, dtype=torch.bfloat16, device=dev)
L.set_debug_flags(0); L.gemm_bf16_nt(A, B, 0, out_bf16=o0)
L.set_debug_flags(flag); L.gemm_bf16_nt(A, B, 0, out_bf16=o1); L.set_debug_flags(0)
torch.cuda.synchronize()
d = (o0.float() - o1.float()).abs()
bm = 191  # plan: nmt = 62 -> bm = ceil(11840/62) = 191
tm = (M + bm - 1) // bm
bad = []
for i in range(tm):
    for j in range(N // 192):
        blk = d[i * bm:(i + 1) * bm, j * 192:(j + 1) * 192]
        if float(blk.max()) > 0:
            bad.append((i, j, float(blk.max()), int((blk > 0).sum()), int((o1[i * bm:(i + 1) * bm, j * 192:(j + 1) * 192] == 777).sum())))
print("tiles", tm, N // 192, "bad tiles", len(bad))
print(bad[:12])
if bad:
    i, j = bad[0][:2]
    blk = d[i * bm:(i + 1) * bm, j * 192:(j + 1) * 192]
    rows = (blk.max(1).values > 0).nonzero().flatten().tolist(); cols = (blk.max(0).values > 0).nonzero().flatten().tolist()
    print("bad rows", rows[:40], "...", len(rows)); print("bad cols", cols[:40], "...", len(cols))
rb = (d.max(1).values > 0).nonzero().flatten()
print("global bad rows: count", rb.numel(), "first", rb[:20].tolist())
import collections
print("row % 185 histogram:", sorted(collections.Counter((rb % 185).tolist()).items())[:40])
cb = (d.max(0).values > 0).nonzero().flatten()
print("bad cols % 4:", collections.Counter((cb % 4).tolist()), " bad cols % 192 first:", sorted(set((cb % 192).tolist()))[:30])
r0 = int(rb[0]); print("row", r0, "bad cols:", (d[r0] > 0).nonzero().flatten()[:24].tolist(), "values v2/v3:", o0[r0, (d[r0] > 0).nonzero().flatten()[:4]].tolist(), o1[r0, (d[r0] > 0).nonzero().flatten()[:4]].tolist())
