"""Yardstick only: which kernels the vendor GEMM library picks for the nav-step shapes (run under rocprofv3 --kernel-trace)."""
import torch
dt = torch.bfloat16
for (M, N, K) in [(2304, 3072, 768), (2304, 2304, 768), (2432, 3072, 768), (1280, 3072, 768), (2304, 768, 3072), (6912, 3072, 128), (4096, 4096, 4096), (5120, 12288, 768)]:
    x = torch.rand(M, K, device='cuda').to(dt); w = torch.rand(N, K, device='cuda').to(dt)
    y = torch.empty(M, N, device='cuda', dtype=dt)
    for _ in range(5):
        torch.mm(x, w.t(), out=y)
    torch.cuda.synchronize()
