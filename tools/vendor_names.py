"""ad-hoc: run the vendor GEMM once per step shape under rocprofv3 to read its kernel names (tile / schedule parameters)."""
import torch
for (m, n, k) in ((8192, 8192, 8192), (22512, 10240, 2560), (22512, 2560, 10240), (22512, 7680, 2560), (22512, 2560, 2560), (98688, 4096, 1024), (98688, 1024, 4096)):
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    for _ in range(5):
        y = torch.matmul(x, w.t())
    torch.cuda.synchronize()
