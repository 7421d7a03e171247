import os, sys
sys.path.insert(0, os.getcwd())
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J = 8192, 5
a = torch.rand((n, n), device="cuda"); b = torch.empty_like(a); c = torch.empty_like(a)
for _ in range(6):
    dwt.transform2d_interleaved("cdf97_s", 0, 0, a, b, n*4, 4, n, n, None, None, J)
    torch.cuda.synchronize()
for _ in range(6):
    dwt.transform2d_interleaved("cdf97_s", 1, 0, b, c, n*4, 4, n, n, None, None, J)
    torch.cuda.synchronize()
