import os, sys
sys.path.insert(0, "/root/repo")
import torch
import libdwt_amd as dwt
dwt.dwt_util_init(); dwt.use_torch_stream()
n, J = 8192, 5
img = torch.rand((4, n, n), device="cuda"); out = torch.empty_like(img)
for i in range(12):
    dwt.transform2d_interleaved("cdf97_s", 0, 0, img[i % 4], out[i % 4], n * 4, 4, n, n, n, n, J)
torch.cuda.synchronize()
