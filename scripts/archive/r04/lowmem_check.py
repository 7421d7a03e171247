import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.getcwd() + "/tests")
import numpy as np, torch
import libdwt_amd as dwt
from oraclelib import Oracle
dwt.dwt_util_init(); dwt.use_torch_stream()
orc = Oracle()
GB = 1 << 30
n, nb, J = 4096, 32, 4      # 2 GiB batches
img = np.random.default_rng(0).random((n, n), dtype=np.float32)
want = img.copy(); orc.fwd("cdf97_2f_s", want, J)
for hog_gb in (0, 240, 272):
    free_b, _ = torch.cuda.mem_get_info()
    hog = torch.empty(min(hog_gb * GB, free_b - 10 * GB), dtype=torch.uint8, device="cuda") if hog_gb else None
    free_b, _ = torch.cuda.mem_get_info()
    src, dst = dwt.alloc_batch("cdf97_s", nb, n, n, J)
    rep = dwt.alloc_batch_report()
    for k in (0, nb - 1):
        assert dwt.lib.dwt_hip_memcpy_h2d(src + k * n * n * 4, img.ctypes.data, img.nbytes) == 0
    dwt.lib.dwt_hip_probe_pair_us(src + n * n * 4, None, (nb - 2) * n * n * 4)
    dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J)
    got = np.empty_like(img)
    ok = True
    for k in (0, nb - 1):
        assert dwt.lib.dwt_hip_memcpy_d2h(got.ctypes.data, dst + k * n * n * 4, got.nbytes) == 0
        ok = ok and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    print(f"hog {hog_gb} GiB: free {free_b/GB:.1f} GiB -> arena {rep['arena_GiB']} GiB, {rep['seconds']} s, bits ok: {ok}", flush=True)
    dwt.lib.dwt_hip_free(src); dwt.lib.dwt_hip_free(dst); dwt.dwt_util_finish()
    del hog; torch.cuda.empty_cache()
