"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the batch
sharding, the scatter/gather used for the batch split, and the max-over-ranks timing.
The per-rank transform is injected (the oracle -- allowed in tests only); the product
path uses the HIP batch entry and no collective inside the transform."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_items, h, w, out_path, wavelet="cdf97_s"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oraclelib import Oracle

        from libdwt_amd import batch as B

        orc = Oracle()
        orc.set_threads(1)
        rng = np.random.default_rng(2024)
        fname = {"cdf97_s": "cdf97_2f_s", "cdf53_i": "cdf53_2f_i", "cdf97_i": "cdf97_2f_i", "cdf97_d": "cdf97_2f_d"}[wavelet]
        if wavelet == "cdf97_s":
            full = torch.from_numpy(rng.random((n_items, h, w), dtype=np.float32))  # same on every rank; only root's is used
        elif wavelet == "cdf97_d":
            full = torch.from_numpy(rng.random((n_items, h, w)))
        else:
            full = torch.from_numpy(rng.integers(-32768, 32768, size=(n_items, h, w), dtype=np.int32))

        def cpu_transform(block, levels):
            assert block.dtype == full.dtype, "the scattered block lost its dtype"
            out = block.clone().numpy()
            for k in range(out.shape[0]):
                orc.fwd(fname, out[k], levels)
            return torch.from_numpy(out)

        lo, hi = B.shard_range(n_items, rank, world)
        got = B.transform_sharded(full if rank == 0 else None, n_items, (h, w), wavelet, 3,
                                  device=torch.device("cpu"), transform=cpu_transform)
        t = B.max_over_ranks(0.5 + rank)
        assert t == 0.5 + (world - 1), t
        if rank == 0:
            want = full.clone().numpy()
            for k in range(n_items):
                orc.fwd(fname, want[k], 3)
            ok = got.dtype == full.dtype and np.array_equal(got.numpy().view(np.uint8), want.view(np.uint8))
            with open(out_path, "w") as f:
                f.write("ok" if ok else "mismatch")
        else:
            assert got is None
        # every rank's block is what the partition says
        assert 0 <= lo <= hi <= n_items
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 5), (3, 4), (2, 1)])
def test_sharded_batch_roundtrip_gloo(tmp_path, world, n_items):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), n_items, 48, 64, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


@pytest.mark.parametrize("wavelet", ["cdf53_i", "cdf97_i", "cdf97_d"])
def test_sharded_int_wavelets_keep_their_dtype(tmp_path, wavelet):
    """int32 wavelets (reversible 5/3, fixed-point 9/7) and the double-precision 9/7 through the batch split:
    blocks keep their element type (and, for double, their 8-byte strides); an unknown name is refused."""
    from libdwt_amd import batch as B

    with pytest.raises(ValueError):
        B.transform_sharded(None, 1, (8, 8), "cdf97_x")
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(2, _free_port(), 3, 48, 64, str(out), wavelet), nprocs=2, join=True)
    assert out.read_text() == "ok"


def test_shard_range_partitions():
    from libdwt_amd.batch import shard_range

    for n in (0, 1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and a <= b and c <= d
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
            # SURVEY.md s8e, literally: item b belongs to rank b*world//n
            for r, (a, b) in enumerate(blocks):
                assert all((i * world) // n == r for i in range(a, b))
    # config 4: 256 images over 8 GPUs -> 32 contiguous images each
    assert shard_range(256, 3, 8) == (96, 128)
    # five images over three ranks: 2, 2, 1 (b*3//5 = 0, 0, 1, 1, 2)
    assert [shard_range(5, r, 3) for r in range(3)] == [(0, 2), (2, 4), (4, 5)]
    # the C entry a one-process caller uses gives the same bounds (no device call inside: runs without a GPU)
    import libdwt_amd as dwt

    for n in (1, 5, 7, 64, 256):
        for world in (1, 2, 3, 8):
            for r in range(world):
                a, cnt = dwt.shard_bounds(n, world, r)
                assert (a, a + cnt) == shard_range(n, r, world)
