"""Batch sharding of independent images across the GPUs of one node.

libdwt has no multi-process anything; images of a batch are independent units, so the
path shards embarrassingly: image b of B goes to rank b*G//B
(contiguous blocks), every rank runs the single-GPU multi-level pipeline on its block,
and there is NO collective inside the transform.  torch.distributed (backend "nccl" =
RCCL over xGMI on the GPU node, "gloo" in the CPU tests) is used only

  * to split a batch that starts on one rank (scatter) and to collect it (gather), and
  * for the barrier / max-over-ranks of the benchmark's timing.

One process per GPU; the caller initialises the process group.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """[lo, hi) of the contiguous block of `n_items` owned by `rank`: item b belongs to rank
    b*world//n_items, so rank r owns [ceil(r*n/world), ceil((r+1)*n/world));
    block sizes differ by at most one.  The C entry dwt_hip_shard_bounds gives the same bounds."""
    lo = (n_items * rank + world - 1) // world
    hi = (n_items * (rank + 1) + world - 1) // world
    return lo, hi


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def _p2p(ops):
    """One grouped launch of point-to-point transfers (ncclGroupStart/End on RCCL: the
    root's sends leave over its 7 xGMI links concurrently instead of one after another;
    on gloo the ops are posted individually)."""
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def _loopback(src, dst):
    """A group of ONE rank: the block goes through the backend's own point-to-point path, rank 0 -> rank 0
    (one grouped isend + irecv).  The rehearsal of the split on a single GPU: the communicator, the grouped
    launch and the copy kernels all run, only the xGMI hop is missing."""
    _p2p([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)])


def scatter_images(batch_on_root, n_items, item_shape, dtype, device, root=0, loopback=False):
    """Split a (B, H, W) batch held by `root` into per-rank blocks.  Every rank passes
    the same n_items/item_shape/dtype; only root passes the tensor.  Returns this
    rank's block (a view of the input on a single rank; with `loopback` and an initialised
    group of one rank, a copy that travelled through the backend's send / recv)."""
    rank, world = _world()
    lo, hi = shard_range(n_items, rank, world)
    if world == 1:
        if loopback and dist.is_available() and dist.is_initialized():
            local = torch.empty((hi - lo,) + tuple(item_shape), dtype=dtype, device=device)
            _loopback(batch_on_root[lo:hi].contiguous(), local)
            return local
        return batch_on_root[lo:hi]
    local = torch.empty((hi - lo,) + tuple(item_shape), dtype=dtype, device=device)
    ops = []
    if rank == root:
        if batch_on_root.dtype != dtype:
            raise TypeError(f"batch dtype {batch_on_root.dtype} does not match the wavelet's {dtype}")
        for r in range(world):
            rlo, rhi = shard_range(n_items, r, world)
            if r == root:
                local.copy_(batch_on_root[rlo:rhi])
            elif rhi > rlo:
                # a contiguous block of a contiguous batch: sent in place, no staging copy
                ops.append(dist.P2POp(dist.isend, batch_on_root[rlo:rhi].contiguous(), r))
    elif hi > lo:
        ops.append(dist.P2POp(dist.irecv, local, root))
    _p2p(ops)
    return local


def gather_images(local, n_items, root=0, loopback=False):
    """Inverse of scatter_images: root gets the (B, ...) batch back, others get None."""
    rank, world = _world()
    if world == 1:
        if loopback and dist.is_available() and dist.is_initialized():
            out = torch.empty_like(local)
            _loopback(local.contiguous(), out)
            return out
        return local
    ops = []
    out = None
    if rank == root:
        out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for r in range(world):
            rlo, rhi = shard_range(n_items, r, world)
            if r == root:
                out[rlo:rhi].copy_(local)
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.irecv, out[rlo:rhi], r))
    elif local.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, local.contiguous(), root))
    _p2p(ops)
    return out


def max_over_ranks(seconds, device="cpu"):
    """Slowest rank's time: what a whole-job throughput has to be divided by."""
    rank, world = _world()
    if world == 1:
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def transform_sharded(batch_on_root, n_items, item_shape, wavelet="cdf97_s", levels=-1, inverse=False,
                      device=None, root=0, transform=None):
    """scatter -> per-rank multi-level transform of the local block -> gather.

    `transform(block, levels)` defaults to the HIP batch entry (device tensors); the CPU
    tests inject the oracle here, the product never does."""
    rank, world = _world()
    dtypes = {"cdf97_s": torch.float32, "cdf53_s": torch.float32, "cdf53_i": torch.int32, "cdf97_i": torch.int32,
              "cdf97_d": torch.float64, "cdf53_d": torch.float64}
    if wavelet not in dtypes:
        raise ValueError(f"transform_sharded: unknown wavelet {wavelet!r} (one of {sorted(dtypes)})")
    dtype = dtypes[wavelet]
    es = 8 if dtype == torch.float64 else 4
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    local = scatter_images(batch_on_root, n_items, item_shape, dtype, device, root)
    if transform is None:
        import libdwt_amd as dwt

        def transform(block, lv):
            if block.shape[0] == 0:
                return block
            h, w = block.shape[1:]
            if block.dtype != dtype:
                raise TypeError(f"block dtype {block.dtype} does not match {wavelet}")
            out = torch.empty_like(block)
            dwt.use_torch_stream()
            dwt.transform2d_batch(wavelet, int(inverse), block, out, h * w * es, block.shape[0], w * es, w, h, lv)
            return out
    local_out = transform(local, levels)
    return gather_images(local_out, n_items, root)
