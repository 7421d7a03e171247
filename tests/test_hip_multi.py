"""GPU tests of the multi-GPU front end behind the C-ABI and of the explicit measurement entry.

The test box has ONE GPU: several slots on device 0 are several host threads with contexts of their own on
the one card -- the same code path as several cards (device binding, streams, peer copies, workspaces), minus
the xGMI hop.  Every result is compared with the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu

NAMES = {"cdf97_s": ("cdf97_2f_s", "cdf97_2i_s", np.float32), "cdf53_i": ("cdf53_2f_i", "cdf53_2i_i", np.int32)}


@pytest.fixture(scope="module")
def dwt():
    import libdwt_amd as d

    d.dwt_util_init()
    yield d
    d.dwt_util_finish()


def rand(rng, shape, dt):
    if dt == np.int32:
        return rng.integers(-32768, 32768, size=shape, dtype=np.int32)
    return rng.random(shape, dtype=np.float32) * 2 - 1


def test_shard_bounds_are_the_partition_of_the_survey(dwt):
    """image b -> slot b*G//B, as SURVEY.md s8e writes it; the same bounds as libdwt_amd.batch.shard_range."""
    from libdwt_amd.batch import shard_range

    for B in (1, 5, 7, 64, 256):
        for G in (1, 2, 3, 8):
            for k in range(G):
                a, n = dwt.shard_bounds(B, G, k)
                assert (a, a + n) == shard_range(B, k, G)
                assert all((b * G) // B == k for b in range(a, a + n))
    assert [dwt.shard_bounds(5, 3, k) for k in range(3)] == [(0, 2), (2, 2), (4, 1)]
    # a slot outside [0, n_slots) or a negative batch owns nothing (it used to return a range past the batch's end)
    assert dwt.shard_bounds(5, 3, 3) == (5, 0) and dwt.shard_bounds(5, 3, 7) == (5, 0) and dwt.shard_bounds(5, 3, -1) == (5, 0)
    assert dwt.shard_bounds(-4, 3, 1) == (0, 0)


@pytest.mark.parametrize("wname", ["cdf97_s", "cdf53_i"])
def test_placed_batch_through_the_sharded_entry(dwt, oracle, wname):
    """dwt_hip_alloc_batch (buffers mapped through the virtual-memory API, placed by measurement) handed to
    dwt_hip_transform2d_batch_sharded with three slots: the slots' peer copies read and write the placed buffers from
    other contexts (round 4 granted the mapping to the allocating device only and never ran the two together).  Seven
    images over three slots: 3 + 2 + 2; each slot's shard crosses in pieces.  Forward and inverse == oracle."""
    ff, fi, dt = NAMES[wname]
    L = dwt.lib
    nb, h, w, J = 7, 260, 520, 3
    rng = np.random.default_rng(5)
    imgs = rand(rng, (nb, h, w), dt)
    want = imgs.copy()
    for k in range(nb):
        oracle.fwd(ff, want[k], J)
    rec = want.copy()
    for k in range(nb):
        oracle.inv(fi, rec[k], J)
    try:
        dwt.set_option("place_min_mib", 0)
        dwt.set_option("place_tries", 2)
        dwt.set_option("place_max_gib", 24)
        dwt.dwt_util_finish()
        src, dst = dwt.alloc_batch(wname, nb, w, h, J)
        if wname == "cdf97_s":  # (the search runs for the 32-bit float wavelet; other wavelets say why not)
            assert dwt.alloc_batch_note() == "" and 8 <= dwt.alloc_batch_report()["arena_GiB"] <= 24, (dwt.alloc_batch_note(), dwt.alloc_batch_report())
        dwt.grant_access(src, [0])
        dwt.grant_access(dst, [0])
        assert L.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
        got = np.empty_like(imgs)
        zeros = np.zeros_like(imgs)  # (kept alive: the address of a temporary would dangle by the time the copy reads it)
        for devices in ([0, 0, 0], [0], [0, 0, 0, 0, 0]):
            assert L.dwt_hip_memcpy_h2d(dst, zeros.ctypes.data, imgs.nbytes) == 0
            assert dwt.transform2d_batch_sharded(wname, 0, src, dst, h * w * 4, nb, w * 4, w, h, J, devices) == J
            assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
            assert np.array_equal(bits(got), bits(want)), devices
        assert dwt.transform2d_batch_sharded(wname, 1, dst, src, h * w * 4, nb, w * 4, w, h, J, [0, 0, 0]) == J
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, src, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(rec))
        # freeing placed buffers while nothing is queued on them, then a fresh call: the context is still usable
        L.dwt_hip_free(src)
        L.dwt_hip_free(dst)
        d = dwt.DeviceImage(h, w).upload(imgs[0])
        getattr(dwt, "dwt_" + ff)(d.ptr, w * 4, 4, w, h, w, h, J)
        assert np.array_equal(bits(d.download(dt)), bits(want[0]))
        d.free()
    finally:
        dwt.set_option("place_min_mib", 1024)
        dwt.set_option("place_tries", 4)
        dwt.set_option("place_max_gib", 0)


@pytest.mark.parametrize("wname", ["cdf97_s", "cdf53_i"])
def test_resident_shards_from_c_abi(dwt, oracle, wname):
    """dwt_hip_transform2d_batch_multi: every shard already lies where it is transformed (the >= 7x case) --
    four shards of unequal size on devices {0, 0, 0, 0}, an empty shard among them, each in buffers of its own; the
    library's slot workers run all of them at once.  Forward and inverse == oracle; a shard that names a device the
    process does not have is refused."""
    ff, fi, dt = NAMES[wname]
    L = dwt.lib
    h, w, J = 300, 1030, 3
    counts = [3, 0, 2, 4, 1]
    rng = np.random.default_rng(17)
    shards = [rand(rng, (n, h, w), dt) for n in counts]
    srcs = [L.dwt_hip_malloc(max(s.nbytes, 16)) for s in shards]
    dsts = [L.dwt_hip_malloc(max(s.nbytes, 16)) for s in shards]
    for p, s in zip(srcs, shards):
        if s.size:
            assert L.dwt_hip_memcpy_h2d(p, s.ctypes.data, s.nbytes) == 0
    devices = [0] * len(counts)
    assert dwt.transform2d_batch_multi(wname, 0, srcs, dsts, counts, devices, h * w * 4, w * 4, w, h, J) == J
    coeffs = []
    for p, s in zip(dsts, shards):
        got = np.empty_like(s)
        if s.size:
            assert L.dwt_hip_memcpy_d2h(got.ctypes.data, p, got.nbytes) == 0
        want = s.copy()
        for k in range(s.shape[0]):
            oracle.fwd(ff, want[k], J)
        assert np.array_equal(bits(got), bits(want))
        coeffs.append(want)
    # inverse: coefficients in dsts -> srcs
    assert dwt.transform2d_batch_multi(wname, 1, dsts, srcs, counts, devices, h * w * 4, w * 4, w, h, J) == J
    for p, c in zip(srcs, coeffs):
        got = np.empty_like(c)
        if c.size:
            assert L.dwt_hip_memcpy_d2h(got.ctypes.data, p, got.nbytes) == 0
        want = c.copy()
        for k in range(c.shape[0]):
            oracle.inv(fi, want[k], J)
        assert np.array_equal(bits(got), bits(want))
    with pytest.raises(dwt.DwtError):
        dwt.transform2d_batch_multi(wname, 0, srcs, dsts, counts, [0, 0, 0, 99, 0], h * w * 4, w * 4, w, h, J)
    with pytest.raises(dwt.DwtError):
        dwt.transform2d_batch_multi(wname, 0, srcs, [None] * 5, counts, devices, h * w * 4, w * 4, w, h, J)
    for p in srcs + dsts:
        L.dwt_hip_free(p)


def test_tuning_in_four_slots_at_once(dwt, oracle):
    """dwt_hip_tune_batch_multi on {0, 0, 0, 0}: four host threads ask for the placement search (spacers of 14 + 28 GiB
    each) and the tile tuner on the one GPU at the same time.  The library runs one measurement at a time per device:
    no deadlock, no out-of-memory, every slot ends up with measured tile heights, and the transforms that follow give
    the oracle's bits (spot-checked per shard) in exactly J launches per shard."""
    L = dwt.lib
    n, J, per = 4096, 3, 8   # 8 x 4096^2 per shard: level 0 reads 512 MiB -> measured
    rng = np.random.default_rng(23)
    shards = [rng.random((per, n, n), dtype=np.float32) for _ in range(4)]
    srcs = [L.dwt_hip_malloc(s.nbytes) for s in shards]
    dsts = [L.dwt_hip_malloc(s.nbytes) for s in shards]
    for p, s in zip(srcs, shards):
        assert L.dwt_hip_memcpy_h2d(p, s.ctypes.data, s.nbytes) == 0
    try:
        dwt.set_option("place_min_mib", 0)
        dwt.set_option("place_tries", 3)
        dwt.dwt_util_finish()
        dwt.tune_batch_multi("cdf97_s", 0, srcs, dsts, [per] * 4, [0] * 4, n * n * 4, n * 4, n, n, J)
        assert len(dwt.placement_report()[0]) >= 1 and dwt.get_option("tile_cache_size") >= 1   # (the caller's own slot)
        for p, s in zip(dsts, shards):
            zero_img = np.zeros_like(s)  # (bound to a name: a temporary's address would dangle)
            assert L.dwt_hip_memcpy_h2d(p, zero_img.ctypes.data, s.nbytes) == 0
        launches = dwt.get_option("stat_launches")
        assert dwt.transform2d_batch_multi("cdf97_s", 0, srcs, dsts, [per] * 4, [0] * 4, n * n * 4, n * 4, n, n, J) == J
        assert dwt.get_option("stat_launches") - launches == J
        for k, (p, s) in enumerate(zip(dsts, shards)):
            got = np.empty_like(s)
            assert L.dwt_hip_memcpy_d2h(got.ctypes.data, p, got.nbytes) == 0
            want = s[k % per].copy()
            oracle.fwd("cdf97_2f_s", want, J)
            assert np.array_equal(bits(got[k % per]), bits(want)), k
    finally:
        dwt.set_option("place_min_mib", 1024)
        dwt.set_option("place_tries", 4)
        for p in srcs + dsts:
            L.dwt_hip_free(p)


def test_a_transform_call_measures_nothing(dwt, oracle):
    """The first forward call on a batch that needs more than 1 GiB of scratch (16 x 8192^2, J = 5): exactly J kernel
    launches and exactly the two scratch allocations -- no placement search, no spacers, no tile tuner, no extra runs of
    the caller's transform (round 4 did all of that inside the call).  dwt_hip_tune is where measurement happens: it
    searches and tunes, the call after it is again J launches and allocates nothing, and the bits are the same -- the
    oracle's (two images spot-checked)."""
    L = dwt.lib
    nb, n, J = 16, 8192, 5
    rng = np.random.default_rng(3)
    one = rng.random((2, n, n), dtype=np.float32)
    src, dst = L.dwt_hip_malloc(nb * n * n * 4), L.dwt_hip_malloc(nb * n * n * 4)
    assert src and dst
    for b in range(nb):  # images 0, 2, 4 ... = one[0], the odd ones = one[1]
        assert L.dwt_hip_memcpy_h2d(src + b * n * n * 4, one[b & 1].ctypes.data, n * n * 4) == 0
    want = one.copy()
    for k in range(2):
        oracle.fwd("cdf97_2f_s", want[k], J)
    try:
        dwt.dwt_util_finish()
        l0, a0 = dwt.get_option("stat_launches"), dwt.get_option("stat_allocs")
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J) == J
        dwt.sync()
        assert dwt.get_option("stat_launches") - l0 == J and dwt.get_option("stat_allocs") - a0 == 2
        assert dwt.placement_report()[0] == [] and dwt.get_option("tile_cache_size") == 0
        first = np.empty((2, n, n), dtype=np.float32)
        for k, b in ((0, 14), (1, 15)):
            assert L.dwt_hip_memcpy_d2h(first[k].ctypes.data, dst + b * n * n * 4, n * n * 4) == 0
        assert np.array_equal(bits(first), bits(want))
        # the explicit measurement, then the same call again
        dwt.dwt_util_finish()
        dwt.tune("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J)
        assert len(dwt.placement_report()[0]) >= 2 and dwt.get_option("tile_cache_size") >= 2
        zero_img = np.zeros((n, n), np.float32)  # (bound to a name: a temporary's address would dangle)
        assert L.dwt_hip_memcpy_h2d(dst + 15 * n * n * 4, zero_img.ctypes.data, n * n * 4) == 0
        l0, a0 = dwt.get_option("stat_launches"), dwt.get_option("stat_allocs")
        assert dwt.transform2d_batch("cdf97_s", 0, src, dst, n * n * 4, nb, n * 4, n, n, J) == J
        dwt.sync()
        assert dwt.get_option("stat_launches") - l0 == J and dwt.get_option("stat_allocs") - a0 == 0
        again = np.empty_like(first)
        for k, b in ((0, 14), (1, 15)):
            assert L.dwt_hip_memcpy_d2h(again[k].ctypes.data, dst + b * n * n * 4, n * n * 4) == 0
        assert np.array_equal(bits(again), bits(want))
    finally:
        L.dwt_hip_free(src)
        L.dwt_hip_free(dst)
        dwt.dwt_util_finish()


TENSOR_SHARDS_SCRIPT = r"""
import sys, numpy as np
import torch                      # first: this process then shares torch's HIP runtime
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import libdwt_amd as dwt
from oraclelib import Oracle
dwt.dwt_util_init()
orc = Oracle()
h, w, J = 128, 260, 2
rng = np.random.default_rng(23)
shards = [rng.random((n, h, w), dtype=np.float32) for n in (2, 1, 3)]
shards[1][0, 0, 0] = 0.0
srcs = [torch.from_numpy(s).to("cuda:0") for s in shards]
dsts = [torch.zeros_like(t) for t in srcs]
torch.cuda.synchronize()
def check(scale):
    for t, s in zip(dsts, shards):
        want = s * scale
        for k in range(s.shape[0]):
            orc.fwd("cdf97_2f_s", want[k], J)
        assert np.array_equal(t.cpu().numpy().view(np.uint32), want.view(np.uint32))
assert dwt.transform2d_batch_multi("cdf97_s", 0, srcs, dsts, [2, 1, 3], [0, 0, 0], h * w * 4, w * 4, w, h, J) == J
check(np.float32(1))
# the producers of a shard on a torch SIDE stream: the slots drain the device before they read
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for t, s in zip(srcs, shards):
        t.copy_(torch.from_numpy(s * 2).to("cuda:0", non_blocking=True))
assert dwt.transform2d_batch_multi("cdf97_s", 0, srcs, dsts, [2, 1, 3], [0, 0, 0], h * w * 4, w * 4, w, h, J) == J
check(np.float32(2))
try:
    dwt.transform2d_batch_multi("cdf97_s", 0, srcs, [dsts[0], None, dsts[2]], [2, 1, 3], [0, 0, 0], h * w * 4, w * 4, w, h, J)
    raise SystemExit("a null shard was accepted")
except dwt.DwtError:
    pass
# two physical devices, where the process sees them (the build box has one)
if dwt.device_count() >= 2:
    nb, hh, ww = 6, 300, 1030
    imgs = rng.random((nb, hh, ww), dtype=np.float32)
    want = imgs.copy()
    for k in range(nb):
        orc.fwd("cdf97_2f_s", want[k], 3)
    s2 = [torch.from_numpy(np.ascontiguousarray(p)).to(f"cuda:{d}") for d, p in enumerate((imgs[:3], imgs[3:]))]
    d2 = [torch.zeros_like(t) for t in s2]
    for d in range(2):
        torch.cuda.synchronize(d)
    assert dwt.transform2d_batch_multi("cdf97_s", 0, s2, d2, [3, 3], [0, 1], hh * ww * 4, ww * 4, ww, hh, 3) == 3
    assert np.array_equal(np.concatenate([t.cpu().numpy() for t in d2]).view(np.uint32), want.view(np.uint32))
    print("two devices OK")
print("tensor shards OK")
"""


def test_resident_shards_given_as_torch_tensors():
    """The pointer arrays of dwt_hip_transform2d_batch_multi built from torch tensors (every other wrapper takes
    them through `_addr`; truth-testing a tensor with several elements used to raise in `_multi_args`, and a
    one-element zero tensor turned into NULL); producers on a torch side stream; and -- where the process sees two
    GPUs -- shards resident on devices 0 and 1.  Own process: torch before the library (one HIP runtime)."""
    import os
    import subprocess
    import sys

    pytest.importorskip("torch")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % root + TENSOR_SHARDS_SCRIPT], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "tensor shards OK" in out.stdout, out.stderr[-2000:]


def test_two_physical_devices_sharded_and_resident(dwt, oracle):
    """Only where the process sees two GPUs (the build box has one: skipped there).  A placed (VMM) batch on device 0
    sharded over devices {0, 1} -- peer copies over xGMI through the granted mapping -- and shards resident on devices
    0 and 1, both bit-compared with the single-GPU call."""
    if dwt.device_count() < 2:
        pytest.skip("needs two GPUs in one process")
    L = dwt.lib
    nb, h, w, J = 6, 300, 1030, 3
    rng = np.random.default_rng(31)
    imgs = rng.random((nb, h, w), dtype=np.float32)
    want = imgs.copy()
    for k in range(nb):
        oracle.fwd("cdf97_2f_s", want[k], J)
    dwt.set_option("place_min_mib", 0)
    dwt.set_option("place_max_gib", 24)
    try:
        src, dst = dwt.alloc_batch("cdf97_s", nb, w, h, J)
        assert L.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
        got = np.empty_like(imgs)
        assert dwt.transform2d_batch_sharded("cdf97_s", 0, src, dst, h * w * 4, nb, w * 4, w, h, J, [0, 1]) == J
        assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
        assert np.array_equal(bits(got), bits(want))
        L.dwt_hip_free(src)
        L.dwt_hip_free(dst)
    finally:
        dwt.set_option("place_min_mib", 1024)
        dwt.set_option("place_max_gib", 0)
    # shards resident on devices 0 and 1 (allocated by threads bound to them)
    import threading

    halves = [np.ascontiguousarray(imgs[:3]), np.ascontiguousarray(imgs[3:])]
    ptrs = [[None, None], [None, None]]

    def alloc(d):
        dwt.set_device(d)
        ptrs[d][0] = L.dwt_hip_malloc(halves[d].nbytes)
        ptrs[d][1] = L.dwt_hip_malloc(halves[d].nbytes)
        assert L.dwt_hip_memcpy_h2d(ptrs[d][0], halves[d].ctypes.data, halves[d].nbytes) == 0

    for d in range(2):
        th = threading.Thread(target=alloc, args=(d,))
        th.start()
        th.join()
    assert dwt.transform2d_batch_multi("cdf97_s", 0, [ptrs[0][0], ptrs[1][0]], [ptrs[0][1], ptrs[1][1]], [3, 3], [0, 1], h * w * 4, w * 4, w, h, J) == J
    got = np.empty_like(imgs)

    def fetch(d):
        dwt.set_device(d)
        assert L.dwt_hip_memcpy_d2h(got[3 * d:3 * d + 3].ctypes.data, ptrs[d][1], halves[d].nbytes) == 0
        L.dwt_hip_free(ptrs[d][0])
        L.dwt_hip_free(ptrs[d][1])

    for d in range(2):
        th = threading.Thread(target=fetch, args=(d,))
        th.start()
        th.join()
    assert np.array_equal(bits(got), bits(want))


def test_bench_distributed_branch_with_one_rank_over_rccl():
    """The one-GPU rehearsal of what the first real N > 1 run executes for the first time: `bench.py --force-dist`
    takes the torch.distributed branch with WORLD_SIZE = 1 -- the gloo + RCCL group is created ("cpu:gloo,cuda:nccl"),
    RCCL's first all_reduce runs on a side stream under its deadline, the barrier / max of the timing go over RCCL
    device tensors, per_rank is gathered, and the batch split sends the batch rank 0 -> rank 0 and back through
    RCCL's grouped send / recv.  The line must say control_plane "rccl" and carry an intact split."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--images", "4", "--size", "2048",
                          "--steps", "2", "--warmup", "1", "--placements", "1", "--no-cpu", "--no-sweep", "--no-single", "--timeout", "240",
                          "--pg-timeout", "120", "--split-timeout", "60"], env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and lines, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads(lines[-1])
    assert line["control_plane"] == "rccl", line["control_plane"]
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["per_rank"] and line["per_rank"][0]["rank"] == 0 and line["per_rank"][0]["images"] == 4
    split = line["batch_split"]
    assert "error" not in split and split["round_trip_intact"] is True and split["scatter_ms"] > 0 and split["gather_ms"] > 0, split
    assert split["bytes_each_way"] == 4 * 2048 * 2048 * 4


def test_placed_batches_allocated_and_freed_repeatedly(dwt, oracle):
    """dwt_hip_alloc_batch / dwt_hip_free cycles: every arena gets virtual addresses this process has not used before.
    hipMemAddressReserve returns a freed range again, and a range reserved again and mapped to OTHER physical chunks was
    accessed through stale translations (kernels wrote the old chunks, copies read the new ones: wrong results from the
    second cycle on -- round 6, scripts/r06/sharded_stress.py).  Eight cycles, both wavelets, one slot and three."""
    L = dwt.lib
    nb, h, w, J = 7, 260, 520, 3
    try:
        dwt.set_option("place_min_mib", 0)
        dwt.set_option("place_tries", 2)
        dwt.set_option("place_max_gib", 16)
        bases = set()
        for rnd in range(8):
            wname = ("cdf97_s", "cdf53_i")[rnd & 1]
            ff, fi, dt = NAMES[wname]
            imgs = rand(np.random.default_rng(900 + rnd), (nb, h, w), dt)
            want = imgs.copy()
            for k in range(nb):
                oracle.fwd(ff, want[k], J)
            dwt.dwt_util_finish()
            src, dst = dwt.alloc_batch(wname, nb, w, h, J)
            assert dwt.alloc_batch_note() == "", dwt.alloc_batch_note()
            assert src not in bases, "an address range was handed out twice"
            bases.add(src)
            assert L.dwt_hip_memcpy_h2d(src, imgs.ctypes.data, imgs.nbytes) == 0
            got = np.empty_like(imgs)
            for devices in ([0], [0, 0, 0]):
                assert dwt.transform2d_batch_sharded(wname, 0, src, dst, h * w * 4, nb, w * 4, w, h, J, devices) == J
                assert L.dwt_hip_memcpy_d2h(got.ctypes.data, dst, got.nbytes) == 0
                assert np.array_equal(bits(got), bits(want)), (rnd, wname, devices)
            L.dwt_hip_free(src)
            L.dwt_hip_free(dst)
    finally:
        dwt.set_option("place_min_mib", 1024)
        dwt.set_option("place_tries", 4)
        dwt.set_option("place_max_gib", 0)
