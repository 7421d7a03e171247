"""GPU parity of the 3-D path (BASELINE config 5): single level against the golden
vectors of cdf97_3f_ip_sep_horizontal_s / cdf97_3i_ip_sep_horizontal_s, multi-level
against the oracle applied on the LLL lattice (strides doubled per level)."""
import numpy as np
import pytest

from conftest import bits, golden_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dwt():
    import libdwt_amd as d

    d.dwt_util_init()
    yield d
    d.dwt_util_finish()


class DevVol:
    def __init__(self, dwt, arr):
        self.dwt = dwt
        self.shape = arr.shape
        self.nbytes = arr.nbytes
        self.ptr = dwt.lib.dwt_hip_malloc(arr.nbytes)
        assert self.ptr
        a = np.ascontiguousarray(arr)
        assert dwt.lib.dwt_hip_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes) == 0

    def get(self):
        out = np.empty(self.shape, np.float32)
        assert self.dwt.lib.dwt_hip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes) == 0
        return out

    def run(self, inverse, levels):
        nz, ny, nx = self.shape
        self.dwt.transform3d(inverse, self.ptr, nx * 4, nx * ny * 4, nx, ny, nz, levels)

    def free(self):
        self.dwt.lib.dwt_hip_free(self.ptr)


def oracle_multilevel(oracle, v, levels, inverse):
    order = range(levels - 1, -1, -1) if inverse else range(levels)
    for j in order:
        s = 1 << j
        view = v[::s, ::s, ::s]
        oracle.vol("cdf97_3i_s" if inverse else "cdf97_3f_s", view)
    return v


def test_golden_single_level(dwt):
    for meta, src, fwd, inv in golden_cases("cdf97_3d_s"):
        d = DevVol(dwt, src)
        d.run(0, 1)
        got = d.get()
        assert np.array_equal(bits(got), bits(fwd)), meta
        d.run(1, 1)
        assert np.array_equal(bits(d.get()), bits(inv)), meta
        d.free()


@pytest.mark.parametrize("shape,levels", [((16, 16, 16), 1), ((33, 65, 129), 1), ((40, 100, 300), 1), ((64, 64, 64), 3),
                                          ((36, 52, 40), 2), ((128, 128, 512), 3), ((256, 256, 256), 3)],
                         ids=lambda v: str(v))
def test_volume_vs_oracle(dwt, oracle, shape, levels):
    rng = np.random.default_rng(sum(shape) + levels)
    vol = rng.random(shape, dtype=np.float32)
    want = oracle_multilevel(oracle, vol.copy(), levels, False)
    d = DevVol(dwt, vol)
    d.run(0, levels)
    got = d.get()
    assert np.array_equal(bits(got), bits(want))
    d.run(1, levels)
    rec = d.get()
    want_rec = oracle_multilevel(oracle, want.copy(), levels, True)
    assert np.array_equal(bits(rec), bits(want_rec))
    assert np.abs(rec - vol).max() < 1e-4
    d.free()
