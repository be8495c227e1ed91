"""The library's own device-wide primitives (deftet_amd/csrc/prims.hpp through deftet_radix_sort / deftet_scan):
stable LSD radix sort and prefix scans against numpy, at sizes around the tile boundaries and at operator scale."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# 512 tiles of 2,048 keys is where the sort changes its pass structure (two launches per pass with the scatter summing the
# tile-major table itself below, hist + scan + scatter above): both sides of 1,048,576 and a size well above it
SIZES = [0, 1, 63, 64, 65, 2047, 2048, 2049, 4096 + 17, 100_003, 1_000_000, 512 * 2048, 512 * 2048 + 1, 2_500_003]


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kdt,vdt,bits", [(np.int32, None, 31), (np.int32, np.int32, 20), (np.int64, np.int32, 63), (np.int64, np.int64, 40),
                                         (np.int32, np.int64, 9), (np.int64, None, 17)])
def test_radix_sort_equals_numpy_stable_sort(cuda, n, kdt, vdt, bits):
    from deftet_amd import hip_ops
    rng = np.random.default_rng(n * 7 + bits)
    hi = min(bits, 62)
    # few distinct keys in the low digits -> long runs of equal keys: stability is visible in the values
    keys = (rng.integers(0, 1 << hi, n, dtype=np.int64) >> rng.integers(0, hi, n)).astype(kdt)
    vals = np.arange(n).astype(vdt) if vdt is not None else None
    k = torch.from_numpy(keys).to(cuda)
    v = torch.from_numpy(vals).to(cuda) if vals is not None else None
    ko, vo = hip_ops.radix_sort(k, v, bits=bits)
    mask = (1 << bits) - 1
    order = np.argsort(keys.astype(np.int64) & mask, kind="stable")
    assert np.array_equal(ko.cpu().numpy(), keys[order])
    if vals is not None:
        assert np.array_equal(vo.cpu().numpy(), vals[order])
    assert torch.equal(k, torch.from_numpy(keys).to(cuda))          # the input is not modified


def test_radix_sort_ignores_bits_above_the_requested_ones_and_honours_a_device_side_count(cuda):
    from deftet_amd import hip_ops
    rng = np.random.default_rng(3)
    n, used = 50_000, 31_337
    keys = rng.integers(0, 1 << 30, n, dtype=np.int64).astype(np.int32)
    vals = rng.standard_normal(n).astype(np.float32)
    k, v = torch.from_numpy(keys).to(cuda), torch.from_numpy(vals).to(cuda)
    ko, vo = hip_ops.radix_sort(k, v, bits=12)                       # only the low 12 bits order the output
    order = np.argsort(keys & 0xFFF, kind="stable")
    assert np.array_equal(ko.cpu().numpy(), keys[order]) and np.array_equal(vo.cpu().numpy(), vals[order])
    cnt = torch.tensor([used], device=cuda, dtype=torch.int32)
    ko, vo = hip_ops.radix_sort(k, v, bits=30, n_valid=cnt)          # the tail beyond the count is not touched / defined
    order = np.argsort(keys[:used], kind="stable")
    assert np.array_equal(ko.cpu().numpy()[:used], keys[:used][order]) and np.array_equal(vo.cpu().numpy()[:used], vals[:used][order])
    # the same with a capacity above 512 tiles (the scan form of the pass) of which only a few tiles hold keys
    n2, used2 = 1_300_000, 5_001
    keys2 = rng.integers(0, 1 << 19, n2, dtype=np.int64).astype(np.int32)
    k2 = torch.from_numpy(keys2).to(cuda)
    v2 = torch.arange(n2, device=cuda, dtype=torch.int32)
    ko, vo = hip_ops.radix_sort(k2, v2, bits=19, n_valid=torch.tensor([used2], device=cuda, dtype=torch.int32))
    order = np.argsort(keys2[:used2], kind="stable")
    assert np.array_equal(ko.cpu().numpy()[:used2], keys2[:used2][order]) and np.array_equal(vo.cpu().numpy()[:used2], order.astype(np.int32))


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("dt", [np.int32, np.int64])
def test_scans_equal_numpy(cuda, n, dt):
    from deftet_amd import hip_ops
    rng = np.random.default_rng(n + 1)
    x = rng.integers(-50, 1000, n).astype(dt)
    t = torch.from_numpy(x).to(cuda)
    incl = np.cumsum(x, dtype=dt)
    assert np.array_equal(hip_ops.scan(t, "inclusive").cpu().numpy(), incl)
    assert np.array_equal(hip_ops.scan(t, "exclusive").cpu().numpy(), (incl - x).astype(dt))
    assert np.array_equal(hip_ops.scan(t, "max").cpu().numpy(), np.maximum.accumulate(x) if n else x)


def test_bad_arguments_are_rejected(cuda):
    from deftet_amd import hip_ops, _lib
    k = torch.arange(10, device=cuda, dtype=torch.int32)
    with pytest.raises(RuntimeError):
        hip_ops.radix_sort(k, bits=33)
    with pytest.raises(RuntimeError):
        hip_ops.radix_sort(k.float())
    with pytest.raises(RuntimeError):
        hip_ops.radix_sort(k, torch.zeros(9, device=cuda, dtype=torch.int32))
    assert issubclass(_lib.DefTetHipError, RuntimeError)
