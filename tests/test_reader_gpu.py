"""Staging ring with a device: file -> pinned host -> HBM on the ring's copy stream,
consumer stream ordered by events (no host synchronisation in the reader)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import bias_gan_amd  # noqa: E402,F401
from bias_gan_amd.data import numpy_reader as nr  # noqa: E402
from bias_gan_amd.data.gpsro_dataset import GPSRODataset  # noqa: E402


def test_device_ring_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    paths = []
    for i in range(5):
        p = str(tmp_path / f"f{i}.npy")
        np.save(p, rng.standard_normal((16, 96, 64)).astype(np.float32) + i)
        paths.append(p)
    r = nr.numpy_reader(False, 0, ring_slots=3)
    r.num_intra_threads = 4
    r.parse(paths[0])
    r.init_file(paths[0])
    t = r.get_sample(0)
    assert t.is_cuda and t.dtype == torch.float32
    np.testing.assert_array_equal(t.cpu().numpy(), np.load(paths[0]))
    # run ahead: two files in flight while the "step" consumes the previous one
    r.prefetch(paths[1]); r.prefetch(paths[2])
    side = torch.cuda.Stream()
    for k in (1, 2, 3, 4):
        with torch.cuda.stream(side):           # consumer on a non-default stream
            x = r.get_prefetched()
            y = x * 2.0
        if k + 2 < len(paths):
            r.prefetch(paths[k + 2])
        side.synchronize()
        np.testing.assert_array_equal(y.cpu().numpy(), 2.0 * np.load(paths[k]))
    # split-axis batches on the device
    rb = nr.numpy_reader(True, 0)
    rb.parse(paths[0]); rb.init_file(paths[0]); rb.set_batchsize(4)
    b = rb.get_batch([3, 1, 15, 0])
    np.testing.assert_array_equal(b.cpu().numpy(), np.load(paths[0])[[3, 1, 15, 0]])


def test_dataset_on_device(tmp_path):
    from test_reader_cpu import _make_dataset
    root = str(tmp_path / "train")
    _make_dataset(root, n=6, c=4, h=19, w=37)
    dev = torch.device("cuda", 0)
    ds = GPSRODataset(root, os.path.join(root, "stats.npz"), [0, 1, 2, 3], normalization_type="MeanVariance", shuffle=True,
                      masks=True, read_device=dev, send_device=dev, num_intra_threads=2)
    cpu = GPSRODataset(root, os.path.join(root, "stats.npz"), [0, 1, 2, 3], normalization_type="MeanVariance", shuffle=True,
                       masks=True)
    for i in range(len(ds)):
        a, b = ds[i], cpu[i]
        assert a[0].is_cuda and a[3] == b[3]
        for u, v in zip(a[:3], b[:3]):
            np.testing.assert_allclose(u.cpu().numpy(), v.numpy(), rtol=1e-6, atol=1e-7)


def test_slot_reuse_with_large_samples(tmp_path):
    """One ring slot, 64 MB samples, back-to-back get_sample() and a batch larger than the ring: the pinned host half of
    a slot must not be overwritten by the next read while the H2D copy of the previous ticket still reads it
    (bg_ring_release waits for that copy's event before the slot changes hands)."""
    rng = np.random.default_rng(7)
    nel = 16 * 1024 * 1024                      # 64 MB of float32 per sample
    paths, arrs = [], []
    for i in range(3):
        a = rng.integers(0, 1 << 20, size=nel, dtype=np.int32).astype(np.float32)
        p = str(tmp_path / f"big{i}.npy")
        np.save(p, a.reshape(4, 2048, 2048))
        paths.append(p)
        arrs.append(a.reshape(4, 2048, 2048))
    r = nr.numpy_reader(False, 0, ring_slots=1)
    r.num_intra_threads = 4
    r.parse(paths[0])
    got = []
    for rep in range(2):
        for p in paths:                         # no host synchronisation between the reads
            r.init_file(p)
            got.append(r.get_sample(0))
    torch.cuda.synchronize()
    for k, t in enumerate(got):
        assert torch.equal(t.cpu(), torch.from_numpy(arrs[k % 3])), f"sample {k} corrupted"
    rb = nr.numpy_reader(True, 0, ring_slots=1)
    rb.num_intra_threads = 4
    rb.parse(paths[1]); rb.init_file(paths[1]); rb.set_batchsize(4)
    b = rb.get_batch([3, 0, 2, 1])              # 4 x 16 MB through one slot
    assert torch.equal(b.cpu(), torch.from_numpy(arrs[1][[3, 0, 2, 1]]))
