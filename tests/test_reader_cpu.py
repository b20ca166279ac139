"""numpy_reader / staging ring / GPSRODataset on the host-only ring (device = -1).
Mirrors the reference's reader tests (src/numpy_reader/tests/reader_test.py):
the oracle is np.load; fixtures are regenerated here with numpy (v1/v2/v3
headers, <f4 <f8 <i8 <i4, C and Fortran order, scalar, truncated payload,
bad magic, corrupt header, unsupported dtype)."""
import os

import numpy as np
import pytest
import torch

import bias_gan_amd  # noqa: F401
from bias_gan_amd.data import numpy_reader as nr
from bias_gan_amd.data.gpsro_dataset import GPSRODataset


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("npy")
    rng = np.random.default_rng(0)
    out = {}

    def save(name, arr, version=None):
        p = os.path.join(d, name)
        if version is None:
            np.save(p, arr)
        else:
            with open(p, "wb") as f:
                np.lib.format.write_array(f, arr, version=version)
        out[name] = p

    save("arr_1d_rm.npy", rng.standard_normal(10).astype("<f4"))
    save("arr_2d_rm.npy", rng.standard_normal((4, 5)).astype("<f8"))
    save("arr_2d_cm.npy", np.asfortranarray(rng.standard_normal((3, 5)).astype("<f4")))
    save("arr_3d_rm.npy", rng.integers(-9, 9, (2, 6, 3)).astype("<i8"))
    save("arr_3d_cm.npy", np.asfortranarray(rng.integers(-9, 9, (2, 6, 3)).astype("<i4")))
    save("arr_one_rm.npy", np.array([3.5], dtype="<f4"))
    save("arr_scalar.npy", np.array(2.25, dtype="<f8"))
    save("arr_v2_rm.npy", rng.standard_normal((7, 3)).astype("<f4"), version=(2, 0))
    save("arr_v3_rm.npy", rng.standard_normal((5,)).astype("<f8"), version=(3, 0))
    save("arr_big_rm.npy", rng.standard_normal((16, 33, 17)).astype("<f4"))
    save("arr_1d_dtype_fail.npy", (rng.standard_normal(4) + 1j).astype("<c16"))
    good = open(out["arr_1d_rm.npy"], "rb").read()
    for name, blob in (("arr_1d_format_fail.npy", b"this is not a numpy file at all, just text\n" * 3),
                       ("arr_1d_header_fail.npy", good[8:]),                 # magic + version bytes lost
                       ("arr_1d_corruption_fail.npy", good[:-12])):          # payload 12 bytes short
        p = os.path.join(d, name)
        open(p, "wb").write(blob)
        out[name] = p
    return out


GOOD = ["arr_1d_rm.npy", "arr_2d_rm.npy", "arr_2d_cm.npy", "arr_3d_rm.npy", "arr_3d_cm.npy", "arr_one_rm.npy",
        "arr_scalar.npy", "arr_v2_rm.npy", "arr_v3_rm.npy", "arr_big_rm.npy"]


@pytest.mark.parametrize("nintra", [1, 2, 4, 8])
@pytest.mark.parametrize("name", GOOD)
def test_single_sample_loads(files, name, nintra):
    r = nr.numpy_reader(False, -1)
    r.num_intra_threads = nintra
    r.parse(files[name])
    r.init_file(files[name])
    t = r.get_sample(0)
    r.finalize_file()
    np.testing.assert_array_equal(np.load(files[name]).reshape(t.shape), t.numpy())   # bit-exact


@pytest.mark.parametrize("nintra", [1, 4])
@pytest.mark.parametrize("name", [n for n in GOOD if "_rm" in n])
def test_multi_sample_loads_row_major(files, name, nintra):
    arr = np.load(files[name])
    r = nr.numpy_reader(True, -1)
    r.num_intra_threads = nintra
    r.parse(files[name])
    r.init_file(files[name])
    assert r.num_samples == arr.shape[0] and r.shape == (list(arr.shape[1:]) or [1])
    for s in range(arr.shape[0]):
        np.testing.assert_array_equal(arr[s].reshape(r.shape), r.get_sample(s).numpy())
    if arr.shape[0] >= 3:
        r.set_batchsize(3)
        b = r.get_batch([2, 0, 1])
        np.testing.assert_array_equal(arr[[2, 0, 1]].reshape(b.shape), b.numpy())
        with pytest.raises(RuntimeError, match="batchsize"):
            r.get_batch([0])
        with pytest.raises(RuntimeError, match="getBatch"):
            r.get_sample(0)
    with pytest.raises(IndexError):
        r.set_batchsize(arr.shape[0] + 1)
    r.finalize_file()


def test_error_paths(files):
    with pytest.raises(RuntimeError, match="reading column-major arrays"):
        nr.numpy_reader(True, -1).parse(files["arr_2d_cm.npy"])
    with pytest.raises(RuntimeError, match="unsupported datatype"):
        nr.numpy_reader(True, -1).parse(files["arr_1d_dtype_fail.npy"])
    with pytest.raises(RuntimeError, match="failed to open file"):
        nr.numpy_reader(True, -1).parse(os.path.join(os.path.dirname(files["arr_1d_rm.npy"]), "never_existed.npy"))
    with pytest.raises(RuntimeError, match="ill formatted or corrupt"):
        nr.numpy_reader(True, -1).parse(files["arr_1d_header_fail.npy"])
    with pytest.raises(RuntimeError, match="not a numpy file"):
        nr.numpy_reader(True, -1).parse(files["arr_1d_format_fail.npy"])
    with pytest.raises(IndexError, match="batch size"):   # std::out_of_range in the reference (numpy_reader.cpp:232)
        r = nr.numpy_reader(False, -1)
        r.parse(files["arr_2d_rm.npy"])
        r.set_batchsize(2)


@pytest.mark.parametrize("nintra", [1, 2, 4, 8])
def test_corrupted_payload_is_an_error_not_a_hang(files, nintra):
    r = nr.numpy_reader(False, -1)
    r.num_intra_threads = nintra
    r.parse(files["arr_1d_corruption_fail.npy"])
    r.init_file(files["arr_1d_corruption_fail.npy"])
    with pytest.raises(IndexError, match="file corruption"):
        r.get_sample(0)


def test_prefetch_ring(files):
    r = nr.numpy_reader(False, -1, ring_slots=2)
    r.prefetch(files["arr_big_rm.npy"])
    r.prefetch(files["arr_big_rm.npy"])
    with pytest.raises(RuntimeError, match="in flight"):
        r.prefetch(files["arr_big_rm.npy"])
    a = r.get_prefetched()
    r.prefetch(files["arr_big_rm.npy"])      # the freed slot is reusable
    b, c = r.get_prefetched(), r.get_prefetched()
    ref = np.load(files["arr_big_rm.npy"])
    for t in (a, b, c):
        np.testing.assert_array_equal(ref, t.numpy())
    with pytest.raises(RuntimeError, match="nothing was prefetched"):
        r.get_prefetched()


def _make_dataset(root, n=8, c=3, h=5, w=7):
    rng = np.random.default_rng(1)
    os.makedirs(root, exist_ok=True)
    for i in range(n):
        np.save(os.path.join(root, f"data_in_{i:03d}.npy"), rng.standard_normal((c, h, w)).astype(np.float32))
        np.save(os.path.join(root, f"data_out_{i:03d}.npy"), rng.standard_normal((c, h, w)).astype(np.float32))
        np.save(os.path.join(root, f"masks_{i:03d}.npy"), (rng.random((c, h, w)) > 0.5).astype(np.float32))
    stats = {k: rng.random(c).astype(np.float32) for k in ("data_minval", "label_minval", "data_mean", "label_mean")}
    stats.update({"data_maxval": stats["data_minval"] + 1.5, "label_maxval": stats["label_minval"] + 2.0,
                  "data_sqmean": stats["data_mean"] ** 2 + 0.7, "label_sqmean": stats["label_mean"] ** 2 + 0.3})
    np.savez(os.path.join(root, "stats.npz"), **stats)
    return stats


@pytest.mark.parametrize("norm", ["MinMax", "MeanVariance"])
def test_gpsro_dataset_contract(tmp_path, norm):
    root = str(tmp_path / "train")
    stats = _make_dataset(root)
    ch = [0, 1, 2]   # the reference indexes only the STATISTICS by `channels` (gpsro_dataset.py:93-111), so they must cover the file
    full = GPSRODataset(root, os.path.join(root, "stats.npz"), ch, normalization_type=norm, shuffle=True, masks=True)
    # sharding: contiguous len//shard_num slices of ONE seeded shuffle, identical on every rank (gpsro_dataset.py:24-33)
    order = sorted(f"{i:03d}.npy" for i in range(8))
    np.random.RandomState(12345).shuffle(order)
    assert full.allfiles == order and len(full) == 8
    for num in (2, 8):
        got = []
        for r in range(num):
            ds = GPSRODataset(root, os.path.join(root, "stats.npz"), ch, normalization_type=norm, shuffle=True, masks=True,
                              shard_idx=r, shard_num=num)
            assert len(ds) == 8 // num
            got += ds.files
        assert got == order
    for idx in (0, 1, 5, 2):            # sequential (prefetch hit) and random (prefetch miss) access
        data, label, mask, fname = full[idx]
        x = np.load(os.path.join(root, "data_in_" + fname))
        y = np.load(os.path.join(root, "data_out_" + fname))
        if norm == "MinMax":
            xs, xsc = stats["data_minval"][ch], 1.0 / (stats["data_maxval"][ch] - stats["data_minval"][ch])
            ys, ysc = stats["label_minval"][ch], 1.0 / (stats["label_maxval"][ch] - stats["label_minval"][ch])
        else:
            xs, xsc = stats["data_mean"][ch], 1.0 / np.sqrt(stats["data_sqmean"][ch] - stats["data_mean"][ch] ** 2)
            ys, ysc = stats["label_mean"][ch], 1.0 / np.sqrt(stats["label_sqmean"][ch] - stats["label_mean"][ch] ** 2)
        # the reference broadcasts [len(channels),1,1] statistics against the [C,H,W] sample: needs len(channels)
        # == C or 1; use the full-channel case for the value check
        assert tuple(full.shapes[0]) == x.shape and fname == full.files[idx]
        np.testing.assert_array_equal(mask.numpy(), np.load(os.path.join(root, "masks_" + fname)))
    allc = GPSRODataset(root, os.path.join(root, "stats.npz"), [0, 1, 2], normalization_type=norm, shuffle=False)
    data, label, fname = allc[3]
    x = np.load(os.path.join(root, "data_in_" + fname)); y = np.load(os.path.join(root, "data_out_" + fname))
    if norm == "MinMax":
        ex = (x - stats["data_minval"][:, None, None]) / (stats["data_maxval"] - stats["data_minval"])[:, None, None]
        ey = (y - stats["label_minval"][:, None, None]) / (stats["label_maxval"] - stats["label_minval"])[:, None, None]
    else:
        ex = (x - stats["data_mean"][:, None, None]) / np.sqrt(stats["data_sqmean"] - stats["data_mean"] ** 2)[:, None, None]
        ey = (y - stats["label_mean"][:, None, None]) / np.sqrt(stats["label_sqmean"] - stats["label_mean"] ** 2)[:, None, None]
    np.testing.assert_allclose(data.numpy(), ex, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(label.numpy(), ey, rtol=1e-5, atol=1e-6)
    # through a DataLoader, like train_gan.py:199
    dl = torch.utils.data.DataLoader(allc, 2, drop_last=True)
    batches = list(dl)
    assert len(batches) == 4 and batches[0][0].shape == (2, 3, 5, 7)
