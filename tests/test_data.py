"""Data path (v1t_amd/data.py): the on-disk Sensorium / Franke layout -> batches, against the reference's MiceDataset +
DataLoader output (golden G10, oracle/gen_golden.py: gen_data) on the recording oracle/fake_sensorium.py writes."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import fake_sensorium as FS

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g10_data.npz")
CASES = [("sensorium", "A", (1, 12, 16), False), ("franke2022", "F", (2, 6, 8), True)]


def _args(ds_name, gray):
    return SimpleNamespace(ds_name=ds_name, behavior_mode=3, seed=1, gray_scale=gray, verbose=0, limit_data=None, batch_size=4)


@pytest.fixture(scope="module")
def root(tmp_path_factory):
    r = str(tmp_path_factory.mktemp("recordings"))
    for ds_name, mouse, shape, _ in CASES:
        FS.write_fake_mouse(r, ds_name, mouse, seed=3, trials=23, image_shape=shape, neurons=9)
    return r


@pytest.mark.parametrize("ds_name,mouse,shape,gray", CASES)
def test_dataset_host_path_vs_reference_golden(root, ds_name, mouse, shape, gray):
    """MiceDataset: tiers, ids, response precision and the per-trial host transforms (no GPU involved)."""
    from v1t_amd.data import MiceDataset

    g = np.load(GOLD)
    for tier in ("train", "validation", "test"):
        ds = MiceDataset(_args(ds_name, gray), tier=tier, data_dir=root, mouse_id=mouse)
        tag = f"g10/{ds_name}/{tier}"
        assert len(ds) == int(g[f"{tag}/n"]) and tuple(ds.image_shape) == tuple(g[f"{tag}/image_shape"])
        np.testing.assert_array_equal(ds._response_precision, g[f"{tag}/precision"])
        for i in range(len(ds)):
            item = ds[i]
            for k in ("image", "response", "behavior", "pupil_center"):
                np.testing.assert_allclose(np.asarray(item[k], dtype=np.float32), g[f"{tag}/{k}"][i], rtol=1e-6, atol=1e-6)
            assert int(item["image_id"]) == int(g[f"{tag}/image_id"][i]) and int(item["trial_id"]) == int(g[f"{tag}/trial_id"][i])
            assert item["mouse_id"] == mouse
    with pytest.raises(FileNotFoundError):
        MiceDataset(_args(ds_name, gray), tier="train", data_dir=os.path.join(root, "nowhere"), mouse_id=mouse)


def test_cycle_dataloaders_and_micro_batching():
    from v1t_amd.data import CycleDataloaders, micro_batching

    ds = {"A": [1, 2, 3], "B": ["x"]}
    seq = list(CycleDataloaders(ds))
    assert len(seq) == 6 == len(CycleDataloaders(ds))
    assert seq == [("A", 1), ("B", "x"), ("A", 2), ("B", "x"), ("A", 3), ("B", "x")]
    b = {"image": torch.arange(7), "response": torch.arange(7) * 2}
    parts = list(micro_batching(b, 3))
    assert [len(p["image"]) for p in parts] == [3, 3, 1] and torch.equal(torch.cat([p["response"] for p in parts]), b["response"])


@pytest.mark.gpu
@pytest.mark.parametrize("ds_name,mouse,shape,gray", CASES)
def test_device_loaders_vs_reference_golden(root, ds_name, mouse, shape, gray):
    """get_training_ds -> DeviceLoader batches (packed store in HBM + v1t_gather_transform) == the reference's collated
    batches: images kept as uint8 (Sensorium) or fp32 + colour-to-gray (Franke); ragged last batch; shuffle = a permutation."""
    from v1t_amd.data import get_training_ds

    dev = torch.device("cuda:0")
    g = np.load(GOLD)
    args = _args(ds_name, gray)
    train, val, test = get_training_ds(args, data_dir=root, mouse_ids=[mouse], batch_size=4, device=dev)
    assert args.output_shapes == {mouse: (9,)} and tuple(args.input_shape) == tuple(g[f"g10/{ds_name}/train/image_shape"])
    for tier, loaders in (("validation", val), ("test", test)):
        tag = f"g10/{ds_name}/{tier}"
        batches = list(loaders[mouse])
        assert len(batches) == len(loaders[mouse]) == -(-int(g[f"{tag}/n"]) // 4)
        for k in ("image", "response", "behavior", "pupil_center"):
            got = torch.cat([b[k] for b in batches]).cpu().numpy()
            assert got.dtype == np.float32 and got.shape == g[f"{tag}/{k}"].shape
            np.testing.assert_allclose(got, g[f"{tag}/{k}"], rtol=1e-6, atol=1e-6)
        assert np.array_equal(torch.cat([b["image_id"] for b in batches]).numpy(), g[f"{tag}/image_id"])
        assert np.array_equal(torch.cat([b["trial_id"] for b in batches]).numpy(), g[f"{tag}/trial_id"])
        assert batches[0]["mouse_id"] == [mouse] * len(batches[0]["image"])
    # shuffled training loader: every epoch is a permutation of the golden rows, two epochs differ
    tag = f"g10/{ds_name}/train"
    ep = [torch.cat([b["trial_id"] for b in train[mouse]]).numpy() for _ in range(2)]
    assert sorted(ep[0]) == sorted(g[f"{tag}/trial_id"]) == sorted(ep[1]) and not np.array_equal(ep[0], ep[1])
    rows = {int(t_): i for i, t_ in enumerate(g[f"{tag}/trial_id"])}
    b0 = next(iter(train[mouse]))
    want = g[f"{tag}/response"][[rows[int(t_)] for t_ in b0["trial_id"]]]
    np.testing.assert_allclose(b0["response"].cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    from v1t_amd.data import MouseStore
    store = MouseStore.get(train[mouse].dataset.mouse_dir, dev)
    assert store.data["image"].dtype == (torch.uint8 if ds_name == "sensorium" else torch.float32)
