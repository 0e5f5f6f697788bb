"""Sensorium / Franke on-disk format -> packed per-mouse store in HBM -> device batches.

Host-side mirror of src/v1t/data.py with the same names, arguments and batch dictionaries:
- `MiceDataset`       <- data.py:275-434 (tiers, statistics, response precision, transforms, `__getitem__`)
- `get_training_ds`   <- data.py:437-491 (returns loaders; sets args.output_shapes / args.input_shape)
- `CycleDataloaders`  <- data.py:71-103, `micro_batching` <- data.py:106-110
The reference reads 4 `.npy` files per trial per epoch through torch DataLoader workers. Here a mouse's recording is read
ONCE (`MouseStore`) into packed `[trials][...]` arrays in HBM, and `DeviceLoader` builds every batch with one gather +
standardise launch per field (`v1t_gather_transform`), so nothing but a few hundred index bytes crosses PCIe per step.
`MiceDataset.__getitem__` keeps the reference's per-trial host path (numpy, same arithmetic) for code that indexes the
dataset directly.
"""
from __future__ import annotations

import os
import typing as t
from glob import glob

import numpy as np
import torch

from . import lib as L

# key - mouse ID, value - directory of the recording (the published file names of the two datasets; data.py:17-41)
SENSORIUM = {
    "S0": "static26872-17-20-GrayImageNet-94c6ff995dac583098847cfecd43e7b6",
    "S1": "static27204-5-13-GrayImageNet-94c6ff995dac583098847cfecd43e7b6",
    "A": "static21067-10-18-GrayImageNet-94c6ff995dac583098847cfecd43e7b6",
    "B": "static22846-10-16-GrayImageNet-94c6ff995dac583098847cfecd43e7b6",
    "C": "static23343-5-17-GrayImageNet-94c6ff995dac583098847cfecd43e7b6",
    "D": "static23656-14-22-GrayImageNet-94c6ff995dac583098847cfecd43e7b6",
    "E": "static23964-4-22-GrayImageNet-94c6ff995dac583098847cfecd43e7b6",
}
FRANKE2022 = {
    "F": "static25311-10-26-ColorImageNet-104e446ed0128d89c639eef0abe4655b",
    "G": "static25340-3-19-ColorImageNet-104e446ed0128d89c639eef0abe4655b",
    "H": "static25704-2-12-ColorImageNet-b23ac8521543becfd382e56c657ba29b",
    "I": "static25830-10-4-ColorImageNet-104e446ed0128d89c639eef0abe4655b",
    "J": "static26085-6-3-ColorImageNet-104e446ed0128d89c639eef0abe4655b",
    "K": "static26142-2-11-ColorImageNet-6a21297215f4dbb802554a60c0e72877",
    "L": "static26426-18-13-ColorImageNet-b23ac8521543becfd382e56c657ba29b",
    "M": "static26470-4-5-ColorImageNet-104e446ed0128d89c639eef0abe4655b",
    "N": "static26644-6-2-ColorImageNet-b23ac8521543becfd382e56c657ba29b",
    "O": "static26872-21-6-ColorImageNet-104e446ed0128d89c639eef0abe4655b",
}
FIELDS = {"image": "images", "response": "responses", "behavior": "behavior", "pupil_center": "pupil_center"}


def get_mouse2path(ds_name: str) -> t.Dict[str, str]:
    assert ds_name in ("sensorium", "franke2022")
    return SENSORIUM if ds_name == "sensorium" else FRANKE2022


def get_num_trials(mouse_dir: str) -> int:
    return len(glob(os.path.join(mouse_dir, "data", "images", "*.npy")))


def micro_batching(batch: t.Dict[str, t.Any], batch_size: int):
    """reference data.py:106-110"""
    for i in range(0, len(batch["image"]), batch_size):
        yield {k: v[i:i + batch_size] for k, v in batch.items()}


def load_mouse_metadata(ds_name: str, mouse_dir: str) -> t.Dict[str, t.Any]:
    """reference data.py:155-227 (without the optional timestamps and the unzip step)."""
    if not os.path.isdir(mouse_dir):
        raise FileNotFoundError(f"{mouse_dir} not found (unpack the dataset archive first)")
    meta = os.path.join(mouse_dir, "meta")
    neuron = lambda a: np.load(os.path.join(meta, "neurons", a))
    trial = lambda a: np.load(os.path.join(meta, "trials", a))
    stat = lambda a, b: np.load(os.path.join(meta, "statistics", a, "all", f"{b}.npy"))
    keys = ["min", "max", "median", "mean", "std"]
    md = {
        "mouse_dir": mouse_dir,
        "num_neurons": len(neuron("unit_ids.npy")),
        "neuron_ids": neuron("unit_ids.npy").astype(np.int32),
        "coordinates": neuron("cell_motor_coordinates.npy").astype(np.float32),
        "tiers": trial("tiers.npy"),
        "stats": {f: {k: stat(d, k) for k in keys} for f, d in FIELDS.items()},
        "image_ids": trial("frame_image_id.npy" if ds_name == "sensorium" else "colorframeprojector_image_id.npy"),
    }
    animal_ids = np.unique(neuron("animal_ids.npy"))
    assert len(animal_ids) == 1, f"Multiple animal ID in {meta}."
    md["animal_id"] = animal_ids[0]
    md["trial_ids"] = trial("trial_idx.npy")
    if np.issubdtype(md["trial_ids"].dtype, np.integer):
        md["trial_ids"] = md["trial_ids"].astype(np.int32)
    return md


class MouseStore:
    """All trials of one mouse as packed device arrays `[trials][...]` (image / response / behavior / pupil_center),
    read from the per-trial .npy files once. Images that are integral in [0, 255] are kept as uint8 (a quarter of the HBM
    footprint and of the gather traffic); everything else fp32. Shared by the train / validation / test datasets."""

    _cache: t.Dict[t.Tuple[str, str], "MouseStore"] = {}

    def __init__(self, mouse_dir: str, device: torch.device):
        n = get_num_trials(mouse_dir)
        if n == 0:
            raise FileNotFoundError(f"no trials under {mouse_dir}/data/images")
        self.device = torch.device(device)
        self.num_trials = n
        self.shapes: t.Dict[str, t.Tuple[int, ...]] = {}
        self.data: t.Dict[str, torch.Tensor] = {}
        for field, d in FIELDS.items():
            arr = np.stack([np.load(os.path.join(mouse_dir, "data", d, f"{i}.npy")) for i in range(n)], axis=0)
            self.shapes[field] = tuple(arr.shape[1:])
            if field == "image" and arr.min() >= 0 and arr.max() <= 255 and np.array_equal(arr, np.rint(arr)):
                arr = arr.astype(np.uint8)
            else:
                arr = arr.astype(np.float32)
            self.data[field] = torch.from_numpy(np.ascontiguousarray(arr.reshape(n, -1))).to(self.device)

    @classmethod
    def get(cls, mouse_dir: str, device: torch.device) -> "MouseStore":
        key = (os.path.abspath(mouse_dir), str(torch.device(device)))
        if key not in cls._cache:
            cls._cache[key] = MouseStore(mouse_dir, device)
        return cls._cache[key]


class MiceDataset:
    """reference data.py:275-434"""

    def __init__(self, args, tier: str, data_dir: str, mouse_id: str):
        assert tier in ("train", "validation", "test", "final_test")
        self.tier, self.mouse_id, self.ds_name = tier, mouse_id, args.ds_name
        assert self.ds_name in ("sensorium", "franke2022")
        mouse_dir = os.path.join(data_dir, get_mouse2path(self.ds_name)[mouse_id])
        md = load_mouse_metadata(self.ds_name, mouse_dir=mouse_dir)
        self.behavior_mode = args.behavior_mode
        if self.behavior_mode and mouse_id == "S0":
            raise ValueError("Mouse S0 does not have behaviour data.")
        self.mouse_dir, self.neuron_ids, self.coordinates, self.stats = md["mouse_dir"], md["neuron_ids"], md["coordinates"], md["stats"]
        indexes = np.where(md["tiers"] == tier)[0].astype(np.int32)
        if tier == "train" and getattr(args, "limit_data", None) and len(indexes) > args.limit_data:
            indexes = np.random.default_rng(seed=args.seed).choice(indexes, size=args.limit_data, replace=False)
        self.indexes = indexes
        self.image_ids = md["image_ids"][self.indexes]
        self.trial_ids = md["trial_ids"][self.indexes]
        self.compute_response_precision()
        self.hashed = self.ds_name == "sensorium" and mouse_id in ("S0", "S1")
        self.image_shape = tuple(np.load(os.path.join(mouse_dir, "data", "images", "0.npy")).shape)
        self.gray_scale = False
        if getattr(args, "gray_scale", False) and self.ds_name == "franke2022":
            self.gray_scale = True
            self.image_shape = (1,) + self.image_shape[1:]
        self._dev: t.Optional[t.Dict[str, t.Any]] = None

    def __len__(self):
        return len(self.indexes)

    image_stats = property(lambda self: self.stats["image"])
    response_stats = property(lambda self: self.stats["response"])
    behavior_stats = property(lambda self: self.stats["behavior"])
    pupil_stats = property(lambda self: self.stats["pupil_center"])
    num_neurons = property(lambda self: len(self.neuron_ids))

    def compute_response_precision(self):
        """1 / std per neuron where std > 1 % of the mean std, else 1 / threshold (data.py:387-397)"""
        std = self.response_stats["std"]
        threshold = 0.01 * np.mean(std)
        idx = std > threshold
        precision = np.ones_like(std) / threshold
        precision[idx] = 1 / std[idx]
        self._response_precision = precision

    # ---- host transforms (per trial, numpy), as the reference applies them in __getitem__
    def transform_image(self, image):
        image = (image - self.image_stats["mean"]) / self.image_stats["std"]
        return np.mean(image, axis=0, keepdims=True) if self.gray_scale else image

    def i_transform_image(self, image):
        if self.behavior_mode == 1:
            image = torch.unsqueeze(image[0], dim=0) if len(image.shape) == 3 else torch.unsqueeze(image[:, 0, :, :], dim=1)
        return (image * self.image_stats["std"]) + self.image_stats["mean"]

    def transform_pupil_center(self, x):
        return (x - self.pupil_stats["mean"]) / self.pupil_stats["std"]

    def i_transform_pupil_center(self, x):
        return (x * self.pupil_stats["std"]) + self.pupil_stats["mean"]

    def transform_behavior(self, x):
        return x / self.behavior_stats["std"]

    def i_transform_behavior(self, x):
        return x * self.behavior_stats["std"]

    def transform_response(self, x):
        return x * self._response_precision

    def i_transform_response(self, x):
        return x / self._response_precision

    def __getitem__(self, idx):
        trial = self.indexes[idx]
        load = lambda d: np.load(os.path.join(self.mouse_dir, "data", d, f"{trial}.npy")).astype(np.float32)
        return {
            "image": self.transform_image(load("images")), "response": self.transform_response(load("responses")),
            "behavior": self.transform_behavior(load("behavior")), "pupil_center": self.transform_pupil_center(load("pupil_center")),
            "image_id": self.image_ids[idx], "trial_id": self.trial_ids[idx], "mouse_id": self.mouse_id,
        }

    # ---- device path
    def _device_state(self, device: torch.device):
        if self._dev is None or self._dev["device"] != torch.device(device):
            store = MouseStore.get(self.mouse_dir, device)
            f32 = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))).to(device)
            # (sub, div, mul) per field; statistics are scalars or arrays of the field's shape (broadcast by period)
            tf = {"image": (f32(self.image_stats["mean"]), f32(self.image_stats["std"]), None),
                  "response": (None, None, f32(self._response_precision)),
                  "behavior": (None, f32(self.behavior_stats["std"]), None),
                  "pupil_center": (f32(self.pupil_stats["mean"]), f32(self.pupil_stats["std"]), None)}
            for field, (s_, d_, m_) in tf.items():
                e = store.data[field].shape[1]
                for a in (s_, d_, m_):
                    if a is not None and a.numel() not in (1, e):
                        raise RuntimeError(f"{field} statistics of {a.numel()} elements do not broadcast over {store.shapes[field]}")
            self._dev = {"device": torch.device(device), "store": store, "tf": tf,
                         "indexes": torch.from_numpy(self.indexes.astype(np.int32)).to(device)}
        return self._dev

    def device_batch(self, positions: torch.Tensor, device: torch.device) -> t.Dict[str, t.Any]:
        """positions: int tensor of dataset positions (0 .. len-1) -> the reference's collated batch, tensors in HBM."""
        st = self._device_state(device)
        store: MouseStore = st["store"]
        pos = positions.to(device=device, dtype=torch.long)
        trial = st["indexes"][pos].contiguous()
        b = int(trial.numel())
        lib = L.load()
        out: t.Dict[str, t.Any] = {}
        for field in FIELDS:
            src = store.data[field]
            e = src.shape[1]
            sub, div, mul = st["tf"][field]
            gray = store.shapes["image"][0] if (field == "image" and self.gray_scale) else 1
            shape = self.image_shape if field == "image" else store.shapes[field]
            y = torch.empty((b, *shape), dtype=torch.float32, device=device)
            n = lambda a: 0 if a is None else a.numel()
            L.check(lib.v1t_gather_transform(src.data_ptr(), int(src.dtype == torch.uint8), trial.data_ptr(), b, e, L.ptr(sub), n(sub), L.ptr(div), n(div),
                                             L.ptr(mul), n(mul), gray, y.data_ptr(), L.stream()), "gather_transform")
            out[field] = y
        p = positions.cpu().numpy()
        out["image_id"] = torch.from_numpy(np.asarray(self.image_ids[p]))
        tid = self.trial_ids[p]
        out["trial_id"] = torch.from_numpy(np.asarray(tid)) if np.issubdtype(np.asarray(tid).dtype, np.number) else list(tid)
        out["mouse_id"] = [self.mouse_id] * b
        return out


class DeviceLoader:
    """Stands where the reference has a torch DataLoader (data.py:470-485): iterable of collated batches, `.dataset`,
    `len()`; batches are built on the device from the packed store."""

    def __init__(self, dataset: MiceDataset, batch_size: int = 1, shuffle: bool = False, device: torch.device = None, seed: int = 0, **_unused):
        if device is None or torch.device(device).type != "cuda":
            raise RuntimeError("DeviceLoader builds batches with the gfx950 gather kernel: it needs a cuda device")
        self.dataset, self.batch_size, self.shuffle, self.device = dataset, int(batch_size), shuffle, torch.device(device)
        self._gen = torch.Generator().manual_seed(int(seed))

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self._gen) if self.shuffle else torch.arange(n)
        for i in range(0, n, self.batch_size):
            yield self.dataset.device_batch(order[i:i + self.batch_size], self.device)


class CycleDataloaders:
    """Cycles through the mice until the longest loader is exhausted (reference data.py:71-103)."""

    def __init__(self, ds: t.Dict[str, t.Any]):
        self.ds = ds
        self.max_iterations = max(len(d) for d in self.ds.values())

    @staticmethod
    def cycle(iterable):
        while True:
            yield from iter(iterable)

    def __iter__(self):
        cycles = {m: self.cycle(d) for m, d in self.ds.items()}
        mice = list(self.ds.keys())
        for i in range(len(self)):
            m = mice[i % len(mice)]
            yield m, next(cycles[m])

    def __len__(self):
        return len(self.ds) * self.max_iterations


def get_training_ds(args, data_dir: str, mouse_ids: t.List[str], batch_size: int = 1, device: torch.device = torch.device("cuda", 0)):
    """reference data.py:437-491"""
    if not hasattr(args, "ds_name"):
        args.ds_name = os.path.basename(args.dataset)
    train_ds, val_ds, test_ds = {}, {}, {}
    args.output_shapes = {}
    seed = int(getattr(args, "seed", 0))
    for k, mouse_id in enumerate(mouse_ids):
        mk = lambda tier, shuffle: DeviceLoader(MiceDataset(args, tier=tier, data_dir=data_dir, mouse_id=mouse_id), batch_size=batch_size, shuffle=shuffle,
                                                device=device, seed=seed * 1000 + k)
        train_ds[mouse_id], val_ds[mouse_id], test_ds[mouse_id] = mk("train", True), mk("validation", False), mk("test", False)
        args.output_shapes[mouse_id] = (train_ds[mouse_id].dataset.num_neurons,)
    args.input_shape = train_ds[mouse_ids[0]].dataset.image_shape
    return train_ds, val_ds, test_ds
