"""Sensorium / Franke on-disk format -> packed per-mouse store in HBM -> device batches.

Host-side mirror of src/v1t/data.py with the same names, arguments and batch dictionaries:
- `MiceDataset`       <- data.py:275-434 (tiers, statistics, response precision, transforms, `__getitem__`)
- `get_training_ds`   <- data.py:437-491 (returns loaders; sets args.output_shapes / args.input_shape)
- `CycleDataloaders`  <- data.py:71-103, `micro_batching` <- data.py:106-110
The reference reads 4 `.npy` files per trial per epoch through torch DataLoader workers. Here a mouse's recording is read
ONCE (`MouseStore`) into packed `[trials][...]` arrays in HBM, and `DeviceLoader` builds every batch with one gather +
standardise launch per field (`v1t_gather_transform`), so nothing but a few hundred index bytes crosses PCIe per step.
The standardisation lives in ONE table (`FieldTransform` per field) read by the device gather, by the host `__getitem__`
(for code that indexes the dataset directly) and by the inverse transforms.
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


class FieldTransform(t.NamedTuple):
    """y = (x - sub) / div * mul for one field of a trial (None = identity for that step); `v1t_gather_transform` applies it on
    the device, `apply` / `invert` on the host. Statistics are scalars or arrays of the field's own shape."""
    sub: t.Optional[np.ndarray] = None
    div: t.Optional[np.ndarray] = None
    mul: t.Optional[np.ndarray] = None

    def apply(self, x):
        if self.sub is not None:
            x = x - self.sub
        if self.div is not None:
            x = x / self.div
        return x if self.mul is None else x * self.mul

    def invert(self, y):
        if self.mul is not None:
            y = y / self.mul
        if self.div is not None:
            y = y * self.div
        return y if self.sub is None else y + self.sub


def _standardisation(stats: t.Dict[str, t.Dict[str, np.ndarray]]) -> t.Dict[str, FieldTransform]:
    """The dataset's standardisation as ONE table (what data.py:357-410 does field by field): images and pupil centres are
    z-scored, behaviour is divided by its std, responses are multiplied by the per-neuron precision 1 / std, where neurons whose
    std is below 1 % of the mean std get 1 / (that threshold) instead (data.py:387-397)."""
    std = stats["response"]["std"]
    floor = 0.01 * np.mean(std)
    precision = 1.0 / np.where(std > floor, std, floor).astype(std.dtype)
    return {"image": FieldTransform(sub=stats["image"]["mean"], div=stats["image"]["std"]),
            "response": FieldTransform(mul=precision),
            "behavior": FieldTransform(div=stats["behavior"]["std"]),
            "pupil_center": FieldTransform(sub=stats["pupil_center"]["mean"], div=stats["pupil_center"]["std"])}


class MiceDataset:
    """One tier of one mouse's recording (the reference's `MiceDataset`, data.py:275-434: same constructor, attributes the
    readouts and metrics read - `coordinates`, `neuron_ids`, `*_stats`, `image_ids`, `trial_ids`, `image_shape`, `hashed` - and
    per-field `transform_* / i_transform_*` methods). Built around the packed `MouseStore`: the standardisation is one
    `FieldTransform` table that the device gather, the host `__getitem__` and the inverse transforms all read."""

    def __init__(self, args, tier: str, data_dir: str, mouse_id: str):
        if tier not in ("train", "validation", "test", "final_test"):
            raise AssertionError(f"unknown tier {tier}")
        self.tier, self.mouse_id, self.ds_name, self.behavior_mode = tier, mouse_id, args.ds_name, args.behavior_mode
        self.mouse_dir = os.path.join(data_dir, get_mouse2path(self.ds_name)[mouse_id])
        if self.behavior_mode and mouse_id == "S0":
            raise ValueError("Mouse S0 does not have behaviour data.")  # the reference's message (data.py:296)
        md = load_mouse_metadata(self.ds_name, mouse_dir=self.mouse_dir)
        self.neuron_ids, self.coordinates, self.stats = md["neuron_ids"], md["coordinates"], md["stats"]
        self.transforms = _standardisation(self.stats)
        self._response_precision = self.transforms["response"].mul
        # the trials of this tier; --limit_data keeps a seeded random subset of the training tier (data.py:310-321)
        rows = np.flatnonzero(md["tiers"] == tier).astype(np.int32)
        limit = getattr(args, "limit_data", None)
        if tier == "train" and limit and len(rows) > limit:
            rows = np.random.default_rng(seed=args.seed).choice(rows, size=limit, replace=False)
        self.indexes, self.image_ids, self.trial_ids = rows, md["image_ids"][rows], md["trial_ids"][rows]
        self.hashed = (self.ds_name, mouse_id) in (("sensorium", "S0"), ("sensorium", "S1"))  # live-test mice: hashed labels
        c, *hw = np.load(os.path.join(self.mouse_dir, "data", FIELDS["image"], "0.npy")).shape
        self.gray_scale = bool(getattr(args, "gray_scale", False)) and self.ds_name == "franke2022"
        self.image_shape = (1 if self.gray_scale else c, *hw)
        self._dev: t.Optional[t.Dict[str, t.Any]] = None

    def __len__(self):
        return len(self.indexes)

    def compute_response_precision(self):
        """Re-derive the response standardisation from `response_stats` (data.py:394-404; public there, called by its constructor):
        refreshes the `response` row of the transform table, which the device gather and the host methods read."""
        self.transforms["response"] = _standardisation(self.stats)["response"]
        self._response_precision = self.transforms["response"].mul
        self._dev = None  # packed device copies carry the old precision

    image_stats = property(lambda self: self.stats["image"])
    response_stats = property(lambda self: self.stats["response"])
    behavior_stats = property(lambda self: self.stats["behavior"])
    pupil_stats = property(lambda self: self.stats["pupil_center"])
    num_neurons = property(lambda self: len(self.neuron_ids))

    # ---- host side: the reference's per-field method names over the table
    def transform_image(self, image):
        z = self.transforms["image"].apply(image)
        return z.mean(axis=0, keepdims=True) if self.gray_scale else z

    def i_transform_image(self, image):
        if self.behavior_mode == 1:  # behaviour planes were appended as channels: keep the stimulus channel (data.py:364-372)
            image = image[:1] if image.ndim == 3 else image[:, :1]
        return self.transforms["image"].invert(image)

    def __getitem__(self, idx):
        """One standardised trial from the .npy files (for code that indexes the dataset; the training path is `device_batch`)."""
        trial = self.indexes[idx]
        item = {f: self.transforms[f].apply(np.load(os.path.join(self.mouse_dir, "data", d, f"{trial}.npy")).astype(np.float32)) for f, d in FIELDS.items()}
        if self.gray_scale:
            item["image"] = item["image"].mean(axis=0, keepdims=True)
        item.update(image_id=self.image_ids[idx], trial_id=self.trial_ids[idx], mouse_id=self.mouse_id)
        return item

    # ---- device path
    def _device_state(self, device: torch.device):
        if self._dev is None or self._dev["device"] != torch.device(device):
            store = MouseStore.get(self.mouse_dir, device)
            tf = {}
            for field, ft in self.transforms.items():
                e = store.data[field].shape[1]
                dev_ft = []
                for a in ft:  # statistics broadcast over the flattened field by period: scalar or the field's own size
                    if a is not None:
                        a = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))).to(device)
                        if a.numel() not in (1, e):
                            raise RuntimeError(f"{field} statistics of {a.numel()} elements do not broadcast over {store.shapes[field]}")
                    dev_ft.append(a)
                tf[field] = tuple(dev_ft)
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


def _field_methods():
    # transform_<field> / i_transform_<field> for the remaining fields (the names evaluate.py / user code call): table look-ups
    for field, stem in (("pupil_center", "pupil_center"), ("behavior", "behavior"), ("response", "response")):
        setattr(MiceDataset, f"transform_{stem}", lambda self, x, _f=field: self.transforms[_f].apply(x))
        setattr(MiceDataset, f"i_transform_{stem}", lambda self, x, _f=field: self.transforms[_f].invert(x))


_field_methods()


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
