"""GPSRODataset with the reference's constructor and item contract
(src/deepCam/data/gpsro_dataset.py:15-175) on the staging-ring reader: while
item k is consumed, the files of item k+1 are already being read and copied to
HBM (the reference reads three files synchronously per item, :130-156)."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import numpy_reader as nr


class GPSRODataset(Dataset):

    def init_files(self, source):
        self.source = source
        self.allfiles = sorted([x.replace("data_in_", "") for x in os.listdir(self.source)
                                if x.endswith(".npy") and x.startswith("data_in_")])
        if self.shuffle:
            self.rng.shuffle(self.allfiles)      # identical on every rank (same seed), gpsro_dataset.py:24-25
        shard_size = len(self.allfiles) // self.shard_num
        start = shard_size * self.shard_idx
        self.files = self.allfiles[start:start + shard_size]
        self.length = len(self.files)

    def __init__(self, source, statsfile, channels, normalization_type="MinMax", shuffle=True, masks=False,
                 augmentation_mode=None, shard_idx=0, shard_num=1, num_intra_threads=1, seed=12345,
                 read_device=torch.device("cpu"), send_device=torch.device("cpu")):
        self.channels = channels
        self.normalization_type = normalization_type
        self.shuffle = shuffle
        self.masks = masks
        self.seed = seed
        self.rng = np.random.RandomState(seed)
        self.shard_idx, self.shard_num = shard_idx, shard_num
        self.read_device, self.send_device = torch.device(read_device), torch.device(send_device)
        self.augmentation_mode = augmentation_mode
        self.init_files(source)
        if self.augmentation_mode == "static":
            self.mixing = self.rng.rand(self.length)
        devindex = -1 if self.read_device.type == "cpu" else (self.read_device.index or 0)
        kinds = ["data_in_", "data_out_"] + (["masks_"] if self.masks else [])
        self._readers = {}
        for kind in kinds:
            r = nr.numpy_reader(split_axis=False, device=devindex)
            r.num_intra_threads = num_intra_threads
            if self.length:
                r.parse(os.path.join(self.source, kind + self.files[0]))
            self._readers[kind] = r
        self.npr_data, self.npr_label = self._readers["data_in_"], self._readers["data_out_"]
        stats = np.load(statsfile)
        if self.normalization_type == "MinMax":
            data_shift = stats["data_minval"][self.channels]
            data_scale = 1. / (stats["data_maxval"][self.channels] - data_shift)
            label_shift = stats["label_minval"][self.channels]
            label_scale = 1. / (stats["label_maxval"][self.channels] - label_shift)
        elif self.normalization_type == "MeanVariance":
            data_shift = stats["data_mean"][self.channels]
            data_scale = 1. / np.sqrt(stats["data_sqmean"][self.channels] - np.square(data_shift))
            label_shift = stats["label_mean"][self.channels]
            label_scale = 1. / np.sqrt(stats["label_sqmean"][self.channels] - np.square(label_shift))
        else:
            raise NotImplementedError(self.normalization_type)
        tt = lambda a: torch.tensor(np.reshape(a, (a.shape[0], 1, 1)).astype(np.float32)).to(self.send_device)  # noqa
        self.data_shift, self.data_scale = tt(data_shift), tt(data_scale)
        self.label_shift, self.label_scale = tt(label_shift), tt(label_scale)
        self._staged = None      # index whose files are in flight
        print("Initialized dataset with ", self.length, " samples.")

    def __len__(self):
        return self.length

    @property
    def shapes(self):
        return self.npr_data.shape, self.npr_label.shape

    def _path(self, kind, idx):
        return os.path.realpath(os.path.join(self.source, kind + self.files[idx]))

    def _stage(self, idx):
        for kind, r in self._readers.items():
            r.prefetch(self._path(kind, idx))
        self._staged = idx

    def __getitem__(self, idx):
        if self._staged != idx:
            if self._staged is not None:          # drop a mispredicted prefetch
                for r in self._readers.values():
                    r.get_prefetched()
            self._stage(idx)
        parts = {kind: r.get_prefetched() for kind, r in self._readers.items()}
        self._staged = None
        if idx + 1 < self.length:                 # sequential access (DataLoader without shuffling): run ahead
            self._stage(idx + 1)
        data, label = parts["data_in_"], parts["data_out_"]
        mask = parts.get("masks_")
        if data.device != self.send_device:
            data = data.to(self.send_device)
        if label.device != self.send_device:
            label = label.to(self.send_device)
        if mask is not None and mask.device != self.send_device:
            mask = mask.to(self.send_device)
        if self.augmentation_mode == "static":
            data = data + self.mixing[idx] * label
            label = (1. - self.mixing[idx]) * label
        elif self.augmentation_mode == "dynamic":
            p = self.rng.rand()
            data = data + p * label
            label = (1. - p) * label
        data = self.data_scale * (data - self.data_shift)
        label = self.label_scale * (label - self.label_shift)
        if self.masks:
            return data, label, mask, self.files[idx]
        return data, label, self.files[idx]
