"""LETOR query files (`train.h5` / `test.h5`) -> the items the four `_trad` twins train on.

One HDF5 file per split, one 2-D float dataset per query id: row = document, column 0 = relevance label, column 1 = query
id, columns 2: = the LETOR features (46 for MQ2008, 136 for MSLR-WEB10K), every query resampled to 20 rows when the file
was made (datasets_trad/convert_to_h5py.py:17-43).  The reference repeats an `LTRDataset` class in each script with a
different sampling rule; the three rules live here and each twin exports its own `LTRDataset(args, path, is_train, ...)`:

    QueryRows    pointwise_trad.py:88-109, pointwise_2data_trad.py:87-108   one item per query, rows as stored
    QueryPairs   ppo_trad.py:63-98                                         training: max_tags random ordered document pairs
                                                                           per query; validation: the whole query
    RewardPairs  reward_trad.py:87-134                                     max_tags label-stratified pairs per query with the
                                                                           4-position chosen / reject index layouts

Queries are visited in the file's key order (increasing name: "10" < "2", as h5py lists them), and the Python / NumPy RNGs are
consumed call for call like upstream, so a run seeded like the reference draws the same pairs (tests/golden/letor_readers.json:
the reference's own classes on the same real HDF5 files).  Files are opened through h5py when it is installed and through
`lr2ppo_amd.h5lite` (libhdf5 via ctypes) otherwise.  Host-side data plumbing: nothing here touches the GPU.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from .. import h5lite


def _open_split(path, is_train):
    return h5lite.open_file(os.path.join(path, "train.h5" if is_train else "test.h5"), "r")


def _labels_and_features(table):
    return table[:, 0], table[:, 2:]


class QueryRows(Dataset):
    """Item = (ground_truths [docs] f64 array, query id (str), features [docs, F] f64 array); the default collate stacks them."""

    def __init__(self, args, path, is_train=False):
        self.is_train, self.data = is_train, _open_split(path, is_train)
        self._keys = None

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        if self._keys is None:
            self._keys = list(self.data.keys())
        query_id = self._keys[index]
        ground_truths, features = _labels_and_features(self.data[query_id][()])
        return ground_truths, query_id, features


class QueryPairs(Dataset):
    """Item = (ground_truths [k], query id, features [k, F]) as tensors: k = 2 drawn documents (training) or all of them."""

    def __init__(self, args, path, is_train=False, max_tags=20):
        self.is_train, self.data = is_train, _open_split(path, is_train)
        self.dataset = []
        for query_id in self.data.keys():
            ground_truths, features = _labels_and_features(self.data[query_id][()])
            order = list(range(len(ground_truths)))
            if not is_train:
                self.dataset.append((ground_truths, query_id, features, order))
                continue
            for _ in range(max_tags):                 # the SAME list is shuffled again for every draw (ppo_trad.py:77-81)
                random.shuffle(order)
                self.dataset.append((ground_truths, query_id, features, order[:2]))

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        ground_truths, query_id, features, rows = self.dataset[index]
        return torch.tensor(ground_truths[rows]), query_id, torch.tensor(features[rows])


class RewardPairs(Dataset):
    """Item = (ground_truths [docs], query id, features [docs, F], chosen [4], reject [4]): one document per relevance class
    present is drawn, two of those form the pair; a pair with equal labels is dropped.  Positions 0-1 of both layouts name the
    pair, positions 2-3 order it better-first (chosen) or worse-first (reject) -- reward_trad.py:106-118."""

    CLASSES = 5

    def __init__(self, args, path, is_train=False, max_tags=20):
        self.is_train, self.data = is_train, _open_split(path, is_train)
        self.dataset = []
        for query_id in self.data.keys():
            ground_truths, features = _labels_and_features(self.data[query_id][()])
            rows_of = [[] for _ in range(self.CLASSES)]
            for row, label in enumerate(ground_truths):
                rows_of[int(label)].append(row)
            present = [rows for rows in rows_of if rows]
            for _ in range(max_tags):
                drawn = [np.random.choice(rows) for rows in present]
                if len(drawn) < 2:
                    continue
                a, b = np.random.choice(drawn, 2, replace=False)
                if ground_truths[a] == ground_truths[b]:
                    continue
                better_first, worse_first = [a, b, a, b], [a, b, b, a]
                if ground_truths[a] < ground_truths[b]:
                    better_first, worse_first = worse_first, better_first
                self.dataset.append((ground_truths, query_id, features, better_first, worse_first))

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        ground_truths, query_id, features, chosen, reject = self.dataset[index]
        return torch.tensor(ground_truths), query_id, torch.tensor(features), torch.tensor(chosen), torch.tensor(reject)


def write_split(path, is_train, tables):
    """{query id: [docs, 2 + F] array} -> `path`/train.h5 | test.h5, one float64 dataset per query like
    datasets_trad/convert_to_h5py.py:41-43 (`hf.create_dataset(str(key), data=value)`)."""
    os.makedirs(path, exist_ok=True)
    with h5lite.open_file(os.path.join(path, "train.h5" if is_train else "test.h5"), "w") as hf:
        for key, value in tables.items():
            hf.create_dataset(str(key), data=np.asarray(value, dtype=np.float64))
