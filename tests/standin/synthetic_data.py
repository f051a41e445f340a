"""Synthetic retrieval dataset for the stand-in training script: numpy tuples in the order the reference's retrieval loaders
return them (dataloader_youcook_retrieval.py:178-189), built with the numpy-1.x dtype aliases (np.float, np.long) the
reference's loaders use (:139) -- which only exist again once run_univl_amd.install_compat() has run."""
import numpy as np
from torch.utils.data import Dataset


class Synthetic(Dataset):
    """Numpy tuples in the order the reference's retrieval loaders return them (dataloader_youcook_retrieval.py:178-189)."""

    def __init__(self, n, W, F, D, seed):
        self.n, self.W, self.F, self.D = n, W, F, D
        self.rng = np.random.RandomState(seed)
        self.items = [self._make() for _ in range(n)]

    def _make(self):
        W, F, D = self.W, self.F, self.D
        lt, lv = self.rng.randint(4, W + 1), self.rng.randint(1, F + 1)
        ids = np.zeros((1, W), dtype=np.long)
        ids[0, :lt] = self.rng.randint(1000, 30522, size=lt)
        ids[0, 0] = 101
        mask = np.zeros((1, W), dtype=np.long)
        mask[0, :lt] = 1
        video = np.zeros((1, F, D), dtype=np.float)                  # np.float: float64, as in the reference loaders
        video[0, :lv] = self.rng.randn(lv, D)
        vmask = np.zeros((1, F), dtype=np.long)
        vmask[0, :lv] = 1
        seg = np.zeros((1, W), dtype=np.long)
        labels = -np.ones((1, W), dtype=np.long)
        vlabels = -np.ones((1, F), dtype=np.long)
        return ids, mask, seg, video, vmask, ids.copy(), labels, video.copy(), vlabels

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return self.items[i]
