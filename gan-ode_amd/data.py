"""The step immediately upstream of the hot path: Rotated-MNIST clips.

Reference: dataset/mnist_rotation.py:7-63 -- `rot-mnist_rand.mat` holds X [n, 16, 784] in [0, 1] and Y; the first
N=500 clips are the training set, viewed as [N, 16, 1, 28, 28]; the image dataset returns one random frame per clip.
The whole training set is 25 MB in fp32, so it lives on the GPU and batches are gathered there (no per-step H2D);
the discriminators read the [B, T, C, H, W] batch in place through strides (no transpose copy).
"""
from __future__ import annotations

import os

import numpy as np
import torch


def load_rot_mnist(path, train=True, N=500, T=16):
    """-> (X float32 [N, T, 1, 28, 28], Y); same slicing as MNISTRotationVideo.__init__."""
    from scipy.io import loadmat
    if not os.path.exists(path):
        raise FileExistsError(f"File {path} does not exists")
    data = loadmat(path)
    X = torch.from_numpy(np.ascontiguousarray(data["X"].squeeze())).float()
    Y = torch.from_numpy(np.ascontiguousarray(data["Y"].squeeze()))
    if train:
        return X[:N].reshape(N, T, 1, 28, 28), Y[:N]
    return X[N:].reshape(-1, T, 1, 28, 28), Y[N:]


class MNISTRotationVideo(torch.utils.data.Dataset):
    """Drop-in for dataset.MNISTRotationVideo (host tensors, for the reference's DataLoader)."""

    def __init__(self, path_to_data, train=True, N=500, T=16, transform=None):
        self.X, self.Y = load_rot_mnist(path_to_data, train, N, T)
        self.transform = transform

    def __len__(self):
        return len(self.Y)

    def __getitem__(self, idx):
        v = self.X[idx]
        if self.transform is not None:
            v = self.transform(v)
        return v.float(), self.Y[idx]


class MNISTRotationImage(MNISTRotationVideo):
    """Drop-in for dataset.MNISTRotationImage: a random frame of clip idx (np.random.randint, as the reference)."""

    def __init__(self, path_to_data, train=True, N=500, T=16, transform=None):
        super().__init__(path_to_data, train, N, T, transform)
        self.T = T

    def __getitem__(self, idx):
        img = self.X[idx, np.random.randint(0, self.T)]
        if self.transform is not None:
            img = self.transform(img)
        return img.float(), self.Y[idx]


class RotMnistOnDevice:
    """Device-resident training set with shuffled, drop_last batch streams (what DataLoader(shuffle=True,
    drop_last=True) + dataGen() deliver in mnist_moco_ode.py:56-71), gathered on the GPU."""

    def __init__(self, path_or_tensor, device="cuda", N=500, T=16, seed=None):
        X = path_or_tensor if torch.is_tensor(path_or_tensor) else load_rot_mnist(path_or_tensor, True, N, T)[0]
        self.X = X.to(device).contiguous()             # [N, T, 1, 28, 28]
        self.N, self.T = self.X.shape[0], self.X.shape[1]
        self.gen = torch.Generator()
        if seed is not None:
            self.gen.manual_seed(seed)

    def _epochs(self, batch):
        while True:
            perm = torch.randperm(self.N, generator=self.gen)
            for i in range(0, self.N - batch + 1, batch):
                yield perm[i:i + batch]

    def videos(self, batch):
        for idx in self._epochs(batch):
            yield self.X[idx.to(self.X.device)]                                  # [B, T, 1, 28, 28]

    def images(self, batch):
        for idx in self._epochs(batch):
            frame = torch.randint(0, self.T, (batch,), generator=self.gen)
            yield self.X[idx.to(self.X.device), frame.to(self.X.device)]         # [B, 1, 28, 28]
