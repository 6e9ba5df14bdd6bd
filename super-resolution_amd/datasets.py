"""Jet-image data path of the reference's ``datasets.py`` on the GPU.

The reference decodes one event at a time in ``Dataset.__getitem__`` with a Python loop over the constituents
(``extract``, datasets.py:136-145) on the CPU, ``n_cpu=0`` workers by default -- a few hundred microseconds per event,
which would cap a 30-images-per-10-ms training loop far below kernel speed.  Here the dataset hands out the RAW sparse rows
and a whole batch is decoded by one kernel launch (``srk_jet_extract``: one workgroup per event, bit-identical to the
sequential loop), followed by the SumPool2d kernels for ``pre_factor`` and the LR image.

Same names and semantics as the reference where they exist::

    extract(data, etaBins, phiBins, channels=1)         datasets.py:136-145   (data: [2, L] = positions, energies)
    ThresholdImageCutter / NHardestCutter / Cutter      datasets.py:170-201
    SparseJetDataset(...)                                datasets.py:232-249   (array-backed: pandas HDF5 needs PyTables,
                                                                               which this image lacks; pass the row array)
    JetDataset(...)                                      datasets.py:203-229   (dense rows * 70)
    get_dataset(dataset_type, ...)                       datasets.py:316-329   ('spjet' / 'jet' from .npy row files)

plus ``extract_batch`` and ``Dataset.decode_batch`` -- the batched forms the training loop uses.
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import _lib as L
from . import ops


def extract_batch(rows, etaBins, phiBins, threshold=None, n_pairs=None):
    """rows: CUDA fp32 [B, >= 2*n_pairs] of interleaved (position, energy) pairs (a SparseJet dataframe row without its
    last column) -> [B, 1, etaBins, phiBins].  ``threshold``: ThresholdImageCutter folded into the store."""
    if not rows.is_cuda:
        raise RuntimeError("super-resolution_amd.datasets.extract_batch runs on the HIP kernel srk_jet_extract: got a CPU tensor")
    rows = rows.contiguous().float()
    B, stride = rows.shape
    n_pairs = stride // 2 if n_pairs is None else n_pairs
    out = torch.empty(B, 1, etaBins, phiBins, dtype=torch.float32, device=rows.device)
    L.check(L.lib().srk_jet_extract(rows.data_ptr(), B, n_pairs, stride, etaBins, phiBins, -1.0 if not threshold else float(threshold),
                                    out.data_ptr(), L.stream_ptr()), "srk_jet_extract")
    return out


def extract(data, etaBins, phiBins, channels=1):
    """datasets.py:136-145 for one event: ``data`` [2, L] (row 0 positions, row 1 energies) -> [channels, etaBins, phiBins]
    (only channel 0 is filled, as in the reference)."""
    img = extract_batch(data.t().contiguous().reshape(1, -1), etaBins, phiBins)[0]
    if channels > 1:
        img = torch.cat([img, torch.zeros(channels - 1, etaBins, phiBins, device=img.device)], 0)
    return img


def _keep_where(x, keep):
    """x with every entry outside the boolean mask ``keep`` set to zero."""
    return x * keep.to(x.dtype)


class ThresholdImageCutter:
    """Zeroes every pixel that does not exceed the energy threshold (datasets.py:170-175)."""

    def __init__(self, threshold):
        self.thres = threshold

    def __call__(self, x):
        return _keep_where(x, x > self.thres)


class NHardestCutter:
    """Keeps the N hardest (largest) pixels of an image, ties included, and zeroes the rest (datasets.py:178-186).  A batch
    [B, C, H, W] with B > 1 is cut image by image; [C, H, W] / [1, C, H, W] is one image, as in the reference."""

    def __init__(self, N):
        self.N = N

    def __call__(self, x):
        per_image = x.dim() == 4 and x.shape[0] > 1
        flat = x.reshape(x.shape[0], -1) if per_image else x.reshape(1, -1)
        nth = torch.topk(flat, self.N, dim=1).values[:, -1]               # the N-th largest value of each image
        nth = nth.reshape(-1, 1, 1, 1) if per_image else nth.reshape(())
        return _keep_where(x, x >= nth)


class Cutter:
    """Dispatches to at most one of the two cutters (datasets.py:189-201); with neither it is the identity."""

    def __init__(self, thres=None, amount=None):
        if thres and amount:
            raise NotImplementedError("only one of thres and amount can be specified")
        self.cutter = ThresholdImageCutter(thres) if thres else (NHardestCutter(amount) if amount else None)

    def __call__(self, x):
        return x if self.cutter is None else self.cutter(x)


class _RowDataset(Dataset):
    """Array-backed event table: ``__getitem__`` returns the raw row; ``decode_batch`` builds {"lr", "hr"} on the GPU."""

    def __init__(self, rows, amount=None, etaBins=80, phiBins=80, factor=2, pre_factor=1, threshold=None, N=None):
        super().__init__()
        rows = np.load(rows) if isinstance(rows, str) else np.asarray(rows)
        if amount is not None:
            rows = rows[:amount]
        self.rows = torch.as_tensor(rows, dtype=torch.float32)
        self.etaBins, self.phiBins, self.factor, self.pre_factor = etaBins, phiBins, factor, pre_factor
        self.threshold, self.N = threshold, N
        self.cutter = Cutter(threshold, N)

    def __len__(self):
        return self.rows.shape[0]

    def __getitem__(self, item):
        return {"rows": self.rows[item]}

    def _finish(self, img):
        if self.pre_factor > 1:
            img = ops.sum_pool(img, self.pre_factor)
        return {"lr": ops.sum_pool(img, self.factor), "hr": img}


class SparseJetDataset(_RowDataset):
    """datasets.py:232-249: rows are (position, energy) pair lists + one trailing column that is dropped (``[:-1]``)."""

    def __init__(self, rows, amount=None, etaBins=80, phiBins=80, factor=2, pre_factor=1, threshold=None, N=None, noise_factor=None):
        super().__init__(rows, amount, etaBins, phiBins, factor, pre_factor, threshold, N)
        self.noise_factor = noise_factor
        self.noise_pixels = 150          # datasets.py:241-242: all but 150 randomly chosen noise pixels are zeroed

    @staticmethod
    def add_noise(img, noise, keep, noise_factor):
        """datasets.py:238-244 for a whole batch, given the random draws: ``noise`` = randn of img's shape, ``keep`` [B, n] = the flat
        pixel indices of each image whose noise survives.  noise <- |noise| / (noise_factor * max |noise| of the image), zero outside
        ``keep``, added to the image."""
        b = img.shape[0]
        a = noise.abs().reshape(b, -1)
        a = a / (noise_factor * a.max(dim=1, keepdim=True).values)
        sparse = torch.zeros_like(a).scatter_(1, keep, a.gather(1, keep))
        return img + sparse.view_as(img)

    def decode_batch(self, rows):
        """rows: [B, row_len] on the GPU -> {"lr": [B,1,h,w], "hr": [B,1,H,W]}."""
        n_pairs = (rows.shape[1] - 1) // 2
        img = extract_batch(rows, self.etaBins * self.pre_factor, self.phiBins * self.pre_factor,
                            threshold=self.threshold if not self.N else None, n_pairs=n_pairs)
        if self.N:
            img = self.cutter(img)
        if self.noise_factor is not None:
            # the reference draws on the host (torch.randn + np.random.choice); here the whole batch is drawn on the GPU: per image
            # |N(0,1)| noise, scaled to a maximum of 1 / noise_factor, on 150 pixels chosen uniformly without replacement
            b, npx = img.shape[0], img[0].numel()
            noise = torch.randn(img.shape, device=img.device)
            keep = torch.rand(b, npx, device=img.device).topk(min(self.noise_pixels, npx), dim=1).indices
            img = self.add_noise(img, noise, keep, float(self.noise_factor))
        return self._finish(img)


class JetDataset(_RowDataset):
    """datasets.py:203-229: dense rows, scaled by 70."""

    def decode_batch(self, rows):
        img = self.cutter(rows.reshape(rows.shape[0], 1, self.etaBins * self.pre_factor, self.phiBins * self.pre_factor) * 70)
        return self._finish(img.contiguous())


def get_dataset(dataset_type, dataset_path, hr_height, hr_width, factor=2, amount=None, pre=1, threshold=None, N=None, noise_factor=None):
    """datasets.py:316-329 for the jet formats; ``dataset_path`` is a ``.npy`` file of the dataframe's rows."""
    if dataset_type == 'jet':
        return JetDataset(dataset_path, amount=amount, etaBins=hr_height, phiBins=hr_width, factor=factor, pre_factor=pre, threshold=threshold, N=N)
    if dataset_type == 'spjet':
        return SparseJetDataset(dataset_path, amount=amount, etaBins=hr_height, phiBins=hr_width, factor=factor, pre_factor=pre,
                                threshold=threshold, N=N, noise_factor=noise_factor)
    raise NotImplementedError(f"dataset_type {dataset_type!r}: only 'jet' and 'spjet' (from .npy row files) are implemented")
