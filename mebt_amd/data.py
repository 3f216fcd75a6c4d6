"""Token-grid data contract of the hot path — counterpart of the `vtokens` branch of reference mebt/data.py:
`HDF5Dataset_vtokens` (:330-414, items `{'video' [T,H,W] int64, 'cbox', 'indices' = randperm(prod(latent_shape))}`)
and the sharding `VideoData._dataloader` gets from `DistributedSampler` (:271-292).  Token files are `.npz` (or HDF5
when h5py is installed) with the reference's keys: `{train,test}_data` [frames,H,W] and `{train,test}_idx` (start frame
of every video plus a trailing sentinel).  Random draws use torch's global generator with the same calls in the same
order as the reference, so a seeded run picks the same clips, crops and permutations."""
import numpy as np
import torch


def _open(path):
    if str(path).endswith((".h5", ".hdf5")):
        import h5py                                    # optional dependency, exactly the reference's container
        return h5py.File(path, "r")
    return np.load(path)


class TokenClipDataset(torch.utils.data.Dataset):
    """reference HDF5Dataset_vtokens (data.py:330-414)"""

    def __init__(self, data_file, sequence_length, train=True, resolution=15, spatial_length=15, sample_every_n_frames=1,
                 latent_shape=()):
        super().__init__()
        self.train, self.sequence_length = train, sequence_length
        self.resolution, self.spatial_length = resolution, spatial_length
        self.sample_every_n_frames = sample_every_n_frames
        self.latent_shape = list(latent_shape)
        self.data_file = data_file
        self.prefix = "train" if train else "test"
        f = _open(data_file)
        self._tokens = np.array(f[f"{self.prefix}_data"])
        self._idx = np.array(f[f"{self.prefix}_idx"][:-1])       # data.py:361 (the sentinel is dropped)
        self.size = len(self._idx)
        if self.resolution is None:                               # token files carry their own grid size
            self.resolution = int(self._tokens.shape[1])
        if self.spatial_length is None:
            self.spatial_length = self.resolution
        ends = np.append(self._idx[1:], len(self._tokens))
        if self.size == 0 or int((ends - self._idx).max()) <= sequence_length:   # the reference would resample forever
            raise ValueError(f"{data_file}: no {self.prefix} video is longer than sequence_length={sequence_length} token frames")

    n_classes = 0                                                 # unconditional token sets (data.py:368-370)

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        start = self._idx[idx]
        end = self._idx[idx + 1] if idx < len(self._idx) - 1 else len(self._tokens)
        if end - start <= self.sequence_length:                   # clip too short: draw another video (data.py:391-392)
            return self.__getitem__(torch.randint(low=0, high=self.size, size=(1,)).item())
        start = start + torch.randint(low=0, high=int(end - start - self.sequence_length), size=(1,)).item()
        if self.spatial_length == self.resolution:
            video = torch.tensor(self._tokens[start:start + self.sequence_length]).long()
            box = 0
        else:
            y0 = torch.randint(low=0, high=self.resolution - self.spatial_length + 1, size=(1,)).item()
            x0 = torch.randint(low=0, high=self.resolution - self.spatial_length + 1, size=(1,)).item()
            video = torch.tensor(self._tokens[start:start + self.sequence_length, y0:y0 + self.spatial_length,
                                              x0:x0 + self.spatial_length]).long()
            box = np.array([y0, y0 + self.spatial_length, x0, x0 + self.spatial_length])
        if self.sample_every_n_frames > 1:
            video = video[::self.sample_every_n_frames]
        return dict(video=video, cbox=box, indices=torch.randperm(int(np.prod(self.latent_shape))))


class SyntheticTokenDataset(torch.utils.data.Dataset):
    """uniform random token grids of `shape` (what bench.py and the launcher use without a token file)"""

    def __init__(self, shape, size=1 << 16, vocab=16384):
        self.shape, self.size, self.vocab = tuple(shape), size, vocab

    def __len__(self):
        return self.size

    def __getitem__(self, idx):
        return dict(video=torch.randint(0, self.vocab, self.shape), cbox=0, indices=torch.randperm(int(np.prod(self.shape))))


class ShardedSampler(torch.utils.data.Sampler):
    """torch.utils.data.distributed.DistributedSampler semantics without a process group: shuffle with seed + epoch,
    pad by wrapping to a multiple of `num_replicas`, rank r takes positions r, r + R, r + 2R, ..."""

    def __init__(self, dataset_len, num_replicas=1, rank=0, shuffle=True, seed=0, drop_last=False):
        self.n, self.R, self.rank, self.shuffle, self.seed, self.drop_last = dataset_len, num_replicas, rank, shuffle, seed, drop_last
        self.epoch = 0
        self.num_samples = (self.n // self.R) if (drop_last and self.n % self.R) else -(-self.n // self.R)
        self.total_size = self.num_samples * self.R

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(idx)
            idx += idx[:pad] if pad <= len(idx) else (idx * -(-pad // len(idx)))[:pad]
        else:
            idx = idx[:self.total_size]
        return iter(idx[self.rank:self.total_size:self.R])


class TokenData:
    """`VideoData` for token grids (data.py:236-305): `args` carries data_path, sequence_length, resolution,
    spatial_length, sample_every_n_frames, latent_shape, batch_size, num_workers like the reference's config `data:` node."""

    def __init__(self, args, shuffle=True, world_size=1, rank=0):
        self.args, self.shuffle, self.world_size, self.rank = args, shuffle, world_size, rank

    def _dataset(self, train):
        a = self.args
        g = (lambda k, d: a[k] if (hasattr(a, "__contains__") and k in a) else getattr(a, k, d))
        if not g("data_path", None):
            return SyntheticTokenDataset(g("latent_shape", [4, 16, 16]))
        return TokenClipDataset(g("data_path", None), g("sequence_length", 4), train=train, resolution=g("resolution", 16),
                                spatial_length=g("spatial_length", g("resolution", 16)),
                                sample_every_n_frames=g("sample_every_n_frames", 1), latent_shape=g("latent_shape", [1]))

    def _dataloader(self, train):
        ds = self._dataset(train)
        a = self.args
        g = (lambda k, d: a[k] if (hasattr(a, "__contains__") and k in a) else getattr(a, k, d))
        sampler = ShardedSampler(len(ds), self.world_size, self.rank) if self.world_size > 1 else None
        nw = g("num_workers", 0)
        return torch.utils.data.DataLoader(ds, batch_size=g("batch_size", 6), num_workers=nw, pin_memory=True, sampler=sampler,
                                           shuffle=sampler is None and self.shuffle, persistent_workers=bool(train and nw > 0))

    def train_dataloader(self):
        return self._dataloader(True)

    def val_dataloader(self):
        return self._dataloader(False)

    test_dataloader = val_dataloader

    @staticmethod
    def add_data_specific_args(parent_parser):
        """the data flags of the reference's scripts (data.py:307-327), so that their command lines parse unchanged"""
        import argparse
        parser = argparse.ArgumentParser(parents=[parent_parser], add_help=False)
        parser.add_argument('--data_path', type=str, default='')
        parser.add_argument('--sequence_length', type=int, default=16)
        parser.add_argument('--resolution', type=int, default=128)
        parser.add_argument('--batch_size', type=int, default=32)
        parser.add_argument('--num_workers', type=int, default=8)
        parser.add_argument('--image_channels', type=int, default=3)
        parser.add_argument('--smap_cond', type=int, default=0)
        parser.add_argument('--smap_only', action='store_true')
        parser.add_argument('--text_cond', action='store_true')
        parser.add_argument('--vtokens', action='store_true')
        parser.add_argument('--vtokens_pos', action='store_true')
        parser.add_argument('--spatial_length', type=int, default=15)
        parser.add_argument('--sample_every_n_frames', type=int, default=1)
        parser.add_argument('--image_folder', action='store_true')
        parser.add_argument('--stft_data', action='store_true')
        parser.add_argument('--preprocessed_hdf5', action='store_true')
        return parser
