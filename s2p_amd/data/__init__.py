"""Data loading for S2P: (previous image, next state, next image) triples.

On-disk schema mirrors the RL side of the reference so the bulk-augmentation caller can reuse it
(`state_transition_rollout.py:62-68,222-243`): `image_observations` uint8 [T,H,W,3] (NHWC),
`observations` fp32 [T,S], optional `next_observations` fp32 [T,S], optional `timeouts` bool [T].
`.npz` always works; `.hdf5` needs h5py (absent in this image -> clear error).
Images are mapped uint8 -> [-1,1] (SPADE convention)."""
import os

import numpy as np
import torch


def load_arrays(path):
    if os.path.isdir(path):
        raise IsADirectoryError(path)
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    if path.endswith(".hdf5") or path.endswith(".h5"):
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError("reading %s needs h5py, which is not installed; convert to .npz" % path) from e
        with h5py.File(path, "r") as f:
            return {k: f[k][()] for k in f.keys()}
    raise ValueError("unsupported dataset file: %s" % path)


def resolve_dataset_file(dataroot, env_type):
    """--dataroot may be a file (README.md:59: ./datasets/cheetah.hdf5) or a directory (README.md:33: ./datasets)."""
    if os.path.isfile(dataroot):
        return dataroot
    for ext in (".npz", ".hdf5", ".h5"):
        p = os.path.join(dataroot, env_type + ext)
        if os.path.isfile(p):
            return p
    raise FileNotFoundError("no %s.{npz,hdf5} under %s" % (env_type, dataroot))


def images_to_tensor(u8_nhwc, size=None):
    """uint8 [T,H,W,3] -> fp32 NCHW in [-1,1] (optionally nearest-resized to size x size)."""
    x = torch.from_numpy(np.ascontiguousarray(u8_nhwc)).permute(0, 3, 1, 2).float() / 127.5 - 1.0
    if size is not None and (x.shape[2] != size or x.shape[3] != size):
        x = torch.nn.functional.interpolate(x, size=(size, size), mode="nearest")
    return x.contiguous()


def tensor_to_images(x_nchw):
    """fp32 NCHW in [-1,1] -> uint8 NHWC (the layout the RL consumer reads, rlkit/torch/slac/algo.py:189-190)."""
    y = ((x_nchw.detach().float().cpu().clamp(-1, 1) + 1.0) * 127.5).round().to(torch.uint8)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


class S2PDataset(torch.utils.data.Dataset):
    def __init__(self, opt):
        self.opt = opt
        arr = load_arrays(resolve_dataset_file(opt.dataroot, opt.env_type))
        self.images = arr["image_observations"]
        self.states = arr["observations"].astype(np.float32)
        T = len(self.images)
        if self.states.shape[1] != opt.state_dim:
            raise ValueError("dataset state dim %d != --state_dim %d" % (self.states.shape[1], opt.state_dim))
        timeouts = arr.get("timeouts")
        self.timeouts = None if timeouts is None else np.asarray(timeouts).astype(bool).reshape(-1)
        valid = np.ones(T - 1, dtype=bool)
        if timeouts is not None:
            valid &= ~self.timeouts[:-1]                          # no pair across an episode boundary
        self.index = np.nonzero(valid)[0][: opt.max_dataset_size]
        self.size = opt.crop_size

    def __len__(self):
        return len(self.index)

    def __getitem__(self, i):
        t = int(self.index[i])
        imgs = images_to_tensor(self.images[t:t + 2], self.size)
        return dict(prev_image=imgs[0], state=torch.from_numpy(self.states[t + 1]), image=imgs[1], index=t)

    def sequence(self, start, length):
        """Frames/states [start, start+length] for an autoregressive rollout."""
        if start + length >= len(self.images):
            raise IndexError("sequence [%d,%d] exceeds dataset length %d" % (start, start + length, len(self.images)))
        if self.timeouts is not None and self.timeouts[start:start + length].any():
            # frame t+1 after a timeout at t belongs to the next episode: the rollout would be scored against it
            end = start + int(np.argmax(self.timeouts[start:start + length]))
            raise IndexError("sequence [%d,%d] crosses an episode boundary (timeout at frame %d); use --seq_len <= %d"
                             % (start, start + length, end, end - start))
        imgs = images_to_tensor(self.images[start:start + length + 1], self.size)
        return imgs, torch.from_numpy(self.states[start:start + length + 1])


def create_dataloader(opt, rank=0, world_size=1):
    ds = S2PDataset(opt)
    sampler = None
    if world_size > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world_size, rank=rank,
                                                                  shuffle=not opt.serial_batches)
    dl = torch.utils.data.DataLoader(ds, batch_size=opt.batchSize, shuffle=(sampler is None and not opt.serial_batches),
                                     sampler=sampler, num_workers=int(opt.nThreads), drop_last=opt.isTrain)
    return dl
