"""Bulk augmentation caller: the step the reference README leaves as TODO
("Then, you should run generation code...", README.md:54).

Input  : the file written by the reference's `state_transition_rollout.py:222-243`
         (`all_state_1step_random_action_dataset_augment.hdf5`, or an .npz with the same keys):
         `image_observations` uint8 [N,H,W,3] (I_t, NHWC) and `next_observations` fp32 [N,S] (predicted s_{t+1}).
Output : the same keys plus `image_observations_tp1` uint8 [N,H,W,3] = G(I_t, s_{t+1}), the key and layout the
         reference's RL consumer reads (`rlkit/torch/slac/algo.py:189-190`, used at :336).

Frames go host uint8 -> device uint8 -> NHWC compute dtype (s2p_u8_to_nhwc) -> generator (explicit HIP forward, no
autograd, frames never leave the kernels' layout) -> uint8 NHWC on the device (s2p_nhwc_to_u8) -> host.
Rows shard over ranks with NO collective (each rank writes its slice): `RANK`/`WORLD_SIZE` or --shard.
"""
import os

import numpy as np
import torch

from . import ops
from ._lib import chunk_elems
from .data import load_arrays

REQUIRED = ("image_observations", "next_observations")


def save_arrays(path, arrays):
    if path.endswith(".npz"):
        np.savez(path, **arrays)
    elif path.endswith(".hdf5") or path.endswith(".h5"):
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError("writing %s needs h5py, which is not installed; use an .npz output" % path) from e
        with h5py.File(path, "w") as f:
            for k, v in arrays.items():
                f.create_dataset(str(k), data=v)
    else:
        raise ValueError("unsupported output file: %s" % path)


def shard_range(n, rank, world):
    """Contiguous, balanced row ranges; concatenating all ranks' slices in rank order restores the dataset."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def check_inputs(arrays, state_dim):
    for k in REQUIRED:
        if k not in arrays:
            raise KeyError("augment: input file has no '%s' (expected the keys of state_transition_rollout.py:222-243)" % k)
    img, st = arrays["image_observations"], arrays["next_observations"]
    if img.dtype != np.uint8 or img.ndim != 4 or img.shape[3] != 3:
        raise ValueError("image_observations must be uint8 [N,H,W,3], got %s %s" % (img.dtype, img.shape))
    if st.ndim != 2 or st.shape[0] != img.shape[0] or st.shape[1] != state_dim:
        raise ValueError("next_observations must be [N,%d] matching image_observations, got %s" % (state_dim, st.shape))
    if img.shape[1] % 4 or img.shape[2] % 4:
        raise ValueError("frame size %dx%d must be a multiple of 4" % (img.shape[1], img.shape[2]))


@torch.no_grad()
def generate_tp1(netG, images_u8, next_states, batch=256):
    """images_u8: uint8 [N,H,W,3] (numpy), next_states: fp32 [N,S] (numpy) -> uint8 [N,H,W,3] numpy."""
    dt = netG.compute_dtype
    dev = netG.store.master.device
    n = images_u8.shape[0]
    out = np.empty_like(images_u8)
    for lo in range(0, n, batch):
        hi = min(lo + batch, n)
        xu8 = torch.from_numpy(np.ascontiguousarray(images_u8[lo:hi])).to(dev, non_blocking=True)
        st = torch.from_numpy(np.ascontiguousarray(next_states[lo:hi], dtype=np.float32)).to(dev, non_blocking=True)
        x = ops.u8_to_nhwc(xu8, dt, chunk_elems(dt))
        y, _ = netG.fwd_nhwc(x, st, save=False)
        out[lo:hi] = ops.nhwc_to_u8(y, 3).cpu().numpy()
    return out


def run(model, in_path, out_path, batch=256, rank=0, world=1):
    arrays = load_arrays(in_path)
    check_inputs(arrays, model.opt.state_dim)
    n = arrays["image_observations"].shape[0]
    lo, hi = shard_range(n, rank, world)
    tp1 = generate_tp1(model.netG, arrays["image_observations"][lo:hi], arrays["next_observations"][lo:hi], batch)
    if world == 1:
        result = dict(arrays)
    else:   # each rank writes its own contiguous slice of every per-row key
        result = {k: (v[lo:hi] if getattr(v, "shape", ()) and v.shape[0] == n else v) for k, v in arrays.items()}
        root, ext = os.path.splitext(out_path)
        out_path = "%s.part%d_of_%d%s" % (root, rank, world, ext)
    result["image_observations_tp1"] = tp1
    save_arrays(out_path, result)
    return out_path, (lo, hi)
