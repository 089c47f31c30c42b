"""Bulk augmentation caller (N1): CPU tests of the on-disk contract, GPU test of the generated frames."""
import os

import numpy as np
import pytest
import torch

import s2p_oracle as O
from s2p_amd import augment as A


def _fake_rollout_file(path, n=10, size=20, S=17, seed=0):
    rng = np.random.default_rng(seed)
    arrays = dict(image_observations=rng.integers(0, 256, (n, size, size, 3), dtype=np.uint8),
                  observations=rng.normal(size=(n, S)).astype(np.float32),
                  next_observations=rng.normal(size=(n, S)).astype(np.float32),
                  actions=rng.uniform(-1, 1, (n, 6)).astype(np.float32),
                  rewards=rng.normal(size=(n,)).astype(np.float32),
                  timeouts=np.zeros(n, bool),
                  slac_observation_indices=np.arange(n * 9).reshape(n, 9))
    np.savez(path, **arrays)
    return arrays


def test_shard_ranges_partition_the_rows():
    for n in (0, 1, 7, 50000, 50001):
        for w in (1, 2, 3, 8):
            r = [A.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_input_contract_errors(tmp_path):
    arr = _fake_rollout_file(os.path.join(tmp_path, "in.npz"))
    A.check_inputs(arr, 17)
    with pytest.raises(KeyError):
        A.check_inputs({k: v for k, v in arr.items() if k != "next_observations"}, 17)
    with pytest.raises(ValueError):
        A.check_inputs(dict(arr, image_observations=arr["image_observations"].astype(np.float32)), 17)
    with pytest.raises(ValueError):
        A.check_inputs(arr, 24)
    with pytest.raises(ValueError):
        A.check_inputs(dict(arr, image_observations=arr["image_observations"][:, :18]), 17)
    with pytest.raises(RuntimeError, match="h5py"):
        A.save_arrays(os.path.join(tmp_path, "o.hdf5"), arr)


@pytest.mark.gpu
def test_u8_round_trip_is_exact(hip_device):
    from s2p_amd import ops
    u8 = torch.arange(256, dtype=torch.uint8).repeat(3).view(1, 16, 16, 3).contiguous().to(hip_device)
    for dt, pitch in ((torch.float32, 4), (torch.bfloat16, 8)):
        x = ops.u8_to_nhwc(u8, dt, pitch)
        assert float(x[..., 3:].float().abs().max()) == 0.0 and float(x.float().min()) >= -1 and float(x.float().max()) <= 1
        ref = u8.float() / 127.5 - 1.0
        assert float((x[..., :3].float() - ref).abs().max()) <= (1e-6 if dt == torch.float32 else 4e-3)
        assert torch.equal(ops.nhwc_to_u8(x, 3), u8)          # exact inverse on all 256 values, both dtypes


@pytest.mark.gpu
def test_augment_writes_tp1_like_the_oracle(hip_device, tmp_path):
    from s2p_amd.models.pix2pix_model import Pix2PixModel
    from s2p_amd.options.test_options import TestOptions
    from test_model_gpu import randomize
    inp, outp = os.path.join(tmp_path, "in.npz"), os.path.join(tmp_path, "out.npz")
    arr = _fake_rollout_file(inp, n=10, size=20)
    opt = TestOptions().parse(["--env_type", "cheetah", "--gpu_ids", "0", "--random_init", "--precision", "fp32",
                               "--checkpoints_dir", str(tmp_path)], quiet=True)
    model = Pix2PixModel(opt)
    spec = O.Spec()
    pg = randomize(O.init_params(O.generator_param_shapes(spec), 1), 11, 1.0)
    model.netG.load_state_dict(pg)
    path, (lo, hi) = A.run(model, inp, outp, batch=4)
    out = dict(np.load(path))
    assert (lo, hi) == (0, 10)
    for k, v in arr.items():                                   # every input key is passed through untouched
        assert np.array_equal(out[k], v), k
    tp1 = out["image_observations_tp1"]
    assert tp1.dtype == np.uint8 and tp1.shape == arr["image_observations"].shape      # NHWC uint8, as the consumer reads
    x = torch.from_numpy(arr["image_observations"]).permute(0, 3, 1, 2).float() / 127.5 - 1.0
    with torch.no_grad():
        y = O.generator_forward(pg, x, torch.from_numpy(arr["next_observations"]), spec)
    ref = ((y.clamp(-1, 1) + 1.0) * 127.5).round().to(torch.uint8).permute(0, 2, 3, 1).numpy()
    diff = np.abs(tp1.astype(np.int16) - ref.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (int(diff.max()), float((diff > 0).mean()))   # fp32: <=1 LSB on <0.1 % of pixels
    # sharded run: two "ranks" reproduce the same rows
    p0, r0 = A.run(model, inp, outp, batch=4, rank=0, world=2)
    p1, r1 = A.run(model, inp, outp, batch=4, rank=1, world=2)
    cat = np.concatenate([np.load(p0)["image_observations_tp1"], np.load(p1)["image_observations_tp1"]], 0)
    assert r0 == (0, 5) and r1 == (5, 10) and cat.shape == tp1.shape
    d2 = np.abs(cat.astype(np.int16) - tp1.astype(np.int16))      # IN moments use fp32 atomics: last-bit run-to-run variation
    assert d2.max() <= 1 and (d2 > 0).mean() < 1e-3
