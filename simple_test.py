"""simple_test.py -- N-step generation demo (README.md:33):
    python simple_test.py --env_type=cheetah --dataroot=./datasets --netG=s2p --start_idx=0 --seq_len=5 --gpu_ids=0
Loads ./checkpoints/<env_type>_<epoch>.pth (README.md:19-26), rolls the generator out autoregressively from frame
`start_idx` and writes a generated-vs-ground-truth strip to --results_dir."""
import os

import numpy as np
import torch

from s2p_amd.data import S2PDataset, tensor_to_images
from s2p_amd.metrics import image_metrics
from s2p_amd.models.pix2pix_model import Pix2PixModel
from s2p_amd.options.test_options import TestOptions
from s2p_amd.rollout import rollout


def main(args=None):
    opt = TestOptions().parse(args)
    model = Pix2PixModel(opt)
    ds = S2PDataset(opt)
    frames, states = ds.sequence(opt.start_idx, opt.seq_len)
    gen = rollout(model.netG, frames[:1], states[1:].unsqueeze(0))[0]          # [T,3,H,W]
    torch.cuda.synchronize()
    gt = frames[1:]
    l1 = float((gen.cpu() - gt).abs().mean())
    print("N-step generation: env=%s start_idx=%d seq_len=%d  mean|gen-gt|=%.4f" % (opt.env_type, opt.start_idx, opt.seq_len, l1))
    if gen.shape[-1] >= 11 and gen.shape[-2] >= 11:          # the paper's fidelity metrics, per generated step (on device)
        psnr, ssim = image_metrics(gen, gt.to(gen.device))
        print("  PSNR per step (dB):", " ".join("%.2f" % v for v in psnr.tolist()))
        print("  SSIM per step     :", " ".join("%.4f" % v for v in ssim.tolist()))
    os.makedirs(opt.results_dir, exist_ok=True)
    top = np.concatenate(list(tensor_to_images(gen)), axis=1)
    bot = np.concatenate(list(tensor_to_images(gt)), axis=1)
    strip = np.concatenate([top, bot], axis=0)
    path = os.path.join(opt.results_dir, "%s_start%d_len%d.png" % (opt.env_type, opt.start_idx, opt.seq_len))
    try:
        from PIL import Image
        Image.fromarray(strip).save(path)
    except ImportError:
        path = path[:-4] + ".npy"
        np.save(path, strip)
    print("saved", path)
    return gen


if __name__ == "__main__":
    main()
