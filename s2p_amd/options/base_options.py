"""BaseOptions: the SPADE-lineage option surface the S2P CLI exposes (README.md:33,59 pin `--env_type --dataroot
--netG --start_idx --seq_len --gpu_ids --batchSize`; the rest follows the lineage defaults)."""
import argparse
import os
import pickle
import sys

import torch

ENV_STATE_DIM = {"cheetah": 17, "walker": 24}   # dmc2gym observation sizes (cheetah verified: SURVEY.md App. C)


class BaseOptions:
    def __init__(self):
        self.initialized = False
        self.isTrain = False

    def initialize(self, parser):
        # experiment specifics
        parser.add_argument("--name", type=str, default="s2p", help="name of the experiment")
        parser.add_argument("--env_type", type=str, default="cheetah", help="cheetah | walker (selects state_dim and checkpoint name)")
        parser.add_argument("--state_dim", type=int, default=0, help="state vector size; 0 = from --env_type")
        parser.add_argument("--gpu_ids", type=str, default="0", help="gpu ids: e.g. 0  0,1,2. -1 is refused (no CPU fallback)")
        parser.add_argument("--checkpoints_dir", type=str, default="./checkpoints", help="models are saved here")
        parser.add_argument("--model", type=str, default="pix2pix", help="which model to use")
        parser.add_argument("--norm_G", type=str, default="matinstance", help="MAT (state+image modulated) instance norm")
        parser.add_argument("--norm_D", type=str, default="instance", help="instance normalization in D")
        parser.add_argument("--phase", type=str, default="train", help="train, val, test, etc")
        parser.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"],
                            help="compute dtype of the HIP kernels (fp32 accumulate either way)")
        # input/output sizes
        parser.add_argument("--batchSize", type=int, default=1, help="input batch size (per GPU/process)")
        parser.add_argument("--load_size", type=int, default=84, help="frames are resized (nearest) to this size")
        parser.add_argument("--crop_size", type=int, default=84, help="frame size fed to the networks")
        parser.add_argument("--output_nc", type=int, default=3, help="# of output image channels")
        # data
        parser.add_argument("--dataroot", type=str, default="./datasets")
        parser.add_argument("--dataset_mode", type=str, default="s2p")
        parser.add_argument("--serial_batches", action="store_true", help="if true, takes samples in order")
        parser.add_argument("--nThreads", default=0, type=int, help="# threads for loading data")
        parser.add_argument("--max_dataset_size", type=int, default=sys.maxsize)
        # generator
        parser.add_argument("--netG", type=str, default="s2p", help="selects model to use for netG (s2p)")
        parser.add_argument("--ngf", type=int, default=64, help="# of gen filters in first conv layer")
        parser.add_argument("--init_type", type=str, default="xavier", help="network initialization [normal|xavier|kaiming|orthogonal]")
        parser.add_argument("--init_variance", type=float, default=0.02, help="variance of the initialization distribution")
        parser.add_argument("--z_dim", type=int, default=256, help="dimension of the state embedding w")
        self.initialized = True
        return parser

    def gather_options(self, args=None):
        if not self.initialized:
            parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
            parser = self.initialize(parser)
        opt, _ = parser.parse_known_args(args)
        # model- and network-specific options (plugin hooks)
        from .. import models
        from ..models import networks
        parser = models.get_option_setter(opt.model)(parser, self.isTrain)
        _real_parse = parser.parse_known_args
        parser.parse_known_args = lambda a=None, n=None: _real_parse(args if a is None else a, n)
        parser = networks.modify_commandline_options(parser, self.isTrain)
        parser.parse_known_args = _real_parse
        self.parser = parser
        return parser.parse_args(args)

    def print_options(self, opt):
        message = "----------------- Options ---------------\n"
        for k, v in sorted(vars(opt).items()):
            default = self.parser.get_default(k)
            comment = "\t[default: %s]" % str(default) if v != default else ""
            message += "{:>25}: {:<30}{}\n".format(str(k), str(v), comment)
        message += "----------------- End -------------------"
        print(message)

    def save_options(self, opt):
        expr_dir = os.path.join(opt.checkpoints_dir, opt.name)
        os.makedirs(expr_dir, exist_ok=True)
        with open(os.path.join(expr_dir, "opt.txt"), "wt") as f:
            for k, v in sorted(vars(opt).items()):
                f.write("{:>25}: {}\n".format(str(k), str(v)))
        with open(os.path.join(expr_dir, "opt.pkl"), "wb") as f:
            pickle.dump(opt, f)

    def parse(self, args=None, save=False, quiet=False):
        opt = self.gather_options(args)
        opt.isTrain = self.isTrain
        if opt.state_dim <= 0:
            if opt.env_type not in ENV_STATE_DIM:
                raise ValueError("unknown --env_type %s: pass --state_dim explicitly" % opt.env_type)
            opt.state_dim = ENV_STATE_DIM[opt.env_type]
        if not quiet:
            self.print_options(opt)
        if opt.isTrain and save:
            self.save_options(opt)
        # set gpu ids
        str_ids = str(opt.gpu_ids).split(",")
        opt.gpu_ids = [int(s) for s in str_ids if int(s) >= 0]
        # one process per GPU (torchrun): LOCAL_RANK overrides the id, gpu_ids keeps single-process semantics
        local_rank = os.environ.get("LOCAL_RANK")
        if local_rank is not None and len(opt.gpu_ids) > 0:
            opt.gpu_ids = [int(local_rank)]
        if len(opt.gpu_ids) > 1:
            raise ValueError("multi-GPU runs use one process per GPU: launch with `python -m torch.distributed.run "
                             "--nproc-per-node N train.py ...` instead of --gpu_ids 0,1,...")
        self.opt = opt
        return self.opt
