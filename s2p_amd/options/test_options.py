from .base_options import BaseOptions


class TestOptions(BaseOptions):
    __test__ = False      # not a pytest test class
    def initialize(self, parser):
        BaseOptions.initialize(self, parser)
        parser.add_argument("--results_dir", type=str, default="./results/", help="saves results here.")
        parser.add_argument("--which_epoch", type=str, default="30", help="checkpoint epoch: loads <checkpoints_dir>/<env_type>_<epoch>.pth")
        parser.add_argument("--start_idx", type=int, default=0, help="index of the first frame of the rollout")
        parser.add_argument("--seq_len", type=int, default=5, help="number of autoregressive generation steps")
        parser.add_argument("--random_init", action="store_true", help="run with seeded random weights when no checkpoint exists")
        parser.set_defaults(serial_batches=True, phase="test", precision="fp32")
        self.isTrain = False
        return parser
