from .base_options import BaseOptions


class TrainOptions(BaseOptions):
    def initialize(self, parser):
        BaseOptions.initialize(self, parser)
        # displays / saving
        parser.add_argument("--print_freq", type=int, default=100, help="frequency of showing training results on console")
        parser.add_argument("--save_latest_freq", type=int, default=5000, help="frequency of saving the latest results")
        parser.add_argument("--save_epoch_freq", type=int, default=10, help="frequency of saving checkpoints at the end of epochs")
        parser.add_argument("--continue_train", action="store_true", help="continue training: load the latest model")
        parser.add_argument("--allow_unsigned_optimizer_state", action="store_true",
                            help="--continue_train: also accept optG / optD states written before the flat-layout signature existed "
                                 "(only if the writing code is known to use this build's parameter order)")
        parser.add_argument("--which_epoch", type=str, default="latest", help="which epoch to load? set to latest to use latest cached model")
        # training
        parser.add_argument("--niter", type=int, default=30, help="# of iter at starting learning rate")
        parser.add_argument("--niter_decay", type=int, default=0, help="# of iter to linearly decay learning rate to zero")
        parser.add_argument("--optimizer", type=str, default="adam")
        parser.add_argument("--beta1", type=float, default=0.0, help="momentum term of adam")
        parser.add_argument("--beta2", type=float, default=0.9, help="momentum term of adam")
        parser.add_argument("--no_TTUR", action="store_true", help="Use TTUR training scheme")
        parser.add_argument("--lr", type=float, default=0.0002, help="initial learning rate for adam")
        parser.add_argument("--D_steps_per_G", type=int, default=1, help="number of discriminator iterations per generator iterations.")
        # discriminators
        parser.add_argument("--ndf", type=int, default=64, help="# of discrim filters in first conv layer")
        parser.add_argument("--lambda_feat", type=float, default=10.0, help="weight for feature matching loss")
        parser.add_argument("--lambda_vgg", type=float, default=10.0, help="weight for vgg loss")
        parser.add_argument("--lambda_l1", type=float, default=10.0, help="weight for the pixel L1 loss (rebuttal.md:71)")
        parser.add_argument("--no_ganFeat_loss", action="store_true", help="if specified, do *not* use discriminator feature matching loss")
        parser.add_argument("--no_vgg_loss", action="store_true", help="if specified, do *not* use VGG feature matching loss")
        parser.add_argument("--vgg_weights", type=str, default="", help="torchvision vgg19 state_dict file; empty = seeded stand-in weights")
        parser.add_argument("--gan_mode", type=str, default="hinge", help="(hinge)")
        parser.add_argument("--netD", type=str, default="multiscale", help="(n_layers|multiscale)")
        parser.add_argument("--hip_graph", action="store_true", help="capture the G+D train step in a hipGraph")
        self.isTrain = True
        return parser
