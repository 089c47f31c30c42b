"""Network plugin registry (SPADE-lineage `models.networks`): `--netG s2p` resolves to the
class whose lower-cased name is 's2p' + 'generator' in `generator.py`, etc."""
import importlib

import torch

from .base_network import BaseNetwork


def find_network_using_name(target_network_name, filename):
    target_class_name = target_network_name + filename
    module = importlib.import_module("s2p_amd.models.networks." + filename)
    network = None
    for name, cls in module.__dict__.items():
        if name.lower() == target_class_name.replace("_", "").lower() and isinstance(cls, type):
            network = cls
    if network is None:
        raise ValueError("In %s, there should be a class whose lower-cased name is %s" % (module.__name__, target_class_name))
    assert issubclass(network, BaseNetwork), "Class %s should be a subclass of BaseNetwork" % network
    return network


def modify_commandline_options(parser, is_train):
    opt, _ = parser.parse_known_args()
    netG_cls = find_network_using_name(opt.netG, "generator")
    parser = netG_cls.modify_commandline_options(parser, is_train)
    if is_train:
        netD_cls = find_network_using_name(opt.netD, "discriminator")
        parser = netD_cls.modify_commandline_options(parser, is_train)
    return parser


def create_network(cls, opt):
    net = cls(opt)
    net.print_network()
    net.init_weights(opt.init_type, opt.init_variance)
    return net


def define_G(opt):
    return create_network(find_network_using_name(opt.netG, "generator"), opt)


def define_D(opt):
    return create_network(find_network_using_name(opt.netD, "discriminator"), opt)
