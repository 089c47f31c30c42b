"""Model factory (SPADE-lineage `models.create_model`; README.md:73)."""
import importlib


def find_model_using_name(model_name):
    modellib = importlib.import_module("s2p_amd.models." + model_name + "_model")
    target = model_name.replace("_", "") + "model"
    for name, cls in modellib.__dict__.items():
        if name.lower() == target and isinstance(cls, type):
            return cls
    raise ValueError(f"In {modellib.__name__}, there should be a class whose lower-cased name is {target}")


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    model = find_model_using_name(opt.model)(opt)
    return model
