"""Drop-in for the host-side helpers of the reference's utils.py that the hot path needs
(import_class utils.py:143-148, save_arg utils.py:191-196).  The numpy Dataset lives in sar_amd.data."""
import os

import yaml


def import_class(name):
    """'models.stgcn' -> module, 'models.resnet.Model' -> class (sub-modules are imported on demand)."""
    import importlib
    components = name.split('.')
    mod = importlib.import_module(components[0])
    for i, comp in enumerate(components[1:], 1):
        if not hasattr(mod, comp):
            importlib.import_module('.'.join(components[:i + 1]))
        mod = getattr(mod, comp)
    return mod


def save_arg(arg):
    arg_dict = vars(arg)
    os.makedirs(arg.log_dir, exist_ok=True)
    with open(os.path.join(arg.log_dir, "config.yaml"), 'w') as f:
        yaml.dump(arg_dict, f)
