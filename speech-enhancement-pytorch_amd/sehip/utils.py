"""Config objects and device helper with the reference's surface (src/utils.py:92-115 prepare_device,
:135-208 load_yaml / Config / dict2obj / obj2dict)."""
import yaml
import torch


class Config:
    """Attribute-access view of a nested dict (the reference's dict2obj result)."""

    def __repr__(self):
        return f"Config({obj2dict(self)})"


def dict2obj(d):
    if isinstance(d, dict):
        o = Config()
        for k, v in d.items():
            setattr(o, k, dict2obj(v))
        return o
    if isinstance(d, list):
        return [dict2obj(v) for v in d]
    return d


def obj2dict(o):
    if isinstance(o, Config):
        return {k: obj2dict(v) for k, v in o.__dict__.items()}
    if isinstance(o, list):
        return [obj2dict(v) for v in o]
    return o


def load_yaml(path, *args, **kwargs):
    with open(path, "r") as f:
        cfg = yaml.safe_load(f)
    cfg["root"] = path
    return dict2obj(cfg)


def set_deterministic(on=True):
    """Process-wide deterministic reductions of libsehip (sehip_set_deterministic, include/sehip.h): what the reference's
    `cudnn_deterministic` flag asks of cuDNN (src/utils.py:108-111) asked of this library."""
    from ._lib import call
    call("sehip_set_deterministic", 1 if on else 0)


def prepare_device(n_gpu, cudnn_deterministic=False):
    """src/utils.py:92-115: 'cuda:0' when GPUs are visible, else CPU.  cudnn_deterministic (src/conf/config.yaml:130 ships True):
    there is no cuDNN here; the flag selects libsehip's deterministic schedule instead (fixed-order reductions, set_deterministic)."""
    if n_gpu == 0:
        return torch.device("cpu")
    if cudnn_deterministic:
        print("Using libsehip's deterministic schedule in the experiment.")
        set_deterministic(True)
    return torch.device("cuda:0")
