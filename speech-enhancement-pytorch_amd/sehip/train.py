"""Entry point with the reference's signature (src/train.py:18): builds model / optimizer / loss / Solver from a
config object and dispatches the mode.  Dataset construction is out of scope: dataloaders are
passed in (any iterable of the reference's 6-tuples, src/distrib.py:97)."""
import random

import numpy as np
import torch

from ._lib import SehipError
from .distrib import get_model, get_optimizer, get_loss_function
from .solver import Solver
from .utils import load_yaml


def main(obj_config, return_solver=False, mode="train", save=False, dev=False, device="gpu", train_dataloader=None,
         validation_dataloader=None, test_dataloader=None, writer=None):
    """The reference's positional order (src/train.py:18-23: obj_config, return_solver, mode, save, dev, device); the dataloaders and the
    scalar writer are keyword extras.  mode: "train" -> Solver.train(); "validation" -> Solver._run_one_epoch(1, 1, train=False)
    (src/train.py:87-90); "test" is the inference / metric path (src/solver.py:534-746), outside the train-step scope: named error.
    `save` / `dev` only matter to that test mode (save enhanced wavs, Clarity development set) and are accepted for call compatibility."""
    if not isinstance(device, (str, torch.device)):
        raise TypeError(f"main(): device must be 'gpu' / 'cpu' / a torch.device, got {device!r} (positional order: obj_config, "
                        f"return_solver, mode, save, dev, device)")
    config = load_yaml(obj_config) if isinstance(obj_config, str) else obj_config
    torch.manual_seed(config.seed)
    np.random.seed(config.seed)
    random.seed(config.seed)
    model = get_model(config.model)
    optimizer = get_optimizer(config.optim, model)
    loss_function = get_loss_function(config.optim, device=device)
    solver = Solver(config=config, model=model, optimizer=optimizer, loss_function=loss_function,
                    train_dataloader=train_dataloader, validation_dataloader=validation_dataloader,
                    test_dataloader=test_dataloader, device=device, writer=writer)
    if return_solver:
        return solver
    if mode == "train":
        solver.train()
    elif mode == "validation":
        solver._run_one_epoch(1, 1, train=False)
    elif mode == "test":
        raise SehipError("main(mode='test'): Solver.inference (metrics, plots, saved wavs; src/solver.py:534-746) is outside the "
                         "train-step scope of sehip (SURVEY section 8); use sehip.evaluate.evaluate + sehip.metric.SI_SDR on the device")
    else:
        raise ValueError(f"unknown mode '{mode}' (train | validation | test)")
    return solver
