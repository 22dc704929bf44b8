"""Entry point with the reference's signature (src/train.py:18): builds model / optimizer / loss / Solver from a
config object and dispatches the mode.  Dataset construction is out of scope: dataloaders are
passed in (any iterable of the reference's 6-tuples, src/distrib.py:97)."""
import random

import numpy as np
import torch

from .distrib import get_model, get_optimizer, get_loss_function
from .solver import Solver
from .utils import load_yaml


def main(obj_config, return_solver=False, mode="train", device="gpu", train_dataloader=None, validation_dataloader=None,
         test_dataloader=None, writer=None):
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
    else:
        raise ValueError(f"mode '{mode}' is outside the train-step scope")
    return solver
