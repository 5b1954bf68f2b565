"""trainer.py of the reference (trainer.py:13-95) with a device-agnostic ``_set_device`` (the reference's compares a list to
-1 and always yields cuda:N).  The reference's own trainer.py also works unchanged with this directory first
on sys.path (tests/test_plugin_surface.py)."""
import copy
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from lpi_amd.retrieval.utils import factory  # noqa: E402
from lpi_amd.retrieval.utils.toolkit import count_parameters  # noqa: E402


def train(args):
    seed_list = copy.deepcopy(args['seed'])
    device = copy.deepcopy(args['device'])
    for seed in seed_list:
        args['seed'] = seed
        args['device'] = device
        _train(args)


def _train(args):
    logging.basicConfig(level=logging.INFO, format='%(asctime)s [%(filename)s] => %(message)s',
                        handlers=[logging.StreamHandler(sys.stdout)])
    _set_random()
    _set_device(args)
    for key, value in args.items():
        logging.info('{}: {}'.format(key, value))
    model = factory.get_model(args['model_name'], args)
    logging.info('All params: {}'.format(count_parameters(model._network)))
    logging.info('Trainable params: {}'.format(count_parameters(model._network, True)))
    model.incremental_train()
    model.after_task()
    return model


def _set_device(args):
    gpus = []
    for device in args['device']:
        if isinstance(device, torch.device):
            gpus.append(device)
        elif str(device) in ("-1", "cpu"):
            gpus.append(torch.device('cpu'))
        else:
            gpus.append(torch.device('cuda:{}'.format(device)))
    args['device'] = gpus


def _set_random():
    torch.manual_seed(1)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(1)
