"""Learner registry — the entry the reference's trainer calls (`factory.get_model(args['model_name'], args)`, trainer.py:46).
Unknown names raise KeyError, as indexing the reference's dict does (utils/factory.py:7)."""
from lpi_amd.retrieval.methods.sprompt import SPrompts

_LEARNERS = {"sprompts": SPrompts}      # the HIP-backed LPI / S-Prompts continual retrieval learner


def get_model(model_name, args):
    learner = _LEARNERS[str(model_name).lower()]
    return learner(args)
