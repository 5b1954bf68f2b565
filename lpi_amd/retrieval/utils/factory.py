"""utils/factory.py:1-7 of the reference: model_name -> learner."""
from lpi_amd.retrieval.methods.sprompt import SPrompts


def get_model(model_name, args):
    name = model_name.lower()
    options = {'sprompts': SPrompts}
    return options[name](args)
