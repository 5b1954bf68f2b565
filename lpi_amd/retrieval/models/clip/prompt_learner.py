"""PromptLearner — host side of the text front end (models/clip/prompt_learner.py:66-218): builds
``"X X ... X" + " " + caption + "."`` , tokenises to [B,77] ids and hands them to the HIP text front end, which does the
embedding lookup, the ctx splice over positions 1..n_ctx (CLASS_TOKEN_POSITION == "end") and the positional add in one
kernel (lpi_txt_embed_fwd).  Accepts a LongTensor of ready token ids in place of the caption list (bench / tests)."""
import torch

from .simple_tokenizer import NativeTokenizer, SimpleTokenizer, tokenize

_tokenizer = None


def get_tokenizer():
    """The C++ tokenizer of liblpi_hip.so (lpi_bpe_*); LPI_TOKENIZER=python selects the Python implementation of the same scheme
    (both are validated against ids captured from the reference, tests/test_tokenizer.py)."""
    global _tokenizer
    if _tokenizer is None:
        import os
        _tokenizer = SimpleTokenizer() if os.environ.get("LPI_TOKENIZER", "native") == "python" else NativeTokenizer()
    return _tokenizer


class cfgc(object):
    backbonename = 'ViT-B/16'
    NCTX = 16
    CTXINIT = ''
    CSC = False
    CLASS_TOKEN_POSITION = 'end'


class PromptLearner:
    def __init__(self, cfg, context_length=77):
        if cfg.CLASS_TOKEN_POSITION != "end":
            raise ValueError("only CLASS_TOKEN_POSITION='end' (every LPI config) is built")   # prompt_learner.py:155-163
        if cfg.CTXINIT:
            raise ValueError("CTXINIT must be '' (LPI configs): ctx comes from the DecomposedPrompt")
        self.n_ctx = cfg.NCTX
        self.prompt_prefix = " ".join(["X"] * self.n_ctx)
        self.context_length = context_length
        self.n_cls = None

    def token_ids(self, captions) -> torch.Tensor:
        if torch.is_tensor(captions):
            return captions
        self.n_cls = len(captions)
        prompts = [self.prompt_prefix + " " + c + "." for c in captions]          # prompt_learner.py:131
        return tokenize(get_tokenizer(), prompts, self.context_length)

    __call__ = token_ids
