"""PromptLearner — host side of the text front end (models/clip/prompt_learner.py:66-218): builds
``"X X ... X" + " " + caption + "."`` , tokenises to [B,77] ids and hands them to the HIP text front end, which does the
embedding lookup, the ctx splice over positions 1..n_ctx (CLASS_TOKEN_POSITION == "end") and the positional add in one
kernel (lpi_txt_embed_fwd).  Accepts a LongTensor of ready token ids in place of the caption list (bench / tests)."""
import torch
from torch import nn

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


class PromptLearner(nn.Module):
    def __init__(self, cfg, context_length=77, ctx_dim=None):
        super().__init__()
        if cfg.CLASS_TOKEN_POSITION != "end":
            raise ValueError("only CLASS_TOKEN_POSITION='end' (every LPI config) is built")   # prompt_learner.py:155-163
        if cfg.CTXINIT:
            raise ValueError("CTXINIT must be '' (LPI configs): ctx comes from the DecomposedPrompt")
        self.n_ctx = cfg.NCTX
        self.prompt_prefix = " ".join(["X"] * self.n_ctx)
        self.context_length = context_length
        self.n_cls = None
        if ctx_dim is not None:
            # the module's own context vectors (prompt_learner.py:100-110: normal(std 0.02), "to be optimized"): never read on the LPI path — the ctx comes
            # from the DecomposedPrompt (slinet.py:130) — and frozen by the trainable filter (sprompt.py:230-237); kept because they are part of the
            # reference network's state (12 x [n_ctx, d_t] of the 149.78 M parameters trainer.py:50 logs)
            self.ctx = nn.Parameter(torch.randn(self.n_ctx, ctx_dim) * 0.02, requires_grad=False)

    def token_ids(self, captions) -> torch.Tensor:
        if torch.is_tensor(captions):
            return captions
        self.n_cls = len(captions)
        prompts = [self.prompt_prefix + " " + c + "." for c in captions]          # prompt_learner.py:131
        return tokenize(get_tokenizer(), prompts, self.context_length)

    forward = token_ids
