"""DecomposedPrompt — same constructor, parameter names and outputs as the reference's module
(models/prompts/prompts.py:4-57); the CP reconstruction and its backward run in HIP kernels
(lpi_prompt_cp_fwd/bwd).  ``r`` is a real argument here (the reference leaves it at 4)."""
import math

import torch
from torch import nn

from lpi_amd.functional import DecomposedPromptFn, InteractFn


class DecomposedPrompt(nn.Module):
    def __init__(self, layer_num, prompt_num, prompt_depth_vis, prompt_depth_text, r=4):
        super().__init__()
        self.d = r
        # five factors ~ N(0, 0.5^2)  (prompts.py:15-25)
        self.dim_1_share = nn.Parameter(torch.randn(layer_num, r) * 0.5)
        self.dim_2_visual = nn.Parameter(torch.randn(prompt_num, r) * 0.5)
        self.dim_2_textual = nn.Parameter(torch.randn(prompt_num, r) * 0.5)
        self.dim_3_visual = nn.Parameter(torch.randn(prompt_depth_vis, r) * 0.5)
        self.dim_3_textual = nn.Parameter(torch.randn(prompt_depth_text, r) * 0.5)
        self.scale = 1

    def forward(self):
        return DecomposedPromptFn.apply(self.dim_1_share, self.dim_2_visual, self.dim_2_textual, self.dim_3_visual,
                                        self.dim_3_textual, float(self.scale))


class InteractModule(nn.Module):
    """The low-rank cross-modal interaction of LPI's grounding branch (grounding/maskrcnn_benchmark/modeling/bert/modeling_bert.py:558-651; the
    optional item (f4) of SURVEY section 8): same constructor, parameter names (dim_{1,2,3}_{v2t,t2v}, visual_norm, textual_norm), initialisation
    (kaiming_uniform_(a = sqrt(5)) on all six factors, :602-608) and forward(visual_out, textual_out, layer_id) -> (visual_out, textual_out) as the
    reference; the arithmetic runs in lpi_interact_fwd / lpi_interact_bwd in the rank-r form (the reference materialises the
    [layer_num, D + 1, D', r] product on every call)."""

    def __init__(self, layer_num=12, visual_dim=96, textual_dim=768, r=4):
        super().__init__()
        self.d = r
        self.visual_dim, self.textual_dim = visual_dim, textual_dim
        self.dim_1_v2t = nn.Parameter(torch.empty(layer_num, r))
        self.dim_2_v2t = nn.Parameter(torch.empty(visual_dim + 1, r))
        self.dim_3_v2t = nn.Parameter(torch.empty(textual_dim, r))
        self.dim_1_t2v = nn.Parameter(torch.empty(layer_num, r))
        self.dim_2_t2v = nn.Parameter(torch.empty(textual_dim + 1, r))
        self.dim_3_t2v = nn.Parameter(torch.empty(visual_dim, r))
        self.visual_norm = nn.LayerNorm(visual_dim)
        self.textual_norm = nn.LayerNorm(textual_dim)
        self.scale = 1
        self.reset_parameters()

    def reset_parameters(self):
        for p in (self.dim_1_v2t, self.dim_2_v2t, self.dim_3_v2t, self.dim_1_t2v, self.dim_2_t2v, self.dim_3_t2v):
            nn.init.kaiming_uniform_(p, a=math.sqrt(5))

    def forward(self, visual_out, textual_out, layer_id):
        return InteractFn.apply(visual_out, textual_out, int(layer_id), self.dim_1_v2t, self.dim_2_v2t, self.dim_3_v2t, self.dim_1_t2v, self.dim_2_t2v,
                                self.dim_3_t2v, self.visual_norm.weight, self.visual_norm.bias, self.textual_norm.weight, self.textual_norm.bias,
                                0.1, self.visual_norm.eps)
