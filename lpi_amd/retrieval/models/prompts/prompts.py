"""DecomposedPrompt — same constructor, parameter names and outputs as the reference's module
(models/prompts/prompts.py:4-57); the CP reconstruction and its backward run in HIP kernels
(lpi_prompt_cp_fwd/bwd).  ``r`` is a real argument here (the reference leaves it at 4)."""
import torch
from torch import nn

from lpi_amd.functional import DecomposedPromptFn


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
