"""FAVOR+ attention module (reference: networks/fast_attention.py:159-205).

Same constructor, same `projection_matrix` buffer (drawn with the same generator
consumption, fast_attention.py:117-146) and the same forward(q, k, v) on [T,H,N,d]
tensors; the arithmetic is mlhot_favor_fwd/_bwd.
"""
import math

import torch
from torch import nn

from mlhot.ops import FavorFunction


def orthogonal_matrix_chunk(cols, device=None):
    block = torch.randn((cols, cols), device=device)
    q, _ = torch.linalg.qr(block.cpu(), mode="reduced")
    return q.to(device).t()


def gaussian_orthogonal_random_matrix(nb_rows, nb_columns, scaling=0, device=None):
    n_full = int(nb_rows / nb_columns)
    blocks = [orthogonal_matrix_chunk(nb_columns, device=device) for _ in range(n_full)]
    rem = nb_rows - n_full * nb_columns
    if rem > 0:
        blocks.append(orthogonal_matrix_chunk(nb_columns, device=device)[:rem])
    final = torch.cat(blocks)
    if scaling == 0:
        mult = torch.randn((nb_rows, nb_columns), device=device).norm(dim=1)
    elif scaling == 1:
        mult = math.sqrt(float(nb_columns)) * torch.ones((nb_rows,), device=device)
    else:
        raise ValueError(f"Invalid scaling {scaling}")
    return torch.diag(mult) @ final


class FastAttention(nn.Module):
    def __init__(self, dim_heads, nb_features=None, ortho_scaling=0, causal=False, generalized_attention=False,
                 kernel_fn=None, no_projection=False):
        super().__init__()
        if causal or generalized_attention or no_projection:
            raise NotImplementedError("mlhot implements the non-causal softmax-kernel FAVOR+ path the "
                                      "reference models use (fast_attention.py:196-204)")
        self.dim_heads = dim_heads
        self.nb_features = nb_features if nb_features is not None else int(dim_heads * math.log(dim_heads))
        self.ortho_scaling = ortho_scaling
        self.causal = causal
        self.register_buffer("projection_matrix",
                             gaussian_orthogonal_random_matrix(self.nb_features, dim_heads, scaling=ortho_scaling))

    @torch.no_grad()
    def redraw_projection_matrix(self, device):
        self.projection_matrix.copy_(gaussian_orthogonal_random_matrix(
            self.nb_features, self.dim_heads, scaling=self.ortho_scaling, device=device))

    def forward(self, q, k, v):
        """q [T,H,Nq,d], k/v [T,H,Nc,d] -> [T,H,Nq,d] (the reference's layout)."""
        T, H, Nq, d = q.shape
        merged = FavorFunction.apply(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3),
                                     self.projection_matrix)
        return merged.view(T, Nq, d, H).permute(0, 3, 1, 2)
