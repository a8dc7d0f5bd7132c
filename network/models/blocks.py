"""Small dense blocks used by HandTrackNet (counterpart of the reference's blocks.py:226-239)."""
from __future__ import annotations

import torch
import torch.nn as nn


class rearrange_module(nn.Module):
    """Mixes each keypoint's feature with four kinematic-neighbour permutations of the 21
    keypoints (child, parent, two cross-finger neighbours) and projects 5C -> C with a 1x1 conv.
    Parameter name `linear` as in the reference."""

    # neighbour tables of the 21-joint hand skeleton (wrist 0, four joints per finger)
    _CHILD = [1, 2, 3, 4, 4, 6, 7, 8, 8, 10, 11, 12, 12, 14, 15, 16, 16, 18, 19, 20, 20]
    _PARENT = [17, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19]
    _PREV_FINGER = [1, 1, 2, 3, 4, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    _NEXT_FINGER = [17, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 17, 18, 19, 20]

    def __init__(self, channel=384, add_points=False, re=5):
        super().__init__()
        self.re = re
        self.linear = nn.Conv1d(channel * re, channel, 1)
        perm = torch.tensor([list(range(21)), self._CHILD, self._PARENT, self._PREV_FINGER, self._NEXT_FINGER])
        self.register_buffer("_perm", perm, persistent=False)

    def forward(self, new_points: torch.Tensor, fast: bool = False) -> torch.Tensor:
        """(B, C, 21) -> (B, C, 21).  fast (eval on GPU): one token-major GEMM instead of a Conv1d call."""
        B, C, J = new_points.shape
        if fast:
            tok = new_points.transpose(1, 2)[:, self._perm.t()]  # (B, 21, 5, C)
            w = self.linear.weight.squeeze(-1)  # (C, 5C)
            return torch.nn.functional.linear(tok.reshape(B * J, self.re * C), w, self.linear.bias).view(B, J, C).transpose(1, 2)
        stacked = new_points[:, :, self._perm]  # (B, C, 5, 21)
        stacked = stacked.permute(0, 2, 1, 3).reshape(B, self.re * C, J)
        return self.linear(stacked)
