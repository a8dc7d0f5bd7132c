""""TransT" blocks of HandTrackNet (counterpart of the reference's transformer.py).

The reference computes multi-head attention in every block and then throws the result away
when called with attn=False (transformer.py:72-82) -- which is how HandTrackNet always calls it
(hand_network.py:139-140).  With `elide_dead=True` (default) the discarded attention, the
position embedding that only feeds it, and whole blocks whose output is never read
(TransT.s12, TransT.c12: ~3.2 GMAC/frame at N=1024) are skipped.  Outputs are bit-identical in
eval mode; in train mode only the dropout RNG stream differs (statistically equivalent).
All parameters are still created so reference checkpoints load and state_dicts match.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn


class attn_module(nn.Module):
    def __init__(self, d_model=384, no_linear=False, only_pos=False, qk_mask=None, nhead=8, dim_feedforward=1024,
                 dropout=0.1, activation="relu", concat=False):
        super().__init__()
        if concat:
            self.attn = nn.MultiheadAttention(72, nhead, vdim=d_model, dropout=dropout)
            self.newlq = nn.Linear(d_model, 72)
            self.newlk = nn.Linear(d_model, 72)
            self.outlv = nn.Linear(72, d_model)
        else:
            self.attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.no_linear, self.only_pos, self.qk_mask, self.concat = no_linear, only_pos, qk_mask, concat
        if not no_linear:
            self.linear1 = nn.Linear(d_model, dim_feedforward)
            self.linear2 = nn.Linear(dim_feedforward, d_model)
            self.dropout2 = nn.Dropout(dropout)
            self.dropout3 = nn.Dropout(dropout)
            self.norm2 = nn.LayerNorm(d_model)
            self.activation = _get_activation_fn(activation)

    @staticmethod
    def with_pos_embed(tensor, pos: Optional[Tensor]):
        return tensor if pos is None else tensor + pos

    def forward(self, src1_ori, pos1_ori, src2_ori, pos2_ori, attn=True, elide_dead=True):
        """src1 (B,C,N) queries, src2 (B,C,M) keys/values, pos* matching embeddings -> (B,C,N)."""
        src1 = src1_ori.permute(2, 0, 1)  # (N,B,C)
        if attn or not elide_dead:
            src2 = src2_ori.permute(2, 0, 1)
            pos1 = None if pos1_ori is None else pos1_ori.permute(2, 0, 1)
            pos2 = None if pos2_ori is None else pos2_ori.permute(2, 0, 1)
            if self.concat:
                out, _ = self.attn(self.with_pos_embed(self.newlq(src1), pos1),
                                   self.with_pos_embed(self.newlk(src2), pos2), value=src2, attn_mask=self.qk_mask)
                mixed = src1 + self.outlv(self.dropout1(out))
            else:
                out, _ = self.attn(self.with_pos_embed(src1, pos1), self.with_pos_embed(src2, pos2), value=src2,
                                   attn_mask=self.qk_mask)
                mixed = src1 + self.dropout1(out)
            x = mixed if attn else src1
        else:
            x = src1
        x = self.norm1(x)
        if not self.no_linear:
            x = self.norm2(x + self.dropout3(self.linear2(self.dropout2(self.activation(self.linear1(x))))))
        return x.permute(1, 2, 0)


class TransT(nn.Module):
    def __init__(self, d_model=384, concat=False):
        super().__init__()
        self.s11 = attn_module(d_model=d_model, no_linear=True, concat=concat)
        self.s12 = attn_module(d_model=d_model, no_linear=True, concat=concat)
        self.c11 = attn_module(d_model=d_model, concat=concat)
        self.c12 = attn_module(d_model=d_model, concat=concat)

    def forward(self, src1, pos1, src2, pos2, attn, elide_dead=True, need_result2=True):
        """Returns (result1, result2).  With attn=False and elide_dead, result1 depends on src1
        only; result2 (only ever consumed as keys/values of a discarded attention) is computed
        only if need_result2."""
        dead = elide_dead and not attn
        src11 = self.s11(src1, pos1, src1, pos1, attn, elide_dead)
        src12 = None
        if not dead or need_result2:
            src12 = self.s12(src2, pos2, src2, pos2, attn, elide_dead)
        result1 = self.c11(src11, pos1, src12, pos2, attn, elide_dead)
        result2 = None
        if not dead or need_result2:
            result2 = self.c12(src12, pos2, src11, pos1, attn, elide_dead)
        return result1, result2


class PositionEmbeddingSine(nn.Module):
    """Sin/cos embedding of batch-normalised coordinates (reference transformer.py:89-123), on
    the input's own device (the reference hard-codes .cuda(), :110)."""

    def __init__(self, num_pos_feats=64, normalize=True):
        super().__init__()
        if normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.normalize = normalize

    def forward(self, coor: Tensor) -> Tensor:
        """coor (B,3,N) -> (B, 6*num_pos_feats, N)."""
        lo, hi = coor.min(), coor.max()
        normal = 2 * ((coor - lo) / (hi - lo)) - 1
        freqs = math.pi * (2 ** torch.arange(self.num_pos_feats, dtype=torch.float, device=coor.device))
        k = normal.unsqueeze(-1) * freqs  # (B,3,N,D)
        x = torch.cat([torch.sin(k), torch.cos(k)], -1)  # (B,3,N,2D)
        return x.transpose(-1, -2).reshape(coor.shape[0], -1, coor.shape[-1])


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")
