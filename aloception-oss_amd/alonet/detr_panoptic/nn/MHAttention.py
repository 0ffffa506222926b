"""Per-head attention MAP between object queries and an encoder feature map (no value multiplication).

Reference: alonet/detr_panoptic/nn/MHAttention.py:10-47.  ``weights[b,q,n,h,w] = softmax_{n,h,w}( <q_lin(q)[b,q,n,:],
k_lin(k)[b,n,:,h,w]> / sqrt(d_head) )`` with padded pixels at -inf.  Here the per-head dot products are one batched
matrix product per image (hipBLASLt) instead of a broadcast multiply + sum over a (B,Q,N,d,H,W) temporary.
"""
import torch
import torch.nn.functional as F
from torch import nn


class MHAttentionMap(nn.Module):
    def __init__(self, query_dim, hidden_dim, num_heads, dropout=0.0, bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.hidden_dim = hidden_dim
        self.dropout = nn.Dropout(dropout)
        self.q_linear = nn.Linear(query_dim, hidden_dim, bias=bias)
        self.k_linear = nn.Linear(query_dim, hidden_dim, bias=bias)
        nn.init.zeros_(self.k_linear.bias)
        nn.init.zeros_(self.q_linear.bias)
        nn.init.xavier_uniform_(self.k_linear.weight)
        nn.init.xavier_uniform_(self.q_linear.weight)
        self.normalize_fact = float(hidden_dim / self.num_heads) ** -0.5

    def forward(self, q, k, mask=None):
        """q (B,Q,C), k (B,C,H,W), mask (B,H,W) bool -> (B,Q,num_heads,H,W)."""
        B, Q, _ = q.shape
        H, W = k.shape[-2:]
        n, d = self.num_heads, self.hidden_dim // self.num_heads
        qh = self.q_linear(q).view(B, Q, n, d).permute(0, 2, 1, 3)                      # (B,n,Q,d)
        kh = F.conv2d(k, self.k_linear.weight[:, :, None, None], self.k_linear.bias).view(B, n, d, H * W)
        weights = torch.matmul(qh * self.normalize_fact, kh).permute(0, 2, 1, 3).reshape(B, Q, n, H, W)
        if mask is not None:
            weights = weights.masked_fill(mask[:, None, None], float("-inf"))
        weights = F.softmax(weights.flatten(2), dim=-1).view(B, Q, n, H, W)
        return self.dropout(weights)
