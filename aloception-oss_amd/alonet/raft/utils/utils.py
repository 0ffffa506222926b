"""Small RAFT helpers (counterparts of alonet/raft/utils/utils.py:21-30)."""
import torch
import torch.nn.functional as F


def coords_grid(batch, ht, wd, device=None):
    """(batch, 2, ht, wd) float grid: channel 0 = x (column index), channel 1 = y (row index)."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def upflow8(flow, mode="bilinear"):
    new_size = (8 * flow.shape[2], 8 * flow.shape[3])
    return 8 * F.interpolate(flow, size=new_size, mode=mode, align_corners=True)
