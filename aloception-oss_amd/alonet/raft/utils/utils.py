"""Small RAFT helpers (counterparts of alonet/raft/utils/utils.py:5-60)."""
import torch
import torch.nn.functional as F


def bilinear_sampler(img, coords, mode="bilinear", mask=False):
    """``grid_sample`` addressed in PIXEL coordinates (reference utils.py:5-19): ``img (N, C, H, W)``, ``coords
    (N, Ho, Wo, 2)`` with (x, y) in pixels, corners aligned.  With ``mask=True`` also returns the float mask of the
    coordinates strictly inside the map.  ``mode`` is accepted and ignored, as in the reference.  The HIP ``CorrBlock`` lookup reproduces exactly this round trip
    (pixel -> [-1, 1] -> pixel) in its kernel; this torch form serves callers outside the correlation path."""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    grid = torch.cat([2 * xgrid / (W - 1) - 1, 2 * ygrid / (H - 1) - 1], dim=-1)
    out = F.grid_sample(img, grid, align_corners=True)  # always bilinear: the reference accepts `mode` and ignores it (utils.py:14)
    if mask:
        inside = (grid[..., :1] > -1) & (grid[..., 1:] > -1) & (grid[..., :1] < 1) & (grid[..., 1:] < 1)
        return out, inside.float()
    return out


def coords_grid(batch, ht, wd, device=None):
    """(batch, 2, ht, wd) float grid: channel 0 = x (column index), channel 1 = y (row index)."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def upflow8(flow, mode="bilinear"):
    new_size = (8 * flow.shape[2], 8 * flow.shape[3])
    return 8 * F.interpolate(flow, size=new_size, mode=mode, align_corners=True)


class Padder:
    """Replicate-pads a frame so that its height and width are multiples of 8 (what RAFT's 1/8-resolution encoders
    need) and crops results back (reference utils.py:33-60).  The reference derives BOTH paddings' target from the
    height (``((h // 8) + 1) * 8 - w``); kept as is so that padded sizes — and therefore flows — match it."""

    def _set_pad(self, frame):
        self.h, self.w = frame.HW
        self.pad_h = (((self.h // 8) + 1) * 8 - self.h) % 8
        self.pad_w = (((self.h // 8) + 1) * 8 - self.w) % 8
        self.top = self.pad_h // 2
        self.bottom = self.pad_h - self.top
        self.left = self.pad_w // 2
        self.right = self.pad_w - self.left

    def pad(self, frame):
        """Pads the frame's pixels (not its labels); returns a tensor of the frame's type."""
        self._set_pad(frame)
        names = getattr(frame, "names", None)
        data = frame.as_tensor() if hasattr(frame, "as_tensor") else frame
        lead = data.dim() < 4  # F.pad(mode="replicate") wants (N, C, H, W)
        out = F.pad(data[None] if lead else data, [self.left, self.right, self.top, self.bottom], mode="replicate")
        out = out[0] if lead else out
        if hasattr(frame, "as_tensor"):
            # the pixels are padded, the frame's labels and properties are not: they ride along unchanged, as with the
            # reference's F.pad on the augmented tensor ("pad frame but not its labels", utils.py:46-51)
            res = out.as_subclass(type(frame))
            return res._inherit(frame, names) if hasattr(res, "_inherit") else res
        return out

    def unpad(self, tensor):
        h, w = tensor.shape[-2:]
        return tensor[..., self.top: h - self.bottom, self.left: w - self.right]
