"""RAFT correlation block on the gfx950 kernels: all-pairs volume + pyramid on the fp16 matrix pipe at fp32 accuracy (scaled
two-term operand split, three products — csrc/corr.hip), windowed lookup as a gather kernel.

Drop-in for the reference's ``CorrBlock`` (alonet/raft/corr.py:12-60) through RAFT's ``corr_block=`` constructor
hook (alonet/raft/raft.py:47-60,168,185): ``CorrBlock(fmap1, fmap2, num_levels=4, radius=4)`` builds
``corr_pyramid`` (list of ``(B*H*W, 1, h_l, w_l)`` float32 tensors) and ``corr_fn(coords)`` returns the
``(B, num_levels*(2r+1)^2, H, W)`` float32 window features.  ``AlternateCorrBlock`` is not provided: the reference's
version needs the absent third-party ``alt_cuda_corr`` extension and is unreachable (corr.py:5-9,86).

Differentiability.  The reference's block is plain autograd-able torch code, so RAFT can be fine-tuned through it.  Here, when
the feature maps require a gradient:

* the pyramid is built by the HIP kernel inside ``_BuildFunction``; its backward is the matmul's own gradient written out —
  ``dF1 = P_l(F2) . dC_l^T / sqrt(C)``, ``dF2 = P_l^T(F1 . dC_l) / sqrt(C)`` summed over the levels that received a gradient, ``P_l`` =
  l-fold 2x2 mean of the feature map (a mean over a cell of the correlation is the correlation with the cell's mean feature) — as
  library GEMMs (``torch.bmm``) over the gradient maps: the volume is never re-evaluated and no second copy of it is held;
* a lookup runs the HIP kernel inside ``_LookupFunction``; its backward is the kernel's exact adjoint
  (``alo_corr_lookup_backward``), ACCUMULATED into gradient maps shared by all lookups of the block (RAFT looks the pyramid up 32
  times; autograd through ``grid_sample`` would materialise a dense gradient of the whole volume for each).  The maps reach
  ``_BuildFunction.backward`` through a scalar token every lookup depends on, not as dense autograd gradients;
* a gradient with respect to the COORDINATES (RAFT detaches them, raft.py:186; the reference's block is differentiable there too)
  is a kernel as well (``alo_corr_lookup_backward_coords``: the forward's gather with the horizontal differences kept, one map per
  level);
* ``corr_pyramid`` tensors used directly in somebody's own graph receive dense gradients the ordinary way; they are added to the
  accumulated maps.
Without gradients nothing of this exists: ``CorrBlock`` is two kernel calls.
"""
import math

import warnings

import torch
import torch.nn.functional as F

import alo_hip

from .utils.utils import bilinear_sampler


# ---- the torch formulation: what the reference computes, op for op (``TorchCorrBlock``; the tests' yardstick for the gradients) ------
def pyramid_torch(fmap1, fmap2, num_levels=4):
    """corr.py:13-27,52-60: ``<fmap1[b,:,i], fmap2[b,:,j]> / sqrt(C)`` as ``(B*H*W, 1, H, W)`` + ``num_levels - 1`` 2x2 means."""
    B, C, H, W = fmap1.shape
    vol = torch.matmul(fmap1.reshape(B, C, H * W).transpose(1, 2), fmap2.reshape(B, C, H * W)) / math.sqrt(C)
    pyramid = [vol.reshape(B * H * W, 1, H, W)]
    for _ in range(num_levels - 1):
        pyramid.append(F.avg_pool2d(pyramid[-1], 2, stride=2))
    return pyramid


def lookup_torch(pyramid, coords, radius):
    """corr.py:29-50: a (2r+1)^2 window around ``coords / 2^l`` on every level; the window's FIRST axis moves x (the reference
    stacks ``meshgrid(dy, dx)`` onto (x, y) coordinates), bilinear with corners aligned and zeros outside."""
    B, _, H, W = coords.shape
    r = radius
    centre = coords.permute(0, 2, 3, 1).reshape(B * H * W, 1, 1, 2)
    steps = torch.linspace(-r, r, 2 * r + 1, device=coords.device, dtype=coords.dtype)
    window = torch.stack(torch.meshgrid(steps, steps, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    feats = [bilinear_sampler(vol, centre / 2 ** lvl + window).view(B, H, W, -1) for lvl, vol in enumerate(pyramid)]
    return torch.cat(feats, dim=-1).permute(0, 3, 1, 2).contiguous().float()


class TorchCorrBlock:
    """The same block on stock torch ops only (any device, differentiable end to end): ``RAFT(corr_block=TorchCorrBlock)``."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        self.corr_pyramid = pyramid_torch(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        return lookup_torch(self.corr_pyramid, coords, self.radius)

    @staticmethod
    def corr(fmap1, fmap2):
        B, _, H, W = fmap1.shape
        return pyramid_torch(fmap1, fmap2, 1)[0].view(B, H, W, 1, H, W)


# ---- HIP forward and backward ----------------------------------------------------------------------------------------------------
def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


# The per-pass bookkeeping below stands on two private hooks of the autograd engine; they are resolved HERE, at import, so that a torch
# build without them fails with a clear message instead of an AttributeError in the middle of a backward pass.
_graph_task_id = getattr(torch._C, "_current_graph_task_id", None)
_queue_callback = getattr(getattr(torch.autograd.Variable, "_execution_engine", None), "queue_callback", None)
if _graph_task_id is None or _queue_callback is None:
    raise ImportError("alonet.raft.corr needs torch._C._current_graph_task_id and the autograd engine's queue_callback "
                      f"(present in torch 2.x; this is torch {torch.__version__})")


class _PyramidState:
    """What the build node and the lookup nodes of one CorrBlock share: the pyramid and, during a backward pass, the gradient maps
    the lookups accumulate into."""

    def __init__(self):
        self.pyramid = None
        self.grad = None
        self.grad_pass = None

    @staticmethod
    def _current_pass():
        """Identity of the autograd pass that is running (the engine's graph-task id; -1 outside a backward)."""
        return _graph_task_id()

    def grad_maps(self):
        """The maps of the backward pass that is running.  They belong to THAT pass: the first lookup backward of a pass creates
        them, tags them with the engine's graph-task id and queues an end-of-pass callback that drops whatever the build node did
        not consume — a pass that never reaches the build node (``autograd.grad(loss, coords)``) must not leave volume-sized maps
        behind to be added to the next pass's gradients.  The engine skips its callbacks when a pass aborts on an exception, so
        maps tagged by another pass are also discarded here, when the next pass first asks for them."""
        now = self._current_pass()
        if self.grad is None or self.grad_pass != now:
            self.grad = [torch.zeros_like(p) for p in self.pyramid]
            self.grad_pass = now
            _queue_callback(self._end_of_pass)
        return self.grad

    def take_grad_maps(self):
        """What the lookups of the running pass accumulated (None if none did); the state is left empty."""
        maps, tag = self.grad, self.grad_pass
        self.grad = self.grad_pass = None
        if maps is not None and tag != self._current_pass():
            # lookups ran in another pass than the build node (a nested / re-entrant backward): their maps are not this pass's
            warnings.warn("CorrBlock: gradient maps accumulated by another autograd pass were discarded; the feature-map gradients of "
                          "this pass do not include those lookups (re-entrant backward through a CorrBlock is not supported)",
                          RuntimeWarning, stacklevel=2)
            return None
        return maps

    def _end_of_pass(self):
        self.grad = self.grad_pass = None


def _pooled_chain(fmap, num_levels):
    """[fmap, 2x2 mean of it, ...]: what the pyramid's level l correlates fmap1 with (avg_pool2d drops an odd last row / column on
    the volume and on the feature map alike)."""
    chain = [fmap]
    for _ in range(num_levels - 1):
        chain.append(F.avg_pool2d(chain[-1], 2, stride=2))
    return chain


class _BuildFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap1, fmap2, num_levels, state):
        ctx.save_for_backward(fmap1, fmap2)
        ctx.num_levels = num_levels
        ctx.state = state
        state.pyramid = alo_hip.corr_build(fmap1, fmap2, num_levels)
        token = fmap1.new_zeros(())   # every lookup takes it as an input: autograd runs them all before this node's backward
        # the outputs are VIEWS: autograd hangs this node on them, and the state (which this node holds) must not hold the node back
        return (token,) + tuple(p.view_as(p) for p in state.pyramid)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _gtoken, *grads):
        fmap1, fmap2 = ctx.saved_tensors
        state = ctx.state
        sparse = state.take_grad_maps()   # a later backward pass through the same graph starts from zero again
        totals = []
        for lvl in range(ctx.num_levels):
            parts = [g for g in (grads[lvl], sparse[lvl] if sparse is not None else None) if g is not None]
            totals.append(None if not parts else (parts[0] if len(parts) == 1 else parts[0] + parts[1]))
        if all(t is None for t in totals):
            return None, None, None, None
        B, C, H, W = fmap1.shape
        scale = 1.0 / math.sqrt(C)
        f1 = fmap1.reshape(B, C, H * W)
        g1 = torch.zeros_like(f1) if ctx.needs_input_grad[0] else None
        g2 = None
        with torch.enable_grad():
            leaf = fmap2.detach().requires_grad_(ctx.needs_input_grad[1])
            chain = _pooled_chain(leaf, ctx.num_levels)
        heads, head_grads = [], []
        for lvl, t in enumerate(totals):
            if t is None:
                continue
            n = chain[lvl].shape[-2] * chain[lvl].shape[-1]
            dvol = t.reshape(B, H * W, n)                                   # d loss / d vol_l[b, i, j]
            if g1 is not None:                                              # sum_j dvol[i, j] * P_l(F2)[c, j]
                g1.baddbmm_(chain[lvl].detach().reshape(B, C, n), dvol.transpose(1, 2), alpha=scale)
            if ctx.needs_input_grad[1]:                                     # sum_i dvol[i, j] * F1[c, i], then back through the means
                heads.append(chain[lvl])
                head_grads.append((torch.bmm(f1, dvol) * scale).reshape(chain[lvl].shape))
        if heads:
            (g2,) = torch.autograd.grad(heads, leaf, head_grads)
        return (g1.reshape(fmap1.shape) if g1 is not None else None), g2, None, None


class _LookupFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, token, radius, state):
        ctx.save_for_backward(coords)
        ctx.radius = radius
        ctx.state = state
        return alo_hip.corr_lookup(state.pyramid, coords, radius)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        (coords,) = ctx.saved_tensors
        state = ctx.state
        gcoords = None
        if ctx.needs_input_grad[0]:
            gcoords = alo_hip.corr_lookup_backward_coords(state.pyramid, coords, grad_out, ctx.radius)
        if ctx.needs_input_grad[1]:
            alo_hip.corr_lookup_backward(state.grad_maps(), coords, grad_out, ctx.radius)
        return gcoords, (coords.new_zeros(()) if ctx.needs_input_grad[1] else None), None, None


class _LookupDenseFunction(torch.autograd.Function):
    """A lookup into pyramid tensors that did not come out of ``_BuildFunction`` under autograd (somebody's own differentiable
    pyramid, or coordinates that want a gradient while the features do not): the same two backward kernels, dense gradient maps."""

    @staticmethod
    def forward(ctx, coords, radius, *pyramid):
        ctx.save_for_backward(coords, *pyramid)
        ctx.radius = radius
        return alo_hip.corr_lookup(list(pyramid), coords, radius)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        coords, *pyramid = ctx.saved_tensors
        gcoords = alo_hip.corr_lookup_backward_coords(pyramid, coords, grad_out, ctx.radius) if ctx.needs_input_grad[0] else None
        gpyr = [None] * len(pyramid)
        if any(ctx.needs_input_grad[2:]):   # dense maps: this pyramid is somebody's own tensors, autograd wants a gradient per tensor
            maps = alo_hip.corr_lookup_backward([torch.zeros_like(p) for p in pyramid], coords, grad_out, ctx.radius)
            gpyr = [m if need else None for m, need in zip(maps, ctx.needs_input_grad[2:])]
        return (gcoords, None) + tuple(gpyr)


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        self._state = self._token = None
        if _needs_grad(fmap1, fmap2):
            self._state = _PyramidState()
            self._token, *pyramid = _BuildFunction.apply(fmap1.contiguous(), fmap2.contiguous(), num_levels, self._state)
            self.corr_pyramid = list(pyramid)
        else:
            self.corr_pyramid = alo_hip.corr_build(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        coords = coords.float()
        if (self._state is not None and torch.is_grad_enabled() and len(self.corr_pyramid) == len(self._state.pyramid)
                and all(a.data_ptr() == b.data_ptr() and a.shape == b.shape for a, b in zip(self.corr_pyramid, self._state.pyramid))):
            return _LookupFunction.apply(coords.contiguous(), self._token, self.radius, self._state)
        if _needs_grad(coords, *self.corr_pyramid):
            return _LookupDenseFunction.apply(coords.contiguous(), self.radius, *self.corr_pyramid)
        return alo_hip.corr_lookup(self.corr_pyramid, coords, self.radius)

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation only: (B,C,H,W) x2 -> (B,H,W,1,H,W), scaled by 1/sqrt(C)."""
        B, _, H, W = fmap1.shape
        fmap1, fmap2 = fmap1.float(), fmap2.float()
        if _needs_grad(fmap1, fmap2):
            _, vol = _BuildFunction.apply(fmap1.contiguous(), fmap2.contiguous(), 1, _PyramidState())
        else:
            (vol,) = alo_hip.corr_build(fmap1, fmap2, 1)
        return vol.view(B, H, W, 1, H, W)
