"""HIP-graph replay of a model's inference forward.

A DETR-family forward at fixed input shape is ~330 kernel launches of 2-250 us; eagerly launched, the GPU idles a few
microseconds between dependent small kernels.  Captured once in a HIP graph (``torch.cuda.CUDAGraph`` is a hipGraph on
ROCm) the same kernels replay with one host call and back-to-back dispatch: 10.1 -> 9.7 ms per step of DeformableDETR-R50
on 8 frames of 1333x800 (MI355X), bit-identical outputs.  No tracing compiler is involved: the graph is the recorded launch
sequence of the eager code, hand-written kernels included.
"""
import weakref

import torch

# Every live wrapper of a model, so that one wrapper's capture does not re-derive (free) the tensors another one's live graphs
# replay on.  Kept OUTSIDE the module object (a weak set inside ``model.__dict__`` made ``torch.save(model)`` fail on the weak
# references and ``copy.deepcopy(model)`` carry ghost wrappers over: round-4 advisor finding).
_WRAPPERS = weakref.WeakKeyDictionary()   # model -> WeakSet of GraphedForward


def _wrappers_of(model):
    return _WRAPPERS.get(model, ())


class GraphedForward:
    """``GraphedForward(model)(*frames, **options)`` == ``model(*frames, **options)`` under ``torch.no_grad()`` for batched
    ``aloscene.Frame`` inputs (one for the detectors, two for RAFT) and hashable keyword options (``iters=32``).

    The first call with a new (shapes, dtypes, device, options) warms the model up on a side stream, captures one forward on
    buffers of its own and replays it; later calls copy the frames (data and padding mask) into the captured inputs and replay.
    The returned tensors are the graph's output buffers: they are overwritten by the next call with the same key, so
    consume them (``model.inference(out)``) before calling again.  Anything that synchronises with the host
    (``inference()``, ``.cpu()``) stays outside the captured region.
    """

    def __init__(self, model, warmup=3, adopt_inputs=False):
        """``adopt_inputs``: capture on the caller's own frame objects instead of on clones — a caller that refills those very
        buffers with each new batch (an H2D copy lands there) then replays without any device-to-device copy."""
        self.model = model
        self.warmup = warmup
        self.adopt_inputs = adopt_inputs
        self._graphs = {}
        self._epoch = 0
        _WRAPPERS.setdefault(model, weakref.WeakSet()).add(self)

    def reset(self):
        """Forget every captured graph (the next call captures again)."""
        self._graphs.clear()

    @staticmethod
    def _key(frames, options):
        return (tuple((tuple(f.shape), f.dtype, str(f.device)) for f in frames), tuple(sorted(options.items())))

    def _capture(self, frames, options):
        import alo_hip

        # Derived tensors (packed / folded / merged weights) are allocated in eager warm-ups, outside any graph's private pool, and
        # every captured graph has their addresses baked in.  So they are re-derived only when NO graph of this object is alive —
        # dropping them under a live graph would leave it replaying on freed memory.  Weight surgery later on goes through
        # alo_hip.invalidate_caches(model) (load_weights calls it), which bumps the model's cache epoch: __call__ then drops
        # every graph and captures again.
        others_live = any(w._graphs and w._epoch == alo_hip.cache_epoch(self.model)
                          for w in _wrappers_of(self.model) if w is not self)
        if not self._graphs and not others_live:
            alo_hip.invalidate_caches(self.model)  # weights edited through .data since the last forward: re-derive before pinning
        # (with another wrapper's graphs alive on the current epoch the derived tensors are kept: re-deriving them would bump the
        # epoch, that wrapper would re-capture on its next call and invalidate this one in turn — a re-capture on every alternation)
        if not self._graphs:
            self._epoch = alo_hip.cache_epoch(self.model)
        device = frames[0].device
        static_in = tuple(frames) if self.adopt_inputs else tuple(f.clone() for f in frames)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # lazy caches (folded / packed weights, level geometry, solver choices) fill here
                self.model(*static_in, **options)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (the RCCL watchdog of a torch.distributed job polls events) may keep
        # making HIP calls while this thread captures
        with torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
            out = self.model(*static_in, **options)
        return static_in, graph, out

    def __call__(self, *frames, **options):
        if not frames or not all(f.is_cuda for f in frames):
            raise RuntimeError("GraphedForward: needs CUDA frames")
        import alo_hip

        if self._graphs and alo_hip.cache_epoch(self.model) != self._epoch:
            self._graphs.clear()   # the tensors those graphs read were invalidated (load_weights, invalidate_caches): capture again
        key = self._key(frames, options)
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._graphs[key] = self._capture(frames, options)
        static_in, graph, out = entry
        for new, static in zip(frames, static_in):
            if new is static:
                continue
            static.as_tensor().copy_(new.as_tensor(), non_blocking=True)
            if getattr(new, "mask", None) is not None and getattr(static, "mask", None) is not None:
                static.mask.as_tensor().copy_(new.mask.as_tensor(), non_blocking=True)
        graph.replay()
        return out
