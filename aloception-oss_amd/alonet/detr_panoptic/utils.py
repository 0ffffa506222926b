"""Selection of the object queries that get a mask (reference: alonet/detr_panoptic/utils.py:7-50)."""
import torch


def get_mask_queries(frames, m_outputs, model, matcher=None, filters=None, **kwargs):
    """Keep, per image, the decoder outputs of the selected queries (score filter of the detector, or the matcher's
    assignment during training) and zero-pad to the largest count: -> ((B, max_kept, C), list of boolean filters)."""
    dec = m_outputs["dec_outputs"][-1]
    if filters is None:
        if matcher is None:
            filters = model.get_outs_filter(m_outputs=m_outputs, **kwargs)
        else:
            filters = [torch.zeros(dec.size(1), dtype=torch.bool, device=dec.device) for _ in range(len(dec))]
            for b, (src, _) in enumerate(matcher(m_outputs=m_outputs, frames=frames, **kwargs)):
                filters[b][src] = True
    sizes = [int(f.sum()) for f in filters]
    width = max(sizes)
    rows = [torch.cat([dec[b:b + 1, f], dec.new_zeros(1, width - n, dec.shape[2])], dim=1)
            for b, (f, n) in enumerate(zip(filters, sizes))]
    return torch.cat(rows, dim=0), filters
