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
    if dec.is_cuda and len(filters) == dec.shape[0] and all(torch.is_tensor(f) and f.dtype == torch.bool and f.dim() == 1 for f in filters):
        # ONE device-to-host transfer of the filters (the per-image `int(f.sum())` and `dec[b, f]` of the straightforward form are two
        # synchronisations per image, in the middle of the forward), then one gather: same rows, same zero padding
        host = torch.stack([f.to(dec.device) for f in filters]).cpu()
        sizes = host.sum(1).tolist()
        width = max(sizes)
        index = torch.zeros((len(filters), width), dtype=torch.long)
        valid = torch.zeros((len(filters), width), dtype=torch.bool)
        for b, h in enumerate(host):
            index[b, :sizes[b]] = torch.nonzero(h).flatten()
            valid[b, :sizes[b]] = True
        index, valid = index.to(dec.device, non_blocking=True), valid.to(dec.device, non_blocking=True)
        rows = dec.gather(1, index.unsqueeze(-1).expand(-1, -1, dec.shape[2]))
        rows = torch.where(valid.unsqueeze(-1), rows, rows.new_zeros(()))      # exact zeros in the padding, whatever row 0 holds
        return rows, filters
    sizes = [int(f.sum()) for f in filters]
    width = max(sizes)
    rows = [torch.cat([dec[b:b + 1, f], dec.new_zeros(1, width - n, dec.shape[2])], dim=1)
            for b, (f, n) in enumerate(zip(filters, sizes))]
    return torch.cat(rows, dim=0), filters
