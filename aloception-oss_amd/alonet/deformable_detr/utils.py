import torch


def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0.0, max=1.0)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))
