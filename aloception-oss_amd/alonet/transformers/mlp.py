"""Plain multi-layer perceptron used by the DETR-family heads (reference: alonet/transformers/mlp.py)."""
import torch.nn.functional as F
from torch import nn

import alo_hip


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        if alo_hip.fusable(x, self.layers[0].weight):
            # inference on the GPU: the ReLU rides in the GEMM epilogue (streaming MFMA kernel or hipBLASLt)
            for layer in self.layers[:-1]:
                x = alo_hip.linear_auto(x, layer.weight, layer.bias, relu=True)
            return self.layers[-1](x)
        for layer in self.layers[:-1]:
            x = F.relu(layer(x))
        return self.layers[-1](x)
