"""L2Norm parameter holder (layers/modules/l2norm.py:5-21 of the reference).  Inside a model the
normalisation runs in libtdrn_hip's l2norm kernel; called stand-alone it is the same formula in torch."""
import torch
import torch.nn as nn


class L2Norm(nn.Module):
    def __init__(self, n_channels, scale):
        super(L2Norm, self).__init__()
        self.n_channels, self.gamma, self.eps = n_channels, scale or None, 1e-10
        self.weight = nn.Parameter(torch.full((n_channels,), float(scale)))

    def forward(self, x):
        norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + self.eps
        return self.weight.view(1, -1, 1, 1) * (x / norm)
