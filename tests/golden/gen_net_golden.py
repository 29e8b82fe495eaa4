"""Generates tests/golden/net_golden_*.npz: an INDEPENDENT PyTorch (CPU, float64 -> float32 outputs)
implementation of the architecture of AlphaZero/cpp_weighted/model.hpp:9-95, seeded weights, a few boards,
and the resulting policy/value.  The reference net is libtorch C++ and cannot be imported here, and no
reference test pins net outputs (SURVEY 8c "net numerics: parity unpinned at the reference"), so these
vectors pin the oracle's C net and the HIP net to PyTorch's own conv/batch-norm/linear/softmax kernels.

    python tests/golden/gen_net_golden.py
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class WeightedPolicy(nn.Module):  # model.hpp:9-50
    def __init__(self, n, C, num_block):
        super().__init__()
        n2 = n * n
        self.n = n
        self.conv = nn.Conv2d(2, C, 3, padding=1)
        self.norm = nn.BatchNorm2d(C)
        self.tower = nn.ModuleList()
        for _ in range(num_block - 1):
            self.tower.append(nn.ModuleList([nn.Conv2d(C, C, 3, padding=1), nn.BatchNorm2d(C), nn.Conv2d(C, C, 3, padding=1), nn.BatchNorm2d(C)]))
        self.policy_conv = nn.Conv2d(C, 2, 1)
        self.policy_norm = nn.BatchNorm2d(2)
        self.policy_linear = nn.Linear(2 * n2, n2)
        self.value_conv = nn.Conv2d(C, 1, 1)
        self.value_norm = nn.BatchNorm1d(n2)
        self.value_linear1 = nn.Linear(n2, 256)
        self.value_linear2 = nn.Linear(256, 1)

    def forward(self, player, board):  # model.hpp:52-95
        n, n2 = self.n, self.n * self.n
        p = player.view(-1, 1).repeat(1, n2).view(-1, 1, n, n)
        x = torch.cat([p, board.view(-1, 1, n, n)], 1)
        x = F.relu(self.norm(self.conv(x)))
        for c0, n0, c1, n1 in self.tower:
            t = F.relu(n0(c0(x)))
            t = n1(c1(t))
            x = F.relu(t + x)
        pol = F.relu(self.policy_norm(self.policy_conv(x)))
        pol = torch.softmax(self.policy_linear(pol.view(-1, 2 * n2)), 1)
        v = self.value_conv(x)
        v = F.relu(self.value_norm(v.view(-1, n2)))
        v = F.relu(self.value_linear1(v))
        v = torch.tanh(self.value_linear2(v))
        return pol, v


def blob_of(m):
    """flat f32 in the documented order (conv.weight, conv.bias, BN gamma, beta, running_mean, running_var, ...)"""
    parts = []

    def bn(b):
        parts.extend([b.weight, b.bias, b.running_mean, b.running_var])

    parts.extend([m.conv.weight, m.conv.bias]); bn(m.norm)
    for c0, n0, c1, n1 in m.tower:
        parts.extend([c0.weight, c0.bias]); bn(n0)
        parts.extend([c1.weight, c1.bias]); bn(n1)
    parts.extend([m.policy_conv.weight, m.policy_conv.bias]); bn(m.policy_norm)
    parts.extend([m.policy_linear.weight, m.policy_linear.bias])
    parts.extend([m.value_conv.weight, m.value_conv.bias]); bn(m.value_norm)
    parts.extend([m.value_linear1.weight, m.value_linear1.bias, m.value_linear2.weight, m.value_linear2.bias])
    return torch.cat([p.detach().reshape(-1) for p in parts]).to(torch.float32).numpy()


def make(n, C, blocks, num, seed, path):
    torch.manual_seed(seed)
    m = WeightedPolicy(n, C, blocks)
    with torch.no_grad():  # non-trivial running statistics so that the BN folding is exercised
        for mod in m.modules():
            if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
    m.eval()
    g = torch.Generator().manual_seed(seed + 1)
    boards = torch.randint(-1, 2, (num, n * n), generator=g)
    player = torch.randint(0, 2, (num,), generator=g) * 2 - 1
    blob = blob_of(m)
    m64 = m.double()
    with torch.no_grad():
        pol, val = m64(player.double(), boards.double())
    np.savez_compressed(path, n=n, C=C, blocks=blocks, blob=blob, boards=boards.numpy().astype(np.int8),
                        player=player.numpy().astype(np.int8), policy=pol.float().numpy(), value=val.float().numpy().reshape(-1))
    print(path, blob.size, "floats")


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    make(9, 64, 3, 6, 1234, os.path.join(here, "net_golden_n9_c64_b3.npz"))
