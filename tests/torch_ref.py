"""PyTorch (CPU) reference of the reference's policy/value net and training step -- TEST INFRASTRUCTURE.

The reference's net is libtorch C++ (AlphaZero/cpp_weighted/model.hpp) and cannot be imported; this file restates its
architecture (model.hpp:9-95), its loss (model.hpp:97-108, including the [B] - [B,1] broadcast of the value term) and
its optimiser (torch::optim::SGD with momentum, main.cpp:107-109,151-157) on PyTorch's own autograd, so that the HIP
trainer (csrc/trainer.cpp, train_kernels.hip) is checked against independent kernels.  Used by
tests/golden/gen_net_golden.py, tests/golden/gen_train_golden.py and tests/test_gpu_train.py.
"""
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

    def loss(self, winner, player, board, pos):  # model.hpp:97-108
        pol, value = self.forward(player, board)
        value_loss = torch.mean((winner - value).pow(2))          # [B] - [B,1] -> [B,B], as in the reference
        one_hot = torch.zeros(pos.shape[0], self.n * self.n, dtype=pol.dtype).scatter_(1, pos.long().view(-1, 1), 1)
        policy_loss = -torch.mean(torch.sum(one_hot * torch.log(pol + 1e-7), -1))
        return value_loss + policy_loss


def _tensors(m, grads=False):
    """the blob order: (tensor, is_parameter) in the layout documented in oracle/c6_oracle_net.c"""
    out = []

    def p(t):
        out.append((t.grad if grads else t, True))

    def bn(b):
        p(b.weight); p(b.bias)
        out.append((torch.zeros_like(b.running_mean) if grads else b.running_mean, False))
        out.append((torch.zeros_like(b.running_var) if grads else b.running_var, False))

    p(m.conv.weight); p(m.conv.bias); bn(m.norm)
    for c0, n0, c1, n1 in m.tower:
        p(c0.weight); p(c0.bias); bn(n0)
        p(c1.weight); p(c1.bias); bn(n1)
    p(m.policy_conv.weight); p(m.policy_conv.bias); bn(m.policy_norm)
    p(m.policy_linear.weight); p(m.policy_linear.bias)
    p(m.value_conv.weight); p(m.value_conv.bias); bn(m.value_norm)
    p(m.value_linear1.weight); p(m.value_linear1.bias); p(m.value_linear2.weight); p(m.value_linear2.bias)
    return out


def blob_of(m, grads=False):
    """flat f32 blob of the parameters/buffers (or of the parameter gradients, zeros at the buffer positions)"""
    return torch.cat([t.detach().reshape(-1) for t, _ in _tensors(m, grads)]).to(torch.float32).numpy()


def param_mask(m):
    """1.0 where the blob holds a trainable parameter, 0.0 for BatchNorm running statistics"""
    return np.concatenate([np.full(t.numel(), 1.0 if isp else 0.0, np.float32) for t, isp in _tensors(m)])


def load_blob(m, blob):
    off = 0
    with torch.no_grad():
        for t, _ in _tensors(m):
            k = t.numel()
            t.copy_(torch.from_numpy(np.asarray(blob[off:off + k], np.float32)).view_as(t).to(t.dtype))
            off += k
    assert off == len(blob)


def randomize_bn(m):
    with torch.no_grad():  # non-trivial running statistics / affine parameters so that every term is exercised
        for mod in m.modules():
            if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm1d)):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.5, 1.5)


def decisive_bn(m):
    """BatchNorm affine parameters that keep every ReLU input far from 0 (beta = +-6 with gamma <= 1): stem, second conv
    of a block (whose ReLU also sees the skip connection), policy norm and half of everything else always on, the other
    half of the first-conv / value-norm channels always off.  A framework-to-framework gradient comparison is only
    meaningful to ~1e-6 where no ReLU mask can flip on a 1e-6 difference in the forward pass (see relu_margin)."""
    with torch.no_grad():
        def setb(b, pattern):
            b.weight.uniform_(0.5, 1.0)
            k = b.bias.numel()
            sign = torch.ones(k) if pattern == "on" else torch.tensor([1.0 if i % 2 == 0 else -1.0 for i in range(k)])
            b.bias.copy_(6.0 * sign)
            b.running_mean.uniform_(-0.2, 0.2)
            b.running_var.uniform_(0.5, 1.5)
        setb(m.norm, "on")
        for c0, n0, c1, n1 in m.tower:
            setb(n0, "half"); setb(n1, "on")
        setb(m.policy_norm, "on"); setb(m.value_norm, "half")
        m.value_linear1.bias.copy_(torch.tensor([3.0 if i % 2 == 0 else -3.0 for i in range(256)]))


def relu_margin(m, batch):
    """smallest |input of a ReLU| in the train-mode forward pass (float64 copy of the model)"""
    import copy
    m = copy.deepcopy(m).double()
    m.train()
    w, p, b, q = [x.double() for x in batch]
    n, n2 = m.n, m.n * m.n
    mins = []
    with torch.no_grad():
        pl = p.view(-1, 1).repeat(1, n2).view(-1, 1, n, n)
        x = torch.cat([pl, b.view(-1, 1, n, n)], 1)
        t = m.norm(m.conv(x)); mins.append(t.abs().min()); x = F.relu(t)
        for c0, n0, c1, n1 in m.tower:
            t = n0(c0(x)); mins.append(t.abs().min()); t = F.relu(t)
            t = n1(c1(t)) + x; mins.append(t.abs().min()); x = F.relu(t)
        t = m.policy_norm(m.policy_conv(x)); mins.append(t.abs().min())
        v = m.value_norm(m.value_conv(x).view(-1, n2)); mins.append(v.abs().min()); v = F.relu(v)
        h = m.value_linear1(v); mins.append(h.abs().min())
    return float(min(mins))


def make_batch(n, B, seed):
    g = torch.Generator().manual_seed(seed)
    boards = torch.randint(-1, 2, (B, n * n), generator=g).float()
    players = (torch.randint(0, 2, (B,), generator=g) * 2 - 1).float()
    winners = (torch.randint(0, 3, (B,), generator=g) - 1).float()
    poses = torch.randint(0, n * n, (B,), generator=g).float()
    return winners, players, boards, poses


def train_steps(m, batches, lr, momentum, dtype=torch.float32):
    """main.cpp:151-157: zero_grad, loss, backward, SGD step.  Returns the losses and the gradients of the LAST step."""
    m = m.to(dtype)
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=lr, momentum=momentum)
    losses = []
    for w, p, b, q in batches:
        opt.zero_grad()
        loss = m.loss(w.to(dtype), p.to(dtype), b.to(dtype), q.to(dtype))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, blob_of(m, grads=True)
