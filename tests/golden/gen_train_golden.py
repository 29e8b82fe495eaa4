"""Generates tests/golden/train_golden_n9_c64_b2.npz: one SGD step of the reference's training loop body
(AlphaZero/cpp_weighted/main.cpp:151-157 on model.hpp:52-108) computed by PyTorch autograd in float64 on the
architecture restated in tests/torch_ref.py -- seeded weights (seed chosen so that no ReLU input lies within 2e-5 of
zero, see tests/test_gpu_train.py), a batch of 8 positions, the loss, every parameter gradient and the blob after the step.

    python tests/golden/gen_train_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch_ref as tr

n, C, blocks, B, seed, lr, momentum = 9, 64, 2, 8, 38, 0.05, 0.9
torch.manual_seed(seed)
m = tr.WeightedPolicy(n, C, blocks)
tr.randomize_bn(m)
blob = tr.blob_of(m)
batch = tr.make_batch(n, B, seed=seed)
margin = tr.relu_margin(m, batch)
losses, grads = tr.train_steps(m, [batch], lr, momentum, dtype=torch.float64)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_golden_n9_c64_b2.npz")
np.savez_compressed(out, n=n, C=C, blocks=blocks, B=B, lr=lr, momentum=momentum, blob=blob, winners=batch[0].numpy(), players=batch[1].numpy(),
                    boards=batch[2].numpy().astype(np.int8), poses=batch[3].numpy(), loss=np.float64(losses[0]), grads=grads,
                    after=tr.blob_of(m), param_mask=tr.param_mask(m), relu_margin=margin)
print(out, "loss", losses[0], "margin", margin, os.path.getsize(out), "bytes")
