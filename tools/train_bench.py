"""Training-step throughput at the reference trainer's shape (AlphaZero/cpp_weighted/main.cpp: model(10) = stem + 9
residual blocks, C = 256, 15x15, mini_batch 2048, SGD momentum 0.9): ms per step and samples/s, both conv precisions.

    python3 tools/train_bench.py [--board 15] [--blocks 10] [--batch 2048] [--steps 5] [--json out.json]
Synthetic batch, random-init weights.  FLOPs per step = 3 x (tower conv FLOPs of the batch): forward, input gradient and
weight gradient each cost 2*9*C^2*N^2 per sample and layer."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from c6amd_loader import load_package
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--board", type=int, default=15); ap.add_argument("--channels", type=int, default=256)
ap.add_argument("--blocks", type=int, default=10); ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--json", default=""); ap.add_argument("--only", default="", help="fp32 | bf16x3: one precision only (profiles)")
a = ap.parse_args()
pkg = load_package()
n, C, nb, B = a.board, a.channels, a.blocks, a.batch
rng = np.random.default_rng(0)
blob = ge.make_random_blob(n, C, nb, rng)
w = rng.integers(-1, 2, B).astype(np.float32); p = rng.choice([-1.0, 1.0], B).astype(np.float32)
b = rng.integers(-1, 2, (B, n * n)).astype(np.float32); q = rng.integers(0, n * n, B).astype(np.float32)
flops = 3.0 * B * 2 * (nb - 1) * (2.0 * 9 * C * C * n * n)
out = {"workload": f"train step: {n}x{n}, stem + {nb-1} residual blocks, C={C}, mini_batch {B}", "conv_flops_per_step": flops, "paths": {}}
for name, prec in (("bf16x3", pkg.PREC_BF16X3), ("fp32", pkg.PREC_FP32)):
    if a.only and a.only != name:
        continue
    t = pkg.Trainer(n, C, nb, B, lr=1e-3, momentum=0.9, precision=prec)
    t.load_weights(blob)
    losses = [t.step(w, p, b, q)]                      # warm-up (includes first-touch allocation effects)
    t0 = time.perf_counter()
    for _ in range(a.steps): losses.append(t.step(w, p, b, q))      # step() returns the loss: synchronises every step
    dt = (time.perf_counter() - t0) / a.steps
    out["paths"][name] = {"ms_per_step": dt * 1e3, "samples_per_s": B / dt, "conv_tflops": flops / dt / 1e12, "loss_first": losses[0], "loss_last": losses[-1]}
    t.close()
print(json.dumps(out))
if a.json: json.dump(out, open(a.json, "w"), indent=1)
