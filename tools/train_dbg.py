"""per-tensor error of the HIP trainer's gradients against PyTorch autograd (debug aid): train_dbg.py n C blocks B [f64]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from c6amd_loader import load_package
pkg = load_package()
import torch, torch_ref as tr
n, C, blocks, B = [int(x) for x in sys.argv[1:5]]
dt = torch.float64 if len(sys.argv) > 5 else torch.float32
torch.manual_seed(n + C)
m = tr.WeightedPolicy(n, C, blocks); (tr.decisive_bn if os.environ.get('DECISIVE') else tr.randomize_bn)(m)
blob = tr.blob_of(m)
batch = tr.make_batch(n, B, seed=B)
t = pkg.Trainer(n, C, blocks, B, lr=0.05, momentum=0.9, precision=int(os.environ.get("PREC", "3")))
t.load_weights(blob)
loss = t.step(*[x.numpy() for x in batch])
g = t.gradients()
rl, rg = tr.train_steps(m, [batch], 0.05, 0.9, dtype=dt)
print("loss", loss, rl[0])
names = []
def bn(p): return [p + ".gamma", p + ".beta", p + ".rm", p + ".rv"]
names += ["stem.w", "stem.b"] + bn("stem")
for i in range(blocks - 1):
    for j in range(2): names += [f"t{i}_{j}.w", f"t{i}_{j}.b"] + bn(f"t{i}_{j}")
names += ["pc.w", "pc.b"] + bn("pn") + ["pl.w", "pl.b", "vc.w", "vc.b"] + bn("vn") + ["v1.w", "v1.b", "v2.w", "v2.b"]
off = 0
for name, (tt, isp) in zip(names, tr._tensors(m)):
    k = tt.numel(); a, b = g[off:off + k], rg[off:off + k]; off += k
    if name == "v1.b" and os.environ.get("DECISIVE"): print("v1.b first entries", a[:6], b[:6], "worst idx", int(np.abs(a-b).argmax()), a[np.abs(a-b).argmax()], b[np.abs(a-b).argmax()])
    if isp: print(f"{name:12s} n={k:7d} max|ref|={np.abs(b).max():.3e} maxerr={np.abs(a-b).max():.3e} rel={np.abs(a-b).max()/max(np.abs(b).max(),1e-12):.2e} l2rel={np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30):.2e}")
off = 0; worst = 0
for name, (tt, isp) in zip(names, tr._tensors(m)):
    k = tt.numel(); a, b = g[off:off + k], rg[off:off + k]; off += k
    if isp and not name.endswith(".b"): worst = max(worst, np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30))
print("WORST l2rel", sys.argv[1:5], worst)
