"""Numerics of the candidate mixed-precision product (CPU emulation, PyTorch float64 accumulate):
    x*w ~= fp16(x)*fp16(w)  +  [fp8(xh / 8)*fp8(wl * 2^(d+11))  +  fp8(xl * 2^8)*fp8(wh * 2^d)] * 2^-(d+8)
with xh = fp16(x), xl = x - xh, wh = fp16(w), wl = w - wh, fp8 = OCP e4m3 (what v_mfma_scale_f32_16x16x128_f8f6f4 takes;
the powers of two are its block scales; d = the layer's weight shift, conv_m8.hip).  Prints the output error of the benchmark-shaped net against float64 and beside the
bf16x3 split's error.    python tools/mixed_err.py [N] [C] [blocks] [boards]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch, torch.nn.functional as F
import torch_ref as tr

def q8(t):
    return t.float().to(torch.float8_e4m3fn).double()


def q6(t, shift):
    """OCP e3m2 ("bf6": 3 significant bits, normal range 0.25 ... 28, subnormal step 0.0625) of t * 2^shift, returned unscaled;
    round to nearest even, saturating -- candidate format of the cross terms (the scaled MFMA runs it at 1.54x the fp8 rate)"""
    x = t.double() * 2.0 ** shift
    a = x.abs().clamp(max=28.0)
    e = torch.floor(torch.log2(a.clamp(min=0.25))).clamp(max=4.0)
    step = 2.0 ** (e - 2.0)
    q = torch.round(a / step) * step
    return torch.sign(x) * q.clamp(max=28.0) * 2.0 ** -shift


def wshift(w, top):
    m = float(w.abs().max())
    return int(np.floor(np.log2(top / m))) if m > 0 else 0


def fold(conv, bn):
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return (conv.weight * s.view(-1, 1, 1, 1)).float(), ((conv.bias - bn.running_mean) * s + bn.bias).float()


def conv_mode(x, w, mode):
    """x float32 NCHW, w float32 folded OIHW -> float64 result of the emulated product"""
    if mode == "f32":
        return F.conv2d(x.double(), w.double(), padding=1)
    if mode == "bf16x3":
        xh = x.bfloat16().float(); xm = (x - xh).bfloat16().float()
        wh = w.bfloat16().float(); wm = (w - wh).bfloat16().float()
        c = lambda a, b: F.conv2d(a.double(), b.double(), padding=1)
        return c(xh, wh) + c(xh, wm) + c(xm, wh)
    if mode == "mixed":      # the shipped scheme (conv_m8.hip): q8(x_h / 8) clamped at 3584, q8(x_l 2^8), per-layer weight shift d
        xh = x.half().float(); xl = x - xh
        wh = w.half().float(); wl = w - wh
        d = wshift(wh, 416.0)
        c = lambda a, b: F.conv2d(a.double(), b.double(), padding=1)
        return (c(xh, wh) + c(q8(xh.clamp(-3584.0, 3584.0) * 2.0 ** -3), q8(wl * 2.0 ** (d + 11))) * 2.0 ** -(d + 8)
                + c(q8(xl * 2.0 ** 8), q8(wh * 2.0 ** d)) * 2.0 ** -(d + 8))
    if mode.startswith("mixed6"):      # fp16 main term + e3m2 cross terms; mode = "mixed6:<x_h shift>"
        sx = int(mode.split(":")[1])
        xh = x.half().float(); xl = x - xh
        wh = w.half().float(); wl = w - wh
        d = wshift(wh, 24.0)
        c = lambda a, b: F.conv2d(a.double(), b.double(), padding=1)
        return c(xh, wh) + c(q6(xh, sx), q6(wl, d + 11)) + c(q6(xl, sx + 11), q6(wh, d))
    raise ValueError(mode)


def forward(m, player, board, mode):
    n = m.n
    n2 = n * n
    with torch.no_grad():
        p = player.view(-1, 1).repeat(1, n2).view(-1, 1, n, n)
        x = torch.cat([p, board.view(-1, 1, n, n)], 1).double()
        x = F.relu(m.norm(m.conv(x))).float()                       # stem in f32
        for c0, n0, c1, n1 in m.tower:
            w0, b0 = fold(c0, n0); w1, b1 = fold(c1, n1)
            t = F.relu(conv_mode(x, w0, mode) + b0.double().view(1, -1, 1, 1)).float()
            t = conv_mode(t, w1, mode) + b1.double().view(1, -1, 1, 1)
            x = F.relu(t + x.double()).float()
        x = x.double()
        pol = F.relu(m.policy_norm(m.policy_conv(x)))
        pol = torch.softmax(m.policy_linear(pol.view(-1, 2 * n2)), 1)
        v = m.value_conv(x)
        v = F.relu(m.value_norm(v.view(-1, n2)))
        v = torch.tanh(m.value_linear2(F.relu(m.value_linear1(v))))
    return pol, v.view(-1), x


def run(n, C, blocks, B, modes=("bf16x3", "mixed"), seed=0):
    """{mode: (policy max error, value max error)} against the float64 evaluation of the same float32 net"""
    torch.manual_seed(seed)
    m = tr.WeightedPolicy(n, C, blocks).double().eval()
    tr.randomize_bn(m)
    rng = np.random.default_rng(1)
    board = torch.tensor(rng.integers(-1, 2, (B, n * n)).astype(np.float64))
    player = torch.tensor(rng.choice([-1.0, 1.0], B))
    ref = forward(m, player, board, "f32")
    out = {"_max_activation": float(ref[2].abs().max())}
    for mode in modes:
        got = forward(m, player, board, mode)
        out[mode] = (float((got[0] - ref[0]).abs().max()), float((got[1] - ref[1]).abs().max()), float((got[2] - ref[2]).abs().max()))
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    res = run(n, C, blocks, B, modes=("bf16x3", "mixed", "mixed6:0", "mixed6:-2", "mixed6:2"))
    print(f"net {n}x{n}, C={C}, stem + {blocks-1} blocks, {B} boards; |activation| max {res['_max_activation']:.2f}")
    for mode, v in res.items():
        if not mode.startswith("_"):
            print(f"{mode:9s}: policy max err {v[0]:.3e}   value max err {v[1]:.3e}   last activation max err {v[2]:.3e}")
