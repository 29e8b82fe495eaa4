"""float32 rounding error of Winograd F(m,3) x F(m,3) layers (Cook-Toom matrices built with sympy, weights transformed in double and rounded once,
transforms and accumulation in float32) against a float64 direct convolution: F(4,3) the engine's, F(5,3) / F(6,3) the candidates of DESIGN 9.
    python tools/wino_err.py"""
import numpy as np
def cook_toom(points, m, r=3):
    # returns AT (m x n), G (n x r), BT (n x n) for F(m, r) with the given finite points + infinity (Lavin's wincnn construction)
    from fractions import Fraction as F
    import sympy
    n = m + r - 1
    a = [sympy.Rational(p) for p in points]  # n-1 finite points
    x = sympy.symbols('x')
    # Vandermonde-based construction
    def Ax(k):  # product over j != k
        v = sympy.Integer(1)
        for j in range(n - 1):
            if j != k: v *= (a[k] - a[j])
        return v
    f = [Ax(k) for k in range(n - 1)]
    AT = sympy.zeros(m, n)
    for i in range(m):
        for k in range(n - 1): AT[i, k] = a[k] ** i
    AT[m - 1, n - 1] = 1
    G = sympy.zeros(n, r)
    for k in range(n - 1):
        for j in range(r): G[k, j] = a[k] ** j / f[k]
    G[n - 1, r - 1] = 1
    # BT from polynomial: rows = coefficients of prod_{j != k}(x - a_j) , last row prod_j (x - a_j)
    BT = sympy.zeros(n, n)
    for k in range(n - 1):
        poly = sympy.Poly(sympy.prod([(x - a[j]) for j in range(n - 1) if j != k]), x)
        co = poly.all_coeffs()[::-1]
        for j, c in enumerate(co): BT[k, j] = c
    poly = sympy.Poly(sympy.prod([(x - a[j]) for j in range(n - 1)]), x)
    co = poly.all_coeffs()[::-1]
    for j, c in enumerate(co): BT[n - 1, j] = c
    return (np.array(AT.tolist(), dtype=np.float64), np.array(G.tolist(), dtype=np.float64), np.array(BT.tolist(), dtype=np.float64))
def check(AT, G, BT, m):
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + 2); g = rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[t] * d[k + t] for t in range(3)) for k in range(m)])
    return np.abs(y - ref).max()
def layer_err(AT, G, BT, m, C=128, tiles=64, seed=1):
    rng = np.random.default_rng(seed)
    n = m + 2
    x = np.maximum(rng.standard_normal((tiles, C, n, n)), 0).astype(np.float32)          # post-ReLU-like activations
    w = (rng.uniform(-1, 1, (C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    ref = np.zeros((tiles, m, m))
    for a in range(3):
        for b in range(3): ref += np.einsum('tcij,c->tij', x[:, :, a:a + m, b:b + m].astype(np.float64), w[:, a, b].astype(np.float64))
    AT32, G64, BT32 = AT.astype(np.float32), G, BT.astype(np.float32)
    U = np.einsum('ia,cab,jb->cij', G64, w.astype(np.float64), G64).astype(np.float32)             # weights transformed in double, rounded once
    V = np.einsum('ia,tcab,jb->tcij', BT32, x, BT32).astype(np.float32)
    M = np.zeros((tiles, n, n), np.float32)
    for c in range(C): M += U[c][None] * V[:, c]                                                    # f32 accumulate
    Y = np.einsum('ia,tab,jb->tij', AT32, M, AT32).astype(np.float32)
    return np.sqrt(np.mean((Y - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)), np.abs(Y - ref).max() / np.abs(ref).max()
if __name__ == "__main__":
    for name, pts, m in (("F(4,3) 0,1,-1,2,-2", [0, 1, -1, 2, -2], 4), ("F(6,3) 0,1,-1,2,-2,1/2,-1/2", [0, 1, -1, 2, -2, "1/2", "-1/2"], 6),
                         ("F(6,3) 0,1,-1,1/2,-1/2,3/2?,-3/2?", [0, 1, -1, "1/2", "-1/2", "3/2", "-3/2"], 6), ("F(5,3) 0,1,-1,2,-2,1/2", [0, 1, -1, 2, -2, "1/2"], 5)):
        AT, G, BT = cook_toom(pts, m)
        print(name, "exactness", check(AT, G, BT, m), "rms rel err, max rel err (f32):", layer_err(AT, G, BT, m))
