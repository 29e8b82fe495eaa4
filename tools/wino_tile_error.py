"""What a LARGER Winograd tile would cost in accuracy (F(6,3) both ways = 729 instead of 788 products per channel pair and board,
-7.5 % matrix work): one 3x3 convolution layer at C = 256 in f32 -- weights transformed in f64 and rounded once (as the packer does), input
transform, channel sum (sequential f32 accumulation, as the matrix unit's accumulator) and output transform in f32 -- against the same layer
in f64, for the tile forms F(4,3) x F(4,3) (rounds 2-3), F(4,3) down x F(6,3) across (the 19x19 tower today) and F(6,3) x F(6,3).
CPU only (numpy); the transforms are the Cook-Toom matrices on the points 0, +-1, +-2 (, +-1/2), inf that conv_wino6.hip uses.

    python tools/wino_tile_error.py [--tiles 96] [--cout 32] [--seed 0] [--json out.json]
"""
import argparse
import json

import numpy as np


def cook_toom(m, pts):
    """F(m, 3) on the finite points `pts` plus infinity: A^T (m x n), G (n x 3), B^T (n x n), n = m + 2 = len(pts) + 1; y = A^T [(G g) * (B^T d)]"""
    r, n = 3, m + 2
    assert len(pts) == n - 1
    p = np.array(pts, dtype=np.float64)
    AT = np.zeros((m, n)); G = np.zeros((n, r))
    for j in range(n - 1):
        nj = np.prod([p[j] - p[l] for l in range(n - 1) if l != j])
        for i in range(m):
            AT[i, j] = p[j] ** i
        for k in range(r):
            G[j, k] = p[j] ** k / nj
    AT[m - 1, n - 1] = 1.0
    G[n - 1, r - 1] = 1.0
    # B^T from the bilinear identity sum_j AT[i][j] G[j][k] BT[j][l] = [l == i + k]  (m*r equations per column l, n unknowns; exact rationals)
    BT = np.zeros((n, n))
    for l in range(n):
        rows, rhs = [], []
        for i in range(m):
            for k in range(r):
                rows.append(AT[i, :] * G[:, k]); rhs.append(1.0 if l == i + k else 0.0)
        sol, res, rank, _ = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)
        assert rank == n
        BT[:, l] = sol
    BT = np.round(BT * 64) / 64      # (the entries are multiples of 1/8 .. 1/4 for these point sets; removes the solver's last-bit noise)
    # check the identity in f64
    rng = np.random.default_rng(1)
    g, d = rng.standard_normal(r), rng.standard_normal(n)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(g[k] * d[i + k] for k in range(r)) for i in range(m)])
    assert np.allclose(y, ref, rtol=1e-10, atol=1e-10), (y, ref)
    return AT, G, BT


PTS = {4: [0, 1, -1, 2, -2], 6: [0, 1, -1, 2, -2, 0.5, -0.5], 3: [0, 1, -1, 2], 2: [0, 1, -1]}


def layer_error(mv, mh, C, cout, tiles, rng, act_gain=1.0):
    ATv, Gv, BTv = cook_toom(mv, PTS[mv]); ATh, Gh, BTh = cook_toom(mh, PTS[mh])
    nv, nh = mv + 2, mh + 2
    # activations after a ReLU, weights at He scale (gain: the hostile-weight tests' larger activations)
    d = np.maximum(rng.standard_normal((tiles, C, nv, nh)), 0.0) * act_gain
    g = rng.standard_normal((cout, C, 3, 3)) * np.sqrt(2.0 / (9 * C))
    # f64 reference: direct correlation
    ref = np.zeros((tiles, cout, mv, mh))
    for a in range(3):
        for b in range(3):
            ref += np.einsum("tcij,oc->toij", d[:, :, a:a + mv, b:b + mh], g[:, :, a, b])
    # f32 Winograd: U in f64 then rounded once; V, the channel sum and Y in f32
    U = np.einsum("ia,ocab,jb->ocij", Gv, g, Gh).astype(np.float32)
    f = np.float32
    d32 = d.astype(f)
    V = np.einsum("ia,tcab->tcib", BTv.astype(f), d32).astype(f)
    V = np.einsum("tcib,jb->tcij", V, BTh.astype(f)).astype(f)
    M = np.zeros((tiles, cout, nv, nh), dtype=f)
    for c in range(C):      # sequential f32 accumulation over the input channels
        M = (M + V[:, None, c] * U[None, :, c]).astype(f)
    Y = np.einsum("ia,toab->toib", ATv.astype(f), M).astype(f)
    Y = np.einsum("toib,jb->toij", Y, ATh.astype(f)).astype(f)
    err = np.abs(Y.astype(np.float64) - ref)
    scale = np.sqrt(np.mean(ref ** 2))
    return {"tile": f"F({mv},3) down x F({mh},3) across", "positions_per_tile": nv * nh, "outputs_per_tile": mv * mh,
            "products_per_output": nv * nh / (mv * mh), "rms_error_over_rms_output": float(np.sqrt(np.mean(err ** 2)) / scale),
            "max_error_over_rms_output": float(err.max() / scale)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=96); ap.add_argument("--cout", type=int, default=32); ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    out = {"what": "one 3x3 convolution layer, C = 256, f32 Winograd against f64 direct; errors relative to the RMS of the layer's outputs", "cases": []}
    for mv, mh in ((2, 2), (4, 4), (4, 6), (6, 6)):
        r = layer_error(mv, mh, 256, a.cout, a.tiles, np.random.default_rng(a.seed))
        out["cases"].append(r)
        print("%-28s %2d positions for %2d outputs (%.2f per output): rms %.2e  max %.2e of the output's RMS" % (
            r["tile"], r["positions_per_tile"], r["outputs_per_tile"], r["products_per_output"], r["rms_error_over_rms_output"], r["max_error_over_rms_output"]), flush=True)
    base = out["cases"][2]
    for r in out["cases"]:
        r["rms_error_vs_todays_tile"] = r["rms_error_over_rms_output"] / base["rms_error_over_rms_output"]
    print("F(6,3) x F(6,3) against today's F(4,3) x F(6,3): %.1f x the rms error" % out["cases"][3]["rms_error_vs_todays_tile"])
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)
