"""Net-vs-net evaluation games (MultiPolicy with two resident nets, policy/multi_policy/mod.rs:61-71): engine A searches
and moves for Black, engine B for White; each receives the other's stones through c6_engine_apply_moves, so both keep
their own trees exactly as two AlphaZero policies inside the reference's MultiPolicy would.

    python3 tools/net_vs_net.py [--board 9] [--channels 64] [--blocks 3] [--games 256] [--sims 32] [--seed-b 1]
Random-init weights for both sides (different seeds); prints the score as JSON."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from c6amd_loader import load_package
import __graft_entry__ as ge


def play_match(pkg, n, games, sims, black_blob, white_blob, C, nb, seed=0, precision=0):
    mk = lambda blob, sd: (lambda e: (e.load_weights(blob), e)[1])(pkg.Engine(board_size=n, num_games=games, num_simulation=sims, seed=sd, evaluator=pkg.EVAL_NET,
                                                                              net_channels=C, net_blocks=nb, net_precision=precision))
    A, B = mk(black_blob, seed), mk(white_blob, seed + 1)
    turn, remain = -1, 1
    while True:
        mover, other = (A, B) if turn == -1 else (B, A)
        mover.step(sims)                                   # S searches + the mover's own stones for every live game
        pos = mover.last_moves()
        if not (pos >= 0).any() and not (pos == -1).any():
            break
        other.apply_moves(np.where(pos >= -1, pos, -2))    # -1 (no move: draw) ends the game on the other side as well
        remain -= 1
        if remain <= 0:
            remain, turn = 2, -turn
        if A.stats()["live_games"] == 0:
            break
    ga, gb = A.fetch_games(), B.fetch_games()
    A.close(); B.close()
    return ga, gb


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=9); ap.add_argument("--channels", type=int, default=64); ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--games", type=int, default=256); ap.add_argument("--sims", type=int, default=32); ap.add_argument("--seed-b", type=int, default=1)
    a = ap.parse_args()
    pkg = load_package()
    blob_a = ge.make_random_blob(a.board, a.channels, a.blocks, np.random.default_rng(0))
    blob_b = ge.make_random_blob(a.board, a.channels, a.blocks, np.random.default_rng(a.seed_b))
    t0 = time.time()
    ga, gb = play_match(pkg, a.board, a.games, a.sims, blob_a, blob_b, a.channels, a.blocks)
    w = np.array([g["winner"] for g in ga])
    print(json.dumps({"games": len(ga), "black(net A)": int((w == -1).sum()), "draw": int((w == 0).sum()), "white(net B)": int((w == 1).sum()),
                      "stones": int(sum(len(g["moves"]) for g in ga)), "seconds": round(time.time() - t0, 2)}))
