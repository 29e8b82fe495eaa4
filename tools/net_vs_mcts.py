"""Evaluation games of the resident net (AlphaZero search, c6_engine) against DefaultPolicy, the reference's pure MCTS with
random playouts (c6_rollout): MultiPolicy(AlphaZero, DefaultPolicy) of policy/multi_policy/mod.rs:61-71 for G games at
once.  The engine moves for its colour with c6_engine_step and receives the opponent's stones through
c6_engine_apply_moves; the rollout policies read the boards the host keeps.

    python3 tools/net_vs_mcts.py [--board 9] [--channels 64] [--blocks 3] [--games 256] [--sims 32] [--iters 50]
Random-init weights (no checkpoint in this image); prints the score for the net as Black and as White as JSON."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from c6amd_loader import load_package
import __graft_entry__ as ge


def play_match(pkg, n, games, sims, iters, blob, C, nb, net_colour, seed=0):
    eng = pkg.Engine(board_size=n, num_games=games, num_simulation=sims, seed=seed, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=nb)
    eng.load_weights(blob)
    dp = pkg.DefaultPolicy(n, games, num_iter=iters, seed=seed + 1)
    boards = np.zeros((games, n * n), np.int8)
    live = np.ones(games, bool)
    winners = np.zeros(games, np.int8)
    stones = 0
    turn, remain = -1, 1
    while live.any():
        if turn == net_colour:
            eng.step(sims)                                  # S searches + one stone for every live game
            pos = eng.last_moves()
        else:
            pos = dp.next(boards, active=live.astype(np.uint8))
            eng.apply_moves(np.where(live, pos, -2))        # -1 (no empty cell) ends the game as a draw there as well
        moved = live & (pos >= 0)
        boards[moved, pos[moved]] = turn
        stones += int(moved.sum())
        w = pkg.search_winner(n, pkg.to_soa(boards))
        winners[moved] = w[moved]
        live &= (pos >= 0) & (w == 0)
        remain -= 1
        if remain <= 0:
            remain, turn = 2, -turn
    eg = eng.fetch_games()
    assert sorted(g["winner"] for g in eg) == sorted(winners.tolist()), "engine and host disagree on the results"
    eng.close(); dp.close()
    return winners, stones


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=9); ap.add_argument("--channels", type=int, default=64); ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--games", type=int, default=256); ap.add_argument("--sims", type=int, default=32); ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    pkg = load_package()
    blob = ge.make_random_blob(a.board, a.channels, a.blocks, np.random.default_rng(0))
    out = {"games_per_colour": a.games, "num_simulation": a.sims, "num_iter": a.iters}
    for name, colour in (("net_black", -1), ("net_white", 1)):
        t0 = time.time()
        w, stones = play_match(pkg, a.board, a.games, a.sims, a.iters, blob, a.channels, a.blocks, colour)
        out[name] = {"net": int((w == colour).sum()), "draw": int((w == 0).sum()), "mcts": int((w == -colour).sum()), "stones": stones,
                     "seconds": round(time.time() - t0, 2)}
    print(json.dumps(out))
