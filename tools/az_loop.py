"""The reference's outer loop (AlphaZero/cpp_weighted/main.cpp:138-168) on one MI355X, everything resident on the device:

    while true:  results = self_play(callback, param)      -> Engine.run() with the resident net
                 buffer.push_game(results)                  -> Engine.push_tuples(ReplayBuffer)   (device to device)
                 for i in batch_size: sample, loss, backward, step -> Trainer.step_replay(...)     (no host round trip)
                 (every ckpt_interval) save                 -> Trainer.export_weights() -> Engine.load_weights()

    python3 tools/az_loop.py [--board 9] [--channels 64] [--blocks 3] [--games 256] [--sims 32] [--epochs 3]
                             [--train-steps 50] [--mini-batch 512]
Small defaults so that it finishes in about a minute; prints one JSON line per epoch."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from c6amd_loader import load_package
import __graft_entry__ as ge

ap = argparse.ArgumentParser()
ap.add_argument("--board", type=int, default=9); ap.add_argument("--channels", type=int, default=64); ap.add_argument("--blocks", type=int, default=3)
ap.add_argument("--games", type=int, default=256); ap.add_argument("--sims", type=int, default=32); ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--train-steps", type=int, default=50); ap.add_argument("--mini-batch", type=int, default=512)
ap.add_argument("--lr", type=float, default=1e-3); ap.add_argument("--max-buffer", type=int, default=100000)
a = ap.parse_args()
pkg = load_package()
n, C, nb = a.board, a.channels, a.blocks
blob = ge.make_random_blob(n, C, nb, np.random.default_rng(0))
rb = pkg.ReplayBuffer(n, a.max_buffer)
tr = pkg.Trainer(n, C, nb, a.mini_batch, lr=a.lr, momentum=0.9)
tr.load_weights(blob)
draw = 0
for epoch in range(a.epochs):
    t0 = time.time()
    e = pkg.Engine(board_size=n, num_games=a.games, num_simulation=a.sims, seed=1000 + epoch, evaluator=pkg.EVAL_NET, net_channels=C, net_blocks=nb)
    e.load_weights(blob)
    e.enable_tuples(a.games * n * n)
    e.run()
    games = e.fetch_games()
    e.push_tuples(rb)
    st = e.stats(); e.close()
    t1 = time.time()
    losses = []
    for _ in range(a.train_steps):
        losses.append(tr.step_replay(rb, seed=7, draw_id=draw)); draw += 1
    blob = tr.export_weights()
    t2 = time.time()
    w = np.array([g["winner"] for g in games])
    print(json.dumps({"epoch": epoch, "games": len(games), "stones": int(st["stones"]), "black/draw/white": [int((w == -1).sum()), int((w == 0).sum()), int((w == 1).sum())],
                      "buffer": len(rb), "self_play_s": round(t1 - t0, 2), "train_s": round(t2 - t1, 2), "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4)}))
tr.close(); rb.close()
