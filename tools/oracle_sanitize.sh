#!/bin/bash
# The CPU oracle (the checker of every parity test) under AddressSanitizer + UndefinedBehaviorSanitizer: builds oracle/_build/libc6oracle_asan.so
# from the same sources and runs the oracle's own tests (known answers K-1 .. K-11, golden vectors, the net against PyTorch) on it.  CPU only.
#   tools/oracle_sanitize.sh [pytest args]        (default: tests/test_oracle_*.py)
set -e
cd "$(dirname "$0")/.."
make -C oracle asan > /dev/null
export C6_ORACLE_SO=$PWD/oracle/_build/libc6oracle_asan.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
if [ $# -eq 0 ]; then set -- tests/test_oracle_game.py tests/test_oracle_simulate.py tests/test_oracle_mcts.py tests/test_oracle_agent.py tests/test_oracle_augment.py tests/test_oracle_rng.py tests/test_oracle_default.py tests/test_oracle_replay.py tests/test_oracle_net.py; fi
python -m pytest "$@" -x -q -m "not gpu" -p no:cacheprovider
# the threaded self-play (games on a pool of threads) under ThreadSanitizer
make -C oracle tsan > /dev/null
C6_ORACLE_SO=$PWD/oracle/_build/libc6oracle_tsan.so LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0" python - <<'PY'
import sys
sys.path.insert(0, "tests")
import c6oracle as o
for n, games, threads, sims in ((9, 16, 8, 16), (15, 8, 8, 8)):
    r = o.self_play(n, num=games, threads=threads, num_simulation=sims, seed=3, want_pi=True)
    assert len(r) == games
    print(f"ThreadSanitizer: {games} games of {n}x{n} on {threads} threads, {sims} simulations per move: no report")
PY
