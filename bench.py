"""bench.py -- self-play games/sec of the MI355X engine (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload at every N (weak scaling: per-GPU work fixed): BASELINE.json configs[1] -- 4096 concurrent 19x19 games per GPU, 400
simulations/move, random-init policy/value net with "4 blocks" = stem + 3 residual blocks, C = 256.

`value` runs on the REFERENCE's arithmetic: f32 operands, f32 products, f32 accumulate (AlphaZero/cpp_weighted/model.hpp:52-95 is an f32
net) -- `--precision fp32`, the library default: Winograd (tiles 4 x {6, 6, 4, 3} of a 19x19 board) on the FP32 matrix cores.  The same f32 arithmetic as a direct
convolution (`fp32_direct_path`) and the reduced-operand arithmetics (`bf16x3_path`, `f16f8_path`, `relaxed_bf16_path`) are measured in
the same run at N = 1 and reported BESIDE it, never as `value`.

A "step" is one simulation ROUND: every live game performs one search() (select -> 8 dihedral net evaluations -> expand -> one-level
backup); game phases are staggered, so each round G/400 games also make their move (policy -> board kernel -> prune / recycle), i.e.
every step carries the steady-state share of all the work.
    games/sec = simulations/sec / (400 * stones_per_game)
stones_per_game is MEASURED on complete self-play games of the same net and noise (tools/stones_probe.py, results committed under
profiles/): at the benchmark's own budget of 400 simulations on a small batch, and at 8 simulations on 4096 games; a smaller live probe
runs here too.  (A 19x19 game at 400 simulations is tens of thousands of rounds: the full-size run cannot finish a game inside a benchmark
of minutes.  The length DOES depend on the budget, and not monotonically -- 170 stones at 2 simulations, 106 at 8, 92 at 32, 139 at 400 --
which is why the value measured at 400 is the one used.)  For the tree-only workload (config5) games complete inside the timed window
and games/sec is counted directly.
Inputs (game state, trees, weights) are resident in HBM before the timed region.
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool: the host driver shares device memory across processes as dmabuf handles only (RCCL's setup needs it); the
# launcher's environment exports this already -- kept here so that a bare `torchrun bench.py` has it too (read when the HIP runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WORKLOADS = {
    # name: board, games/GPU, sims, net blocks (reference num_block), channels, evaluator, stones/game PRIOR (SURVEY 8d scratch estimate;
    # only used when no measurement is available, and then labelled as a prior)
    "config2": dict(n=19, games=4096, sims=400, blocks=4, channels=256, evaluator="net", stones_prior=208.0),
    "config3": dict(n=19, games=16384, sims=800, blocks=8, channels=256, evaluator="net", stones_prior=208.0),
    "config4": dict(n=19, games=8192, sims=400, blocks=4, channels=256, evaluator="net", stones_prior=208.0),
    "config5": dict(n=9, games=32768, sims=100, blocks=0, channels=0, evaluator="uniform", stones_prior=68.0),
    # the REFERENCE's own shape at batch (missing 3): BOARD_SIZE = 15 (Connect6/libconnect6/src/lib.rs:21), model(10) = stem + 9 residual blocks
    # (AlphaZero/cpp_weighted/main.cpp:12, model.hpp:44-49), HyperParameter::default's 800 simulations per move (alphazero_policy/mod.rs:98-108)
    "ref15": dict(n=15, games=4096, sims=800, blocks=10, channels=256, evaluator="net", stones_prior=139.0),
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, Peak BF16 / FP16 MFMA, dense
PEAK_HBM_GBS = 8000.0
KERNELS = {      # precision name -> (dominant kernel as rocprofv3 names it, description)
    "fp32": ("c6_w6_gemm_kernel<false, false, false>", "FP32 Winograd, tiles 4 x {6, 6, 4, 3} across a 19-wide board (F(4,3) down, F(6,3) / F(4,3) / F(3,3) across; F(3,3) down in the "
             "last tile row): the 48 / 40 / 36 / 30 GEMMs of a 64-tile block on v_mfma_f32_32x32x2_f32, the F(4) half of the output transform accumulated in "
             "registers; the transforms between the layers' GEMMs are HBM-bound kernels of their own (c6_w6_mid_kernel: output transform of a layer + "
             "input transform of the next)"),
    "fp32_direct": ("c6_conv3x3_f32n_kernel<256, 19>", "FP32 direct implicit GEMM, v_mfma_f32_32x32x2_f32"),
    "bf16x3": ("c6_conv3x3_b3s_kernel<256, 19, true, false>", "split-bf16 implicit GEMM, v_mfma_f32_16x16x32_bf16, 3 MFMAs per product"),
    "f16f8": ("c6_conv3x3_m8_kernel<256, 19, false>", "fp16 main term + MX-fp8 cross terms, v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4"),
}


def conv_kernel_of(name, n):
    """(kernel as rocprofv3 names it, description) of the 3x3 convolution's dominant kernel for a precision and a board size"""
    if name == "fp32" and n == 15:
        return ("c6_w6_gemm_kernel<false, false, false>",
                "FP32 Winograd, conv_wino6.hip's tiling of the reference's 15x15 board: tiles 4 x {6, 6, 3} (F(4,3) down, F(6,3) / F(3,3) across; F(3,3) down in the last tile "
                "row): 6 x 48 + 2 x 40 + 3 x 30 + 30 = 488 positions per channel pair and board (conv_wino4.hip's F(4x4) tiling: 534), the same GEMM kernels as the 19x19 "
                "benchmark (c6_w6_gemm_kernel<false> for the rows of six positions, <true> for class D') on v_mfma_f32_32x32x2_f32; HBM-bound boundary kernels between "
                "the layers (c6_w6_mid_kernel<15, .>)")
    if name == "fp32" and n != 19:
        tw = (n + 3) // 4
        edge = 2 * tw - 1 if n % 4 == 3 else 0
        return ("c6_wino4_gemm_kernel<%d, false, false>" % n,
                f"FP32 Winograd F(4x4,3x3) ({tw} x {tw} tiles per {n}x{n} board, F(3,3) in the last tile row / column: {36 * (tw * tw - edge) + 30 * edge} positions per channel "
                "pair and board): the 36 / 30 GEMMs of a 64-tile block on v_mfma_f32_32x32x2_f32, the F(4) half of the output transform accumulated in registers "
                "(conv_wino4.hip); HBM-bound boundary kernels between the layers (c6_wino4_mid_kernel)")
    k, d = KERNELS[name]
    return k.replace("256, 19", "256, %d" % n), d


def net_flops_per_board(n, C, blocks):
    """F_net(N, k) = 2*9*2*C*N^2 + (k-1)*2*(2*9*C^2*N^2) + heads"""
    n2 = n * n
    heads = 2 * C * 2 * n2 + 2 * (2 * n2) * n2 + 2 * C * n2 + 2 * n2 * 256 + 2 * 256
    return 2 * 9 * 2 * C * n2 + (blocks - 1) * 2 * (2 * 9 * C * C * n2) + heads


CONV_SOURCES = ("conv_wino6.hip", "conv_wino6_gemm.inc", "c6_wino.h", "net_kernels.hip")      # what the FP32 tower's kernels (the ones the PMC summary describes) are compiled from


def kernel_source_hash(names=None):
    """identity of kernel sources (all of csrc/, or the named files): a committed PMC summary is only replayed while it describes THESE kernels
    (`names=CONV_SOURCES`: an edit of the tree kernels does not make the convolution kernels' counters stale)"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "alphazero-connect6_amd", "csrc", "*"))):
        if f.endswith((".hip", ".h", ".cpp", ".inc")) and (names is None or os.path.basename(f) in names):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_record(kernel_substr):
    """The dominant kernel's row of the newest committed rocprofv3 PMC summary (separate --pmc passes of this same command,
    tools/profile_round.sh): HBM bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), matrix-core busy
    fraction, effective clock.  {} -- and so `traffic: null` -- when the summary was taken from other kernel sources than the ones in
    this tree (it would be stale), or when there is none."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
        if d.get("src_hash") != kernel_source_hash(CONV_SOURCES) and d.get("src_hash") != kernel_source_hash():
            return {}
        for k, v in d["kernels"].items():
            if kernel_substr in k:
                return dict(v, summary=os.path.basename(files[-1]))
    except Exception:
        return {}
    return {}


def committed_stones_probe(workload, sims, rng, ignored=None):
    """profiles/r*_stones_probe*.json (tools/stones_probe.py): complete games of the benchmark's net.  A probe turns simulations/s into
    `value` only if it was played (i) at the workload's OWN simulation budget -- the game length depends on it, and not monotonically --
    (ii) in the reference's f32 arithmetic and (iii) under the random-stream specification this library implements (c6_rng_spec): the
    games an engine plays are a function of (specification, seed, weights, arithmetic, hyper-parameters), pinned to the CPU oracle stone
    for stone by the whole-game parity tests, so a probe stays valid across kernel edits that keep the specification and is void after
    one that changes it.  Among the qualifying probes the one taken on exactly these kernel sources wins (`src_hash`), then the newest.
    Records from before round 4 carry no `rng` field: they were played under C6RNG v4.  Everything else is listed in `ignored`."""
    best = None
    h = kernel_source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_stones_probe*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload:
            continue
        why = None
        if d.get("num_simulation") != sims:
            why = f"played at {d.get('num_simulation')} simulations per move, not the workload's {sims}"
        elif d.get("precision") != "fp32":
            why = f"precision {d.get('precision')}"
        elif d.get("rng", "C6RNG v4") != rng:
            why = f"random-stream specification {d.get('rng', 'C6RNG v4')}, this library implements {rng}"
        if why:
            if ignored is not None:
                ignored.append(f"{os.path.basename(f)}: {why}")
            continue
        d = dict(d, file=os.path.basename(f), same_sources=d.get("src_hash") == h)
        key = (d["same_sources"], os.path.basename(f))
        if best is None or key > best[0]:
            best = (key, d)
    return best[1] if best else None


SAME_LAW = ("C6RNG v4", "C6RNG v5")      # specifications whose noise / tie-break / evaluator draws follow the same distributions (v5 changed the
                                         # Gamma sampler's tail proposal and the noise draws' round count, not a law; tests/test_oracle_rng.py)


def pooled_stones_probes(workload, sims):
    """every committed probe at the workload's own budget in f32 arithmetic under a specification of SAME_LAW, pooled by games: the game
    length is a statistic of (net, budget, noise LAW), not of the stream -- reported BESIDE `value`, which uses the current specification's probe only"""
    n, s1, s2, files = 0, 0.0, 0.0, []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_stones_probe*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload or d.get("num_simulation") != sims or d.get("precision") != "fp32" or d.get("rng", "C6RNG v4") not in SAME_LAW:
            continue
        g = d["games"]
        n += g
        s1 += g * d["stones_per_game"]
        s2 += g * (d["stones_std"] ** 2 + d["stones_per_game"] ** 2)
        files.append(f"{os.path.basename(f)} ({d.get('rng', 'C6RNG v4')}: {g} games, {d['stones_per_game']:.1f})")
    if n == 0:
        return None
    mean = s1 / n
    return {"games": n, "stones_per_game": mean, "standard_error": (max(s2 / n - mean * mean, 0.0) / n) ** 0.5, "probes": files}


def usable_cores():
    """host cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box shows 256
    logical CPUs but grants cpu.max = 16 CPUs of time; more threads than that only fight over the quota)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def stones_probe(pkg, wl, games, sims, precision, seed=7, device=0, progress=False, partial_path=None):
    """complete self-play games of the workload's net (same random-init weights, same noise) at `sims` simulations per move: the
    measured game length that turns simulations/s into games/s.  partial_path: the games finished so far are written there at every progress
    tick (a run is bounded by its LONGEST game; a call cut off by its time limit then still leaves its finished games and the number of
    unfinished ones behind -- a censored sample, labelled as such, never mixed into a mean)"""
    import __graft_entry__ as ge
    n = wl["n"]
    ev = pkg.EVAL_NET if wl["evaluator"] == "net" else pkg.EVAL_UNIFORM
    eng = pkg.Engine(board_size=n, num_games=games, num_simulation=sims, seed=seed, evaluator=ev, net_channels=max(wl["channels"], 64),
                     net_blocks=max(wl["blocks"], 1), device=device, net_precision=precision)
    if ev == pkg.EVAL_NET:
        eng.load_weights(ge.make_random_blob(n, wl["channels"], wl["blocks"], np.random.default_rng(0)))
    t0 = time.perf_counter()
    gs = []
    if progress:      # long probes report as they go (a silent GPU job is taken to be hung)
        while True:
            eng.run(max_rounds=256)
            st = eng.stats()
            print(f"[stones_probe] round {st['rounds']}: {st['games_finished']} / {games} games finished, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
            if partial_path and (st["rounds"] // 256) % 16 == 0:
                gs += eng.fetch_games()
                json.dump({"censored": True, "games_started": games, "games_finished": len(gs), "rounds": st["rounds"], "num_simulation": sims, "seed": seed,
                           "lengths_of_finished": [len(g["moves"]) for g in gs], "winners_of_finished": [g["winner"] for g in gs],
                           "unfinished_have_at_least_stones": int(st["rounds"] // sims)}, open(partial_path, "w"))
            if st["live_games"] == 0:
                break
    eng.run()
    dt = time.perf_counter() - t0
    st = eng.stats()
    gs += eng.fetch_games()
    eng.close()
    lens = np.array([len(g["moves"]) for g in gs], np.float64)
    wins = np.array([g["winner"] for g in gs])
    return {"games": int(lens.size), "num_simulation": sims, "stones_per_game": float(lens.mean()), "stones_std": float(lens.std()),
            "stones_min": int(lens.min()), "stones_max": int(lens.max()), "black_wins": int((wins == -1).sum()), "white_wins": int((wins == 1).sum()),
            "draws": int((wins == 0).sum()), "seconds": dt, "overflow_games": st["overflow_games"]}


def cpu_baseline(wl, stones, seconds_budget=20.0):
    """The CPU oracle (a port of the reference algorithm: the Rust path cannot be built, no rustc) on the host cores, as the reference
    runs it: one game per task on a pool of `cores` workers, no cross-game batching (AsyncAgent::run, agent/async_agent/mod.rs:100-137).
    Bounded sample.  Tree-only workload: FULL games on all cores.  Net workloads: a full move is 400 searches of 8 FP32 net evaluations --
    minutes per move on a core -- so the sample is searches of each game's opening move and games/s = searches/s / (S * stones/game)
    with the same measured stones/game as the GPU line (tools/cpu_baseline_full.py runs BASELINE.md 3's >= 3 full moves per game; its
    result is committed under profiles/)."""
    import threading

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge
    import c6oracle as o

    cores = usable_cores()
    n = wl["n"]
    if wl["evaluator"] != "net":
        done, stones_played = 0, 0
        t0 = time.time()
        while time.time() - t0 < seconds_budget:
            res = o.self_play(n, num=cores, threads=cores, num_simulation=wl["sims"], seed=1000 + done, eval_kind=o.EVAL_UNIFORM)
            done += len(res)
            stones_played += sum(len(r["path"]) for r in res)
        dt = time.time() - t0
        return {"value": done / dt, "unit": "games/s", "cores": cores, "kind": "port", "stones_per_game": stones_played / max(done, 1),
                "sims_per_s": stones_played * wl["sims"] / dt,
                "sample": f"{done} FULL games ({cores} at a time, one per core, {dt:.1f} s), uniform evaluator, {wl['sims']} simulations per move"}
    blob = ge.make_random_blob(n, wl["channels"], wl["blocks"], np.random.default_rng(0))
    net = o.Net(n, wl["channels"], wl["blocks"], blob)
    fn, ctx = net.eval_fn()
    done = [0] * cores
    stop = time.time() + seconds_budget

    def work(i):
        az = o.AlphaZero(n, num_simulation=wl["sims"], eval_kind=o.EVAL_CUSTOM, eval_fn=fn, ctx=ctx, seed=0, game_uid=i)
        s = o.sim_from_game(o.new_game(n))
        az.init(s)
        while time.time() < stop:
            az.search(s)
            done[i] += 1
        az.close()

    t0 = time.time()
    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    sims = sum(done)
    net.close()
    sims_s = sims / dt
    return {"value": sims_s / (wl["sims"] * stones) if stones else None, "unit": "games/s", "cores": cores, "kind": "port", "sims_per_s": sims_s,
            "sample": f"{sims} searches of the opening move ({cores} games, one per core, {dt:.1f} s) with the oracle's FP32 net; "
                      + (f"games/s = searches/s / ({wl['sims']} * {stones:.1f} measured stones/game)" if stones else "no game length measured at this budget: searches/s only")}


def rank_topology(torch, dist, world, local_rank, backend):
    """{ranks, backend, distinct_devices, rccl_ranks, devices}: every rank's (host, device identity) all-gathered; distinct_devices is what
    `n_gpus` reports.  rccl_ranks = the size of the RCCL communicator the tuple exchange runs on (0: no RCCL collective in this run)."""
    import socket
    p = torch.cuda.get_device_properties(local_rank)
    ident = (socket.gethostname(), str(getattr(p, "uuid", "")), int(getattr(p, "pci_domain_id", -1)), int(getattr(p, "pci_bus_id", -1)),
             int(getattr(p, "pci_device_id", -1)), local_rank if not hasattr(p, "pci_bus_id") and not hasattr(p, "uuid") else -1)
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)
    else:
        idents = [ident]
    distinct = len({tuple(i) for i in idents})
    is_rccl = world > 1 and backend == "nccl"
    return {"ranks": world, "backend": ("nccl (RCCL)" if backend == "nccl" else backend) if world > 1 else "none (one rank, no collective)",
            "distinct_devices": distinct, "rccl_ranks": dist.get_world_size() if is_rccl else 0,
            "rehearsal": bool(world > 1 and (distinct < world or not is_rccl)),
            "devices": sorted({f"{i[0]}:{i[2]:04x}:{i[3]:02x}:{i[4]:02x}" if i[3] >= 0 else f"{i[0]}:{i[1] or i[5]}" for i in idents})}


def under_profiler():
    """rocprofv3 preloads its tool library into the process it runs"""
    return "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF_", "ROCPROFILER_")) for k in os.environ)


def main():
    # A `rocprofv3 --kernel-trace` run of a process whose record pools are mapped with HIP virtual memory management did not come back on this
    # pool's runtime: under the profiler the pools are plain allocations (nothing the timed kernels do depends on it)
    if under_profiler():
        os.environ.setdefault("C6_NO_VMM", "1")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--games", type=int, default=0, help="override games per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32_direct", "bf16x3", "f16f8"],
                    help="arithmetic of the 3x3 convolutions for `value` (default: the reference's f32); the others are measured too at N = 1 "
                         "and reported beside it unless --single-precision")
    ap.add_argument("--single-precision", action="store_true")
    ap.add_argument("--no-probe", action="store_true", help="skip the live stones-per-game probe and the steady-state tree measurement")
    ap.add_argument("--no-side-reports", action="store_true", help="skip the side reports that build engines of their own (small_batches: 1 and 64 games; many_games: 16 384 games) "
                    "-- ADVICE r5: they lengthen the default run by about 15 s")
    ap.add_argument("--probe-games", type=int, default=256)
    ap.add_argument("--gather-every", type=int, default=0, help="rounds between tuple exchanges at N > 1 (0: 1 for net workloads, 64 otherwise)")
    ap.add_argument("--exchange-always", action="store_true", help="run the tuple exchange every --gather-every rounds at N = 1 too (its cost on one GPU: "
                    "compare ms_per_step with and without)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo only to "
                    "exercise the N > 1 path on a 1-GPU box together with C6_BENCH_SHARE_GPU=1)")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if args.games:
        wl["games"] = args.games

    import torch
    import torch.distributed as dist

    from c6amd_loader import load_package
    pkg = load_package()
    from alphazero_connect6_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libconnect6_amd has no CPU fallback)")
    if os.environ.get("C6_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    ndev_visible = torch.cuda.device_count()
    if local_rank >= ndev_visible:      # a launcher that gives every rank its own visibility mask (HIP_VISIBLE_DEVICES = one GPU): the rank's device is ordinal 0 there
        local_rank = local_rank % max(ndev_visible, 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # what the ranks really run on: the line says it, so that a rehearsal (several ranks sharing a GPU, or gloo carrying the all-gathers)
    # can never be read as a multi-GPU RCCL measurement
    topo = rank_topology(torch, dist, world, local_rank, args.backend)

    n, G, S = wl["n"], wl["games"], wl["sims"]
    is_net = wl["evaluator"] == "net"
    uid_base, uid_stride = shard.uid_plan(G, world, rank)
    ev = {"net": pkg.EVAL_NET, "uniform": pkg.EVAL_UNIFORM, "random": pkg.EVAL_RANDOM}[wl["evaluator"]]
    gather_every = args.gather_every or (1 if is_net else 64)
    cap = 8192      # tuple records per exchange buffer (8.9 MB at 19x19): a round finishes ~G / (S * stones) games = a handful of records

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(precision, steps, warmup, evaluator=ev, settle_rounds=0, geom=None, chunk=0):
        """one engine, `warmup` untimed + `steps` timed rounds; returns a dict.  geom = (n, G, S): another geometry than the workload's
        (the tree-kernel side report)"""
        n, G, S = geom or (wl["n"], wl["games"], wl["sims"])
        eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=2024, evaluator=evaluator, net_channels=max(wl["channels"], 64),
                         net_blocks=max(wl["blocks"], 1), recycle=True, stagger=True, device=local_rank, game_uid_base=uid_base if not geom else 0,
                         game_uid_stride=uid_stride if not geom else 0, net_precision=precision)
        if evaluator == pkg.EVAL_NET:
            import __graft_entry__ as ge
            eng.load_weights(ge.make_random_blob(n, wl["channels"], wl["blocks"], np.random.default_rng(0)))
        # training tuples: double-buffered torch-owned device buffers, pooled over the ranks on a side stream (shard.TuplePool)
        pool = shard.TuplePool(eng, cap, world, backend=args.backend if world > 1 else "nccl", keep=False)

        trace = os.environ.get("C6_BENCH_TRACE") == "1"

        def run_rounds(k):
            # the exchange is pipelined: rounds are enqueued, THEN the buffer retired one interval ago is pooled (the host waits only
            # for its counts, which sit behind work already enqueued) -- no per-round synchronisation of the engine's stream
            done = 0
            while done < k:
                c = min(chunk or gather_every, k - done)
                ta = time.perf_counter()
                eng.step(c)
                tb = time.perf_counter()
                done += c
                if world > 1 or args.exchange_always or done == k:
                    pool.exchange()
                if trace:
                    print(f"[trace] rounds {done - c}..{done}: step {1000 * (tb - ta):.1f} ms, exchange {1000 * (time.perf_counter() - tb):.1f} ms", file=sys.stderr, flush=True)

        if settle_rounds:
            left = settle_rounds
            while left > 0:      # long tree-only runs: let the pools grow if they must (host check every now and then)
                c = min(left, 4096)
                eng.step(c)
                left -= c
                eng.maintain()
        run_rounds(warmup)
        eng.sync()
        s0 = eng.stats()
        eng.profile(True)
        eng.profile_read(reset=True)
        barrier()
        t0 = time.perf_counter()
        run_rounds(steps)
        t_enq = time.perf_counter()
        eng.sync()
        t_sync = time.perf_counter()
        barrier()
        dt = time.perf_counter() - t0
        if trace:
            print(f"[trace] timed region: enqueue {t_enq - t0:.3f} s, engine sync {t_sync - t_enq:.3f} s, barrier {t0 + dt - t_sync:.3f} s", file=sys.stderr, flush=True)
        prof = eng.profile_read(reset=True)
        eng.profile(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        s1 = eng.stats()
        pool.exchange()
        pool.exchange()
        pooled = pool.pooled_total()
        free_b, total_b = torch.cuda.mem_get_info()
        r = dict(dt=dt, prof=prof, st=s1, pooled=pooled, hbm_used_bytes=total_b - free_b, hbm_total_bytes=total_b, exchanges=pool.exchanges, payload_gathers=pool.payload_gathers,
                 games_in_window=s1["games_finished"] - s0["games_finished"], stones_in_window=s1["stones"] - s0["stones"],
                 sims_in_window=s1["simulations"] - s0["simulations"])
        eng.close()
        del pool
        torch.cuda.empty_cache()
        return r

    PRECS = {"fp32": pkg.PREC_FP32, "fp32_direct": pkg.PREC_FP32_DIRECT, "bf16x3": pkg.PREC_BF16X3, "f16f8": pkg.PREC_F16F8}
    DTYPES = {"fp32": "f32 (f32 operands, f32 products, f32 accumulate on v_mfma_f32_32x32x2_f32; Winograd form: F(4,3) / F(3,3) down, F(6,3) / F(4,3) / F(3,3) across)",
              "fp32_direct": "f32 (the same arithmetic as a direct implicit GEMM)",
              "bf16x3": "bf16x3 (f32 operands split hi+mid in bf16, 3 bf16 MFMAs per product, f32 accumulate; net outputs within 1e-5)",
              "f16f8": "f16+f8 (f32 operands split hi+lo in fp16: x_hi*w_hi on fp16 MFMAs, cross terms on MX-scaled fp8 MFMAs, f32 accumulate; net outputs "
                       "within 1e-5 inside its guarded range)"}
    prec = PRECS[args.precision]
    settle = 0 if is_net else int(S * wl["stones_prior"])      # tree-only workload: one game length of rounds, so that mid-game trees are timed
    head = measure(prec, args.steps, args.warmup, settle_rounds=settle)
    others, relaxed = [], None
    if is_net and world == 1 and not args.single_precision and n == 19:      # (the other arithmetics' side reports are the 19x19 benchmark's)
        for name, o_prec in PRECS.items():      # measured in the same run, reported beside `value`
            if o_prec != prec:
                try:
                    others.append((name, measure(o_prec, args.steps, args.warmup)))
                except pkg.C6Error as e:      # e.g. C6_ERR_RANGE from the guarded arithmetic: reported, never hidden
                    others.append((name, {"error": str(e)}))
        relaxed = measure(pkg.PREC_BF16, args.steps, args.warmup)      # side report only, never `value`

    # ---- stones per game: measured --------------------------------------------------------------------------------------
    probes_ignored = []
    probe_file = committed_stones_probe(args.workload if args.workload != "config4" else "config2", S, pkg.rng_spec(), probes_ignored) if is_net else None
    live = None
    if world == 1 and is_net and not args.no_probe:
        live = stones_probe(pkg, wl, args.probe_games, 8, prec, device=local_rank)
    if is_net:
        if probe_file:
            stones, stones_src = probe_file["stones_per_game"], (f"measured: {probe_file['games']} complete games of this net at {probe_file['num_simulation']} "
                                                                  f"simulations/move on MI355X (profiles/{probe_file['file']}, tools/stones_probe.py; {pkg.rng_spec()}, "
                                                                  + ("these kernel sources)" if probe_file["same_sources"] else
                                                                     "played on earlier kernel sources under the same random-stream specification and arithmetic: the same process -- the parity tests pin the engine's "
                                                                     "games to the oracle's, stone for stone with the built-in evaluators, to 1e-5 in the priors with the net)"))
        else:
            # No probe at this workload's own budget: `value` is NOT derived from a probe at another budget (game length at 8 simulations
            # per move is 106 stones, at 400 it is 136) -- simulations/s is the measurement, games/s stays null
            stones, stones_src = None, (f"none: no committed probe of complete games at {S} simulations/move for this net under {pkg.rng_spec()}; value is null, "
                                        "sims_per_s is the measurement (value_with_live_probe: the 8-simulation live probe, for orientation only)")
    else:
        # tree-only workload: complete games are cheap -- play a batch to the end (no recycling: every game counts, short and long alike).
        # The ratio stones / finished games inside a timed window is NOT the game length while the games' phases are still correlated
        # (all slots started together: 45 in a window one game length after the start, against 65.5 for complete games)
        live = stones_probe(pkg, wl, min(G, 8192), S, 0, device=local_rank) if (rank == 0 or world == 1) and not args.no_probe else None
        if live:
            stones, stones_src = live["stones_per_game"], f"measured in this run: {live['games']} complete games on the engine (same evaluator, same noise), mean length"
        else:
            stones, stones_src = wl["stones_prior"], "PRIOR (--no-probe, or a rank other than 0)"

    # ---- steady state of the tree kernels at this workload's geometry (N = 1) ---------------------------------------------
    tree_steady = None
    if world == 1 and is_net and not args.no_probe:
        k = 200
        op = measure(0, k, 20, evaluator=pkg.EVAL_UNIFORM)
        ss = measure(0, k, 20, evaluator=pkg.EVAL_UNIFORM, settle_rounds=min(int(S * (stones or live["stones_per_game"])), 60000))
        tree_steady = {"evaluator": "uniform (tree + board kernels only, same G / N / S)", "opening_ms_per_round": 1000.0 * op["dt"] / k,
                       "steady_ms_per_round": 1000.0 * ss["dt"] / k, "steady_max_records_used": ss["st"]["max_records_used"],
                       "records_per_game": ss["st"]["records_per_game"], "pool_grows": ss["st"]["pool_grows"],
                       "steady_select_steps_per_sim": ss["st"]["select_steps"] / max(ss["st"]["simulations"], 1),
                       "abandoned_games": ss["st"]["abandoned_games"]}

    # ---- K1 (board kernel) and K2 (tree kernel) against the HBM roofline, in every default run (N = 1) ----------------------------
    board_side, tree_side = None, None
    if world == 1 and not args.no_probe:
        board_side = {"kernel": "c6_board_kernel<N, true> (Game::set + winner scan, one stone per game) / <N, false> (winner scan only)",
                      "algorithmic_bytes_per_board_step": n * n + 2, "peak_gbs": PEAK_HBM_GBS, "cases": []}
        for games in sorted({G, 1 << 20, 1 << 22}):
            for scan in (False, True):
                ms = pkg.bench_board(n, games, 30, scan_only=scan)
                b = (n * n + (1 if scan else 2)) * games
                board_side["cases"].append({"games": games, "launch": "search-only" if scan else "board-step", "us_per_launch": 1e3 * ms,
                                            "algorithmic_gbs": b / ms / 1e6, "frac": b / ms / 1e6 / PEAK_HBM_GBS})
        c5 = WORKLOADS["config5"]
        k = 2048      # (r4: 256 rounds were 32 ms -- the ~2 ms of synchronisation around the timed region read as +7 % per round)
        tm = measure(0, k, 64, evaluator=pkg.EVAL_UNIFORM, settle_rounds=int(c5["sims"] * c5["stones_prior"]), geom=(c5["n"], c5["games"], c5["sims"]), chunk=64)
        d_ = tm["st"]["select_steps"] / max(tm["st"]["simulations"], 1)
        bsim = (12.0 * d_ + 13.0) * c5["n"] ** 2
        tree_ms = tm["prof"]["select_ms"] + tm["prof"]["expand_ms"]
        tree_side = {"kernel": "c6_search_kernel<2, 2> at config 5's geometry (32 768 9x9 games, S = 100, uniform evaluator), steady state",
                     "ms_per_round": 1000.0 * tm["dt"] / k, "kernel_ms_per_round": tree_ms / k, "select_steps_per_sim": d_, "algorithmic_bytes_per_sim": bsim,
                     "algorithmic_gbs": bsim * c5["games"] * k / max(tree_ms * 1e-3, 1e-9) / 1e9, "peak_gbs": PEAK_HBM_GBS,
                     "frac": bsim * c5["games"] * k / max(tree_ms * 1e-3, 1e-9) / 1e9 / PEAK_HBM_GBS,
                     "sims_per_s": c5["games"] * k / tm["dt"], "abandoned_games": tm["st"]["abandoned_games"],
                     "note": "bound by the vector ALU (noise sampler) and small HBM transactions, not by streaming bandwidth (DESIGN 5.3)"}

    # ---- the same net at the small end of the batch axis (N = 1, net workloads): what a round costs for one game and for 64 (r5: the GEMM's split form) ------
    small_side = None
    if world == 1 and is_net and not args.no_probe and not args.games and not args.no_side_reports:
        small_side = {"what": "ms per simulation round of this workload's net with few concurrent games (same kernels; below ~800 games the convolution's GEMM "
                              "hands a row of a block's positions to a workgroup instead of the block, for a handful of games to four waves: DESIGN 5.1b)",
                      "cases": []}
        try:      # (a side report: its failure must not cost the line)
            for g_small in (1, 64):
                sm = measure(prec, 100, 20, geom=(n, g_small, S))
                small_side["cases"].append({"games": g_small, "ms_per_round": 1000.0 * sm["dt"] / 100, "sims_per_s": g_small * 100 / sm["dt"]})
        except Exception as ex:
            small_side["error"] = repr(ex)

    # ---- north_star's target in the driver's own line: >= 10 k concurrent 19x19 games at 400 simulations per move on ONE MI355X ----------------
    many_side = None
    if world == 1 and is_net and not args.no_probe and not args.games and args.workload == "config2" and not args.no_side_reports:
        many_side = {"what": "this workload's net and budget with 16 384 concurrent games on the one GPU (4 net passes of 32 768 dihedral images per round): "
                             "north_star's target is >= 10 000 concurrent 19x19 games at 400 simulations per move", "games": 16384}
        try:      # (a side report: its failure must not cost the line)
            mg = measure(prec, 5, 2, geom=(n, 16384, S))
            sims_mg = mg["sims_in_window"] / mg["dt"]
            many_side.update(ms_per_round=1000.0 * mg["dt"] / 5, sims_per_s=sims_mg, rounds_timed=5,
                             pool_bytes=mg["st"]["pool_bytes"], hbm_used_bytes=mg["hbm_used_bytes"], hbm_total_bytes=mg["hbm_total_bytes"],
                             live_games=mg["st"]["live_games"], overflow_games=mg["st"]["overflow_games"], abandoned_games=mg["st"]["abandoned_games"])
        except Exception as ex:
            many_side["error"] = repr(ex)

    if rank == 0:
        def line(m):
            sims = m["sims_in_window"] * world / m["dt"] if world == 1 else args.steps * G * world / m["dt"]
            return sims, (sims / (S * stones) if stones else None)

        def conv_roofline(m, name):
            prof, dt_ = m["prof"], m["dt"]
            kern, desc = conv_kernel_of(name, n)
            gemm_ms = prof["conv_ms"] - prof["conv_pre_ms"] - prof.get("conv_post_ms", 0.0)
            peak = PEAK_FP32_MFMA_TFLOPS if name.startswith("fp32") else PEAK_BF16_MFMA_TFLOPS
            L = max(prof["conv_launches"], 1)
            alg = prof["conv_flops"] / (gemm_ms * 1e-3) / 1e12      # algorithmic (direct-convolution) FLOPs over the dominant kernel's time
            exe = prof["conv_exec_flops"] / (gemm_ms * 1e-3) / 1e12 # matrix FLOPs the kernel executes over the same time
            std = args.workload == "config2" and not args.games
            rec = pmc_record(kern) if std else {}
            # `achieved` / `frac`: what the named kernel DOES against the roof that bounds it (executed matrix FLOP/s over the dense peak --
            # a fraction of a roof, <= 1).  The algorithmic count of SURVEY 8d (2*9*C^2*N^2 per board and layer) is reported beside it:
            # Winograd executes 3.8x fewer products than that, so algorithmic / peak is a ratio, not a roofline fraction.
            ach = exe if name.startswith("fp32") else alg
            traffic = rec.get("hbm_bytes_per_launch")
            if name == "fp32" and traffic is not None:      # a layer's GEMM is two launches: the blocks with rows of six positions (<false>, 708 of a board's 788 positions) and of five (<true>, 80)
                traffic += pmc_record("c6_w6_gemm_kernel<true, false, false>").get("hbm_bytes_per_launch", 0.0)
            r = {"bound": "mfma", "kernel": f"{kern} ({desc})", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                 "traffic": traffic, "avg_launch_ms": gemm_ms / L, "launches": prof["conv_launches"],
                 "algorithmic_flops_per_launch": prof["conv_flops"] / L, "executed_flops_per_launch": prof["conv_exec_flops"] / L,
                 "algorithmic_tflops": alg, "algorithmic_ratio": alg / peak,
                 "executed_mfma_tflops": exe, "conv_share_of_step": prof["conv_ms"] / (1000.0 * dt_)}
            r["frac_executed"] = exe / peak
            # the layer as a whole: the same executed FLOPs over GEMM + boundary-kernel time (an MFMA-bound phase and an HBM-bound phase that
            # alternate: their times add)
            r["layer_frac"] = prof["conv_exec_flops"] / (prof["conv_ms"] * 1e-3) / 1e12 / peak
            if name == "fp32" and n != 19:
                r["note"] = ("achieved / frac = EXECUTED matrix FLOP/s of the GEMM kernel over the dense FP32-MFMA peak; layer_frac = the same FLOPs over GEMM + boundary-kernel "
                             "time; algorithmic_tflops counts 2*9*C^2*N^2 per board and layer (SURVEY 8d)")
            if name == "fp32" and n == 19:
                r["note"] = ("a layer's GEMM is two launches of one kernel template -- c6_w6_gemm_kernel<false, false, false> (the tile classes with rows of six positions: 708 of a board's "
                             "788 products) then <true> (class D', rows of five: 80) -- timed together: avg_launch_ms, achieved and traffic are per layer = the pair; "
                             "achieved / frac = EXECUTED matrix FLOP/s of the GEMM kernel over the dense FP32-MFMA peak; layer_frac = the same FLOPs over GEMM + "
                             "boundary-kernel time; algorithmic_tflops / algorithmic_ratio count 2*9*C^2*N^2 per board and layer (SURVEY 8d), of which this Winograd "
                             "tiling executes 1 / 4.12 (788 products per channel pair and board instead of 3249)")
                nbd, Cc, tiles = min(G * 8, 32768), wl["channels"], ((n + 3) // 4) ** 2
                # between the GEMM kernels: one input transform in front of the tower, then ONE kernel per layer boundary (output transform of
                # layer l + input transform of layer l + 1; the activations between the two convolutions of a block are never stored), and
                # the last output transform, which applies the heads' 1x1 filters.  HBM-bound kernels.
                nl = 2 * (wl["blocks"] - 1)
                tw = (n + 3) // 4
                edge = 2 * tw - 1 if n % 4 == 3 else 0      # conv_wino4.hip's tiling: the last tile row / column is F(3,3) in one direction, 30 positions / 20 row sums
                vpos, zrows = 36 * (tiles - edge) + 30 * edge, 24 * (tiles - edge) + 20 * edge
                if n == 19 and Cc == 256 and wl["blocks"] - 1 >= 2:      # conv_wino6.hip: 8 tiles of 48 positions / 32 row sums, 2 of 40 / 32, 4 of 36 / 24, 6 of 30 / 20
                    vpos, zrows = 8 * 48 + 2 * 40 + 4 * 36 + 6 * 30, 10 * 32 + 4 * 24 + 6 * 20
                act_b, v_b, z_b = n * n * Cc * 4 * nbd, vpos * Cc * 4 * nbd, zrows * Cc * 4 * nbd
                # V: the transformed positions of a tile, GEMM input; Z: the row sums the GEMM kernel leaves (the F(4) half of the output
                # transform applied in its registers).  First layer: V written from the boards (the stem is computed, not read); inside a block:
                # Z in, V out; across blocks: Z + residual in, activations + V out -- except behind the first block, whose residual (the stem) is
                # computed too; last layer: Z + residual in
                tr_bytes = v_b + (nl // 2) * (z_b + v_b) + (nl // 2 - 1) * (z_b + v_b + 2 * act_b) - act_b + (z_b + act_b)
                tr_ms = (prof["conv_pre_ms"] + prof.get("conv_post_ms", 0.0)) / L
                r["transforms"] = {"kernels": "c6_w6_stem_in_kernel (first layer: stem + input transform), c6_w6_mid_kernel (between layers), "
                                              "c6_w6_out_kernel (last layer + heads): HBM bound",
                                   "avg_ms_per_layer": tr_ms, "algorithmic_bytes_per_layer": tr_bytes / nl,
                                   "achieved_gbs": tr_bytes / nl / max(tr_ms, 1e-9) / 1e6,
                                   "traffic_mid_kernel": pmc_record("c6_w6_mid_kernel<19, false>").get("hbm_bytes_per_launch") if std else None}
                r["conv_layer_tflops_algorithmic"] = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
            if name == "bf16x3":
                r.update(mfma_flops_per_algorithmic_flop=3, executed_mfma_tflops=3 * alg, frac_executed=3 * alg / peak)
            if name == "f16f8":
                r.update(scheme_ceiling_tflops=peak / (1.0 + 10.0 / 9.0), frac_of_scheme_ceiling=alg * (1.0 + 10.0 / 9.0) / peak)
            if "mfma_util" in rec:
                r["pmc_mfma_busy_frac"] = rec["mfma_util"]
                r["pmc_effective_clock_ghz"] = rec.get("effective_clock_ghz")
                r["pmc_summary"] = rec.get("summary")
            return r

        sims_s, games_s = line(head)
        st = head["st"]
        out = {
            "metric": "self-play games/sec at 400 sims/move, 19x19, 1/2/4/8 MI355X" if args.workload in ("config2", "config4")
                      else f"self-play games/sec ({args.workload})",
            "value": games_s, "unit": "games/s", "n_gpus": topo["distinct_devices"], "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * head["dt"] / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[args.precision] if is_net else "f32/f64 (tree arithmetic), int8 boards",
            "data": "synthetic (random-init weights, self-generated positions)",
            "config": {"workload": f"{args.workload}: {G} concurrent {n}x{n} games per GPU, {S} sims/move, "
                                   + (f"stem + {wl['blocks'] - 1} residual blocks, C={wl['channels']}" if is_net else "on-device uniform evaluator"),
                       "games_per_gpu": G, "board": n, "num_simulation": S, "net_blocks": wl["blocks"], "net_channels": wl["channels"],
                       "stones_per_game": stones, "stones_per_game_source": stones_src,
                       "step": ("one simulation round over all games (8 dihedral net evaluations per game), staggered move phases" if is_net else
                                "one simulation round over all games, staggered move phases, measured in the steady state (after one game length of rounds)"),
                       "parallelism": "1 GPU" if world == 1 else
                                      (f"games sharded over {world} ranks on {topo['distinct_devices']} distinct GPU(s); training tuples pooled every {gather_every} round(s) by "
                                       + ("an RCCL all-gather on a side stream" if topo["rccl_ranks"] else f"a {args.backend} all-gather through host copies")
                                       + " (shard.TuplePool)"
                                       + (" -- a REHEARSAL of the N > 1 path, not a multi-GPU measurement: " + ("the ranks share a GPU" if topo["distinct_devices"] < world else "")
                                          + ("; " if topo["distinct_devices"] < world and not topo["rccl_ranks"] else "") + ("no RCCL" if not topo["rccl_ranks"] else "")
                                          if topo["rehearsal"] else ""))},
            "ranks": topo["ranks"], "backend": topo["backend"], "distinct_devices": topo["distinct_devices"], "rccl_ranks": topo["rccl_ranks"],
            "rehearsal": topo["rehearsal"], "devices": topo["devices"],
            "sims_per_s": sims_s, "boards_per_s": sims_s * 8,
            "measured_in_window": {"games_finished": head["games_in_window"], "stones": head["stones_in_window"], "simulations": head["sims_in_window"],
                                   "seconds": head["dt"]},
            "engine": {"select_steps_per_sim": st["select_steps"] / max(st["simulations"], 1), "max_records_used": st["max_records_used"],
                       "records_per_game": st["records_per_game"], "pool_grows": st["pool_grows"], "overflow_games": st["overflow_games"],
                       "abandoned_games": st["abandoned_games"], "stones": st["stones"], "tuples_pooled": head["pooled"],
                       "exchanges": head["exchanges"], "payload_gathers": head["payload_gathers"]},
        }
        if not is_net and head["games_in_window"]:
            # `value` = simulations/s / (S * mean length of complete games).  The completions counted in the window are reported
            # beside it: they converge to `value` only once the slots' game phases have decorrelated (many game lengths)
            out["games_per_s_counted_in_window"] = head["games_in_window"] / head["dt"]
            out["stones_per_finished_game_in_window"] = head["stones_in_window"] / head["games_in_window"]
        if is_net and probe_file:      # the divisor's own uncertainty travels with it (config 3's probe is 70-odd games, config 2's 1024)
            se = probe_file.get("standard_error") or (probe_file["stones_std"] / max(probe_file["games"], 1) ** 0.5)
            out["config"]["stones_per_game_games"] = probe_file["games"]
            out["config"]["stones_per_game_standard_error"] = se
            out["value_relative_standard_error_from_stones"] = se / probe_file["stones_per_game"]
        if live:
            out["config"]["stones_per_game_live_probe"] = live
            if is_net:
                out["value_with_live_probe"] = sims_s / (S * live["stones_per_game"])
        if probes_ignored:
            out["config"]["stones_probes_ignored"] = probes_ignored
        if is_net:
            pooled_p = pooled_stones_probes(args.workload if args.workload != "config4" else "config2", S)
            if pooled_p and len(pooled_p["probes"]) > 1:
                out["config"]["stones_per_game_all_probes_at_this_budget"] = pooled_p
                out["value_with_all_probes_at_this_budget"] = sims_s / (S * pooled_p["stones_per_game"])
        if is_net and head["prof"]["conv_launches"]:
            out["roofline"] = conv_roofline(head, args.precision)
            out["net_tflops_end_to_end"] = net_flops_per_board(n, wl["channels"], wl["blocks"]) * 8 * sims_s / 1e12
            for name, m in others:
                if "error" in m:
                    out[name + "_path"] = m
                    continue
                o_sims, o_games = line(m)
                out[name + "_path"] = {"value": o_games, "unit": "games/s", "ms_per_step": 1000.0 * m["dt"] / args.steps, "sims_per_s": o_sims,
                                       "dtype": DTYPES[name], "roofline": conv_roofline(m, name)}
            if relaxed:      # C6_PREC_BF16: NOT parity grade (value error ~1e-4..1e-3 > the 1e-5 bar) -- a side report, never `value`
                r_sims, r_games = line(relaxed)
                out["relaxed_bf16_path"] = {"value": r_games, "unit": "games/s", "ms_per_step": 1000.0 * relaxed["dt"] / args.steps, "sims_per_s": r_sims,
                                            "dtype": "bf16 (one MFMA per product; net outputs within ~1e-3 of FP32: below the parity bar, reported for headroom only)",
                                            "conv_tflops_algorithmic": relaxed["prof"]["conv_flops"] / (relaxed["prof"]["conv_ms"] * 1e-3) / 1e12}
            if tree_steady:
                d_ms = tree_steady["steady_ms_per_round"] - tree_steady["opening_ms_per_round"]
                tree_steady["steady_state_ms_per_step_estimate"] = out["ms_per_step"] + d_ms
                tree_steady["value_at_steady_state_estimate"] = games_s * out["ms_per_step"] / (out["ms_per_step"] + d_ms) if games_s is not None else None
                out["steady_state"] = tree_steady
        else:
            # tree-bound workload: algorithmic bytes per simulation, SURVEY 8d: (12*d + 13) * N^2
            d = st["select_steps"] / max(st["simulations"], 1)
            prof = head["prof"]
            tree_ms = prof["select_ms"] + prof["expand_ms"]
            bytes_sim = (12.0 * d + 13.0) * n * n
            ach = bytes_sim * G * args.steps / max(tree_ms * 1e-3, 1e-9) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "c6_search_kernel<2, 2> (a whole round -- search, and the move when its budget is spent -- per game and wave; "
                                                          "up to 64 rounds of every game per launch)", "achieved": ach, "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None, "bytes_per_sim": bytes_sim,
                               "avg_launch_ms": tree_ms / max(prof["select_launches"], 1), "rounds_per_launch": args.steps / max(prof["select_launches"], 1),
                               "note": "algorithmic bytes (SURVEY 8d: (12 d + 13) N^2 per simulation) over the kernel's time; the kernel is bound by the vector "
                                       "ALU (the Gamma sampler of the Dirichlet noise is half of it) and by small HBM transactions, not by streaming bandwidth: "
                                       "profiles/r02_tree_ablation.txt"}
        if board_side:
            out["board_kernel"] = board_side
        if tree_side:
            out["tree_kernel"] = tree_side
        if small_side:
            out["small_batches"] = small_side
        if many_side:
            if "sims_per_s" in many_side:
                many_side["value"] = many_side["sims_per_s"] / (S * stones) if stones else None
                many_side["per_game_rate_vs_headline"] = many_side["sims_per_s"] / sims_s
            out["many_games"] = many_side
        out["phase_ms"] = {k: head["prof"][k] for k in ("conv_ms", "conv_pre_ms", "conv_post_ms", "select_ms", "expand_ms", "eval_other_ms", "move_ms")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, stones, args.cpu_seconds)      # (stones None: its value is null too, sims_per_s stays)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
