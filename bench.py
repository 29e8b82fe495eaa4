"""bench.py -- self-play games/sec of the MI355X engine (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload at every N (weak scaling: per-GPU work fixed): BASELINE.json configs[1] --
4096 concurrent 19x19 games per GPU, 400 simulations/move, random-init policy/value net with
"4 blocks" = stem + 3 residual blocks, C = 256.  The 3x3 convolutions run in one of three parity-grade
arithmetics (net outputs within 1e-5 of the FP32 oracle, tests/test_gpu_net.py): `f16f8` (default: fp16 main term + MX-scaled
fp8 cross terms), `bf16x3` (operands split into bf16 hi+mid, three bf16 MFMAs per product) and `fp32` (exact-FP32 MFMAs).  The
headline `value` is the --precision path; the other two are measured in the same run: `bf16x3_path`, `fp32_path`.

A "step" is one simulation ROUND: every live game performs one search() (select -> 8 dihedral
net evaluations -> expand -> one-level backup); game phases are staggered, so each round
G/400 games also make their move (policy -> board kernel -> prune / recycle), i.e. every step
carries the steady-state share of all the work.  games/sec = rounds/sec * G / (400 * stones_per_game);
stones_per_game cannot be measured inside a benchmark of minutes (one 19x19 game is ~83 000 rounds), the
figure used is stated in config.stones_per_game (uniform-random 19x19 play, mean 208).
Inputs (game state, trees, weights) are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (board, games/GPU, sims, net blocks (reference num_block), channels, evaluator, stones/game prior)
    "config2": dict(n=19, games=4096, sims=400, blocks=4, channels=256, evaluator="net", stones=208.0),
    "config3": dict(n=19, games=16384, sims=800, blocks=8, channels=256, evaluator="net", stones=208.0),
    "config4": dict(n=19, games=8192, sims=400, blocks=4, channels=256, evaluator="net", stones=208.0),
    "config5": dict(n=9, games=32768, sims=100, blocks=0, channels=0, evaluator="uniform", stones=68.0),
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, Peak BF16 MFMA, dense
PEAK_HBM_GBS = 8000.0
# c6_conv3x3_b3s_kernel<256, 19, SPLIT = true, HEAD = false>: the split (parity-grade) instantiation that stores activations --
# five of the six conv launches of a round; the sixth (HEAD = true) hands the heads their 1x1 partial sums instead
B3S_SPLIT = "19, true, false>"
M8_PLAIN = "m8_kernel<256, 19, false>"      # c6_conv3x3_m8_kernel<256, 19, HEAD = false>: five of the six conv launches of a round


def net_flops_per_board(n, C, blocks):
    """F_net(N, k) = 2*9*2*C*N^2 + (k-1)*2*(2*9*C^2*N^2) + heads"""
    n2 = n * n
    heads = 2 * C * 2 * n2 + 2 * (2 * n2) * n2 + 2 * C * n2 + 2 * n2 * 256 + 2 * 256
    return 2 * 9 * 2 * C * n2 + (blocks - 1) * 2 * (2 * 9 * C * C * n2) + heads


def pmc_record(kernel_substr):
    """The dominant kernel's row of the committed rocprofv3 PMC summary (separate --pmc passes of this same command,
    tools/profile_round.sh): HBM bytes per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950),
    matrix-core busy fraction and effective clock."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
        for k, v in d["kernels"].items():
            if kernel_substr in k:
                return v
    except Exception:
        return {}
    return {}


def pmc_traffic(kernel_substr):
    return pmc_record(kernel_substr).get("hbm_bytes_per_launch")


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


def cpu_baseline(wl, seconds_budget=20.0):
    """The CPU oracle (a port of the reference algorithm: the Rust path cannot be built, no rustc) on the
    host cores, bounded sample: every core runs searches of the first move of one game of the same workload."""
    import threading

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as ge
    import c6oracle as o

    cores = usable_cores()
    n = wl["n"]
    net = None
    if wl["evaluator"] == "net":
        blob = ge.make_random_blob(n, wl["channels"], wl["blocks"], np.random.default_rng(0))
        net = o.Net(n, wl["channels"], wl["blocks"], blob)
        fn, ctx = net.eval_fn()
        kind = o.EVAL_CUSTOM
    else:
        fn, ctx, kind = None, None, o.EVAL_UNIFORM
    done = [0] * cores
    stop = time.time() + seconds_budget

    def work(i):
        az = o.AlphaZero(n, num_simulation=wl["sims"], eval_kind=kind, eval_fn=fn, ctx=ctx, seed=0, game_uid=i)
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
    if net:
        net.close()
    sims_s = sims / dt
    return {"value": sims_s / (wl["sims"] * wl["stones"]), "unit": "games/s", "cores": cores, "kind": "port",
            "sims_per_s": sims_s,
            "sample": f"{sims} searches of the opening move ({cores} games, one per core, {dt:.1f} s) with the oracle's FP32 net; "
                      f"games/s = searches/s / ({wl['sims']} * {wl['stones']:.0f})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--games", type=int, default=0, help="override games per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--precision", default="f16f8", choices=["f16f8", "bf16x3", "fp32"],
                    help="arithmetic of the 3x3 convolutions for the headline value; the other one is measured too and reported "
                         "beside it (fp32_path / bf16x3_path) unless --single-precision")
    ap.add_argument("--single-precision", action="store_true")
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
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    n, G, S = wl["n"], wl["games"], wl["sims"]
    uid_base, uid_stride = shard.uid_plan(G, world, rank)
    ev = {"net": pkg.EVAL_NET, "uniform": pkg.EVAL_UNIFORM, "random": pkg.EVAL_RANDOM}[wl["evaluator"]]
    stride = pkg.lib().c6_tuple_stride(n)
    cap = 4096

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(precision, steps, warmup):
        """one engine, `warmup` untimed + `steps` timed rounds; returns (seconds, profile, stats, pooled tuples)"""
        eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=2024, evaluator=ev, net_channels=max(wl["channels"], 64),
                         net_blocks=max(wl["blocks"], 1), recycle=True, stagger=True, device=local_rank, game_uid_base=uid_base,
                         game_uid_stride=uid_stride, net_precision=precision)
        if ev == pkg.EVAL_NET:
            import __graft_entry__ as ge
            eng.load_weights(ge.make_random_blob(n, wl["channels"], wl["blocks"], np.random.default_rng(0)))
        # training tuples go to a torch-owned device buffer that the all-gather reads in place
        tbuf = torch.zeros((cap, stride), dtype=torch.uint8, device="cuda")
        tcount = torch.zeros(1, dtype=torch.int32, device="cuda")
        eng.set_tuple_buffer(tbuf.data_ptr(), cap, tcount.data_ptr())
        pooled_tuples = 0

        def run_rounds(k):
            nonlocal pooled_tuples
            if world == 1:
                eng.step(k)
                eng.sync()
                return
            for _ in range(k):
                eng.step(1)
                eng.sync()
                if args.backend == "nccl":
                    pooled, counts = shard.gather_tuples(tbuf, tcount, world)     # RCCL all-gather over xGMI
                else:
                    pooled, counts = shard.gather_tuples(tbuf.cpu(), tcount.cpu(), world)
                pooled_tuples += int(counts.clamp(max=cap).sum().item())
                tcount.zero_()
                torch.cuda.current_stream().synchronize()

        if ev != pkg.EVAL_NET and world == 1:
            # tree-only workload: a round costs more in the middle game (deeper selects, fuller record pools) than in the
            # opening all games start in -- settle into the steady state (one full game length of rounds, a few seconds)
            # before the warm-up; the net workloads are conv-bound and phase-independent
            eng.step(int(S * wl["stones"]))
            eng.sync()
        run_rounds(warmup)
        eng.profile(True)
        eng.profile_read(reset=True)
        barrier()
        t0 = time.perf_counter()
        run_rounds(steps)
        barrier()
        dt = time.perf_counter() - t0
        prof = eng.profile_read(reset=True)
        eng.profile(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        st = eng.stats()
        eng.close()
        del tbuf, tcount
        torch.cuda.empty_cache()
        return dt, prof, st, pooled_tuples

    def conv_roofline(prof, dt, precision):
        ach = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
        if precision == pkg.PREC_FP32:
            f32k = "c6_conv3x3_kernel<256,3>" if os.environ.get("C6_CONV_VARIANT") == "1" else "c6_conv3x3_f32n_kernel<256,19>"
            name, peak, note, key = f32k + " (FP32 MFMA implicit GEMM, v_mfma_f32_32x32x2_f32)", PEAK_FP32_MFMA_TFLOPS, {}, "f32n"
        elif precision == pkg.PREC_F16F8:
            # per algorithmic flop: one fp16 MFMA flop (dense peak 2 500 TF) + the two cross terms on fp8 MFMAs (dense peak 5 000 TF)
            # whose K = 128 covers two taps, the ninth tap alone: 10/9 * 2 fp8 flops.  Matrix-pipe time per algorithmic flop is
            # (1 + 10/9) / 2 500 TF, i.e. the scheme's ceiling is 1 184 TF algorithmic
            name, peak, key = ("c6_conv3x3_m8_kernel<256,19> (mixed fp16 + MX-fp8 implicit GEMM, v_mfma_f32_16x16x32_f16 + "
                               "v_mfma_scale_f32_16x16x128_f8f6f4)"), PEAK_BF16_MFMA_TFLOPS, M8_PLAIN
            note = {"matrix_pipe_time_per_algorithmic_flop_vs_fp16": 1.0 + 10.0 / 9.0, "scheme_ceiling_tflops": peak / (1.0 + 10.0 / 9.0),
                    "frac_of_scheme_ceiling": ach * (1.0 + 10.0 / 9.0) / peak}
        else:
            b3 = "b3n" if os.environ.get("C6_CONV_VARIANT") == "6" else "b3s"
            name, peak, key = ("c6_conv3x3_%s_kernel<256,19> (split-bf16 implicit GEMM, v_mfma_f32_%s_bf16)" % (b3, "32x32x16" if b3 == "b3n" else "16x16x32")), PEAK_BF16_MFMA_TFLOPS, B3S_SPLIT
            note = {"mfma_flops_per_algorithmic_flop": 3, "executed_mfma_tflops": 3 * ach, "frac_executed": 3 * ach / peak}
        std = args.workload == "config2" and not args.games
        r = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
             "traffic": pmc_traffic(key) if std else None,
             "avg_launch_ms": prof["conv_ms"] / prof["conv_launches"], "launches": prof["conv_launches"],
             "conv_share_of_step": prof["conv_ms"] / (1000.0 * dt)}
        if std:
            rec = pmc_record(key)
            if "mfma_util" in rec:      # rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (profiles/)
                r["pmc_mfma_busy_frac"] = rec["mfma_util"]
                r["pmc_effective_clock_ghz"] = rec.get("effective_clock_ghz")
        r.update(note)
        return r

    PRECS = {"f16f8": pkg.PREC_F16F8, "bf16x3": pkg.PREC_BF16X3, "fp32": pkg.PREC_FP32}
    prec = PRECS[args.precision]
    dt, prof, st, pooled_tuples = measure(prec, args.steps, args.warmup)
    others = []
    relaxed = None
    if ev == pkg.EVAL_NET and not args.single_precision:
        for name, o_prec in PRECS.items():      # the other parity-grade paths, measured in the same run and reported beside `value`
            if o_prec != prec:
                others.append((name, o_prec) + measure(o_prec, args.steps, args.warmup))
        relaxed = measure(pkg.PREC_BF16, args.steps, args.warmup)      # side report only, never `value`

    if rank == 0:
        def line(dt_):
            rounds_s = args.steps / dt_
            sims = rounds_s * G * world
            return sims, sims / (S * wl["stones"])
        sims_s, games_s = line(dt)
        DTYPES = {pkg.PREC_F16F8: "f16+f8 (f32 operands split hi+lo in fp16: x_hi*w_hi on fp16 MFMAs, the cross terms x_hi*w_lo + x_lo*w_hi on MX-scaled fp8 MFMAs, f32 accumulate; net outputs within 1e-5)",
                  pkg.PREC_BF16X3: "bf16x3 (f32 operands split hi+mid in bf16, 3 bf16 MFMAs per product, f32 accumulate; net outputs within 1e-5)",
                  pkg.PREC_FP32: "f32 (v_mfma_f32_32x32x2_f32, exact f32 products)"}
        dtype = DTYPES[prec] if ev == pkg.EVAL_NET else "f32/f64 (tree arithmetic), int8 boards"
        out = {
            "metric": "self-play games/sec at 400 sims/move, 19x19, 1/2/4/8 MI355X" if args.workload in ("config2", "config4")
                      else f"self-play games/sec ({args.workload})",
            "value": games_s, "unit": "games/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic (random-init weights, self-generated positions)",
            "config": {"workload": f"{args.workload}: {G} concurrent {n}x{n} games per GPU, {S} sims/move, "
                                   + (f"stem + {wl['blocks'] - 1} residual blocks, C={wl['channels']}" if ev == pkg.EVAL_NET else "on-device uniform evaluator"),
                       "games_per_gpu": G, "board": n, "num_simulation": S, "net_blocks": wl["blocks"], "net_channels": wl["channels"],
                       "stones_per_game": wl["stones"], "step": ("one simulation round over all games (8 dihedral net evaluations per game), staggered move phases" if ev == pkg.EVAL_NET else
                                "one simulation round over all games, staggered move phases, measured in the steady state (after one game length of rounds)"),
                       "parallelism": f"games sharded over {world} GPU(s), RCCL all-gather of training tuples per round" if world > 1 else "1 GPU"},
            "sims_per_s": sims_s, "boards_per_s": sims_s * 8,
            "engine": {"select_steps_per_sim": st["select_steps"] / max(st["simulations"], 1), "max_records_used": st["max_records_used"],
                       "overflow_games": st["overflow_games"], "stones": st["stones"], "tuples_pooled": pooled_tuples},
        }
        if ev == pkg.EVAL_NET and prof["conv_launches"]:
            out["roofline"] = conv_roofline(prof, dt, prec)
            out["net_tflops_end_to_end"] = net_flops_per_board(n, wl["channels"], wl["blocks"]) * 8 * sims_s / 1e12
            for name, o_prec, o_dt, o_prof, o_st, _ in others:
                o_sims, o_games = line(o_dt)
                out[name + "_path"] = {"value": o_games, "unit": "games/s", "ms_per_step": 1000.0 * o_dt / args.steps, "sims_per_s": o_sims,
                                       "dtype": DTYPES[o_prec], "roofline": conv_roofline(o_prof, o_dt, o_prec)}
            if relaxed:      # C6_PREC_BF16: NOT parity grade (value error ~1e-4..1e-3 > the 1e-5 bar) -- a side report, never `value`
                r_dt, r_prof, _, _ = relaxed
                r_sims, r_games = line(r_dt)
                out["relaxed_bf16_path"] = {"value": r_games, "unit": "games/s", "ms_per_step": 1000.0 * r_dt / args.steps, "sims_per_s": r_sims,
                                            "dtype": "bf16 (one MFMA per product; net outputs within ~1e-3 of FP32: below the parity bar, reported for headroom only)",
                                            "conv_tflops_algorithmic": r_prof["conv_flops"] / (r_prof["conv_ms"] * 1e-3) / 1e12}
        else:
            # tree-bound workload: algorithmic bytes per simulation, SURVEY 8d: (12*d + 13) * N^2
            d = st["select_steps"] / max(st["simulations"], 1)
            tree_ms = prof["select_ms"] + prof["expand_ms"]
            bytes_sim = (12.0 * d + 13.0) * n * n
            ach = bytes_sim * G * args.steps / max(tree_ms * 1e-3, 1e-9) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "c6_select_kernel + c6_expand_kernel", "achieved": ach, "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None, "bytes_per_sim": bytes_sim}
        out["phase_ms"] = {k: prof[k] for k in ("conv_ms", "select_ms", "expand_ms", "eval_other_ms", "move_ms")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
