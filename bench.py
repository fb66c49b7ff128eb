#!/usr/bin/env python3
"""bench.py — self-play throughput of the MI355X engine on BASELINE.json's north-star configuration.

    python bench.py --gpus N --steps K --warmup W

Workload (configs[1]): 5x5 Tak, half-komi 4, 4096 concurrent games per GPU,
400 simulations per move, classic PUCT + Dirichlet(alpha 0.05, ratio 0.2), beta 0, net5 (20 residual
blocks x 256 filters + RND) with random-init weights (seed 123), synthetic openings.  One "step" is one
self-play move for every game of the shard: 1 root expansion + 400 lock-step simulations, move choice,
subtree-reuse step, restart of finished games, target completion.  Everything a simulation needs
(tree descent, move generation, plane encoding, the net forward, expansion, backup) runs on the GPU with
positions resident in HBM; per move the host only draws the Dirichlet noise and reads root statistics.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL, for the barriers and the final reduction of the
counters), independent shards of 4096 games (weak scaling), and after every move the hand-over of the finished games'
packed target records and replay lines — an all-gather of counts, then of the padded records — the only collective
on the path; by default through the library's own RCCL communicator (tz_comm_*, ncclAllGather on the shard's GPU;
TZ_BENCH_EXCHANGE=torch runs it through torch.distributed instead).  The JSON line names the backend, the world size
and the transport; if RCCL cannot form the ring the run ends with a non-zero exit code (no silent fallback;
TZ_BENCH_BACKEND=gloo is an explicit rehearsal switch).  If the ring is up but the library's own communicator does not come
up on some rank, all ranks agree (one all-reduce) to hand the targets over through the process group instead - the same RCCL
ring - and the line's `exchange` says so with every rank's error; TZ_BENCH_EXCHANGE=native makes that an exit code 4 too.

Prints ONE JSON line (rank 0).  `value` = MCTS simulations/s summed over all ranks.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_BOARD, HALF_KOMI, GAMES, SIMS = 5, 4, 4096, 400
FLOP_PER_POSITION = 1.2071e9          # net5, SURVEY.md §8d
CONV_FLOP_PER_POSITION = 2 * 25 * 256 * 2304   # one 3x3 256->256 conv on one 5x5 board
TOWER_LAYERS = 40                              # net5: 20 residual blocks x 2 convs
# dominant kernel by TZ_TOWER: 2 (default) = whole trunk + heads in one launch, 1 = residual tower in one launch,
# 0 = one launch per conv.  Algorithmic FLOPs per position per launch (2 x MACs, real channel counts):
FUSED_MODE = int(os.environ.get("TZ_TOWER", "2"))
FLOP_PER_LAUNCH_POS = {0: CONV_FLOP_PER_POSITION,
                       1: CONV_FLOP_PER_POSITION * TOWER_LAYERS,
                       2: 2 * 25 * 256 * 32 * 9 + CONV_FLOP_PER_POSITION * TOWER_LAYERS + 2 * 25 * 123 * 2304 + 4 * 25 * 256}[FUSED_MODE]
KERNEL_NAME = {0: "conv_mfma_kernel<5,8,2,9,false,0,true,8,1> (one 3x3 256->256 conv)",
               1: "tower_mfma_kernel<5,8> (20 residual blocks = 40 3x3 256->256 convs, one persistent launch)",
               2: "net_mfma_kernel<5,8,1> (game_repr + first conv + 20 residual blocks + policy conv + value/UBE heads, one persistent launch)"}[FUSED_MODE]
# Of those FLOPs the net kernel does not issue the ones that multiply zero padding: with its square-major row order
# (csrc/tz_nn.hip RowMap) 26 of the 117 (tap, 16-row tile) pairs of a tower conv are all padding on 5x5 and are
# left out at compile time.  Issued MFMA FLOPs per position per launch (incl. the 208-for-200 row padding), for the
# `issued` figures beside the algorithmic ones:
SQUARE_MAJOR = os.environ.get("TZ_NET_ROWS", "square") != "board"
TOWER_TILE_TAPS = (91 if SQUARE_MAJOR else 117, 117)
ISSUED_FLOP_PER_LAUNCH_POS = (2 * 16 * 256 * 32 * (9 * 13 * 1 + TOWER_LAYERS * TOWER_TILE_TAPS[0] * 8 + TOWER_TILE_TAPS[0] * 8 * 0.5)) / 8.0   # first conv (one 32-plane chunk) + tower + policy conv (128 of 256 columns)
FUSED_TOWER = FUSED_MODE >= 1
PEAK_BF16_TFLOPS = 2500.0             # MI355X dense MFMA peak of the 16-bit types (f16 = bf16), MI355X_MICROARCH.md


def cpu_baseline(seconds=12.0):
    """Stand-in for the reference's CPU tch path: the single-threaded CPU oracle search
    (oracle/, restating batched.rs:63-128) driving a LibTorch CPU forward of the same net5 graph, 128 games
    (the reference's BATCH_SIZE).  Returns sims/s on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch

    import nets_torch as T
    import oracle_lib as O
    from takzero_amd import weights as W

    lib = O.load()
    w = W.init_weights(W.ARCH_NET5, seed=123)
    B = 128

    def agent(user, n_envs, states, legal_idx, legal_count, amax, logits_out, value_out, variance_out):
        planes = np.zeros((n_envs, 32 * 25), np.float32)
        for i in range(n_envs):
            lib.tzo_game_repr(C.byref(states[i]), planes[i].ctypes.data_as(C.POINTER(C.c_float)))
        planes = planes.reshape(n_envs, 32, 5, 5)
        pol, val, ube = T.forward(w, planes, 20)
        var = T.variance(w, planes, ube, 5)
        pol = pol.reshape(n_envs, -1).numpy()
        for i in range(n_envs):
            k = legal_count[i]
            idx = np.ctypeslib.as_array(legal_idx, shape=(n_envs * amax,))[i * amax:i * amax + k]
            np.ctypeslib.as_array(logits_out, shape=(n_envs * amax,))[i * amax:i * amax + k] = pol[i, idx]
            value_out[i] = float(val[i])
            variance_out[i] = float(var[i])

    s = O.OracleSearch(lib, B, N_BOARD, HALF_KOMI, agent_kind=0, agent_fn=agent)
    rng = np.random.default_rng(0)
    s.new_openings(rng.integers(0, 16, B))
    betas = np.zeros(B, np.float32)
    s.simulate(betas, 1)  # root expansion, untimed warm-up of torch
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds or steps < 2:
        s.simulate(betas, 1)
        steps += 1
    dt = time.perf_counter() - t0
    return dict(value=B * steps / dt, unit="sims/s", cores=int(torch.get_num_threads()), kind="port",
                sample="%d lock-step simulations of 128 games (5x5, net5 fp32 on LibTorch CPU, oracle search single-threaded), %.1f s"
                       % (steps, dt))


# TZ_PREC_F16C6 (csrc/tz_nn_c6.hip): per (tap, row tile, 16 outputs) of a tower conv 8 fp16 MFMAs (16x16x32) and 4 FP6-scaled
# MFMAs (16x16x128), on 8 boards per workgroup (13 row tiles, 91 of 117 pairs); first conv in the split form (3 fp16 products)
C6_ISSUED_F16 = 2 * 16 * 16 * 32 * 8 * 16 * (TOWER_LAYERS * TOWER_TILE_TAPS[0] + TOWER_TILE_TAPS[0] * 0.5) / 8.0 + 3 * 2 * 16 * 256 * 32 * 9 * 13 / 8.0
C6_ISSUED_FP6 = 2 * 16 * 16 * 128 * 4 * 16 * (TOWER_LAYERS * TOWER_TILE_TAPS[0] + TOWER_TILE_TAPS[0] * 0.5) / 8.0
PEAK_FP6_TFLOPS = 10000.0             # dense FP6 / FP4 MFMA peak (MI355X_MICROARCH.md, Matrix cores)


def roofline_of(precision, prof, sims, evals, games):
    """The `roofline` object of a measured run: algorithmic FLOPs per launch of the fused net kernel (1 197.5 MFLOP x the positions a
    launch evaluated, from the device counters) over its average duration by HIP events on the engine's stream (mcts.profile)."""
    per_launch_positions = evals / max(1.0, sims / games)
    avg_ms = prof["conv_ms"] / prof["conv_launches"]
    achieved = FLOP_PER_LAUNCH_POS * per_launch_positions / (avg_ms * 1e-3) / 1e12
    split = precision in ("f16x2", "f16c8")
    kernel_name, rows, issued, mix_peak = KERNEL_NAME, "square-major, %d of %d (tap, row tile) pairs per tower conv issued" % TOWER_TILE_TAPS, ISSUED_FLOP_PER_LAUNCH_POS, PEAK_BF16_TFLOPS
    if split and FUSED_MODE == 2:
        # 4 boards per workgroup (7 row tiles, 49 of 63 pairs issued) and three products per MAC: hi*hi on fp16 MFMAs, the two
        # corrections on fp16 MFMAs (f16x2) or on FP8 MFMAs of 4x the K at twice the rate (f16c8) - the same issued FLOP count
        kernel_name = ("net_mfma_kernel<5,4,1,SP=%d> (%s; same fusion)" %
                       ((1, "hi/lo fp16 operands, 3 fp16 MFMAs per product") if precision == "f16x2"
                        else (2, "fp16 product + 2 correction products on FP8 E4M3 copies, v_mfma_f32_16x16x128_f8f6f4")))
        rows = "square-major, 49 of 63 (tap, row tile) pairs per tower conv issued"
        issued = 3 * (2 * 16 * 256 * 32 * (9 * 7 * 1 + TOWER_LAYERS * 49 * 8 + 49 * 8 * 0.5)) / 4.0
        # fp16 MFMAs at 2.5 PFLOP/s; f16c8 issues a third of its FLOPs there and two thirds on FP8 MFMAs at 5 PFLOP/s
        mix_peak = 3750.0 if precision == "f16c8" else PEAK_BF16_TFLOPS
    elif precision == "f16c6" and FUSED_MODE == 2:
        kernel_name = ("net_c6_kernel<5,8,1> (fp16 product + 2 correction products on FP6 E2M3 block-scaled copies, "
                       "v_mfma_scale_f32_16x16x128_f8f6f4; same fusion, 8 boards per workgroup)")
        issued = C6_ISSUED_F16 + C6_ISSUED_FP6
        mix_peak = issued / (C6_ISSUED_F16 / PEAK_BF16_TFLOPS + C6_ISSUED_FP6 / PEAK_FP6_TFLOPS)
    traffic = None   # HBM-side bytes per launch of that kernel from the rocprofv3 PMC passes (profiles/)
    tpath = os.path.join(ROOT, "profiles", "tower_pmc_traffic.json")
    if FUSED_MODE == 2 and games == GAMES and os.path.exists(tpath):
        stored = json.load(open(tpath))
        traffic = (stored.get(precision) or {}).get("hbm_bytes_per_launch") if precision != "f16" else stored.get("hbm_bytes_per_launch")
    issued_tf = issued * per_launch_positions / (avg_ms * 1e-3) / 1e12 if FUSED_MODE == 2 else None
    return {"bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
            "traffic_source": ("stored figure, not measured in this run: profiles/tower_pmc_traffic.json (rocprofv3 --pmc "
                               "passes of the same command; 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction)"
                               if traffic is not None else None),
            "kernel": kernel_name,
            "rows": rows if FUSED_MODE == 2 else "board-major",
            "issued_tflops": issued_tf,
            # the MFMA mix's own ceiling: every issued FLOP priced at the dense peak of the instruction that issues it
            "issued_frac_of_mfma_mix_peak": (issued_tf / mix_peak if FUSED_MODE == 2 else None),
            # the chip's own best case beside the sheet figure: a bare loop of independent fp16 MFMAs with every operand in
            # registers reaches 1.99 PFLOP/s (power-limited clock; tools/mfma_f8_probe.hip, profiles/r02_mfma_f8_probe.txt)
            "bare_mfma_loop_tflops": {"value": 1986.0, "source": "stored figure: profiles/r02_mfma_f8_probe.txt (tools/mfma_f8_probe.hip, mode 0)"},
            "avg_launch_ms": avg_ms, "launches": prof["conv_launches"],
            "positions_per_launch": per_launch_positions}


TOLERANCE_PRECISION = "f16c6"   # the cheapest arithmetic that holds the north star's 1e-3 on trained nets: timed at length, with its own roofline


def precision_report(A, SP, W, args, primary_sims_per_s, moves=2):
    """Throughput of the precisions the main measurement did not run (same workload) and the measured output errors of all of them
    against the library's fp32 path at random-init and at trained logit scale (takzero_amd/precision.py).  The north star's
    tolerance (logits within 1e-3 of the fp32 path) is met by f16c6 / f16c8 (fp16 products + FP6 / FP8 correction products) and f16x2
    (hi / lo fp16 operands) at any scale, by f16 only while |logit| <~ 1.  The cheapest of those (f16c6) is timed over 10 moves after
    2 warm-up moves and gets a `roofline` object of its own (HIP events around every 8th launch of its net kernel, as for the
    headline); the others over `moves` moves after one."""
    from takzero_amd import precision as P

    names = ("f16", "f16c6", "f16c8", "f16x2")
    rates = {args.precision: primary_sims_per_s}
    extra = {}
    for other in names:
        if other in rates:
            continue
        long_run = other == TOLERANCE_PRECISION
        net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[other])
        net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
        mcts = A.BatchedMCTS(args.games, N_BOARD, HALF_KOMI, agent=net, node_capacity=args.capacity)
        sp = SP.NativeSelfPlay(mcts, args.sims, seed=0, shard=0, search=args.search, sampled_actions=64)
        for _ in range(2 if long_run else 1):
            sp.play_move()
        mcts.sync()
        mcts.profile(reset=1 if long_run else 2)
        s0, e0 = mcts.counters()
        t0 = time.perf_counter()
        n_moves = 10 if long_run else moves
        for _ in range(n_moves):
            sp.play_move()
        mcts.sync()
        dt = time.perf_counter() - t0
        s1, e1 = mcts.counters()
        rates[other] = (s1 - s0) / dt
        if long_run:
            prof = mcts.profile(reset=2)
            extra[other] = {"timed_moves": n_moves, "warmup_moves": 2, "ms_per_step": 1000.0 * dt / n_moves,
                            "nn_leaf_evals_per_s": (e1 - e0) / dt}
            if prof["conv_launches"]:
                extra[other]["roofline"] = roofline_of(other, prof, s1 - s0, e1 - e0, args.games)
                extra[other]["time_split_ms_per_sim"] = {"tree_kernels": prof["tree_ms"] / max(1, prof["steps"]),
                                                         "dominant_kernel": prof["conv_ms"] / max(1, prof["steps"]),
                                                         "wall": 1000.0 * dt / max(1.0, (s1 - s0) / args.games)}
        sp.close()
        mcts.close()
        net.close()
    states = P.sample_positions(N_BOARD, HALF_KOMI, 64, seed=7)
    w0 = W.init_weights(W.ARCH_NET5, seed=123)
    e0 = P.errors_against_f32(A.ARCH_NET5, w0, states, precisions=names)
    e1 = P.errors_against_f32(A.ARCH_NET5, P.trained_scale_weights(A.ARCH_NET5, states, seed=123), states, precisions=names)
    out = {"reference": e0["reference"], "positions": e0["positions"], "tolerance": "north star: logits within 1e-3 (absolute) of the fp32 path",
           "timed_moves_of_the_other_precisions": moves, "timed_moves_of_" + TOLERANCE_PRECISION: 10}
    for p in names:
        out[p] = {"sims_per_s": rates[p],
                  "random_init_scale": dict(e0[p], logit_scale=e0["logit_scale"]),
                  "trained_scale": dict(e1[p], logit_scale=e1["logit_scale"]),
                  "meets_1e-3_at_trained_scale": bool(e1[p]["max_abs_logit_err"] < 1e-3 and e1[p]["max_abs_value_err"] < 1e-3)}
        out[p].update(extra.get(p, {}))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=GAMES)
    ap.add_argument("--sims", type=int, default=None, help="simulations per move (default %d; 768 with --search gumbel, whose budget must be a multiple of k*log2 k)" % SIMS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--capacity", type=int, default=0)
    ap.add_argument("--search", choices=["puct", "gumbel"], default="puct",
                    help="puct = the north star's PUCT + Dirichlet loop (default); gumbel = what the reference's selfplay "
                         "binary runs today (sequential halving, 64 sampled actions; use --sims 768 for its budget)")
    ap.add_argument("--driver", choices=["native", "python"], default="native",
                    help="native = the self-play outer loop in csrc/tz_host.cpp (tz_selfplay_*); python = its mirror in "
                         "takzero_amd/selfplay.py")
    ap.add_argument("--precision", choices=["bf16", "f16", "f16c6", "f16c8", "f16x2"], default=os.environ.get("TZ_PRECISION", "f16"),
                    help="arithmetic of the MFMA path: f16 (default) = fp16 storage, fp32 accumulate, ~1e-3 relative logit error "
                         "through the 41 convs; f16c6 = the fp16 product plus FP6 (E2M3, block-scaled) correction products, 8 boards per "
                         "workgroup (1.4e-4 absolute at trained logit scale); f16c8 = the fp16 product plus FP8 (E4M3) correction products (1.4e-4 absolute at "
                         "trained logit scale, ~2.3x the f16 kernel time); f16x2 = split precision (hi/lo fp16 operands, 3 MFMAs per "
                         "product: 2.4e-5, ~3x); bf16 = the f16 kernels 5 %% faster at 1e-3 .. 7e-3 (random-init scale)")
    ap.add_argument("--no-precision-report", action="store_true",
                    help="skip the second measurement (N = 1 only): throughput of the other precision and the measured logit errors")
    args = ap.parse_args()
    if args.sims is None:
        args.sims = 768 if args.search == "gumbel" else SIMS

    if args.gpus > 1 and "RANK" not in os.environ:
        # started as plain `python bench.py --gpus N`: run the N ranks as a child torch.distributed.run job (nothing in
        # this process has touched the GPU yet) and hand its exit code back
        import subprocess

        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    dist = None
    backend = os.environ.get("TZ_BENCH_BACKEND", "nccl")   # "gloo" (+ TZ_BENCH_DEVICE=0): explicit rehearsal of N ranks without RCCL
    if "TZ_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["TZ_BENCH_DEVICE"])
    if world > 1 or os.environ.get("TZ_BENCH_FORCE_DIST"):   # TZ_BENCH_FORCE_DIST=1: rehearse the N > 1 set-up with one rank
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            try:
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                probe = torch.ones(1, device="cuda:%d" % local_rank)
                dist.all_reduce(probe)          # fail here, not in the timed region, if RCCL cannot form the ring
                torch.cuda.synchronize()
                assert int(probe.item()) == world, "all_reduce over %d ranks returned %r" % (world, probe.item())
            except Exception as e:              # no fallback: a SCALE record must show what actually ran
                sys.stderr.write("rank %d: backend nccl (RCCL) requested and unavailable: %r\n" % (rank, e))
                sys.stderr.flush()
                os._exit(3)
        else:
            dist.init_process_group(backend)
    import takzero_amd.api as A
    from takzero_amd import selfplay as SP
    from takzero_amd import weights as W

    net = A.Net(arch=A.ARCH_NET5, device=local_rank, precision=A.PREC_NAMES[args.precision])
    net.load_tensors(W.init_weights(W.ARCH_NET5, seed=123))
    mcts = A.BatchedMCTS(args.games, N_BOARD, HALF_KOMI, agent=net, node_capacity=args.capacity)
    if args.driver == "native":
        sp = SP.NativeSelfPlay(mcts, args.sims, seed=0, shard=rank, search=args.search, sampled_actions=64)
    else:
        sp = SP.SelfPlay(mcts, args.sims, seed=0, shard=rank, search=args.search, sampled_actions=64)
    dev = "cuda:%d" % local_rank if backend == "nccl" else "cpu"
    # the hand-over of the finished targets between the shards
    exchange = os.environ.get("TZ_BENCH_EXCHANGE", "native" if args.driver == "native" else "torch")
    comm, native_error = None, None
    if dist is not None and exchange == "native":
        from takzero_amd import comm as CM

        try:
            if backend == "nccl" or os.environ.get("TZ_BENCH_COMM") == "rccl":   # TZ_BENCH_COMM=rccl: the RCCL communicator in a gloo rehearsal
                ident = torch.zeros(CM.ID_BYTES, dtype=torch.uint8, device=dev)
                if rank == 0:
                    ident = torch.frombuffer(bytearray(CM.unique_id()), dtype=torch.uint8).to(dev)
                dist.broadcast(ident, 0)
                comm = CM.Comm.rccl(bytes(ident.cpu().numpy().tobytes()), rank, world, local_rank)
            else:                               # rehearsal without RCCL: the same packing over the shared-directory transport
                box = [None]
                if rank == 0:
                    import tempfile

                    box[0] = tempfile.mkdtemp(prefix="tz_bench_xch_")
                dist.broadcast_object_list(box, 0)
                comm = CM.Comm.fs(box[0], rank, world)
            comm.barrier()
            probe = comm.all_gather(b"rank %d" % rank)    # fail here, not in the timed region, if the ring does not carry data
            assert probe == [b"rank %d" % r for r in range(world)], probe
        except Exception as e:
            native_error = repr(e)
            sys.stderr.write("rank %d: the native exchange (tz_comm over %s) failed: %s\n" % (rank, backend, native_error))
            sys.stderr.flush()
        # every rank takes the same road: if the library's communicator did not come up on any one of them, all of them hand the
        # targets over through the process group that is already up (same RCCL ring, torch's calls), and the line says so
        ok = torch.tensor([0 if native_error else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            sp.set_comm(comm, writer_rank=-1)   # every rank ends up with all the lines, as `learn` on any rank would need
        else:
            if os.environ.get("TZ_BENCH_EXCHANGE") == "native":   # asked for by name: no other road
                os._exit(4)
            errors = [None] * world
            dist.all_gather_object(errors, native_error)
            native_error = "; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errors) if e)
            if comm is not None:
                comm.close()
            comm = None

    def barrier():
        mcts.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def one_move():
        """One outer-loop iteration + the exchange of the finished targets (every rank ends up with all of them)."""
        if args.driver == "native":
            sp.play_move()
            if comm is not None:
                sp.exchange()                # all-gather of counts + packed records over the library's communicator
            lines = sp.take_text(0)          # target lines finished this move, as learn reads them
            sp.take_text(1)
            if dist is not None and comm is None:
                lines = SP.all_gather_bytes(lines, dev)
            return lines.count(b"\n")
        t, _r = sp.play_move()
        return len(SP.all_gather_targets(t, N_BOARD, dev) if dist is not None else t)

    gathered = 0
    for _ in range(args.warmup):
        gathered += one_move()
    barrier()
    mcts.profile(reset=2 if args.no_profile else 1)
    sims0, evals0 = mcts.counters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered += one_move()
    barrier()
    dt = time.perf_counter() - t0
    sims1, evals1 = mcts.counters()
    prof = mcts.profile(reset=2)

    local = np.array([dt, sims1 - sims0, evals1 - evals0, args.games * args.steps], dtype=np.float64)
    if dist is not None:
        tl = torch.tensor(local, device=dev)
        tmax = tl.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tl, op=dist.ReduceOp.SUM)
        dt_max = float(tmax[0])
        sims, evals, positions = float(tl[1]), float(tl[2]), float(tl[3])
    else:
        dt_max, sims, evals, positions = dt, local[1], local[2], local[3]

    if rank == 0:
        out = {
            "metric": "mcts_simulations_per_s",
            "value": sims / dt_max,
            "unit": "sims/s",
            "n_gpus": world,
            "world": world,
            "backend": (backend if dist is not None else "none (single process)"),
            "exchange": ((dict(comm.info(), api="tz_comm (csrc/tz_comm.cpp)") if comm is not None
                          else dict({"transport": "torch.distributed " + backend},
                                    **({"native_exchange_error": native_error} if native_error else {}))) if dist is not None else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt_max / max(1, args.steps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (random-init net5 weights seed 123, random symmetric openings)",
            "config": {"workload": "5x5 Tak self-play, %d concurrent games/GPU, %d sims/move, %s, net5 (BASELINE configs[1])"
                                   % (args.games, args.sims,
                                      "PUCT+Dirichlet" if args.search == "puct" else "Gumbel sequential halving k=64"),
                       "games_per_gpu": args.games, "sims_per_move": args.sims, "parallelism": "shard%d" % world},
            "selfplay_positions_per_s": positions / dt_max,
            "nn_leaf_evals_per_s": evals / dt_max,
            "targets_gathered": gathered,
            "host_driver": "native (csrc/tz_host.cpp)" if args.driver == "native" else "python (takzero_amd/selfplay.py)",
        }
        if not args.no_profile and prof["conv_launches"]:
            out["roofline"] = roofline_of(args.precision, prof, sims1 - sims0, evals1 - evals0, args.games)
            out["time_split_ms_per_sim"] = {"tree_kernels": prof["tree_ms"] / max(1, prof["steps"]),
                                            "dominant_kernel": prof["conv_ms"] / max(1, prof["steps"]),
                                            "wall": 1000.0 * dt / max(1.0, (sims1 - sims0) / args.games)}
            out["net_flops_frac_of_peak"] = (evals / dt_max) * FLOP_PER_POSITION / (world * PEAK_BF16_TFLOPS * 1e12)
        if world == 1 and not args.no_precision_report and args.games >= 256:
            # the Agent surface on its own (tz_net_eval: host states and legal moves in, host logits / value / variance out), the drop-in point
            # for a host that keeps the reference's CPU BatchedMCTS: per call at the reference's batch of 128 (selfplay/src/main.rs:37) and at 1.
            # Not part of `value`.
            try:
                states = mcts.get_positions()
                ch, info = mcts.root_children(), mcts.root_info()
                acts = [ch["move_idx"][g, :info["n_children"][g]] for g in range(256)]
                out["agent_surface"] = {"what": "the C ABI call tz_net_eval (arrays packed once, as a C / Rust host holds them), PCIe and staging "
                                                "included; up to 256 positions run on four CUs per board group"}

                for b in (128, 1):
                    st_b = A._states(states[:b])
                    amax = max(1, max(len(a) for a in acts[:b]))
                    idx, cnt = np.zeros((b, amax), np.uint16), np.zeros(b, np.int32)
                    for i, a in enumerate(acts[:b]):
                        cnt[i] = len(a)
                        idx[i, :len(a)] = a
                    logits, value, var = np.zeros((b, amax), np.float32), np.zeros(b, np.float32), np.zeros(b, np.float32)

                    def call():
                        A.check(net.lib.tz_net_eval(net.h, b, st_b.ctypes.data, idx.ctypes.data, cnt.ctypes.data, amax, logits.ctypes.data,
                                                    value.ctypes.data, var.ctypes.data))

                    call()
                    t0 = time.perf_counter()
                    for _ in range(40):
                        call()
                    per = (time.perf_counter() - t0) / 40
                    out["agent_surface"]["batch_%d" % b] = {"ms_per_call": 1000.0 * per, "positions_per_s": b / per}
            except Exception as e:
                out["agent_surface"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the checker must not take the measurement down with it
                out["cpu_baseline"] = {"error": repr(e)}
        if world == 1 and not args.no_precision_report and args.games == GAMES and args.precision in ("f16", "f16c6", "f16c8", "f16x2"):
            # the other precision's throughput and both measured errors: close the first engine (66 GB of pools)
            sp.close()
            mcts.close()
            net.close()
            mcts = net = None
            try:
                out["precisions"] = precision_report(A, SP, W, args, out["value"])
                # the fastest precision whose measured errors hold the north star's 1e-3 at trained logit scale, named once at the top level
                ok = [(v["sims_per_s"], k) for k, v in out["precisions"].items() if isinstance(v, dict) and v.get("meets_1e-3_at_trained_scale")]
                if ok:
                    rate, name = max(ok)
                    out["within_1e-3_at_trained_scale"] = {"precision": name, "sims_per_s": rate,
                                                           "max_abs_logit_err": out["precisions"][name]["trained_scale"]["max_abs_logit_err"]}
            except Exception as e:
                out["precisions"] = {"error": repr(e)}
        print(json.dumps(out))
    if comm is not None:
        sp.close()
        comm.close()
    if mcts is not None:
        mcts.close()
        net.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
