"""The native self-play driver (csrc/tz_host.cpp, tz_selfplay_*): the outer loop of selfplay::main below the ABI.
Its output is held to what the reference's own consumers require: every target line parses and lists exactly the legal
moves of its position, values are discounted game results, every replay re-validates move by move through the oracle
and ends in the terminal result it names; same seed -> same bytes."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import random_positions, require_gpu

pytestmark = pytest.mark.gpu


def _setup(A, n=4, B=48, seed=7):
    from takzero_amd import weights as W

    net = A.Net(arch=A.ARCH_TEST, n=n, precision=A.PREC_BF16, blocks=1).load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=seed))
    return net, A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 13)


def _check_lines(oracle, n, targets, replays, search, game_values=True):
    from takzero_amd import formats as F

    assert targets and (replays or replays is None)
    for line in targets.decode().splitlines(keepends=True):
        st, mv, pol, value, ube = F.parse_target(line, n, 4)
        legal = O.possible_moves(oracle, O.TzState.from_buffer_copy(np.array([st]).tobytes()))
        assert [int(m) for m in mv] == list(legal)          # exactly the legal moves, in possible_moves order
        total = float(pol.sum(dtype=np.float64))
        assert abs(total - 1.0) < (0.15 if search == "puct" else 1e-3)
        assert 0.0 <= ube <= 4.0
        if not game_values:                                 # reanalyze: the search's own root value (reanalyze/src/main.rs:214-224)
            assert -1.0 <= value <= 1.0
        elif value != 0.0:                                  # +-0.997^k, k >= 1 plies before the end
            k = math.log(abs(float(value))) / math.log(0.997)
            assert k > 0.5 and abs(k - round(k)) < 1e-2, value
        assert F.format_target(n, st, mv, pol, value, ube) == line
    moves_total = 0
    for line in (replays or b"").decode().splitlines():
        start, moves = F.parse_replay(line, n, 4)
        s = O.TzState.from_buffer_copy(np.array([start]).tobytes())
        for m in moves:
            assert oracle.tzo_terminal(C.byref(s)) == -1
            s = O.play(oracle, s, int(m))
        assert oracle.tzo_terminal(C.byref(s)) != -1
        tag = line.split()[-1]
        assert tag in {1: ("R-0", "F-0"), 2: ("0-R", "0-F"), 3: ("1/2-1/2",)}[oracle.tzo_result(C.byref(s))]
        moves_total += len(moves)
    return moves_total


@pytest.mark.parametrize("search,sims,k", [("puct", 12, 64), ("gumbel", 16, 4), ("random", 0, 64)])
def test_native_driver_output_is_what_learn_and_reanalyze_accept(oracle, search, sims, k):
    A = require_gpu()
    from takzero_amd import selfplay as SP

    n = 4
    outs = []
    for run in range(2):
        net, mcts = _setup(A, n)
        if search == "random":
            mcts = A.BatchedMCTS(48, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
        sp = SP.NativeSelfPlay(mcts, sims, seed=3, shard=0, search=search, sampled_actions=k)
        for _ in range(70):
            sp.play_move()
        outs.append((sp.take_text(0), sp.take_text(1), sp.counters()))
        assert sp.take_text(0) == b"" and sp.take_text(1) == b""   # taken means gone
    assert outs[0][:2] == outs[1][:2], "same seed, same shard: same bytes"
    targets, replays, counters = outs[0]
    assert counters["moves"] == 70 and counters["replays"] == replays.count(b"\n") and counters["targets"] == targets.count(b"\n")
    moves_total = _check_lines(oracle, n, targets, replays, search)
    assert counters["targets"] == moves_total   # beta = 0: every position of every finished game is a target
    net, mcts = _setup(A, n)
    other = SP.NativeSelfPlay(mcts if search != "random" else A.BatchedMCTS(48, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10),
                              sims, seed=3, shard=1, search=search, sampled_actions=k)
    for _ in range(70):
        other.play_move()
    assert other.take_text(1) != replays, "another shard draws another stream"


def test_native_directory_loop_with_exploration_and_back_pressure(oracle, tmp_path):
    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import runner as R

    d, n = str(tmp_path), 4
    net, mcts = _setup(A, n, B=32)
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(10, 20))
    sp = R.run_selfplay(d, mcts, 16, moves=60, seed=5, search="gumbel", sampled_actions=4, watch_model=False, exploration=True,
                        native=True, max_wait=5)
    assert sp.counters()["moves"] == 60
    targets = open(os.path.join(d, "targets-selfplay.txt"), "rb").read()
    replays = open(os.path.join(d, "replays.txt"), "rb").read()
    expl = open(os.path.join(d, "replays-exploration.txt")).read().splitlines()
    moves_total = _check_lines(oracle, n, targets, replays, "gumbel")
    assert 0 < len(expl) < replays.count(b"\n")
    full = {tuple(line.split()[:2 + 10 + 3]) for line in replays.decode().splitlines()}
    for line in expl:
        toks = line.split()
        assert len(toks) <= 5 + 10      # '[TPS', three TPS fields ... plus at most ten moves, no result
    assert targets.count(b"\n") < moves_total   # exploratory games only give targets after ply 10
    # learn says its self-play buffer is full: nothing is played
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(R.MAX_SELFPLAY_BUFFER_LEN + 1, 0))
    before = os.path.getsize(os.path.join(d, "replays.txt"))
    with pytest.raises(TimeoutError):
        R.run_selfplay(d, mcts, 16, moves=1, watch_model=False, native=True, max_wait=0.1)
    assert os.path.getsize(os.path.join(d, "replays.txt")) == before


def test_native_reanalyze_feeds_samples_and_writes_targets(oracle, tmp_path):
    """tz_reanalyze_*: the native position buffer holds exactly the positions the Python one derives from the same
    replay file (deterministic, no draws involved); an iteration gives one valid target line per sampled position;
    ranks split the replay lines; the directory loop appends targets-reanalyze.txt."""
    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import reanalyze as RA
    from takzero_amd import runner as R
    from takzero_amd import selfplay as SP

    d, n = str(tmp_path), 4
    net, mcts = _setup(A, n, B=32)
    sp = SP.NativeSelfPlay(mcts, 12, seed=1, search="puct")
    for _ in range(60):
        sp.play_move()
    rpath = os.path.join(d, "replays.txt")
    replays = sp.take_text(1)
    open(rpath, "wb").write(replays + b"[TPS \"x4/x4/x4/x4 1 1\"] a1 a1\nnot a replay\n[TPS \"x4/x4/x4/x4 1 1\"] a1 b")  # illegal, junk, half
    ref = RA.PositionBuffer(mcts, n, 4)
    want = ref.read_new(rpath)
    nat = RA.NativeReanalyze(mcts, 16, seed=2, search="gumbel", sampled_actions=4)
    assert nat.feed(rpath) == want == nat.positions and nat.feed(rpath) == 0
    with open(rpath, "ab") as f:
        f.write(b"1\n")                        # the half-written line is completed ("a1 b1"): one more replay, two positions
    assert nat.feed(rpath) == ref.read_new(rpath) == 2
    for search, sims in (("gumbel", 16), ("puct", 24)):
        nat2 = RA.NativeReanalyze(mcts, sims, seed=3, search=search, sampled_actions=4)
        nat2.feed(rpath)
        nat2.iterate()
        lines = nat2.take_text().decode().splitlines(keepends=True)
        assert len(lines) == mcts.batch and nat2.take_text() == b""
        for line in lines:
            st, mv, pol, value, ube = F.parse_target(line, n, 4)
            legal = O.possible_moves(oracle, O.TzState.from_buffer_copy(np.array([st]).tobytes()))
            assert [int(m) for m in mv] == list(legal)
            assert abs(float(pol.sum(dtype=np.float64)) - 1.0) < 1e-3 and -1.0 <= value <= 1.0 and 0.0 <= ube <= 4.0
    a = RA.NativeReanalyze(mcts, 16, seed=2, rank=0, world=2)
    b = RA.NativeReanalyze(mcts, 16, seed=2, rank=1, world=2)
    assert a.feed(rpath) + b.feed(rpath) == nat.positions and a.positions > 0 and b.positions > 0
    # the directory loop
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
    nra = R.run_reanalyze(d, mcts, 16, iterations=2, seed=4, search="gumbel", sampled_actions=4, min_positions=64,
                          watch_model=False, native=True, max_wait=5)
    out = open(os.path.join(d, "targets-reanalyze.txt")).read().splitlines()
    assert len(out) == 2 * mcts.batch
    assert nra.feed(rpath) == 0   # the loop has consumed the whole replay file
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, R.MAX_REANALYZE_BUFFER_LEN + 1))
    with pytest.raises(TimeoutError):
        R.run_reanalyze(d, mcts, 16, iterations=1, watch_model=False, native=True, min_positions=64, max_wait=0.1)


def _build_example(tmp_path, name):
    import subprocess

    from takzero_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    r = subprocess.run(["g++", "-std=c++17", "-O2", os.path.join(root, "examples", name + ".cpp"), "-I" + os.path.join(root, "include"),
                        "-L" + os.path.dirname(_lib.LIB_PATH), "-ltakzero_hip", "-Wl,-rpath," + os.path.dirname(_lib.LIB_PATH), "-o", exe],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the example here: " + r.stderr[-300:])
    return exe


def _fields(stdout):
    return dict(zip(stdout.split()[::2], stdout.split()[1::2]))


@pytest.mark.parametrize("reload_flags", [[], ["--async-reload"]], ids=["sync", "async"])
def test_cpp_program_over_the_c_abi_alone(oracle, tmp_path, reload_flags):
    """examples/selfplay_cli.cpp: the selfplay binary as a plain C++ program linked against libtakzero_hip.so — no
    Python, no torch in the process.  It must produce the same kind of files, and pick up a new model_latest.ot (the
    LibTorch archive the reference's learn writes: read by the library itself) — at once, or with --async-reload prepared on
    another thread while the games go on and swapped in one move later (tz_net_load_prepare / tz_net_load_commit)."""
    import subprocess

    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import ot
    from takzero_amd import weights as W

    exe = _build_example(tmp_path, "selfplay_cli")
    d, n = str(tmp_path), 4
    W.save_tzw(os.path.join(d, "start.tzw"), W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=7))
    ot.save_ot(os.path.join(d, "model_latest.ot"), W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=8))
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
    r = subprocess.run([exe, "--directory", d, "--model", os.path.join(d, "start.tzw"), "--arch", "100", "--n", str(n), "--blocks", "1",
                        "--games", "48", "--sims", "16", "--sampled-actions", "4", "--search", "gumbel", "--moves", "60", "--wait-limit", "5"] + reload_flags,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr[-1500:])
    fields = _fields(r.stdout)
    assert fields["moves"] == "60" and fields["model_reloads"] == "1" and int(fields["replays"]) > 0
    targets = open(os.path.join(d, "targets-selfplay.txt"), "rb").read()
    replays = open(os.path.join(d, "replays.txt"), "rb").read()
    assert targets.count(b"\n") == int(fields["targets"]) and replays.count(b"\n") == int(fields["replays"])
    _check_lines(oracle, n, targets, replays, "gumbel")


def test_closed_loop_of_the_three_cpp_programs_over_ot_files(oracle, tmp_path):
    """learn_cli (tz_learn_run with its own save points) writes model_0000000.ot, model_<pre>.ot and
    model_latest.ot as LibTorch archives; selfplay_cli (started from Net::new) and reanalyze_cli reload model_latest.ot while they
    run, selfplay feeds learn through targets-selfplay.txt and reanalyze through replays.txt, reanalyze feeds learn through
    targets-reanalyze.txt (learn switches to half-and-half batches at --steps-before-reanalyze) - the reference's deployment
    (README.md:130) as three C++ processes on one directory and one GPU, no Python, no torch, no LibTorch in any of them.  The
    archives are then read back by LibTorch itself (torch.jit.load)."""
    import subprocess
    import time

    require_gpu()
    from takzero_amd import ot

    learn, selfplay, reanalyze = (_build_example(tmp_path, name) for name in ("learn_cli", "selfplay_cli", "reanalyze_cli"))
    d, n = str(tmp_path / "run"), 4
    os.makedirs(d)
    common = ["--arch", "100", "--n", str(n), "--blocks", "1"]
    lp = subprocess.Popen([learn, "--directory", d, "--batch", "64", "--steps", "90", "--seed", "5", "--pre-training-steps", "20",
                           "--initial-targets", "2000", "--min-selfplay", "300", "--min-reanalyze", "128", "--steps-per-save", "30",
                           "--steps-per-checkpoint", "60", "--steps-before-reanalyze", "60", "--read-interval", "0.2", "--sleep", "0.2",
                           "--wait-limit", "240"] + common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not os.path.exists(os.path.join(d, "buffer_lengths.txt")) and time.time() - t0 < 120 and lp.poll() is None:
        time.sleep(0.1)
    assert os.path.exists(os.path.join(d, "buffer_lengths.txt")), lp.communicate()[1][-1500:]
    rp = subprocess.Popen([reanalyze, "--directory", d, "--games", "64", "--sims", "16", "--sampled-actions", "4", "--search", "gumbel",
                           "--iterations", "12", "--min-positions", "500", "--wait-limit", "120", "--seed", "4", "--async-reload"] + common,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    sp = subprocess.run([selfplay, "--directory", d, "--games", "64", "--sims", "16", "--sampled-actions", "4", "--search", "gumbel",
                         "--moves", "150", "--wait-limit", "60", "--seed", "9"] + common, capture_output=True, text=True, timeout=600)
    rout, rerr = rp.communicate(timeout=600)
    lout, lerr = lp.communicate(timeout=600)
    assert sp.returncode == 0, (sp.stdout, sp.stderr[-1500:])
    assert rp.returncode == 0, (rout, rerr[-1500:])
    assert lp.returncode == 0, (lout, lerr[-1500:])
    lf, sf, rf = _fields(lout), _fields(sp.stdout), _fields(rout)
    assert lf["model_steps"] == "110" and lf["starting_steps"] == "20" and lf["rc"] == "0"
    assert int(sf["model_reloads"]) >= 2 and sf["moves"] == "150"     # the initial model_latest.ot and at least one save point
    assert rf["rc"] == "0" and int(rf["model_reloads"]) >= 1 and int(rf["simulations"]) >= 12 * 64 * 16
    names = sorted(f for f in os.listdir(d) if f.endswith(".ot"))
    assert names == ["model_0000000.ot", "model_0000020.ot", "model_0000060.ot", "model_latest.ot"], names
    assert not [f for f in os.listdir(d) if f.endswith(".part")]
    first, last = ot.load_ot_libtorch(os.path.join(d, "model_0000000.ot")), ot.load_ot_libtorch(os.path.join(d, "model_latest.ot"))
    assert set(first) == set(last) == set(ot.load_ot(os.path.join(d, "model_latest.ot")))
    assert not np.array_equal(first["core.res_block_0.b.conv2d.weight"], last["core.res_block_0.b.conv2d.weight"])   # it trained
    assert not np.array_equal(first["core.batch_norm.running_mean"], last["core.batch_norm.running_mean"])
    targets = open(os.path.join(d, "targets-selfplay.txt"), "rb").read()
    replays = open(os.path.join(d, "replays.txt"), "rb").read()
    _check_lines(oracle, n, targets, replays, "gumbel")
    re_targets = open(os.path.join(d, "targets-reanalyze.txt"), "rb").read()
    assert re_targets.count(b"\n") == 12 * 64
    _check_lines(oracle, n, re_targets, None, "gumbel", game_values=False)


def test_reanalyze_program_retries_a_torn_model_in_both_reload_modes(oracle, tmp_path):
    """reanalyze retries every failed load on its next iteration (reanalyze/src/main.rs:93-105).  model_latest.ot is torn (the
    reference's learn does not write it atomically) for the whole run: with and without --async-reload the program keeps the net
    it has, says so at every iteration rather than once, and finishes its iterations."""
    import subprocess

    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import ot
    from takzero_amd import selfplay as SP
    from takzero_amd import weights as W

    exe = _build_example(tmp_path, "reanalyze_cli")
    n = 4
    mcts = A.BatchedMCTS(48, n, 4, agent_kind=A.AGENT_DUMMY, node_capacity=1 << 10)
    sp = SP.NativeSelfPlay(mcts, 0, seed=3, shard=0, search="random", sampled_actions=64)
    for _ in range(70):
        sp.play_move()
    replays = sp.take_text(1)
    sp.close()
    mcts.close()
    for mode in ([], ["--async-reload"]):
        d = str(tmp_path / ("run" + str(len(mode))))
        os.makedirs(d)
        open(os.path.join(d, "replays.txt"), "wb").write(replays)
        open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
        W.save_tzw(os.path.join(d, "start.tzw"), W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=7))
        ot.save_ot(os.path.join(d, "whole.ot"), W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=8))
        whole = open(os.path.join(d, "whole.ot"), "rb").read()
        open(os.path.join(d, "model_latest.ot"), "wb").write(whole[:len(whole) // 2])
        r = subprocess.run([exe, "--directory", d, "--model", os.path.join(d, "start.tzw"), "--arch", "100", "--n", str(n), "--blocks", "1",
                            "--games", "32", "--sims", "16", "--sampled-actions", "4", "--search", "gumbel", "--iterations", "6",
                            "--min-positions", "100", "--wait-limit", "20", "--seed", "4"] + mode, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout, r.stderr[-1500:])
        f = _fields(r.stdout)
        assert f["rc"] == "0" and f["model_reloads"] == "0", r.stdout
        assert r.stderr.count("Cannot load model") >= 2, r.stderr[-1500:]
        assert open(os.path.join(d, "targets-reanalyze.txt"), "rb").read().count(b"\n") == 6 * 32


def test_evaluation_cpp_program_matches_models_up(oracle, tmp_path):
    """examples/evaluation_cli.cpp = the reference's `evaluation` binary (evaluation/src/main.rs:131-222) over the C ABI: picks
    two of the directory's model_<steps>.ot files (never model_latest.ot), loads them with load_partial semantics, starts the
    games from an opening book or from openings with two or three random moves, and plays tz_compete both ways round (tz_compete
    itself is checked against the oracle in test_puzzle.py / test_gpu_drivers.py)."""
    import re
    import subprocess

    require_gpu()
    from takzero_amd import ot
    from takzero_amd import weights as W

    exe = _build_example(tmp_path, "evaluation_cli")
    d, n = str(tmp_path / "models"), 4
    os.makedirs(d)
    for steps, seed in ((0, 1), (100, 2), (200, 3)):
        ot.save_ot(os.path.join(d, "model_%07d.ot" % steps), W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=seed))
    ot.save_ot(os.path.join(d, "model_latest.ot"), W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=3))
    common = [exe, "--model-path", d, "--arch", "100", "--n", str(n), "--blocks", "1", "--games", "16", "--sampled-actions", "4",
              "--budget", "16", "--max-moves", "80", "--rounds", "3", "--seed", "5"]
    book = tmp_path / "book.tps"
    states = random_positions(oracle, O, n, 4, 24, 11, max_ply=6)
    book.write_text("".join(O.to_tps(oracle, s) + "\n" for s in states))
    for extra in ([], ["--opening-book", str(book)]):
        r = subprocess.run(common + extra, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout, r.stderr[-1500:])
        lines = [ln for ln in r.stdout.splitlines() if " vs. " in ln]
        assert len(lines) == 6, r.stdout                      # three match-ups, each played both ways round
        for i in range(0, 6, 2):
            m1 = re.fullmatch(r"(\S+) vs\. (\S+): Evaluation \{ wins: (\d+), losses: (\d+), draws: (\d+) \} ([0-9.]+)%", lines[i])
            m2 = re.fullmatch(r"(\S+) vs\. (\S+): Evaluation \{ wins: (\d+), losses: (\d+), draws: (\d+) \} ([0-9.]+)%", lines[i + 1])
            assert m1 and m2, lines[i:i + 2]
            assert m1.group(1) == m2.group(2) and m1.group(2) == m2.group(1) and m1.group(1) != m1.group(2)
            for m in (m1, m2):
                assert m.group(1) in ("model_0000000.ot", "model_0000100.ot", "model_0000200.ot") and "latest" not in m.group(2)
                w, l, dr = int(m.group(3)), int(m.group(4)), int(m.group(5))
                assert 0 < w + l + dr <= 16
                assert abs(float(m.group(6)) - 100.0 * w / (w + l + dr)) < 0.06
    # too few models: the reference sleeps and looks again; a bounded run with --sleep 0 says so and stops
    os.remove(os.path.join(d, "model_0000100.ot"))
    os.remove(os.path.join(d, "model_0000200.ot"))
    r = subprocess.run(common + ["--sleep", "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Too few models" in r.stderr


def test_two_shards_of_the_cpp_program_hand_over_to_rank_0(oracle, tmp_path):
    """N shards natively (VERDICT r1 #2b): two selfplay_cli processes (both on this box's one GPU, "fs" transport — on a node
    it is one process per GPU over RCCL, same code above the transport) exchange after every move; rank 0 appends
    everybody's targets and replays to the un-suffixed files `learn` and `reanalyze` read, rank 1 writes nothing."""
    import subprocess

    require_gpu()
    from takzero_amd import formats as F

    exe = _build_example(tmp_path, "selfplay_cli")
    d, n = str(tmp_path / "run"), 4
    os.makedirs(d)
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
    os.makedirs(tmp_path / "xch")
    procs = [subprocess.Popen([exe, "--directory", d, "--arch", "100", "--n", str(n), "--blocks", "1", "--games", "32", "--sims", "16",
                               "--sampled-actions", "4", "--search", "gumbel", "--moves", "70", "--wait-limit", "30", "--seed", "3",
                               "--rank", str(r), "--world", "2", "--comm", "fs", "--comm-dir", str(tmp_path / "xch"), "--device", "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1200:] for o in outs]
    f0, f1 = _fields(outs[0][0]), _fields(outs[1][0])
    targets = open(os.path.join(d, "targets-selfplay.txt"), "rb").read()
    replays = open(os.path.join(d, "replays.txt"), "rb").read()
    assert int(f0["targets"]) > 0 and int(f1["targets"]) > 0 and f0["targets"] != f1["targets"]   # two different shards
    assert targets.count(b"\n") == int(f0["targets"]) + int(f1["targets"])
    assert replays.count(b"\n") == int(f0["replays"]) + int(f1["replays"])
    assert sorted(os.listdir(d)) == ["buffer_lengths.txt", "replays.txt", "targets-selfplay.txt"]
    _check_lines(oracle, n, targets, replays, "gumbel")


def test_two_shards_through_the_rccl_side_of_the_communicator(oracle, tmp_path):
    """The RCCL side of csrc/tz_comm.cpp with two ranks on this box's one GPU.  RCCL itself refuses two ranks on one device, so
    the library is pointed (TZ_RCCL_LIB) at a stand-in with RCCL's entry points and semantics on device buffers
    (tests/mock_rccl.cpp: bytes move through files): everything above it is the code an N-GPU job runs - the unique-id rendezvous,
    ncclCommInitRank, the staging buffers on the device, the stream order, all-gather of counts then of padded records at world 2,
    the broadcast of a reloaded model from rank 0 - and the result must be what the fs transport gives: rank 0 holds everybody's
    lines, the two shards play different games, rank 1 writes nothing, and the model rank 0 picked up reached rank 1."""
    import subprocess

    require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import ot
    from takzero_amd import weights as W

    mock = str(tmp_path / "libmockrccl.so")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-w", os.path.join(root, "tests", "mock_rccl.cpp"), "-o", mock],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("cannot build the stand-in library here: " + r.stderr[-300:])
    exe = _build_example(tmp_path, "selfplay_cli")
    d, n = str(tmp_path / "run"), 4
    os.makedirs(d)
    os.makedirs(tmp_path / "xch")
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
    # a model for rank 0 to find at its first look: it has to reach rank 1 through the broadcast
    ot.save_ot(os.path.join(d, "model_latest.ot"), W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=8))
    env = dict(os.environ, TZ_RCCL_LIB=mock)
    procs = [subprocess.Popen([exe, "--directory", d, "--arch", "100", "--n", str(n), "--blocks", "1", "--games", "32", "--sims", "16",
                               "--sampled-actions", "4", "--search", "gumbel", "--moves", "70", "--wait-limit", "30", "--seed", "3",
                               "--rank", str(rk), "--world", "2", "--comm", "rccl", "--comm-dir", str(tmp_path / "xch"), "--device", "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for rk in (0, 1)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1200:] for o in outs]
    f0, f1 = _fields(outs[0][0]), _fields(outs[1][0])
    assert f0["model_reloads"] == "1"                    # rank 0 loaded the archive; rank 1 got the variables over the communicator
    targets = open(os.path.join(d, "targets-selfplay.txt"), "rb").read()
    replays = open(os.path.join(d, "replays.txt"), "rb").read()
    assert int(f0["targets"]) > 0 and int(f1["targets"]) > 0 and f0["targets"] != f1["targets"]
    assert targets.count(b"\n") == int(f0["targets"]) + int(f1["targets"])
    assert replays.count(b"\n") == int(f0["replays"]) + int(f1["replays"])
    assert sorted(os.listdir(d)) == ["buffer_lengths.txt", "model_latest.ot", "replays.txt", "targets-selfplay.txt"]
    _check_lines(oracle, n, targets, replays, "gumbel")
    # the same job over the fs transport writes the same bytes (same seeds, same model): the transports are interchangeable
    d2 = str(tmp_path / "run_fs")
    os.makedirs(d2)
    os.makedirs(tmp_path / "xch_fs")
    open(os.path.join(d2, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
    ot.save_ot(os.path.join(d2, "model_latest.ot"), W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=8))
    procs = [subprocess.Popen([exe, "--directory", d2, "--arch", "100", "--n", str(n), "--blocks", "1", "--games", "32", "--sims", "16",
                               "--sampled-actions", "4", "--search", "gumbel", "--moves", "70", "--wait-limit", "30", "--seed", "3",
                               "--rank", str(rk), "--world", "2", "--comm", "fs", "--comm-dir", str(tmp_path / "xch_fs"), "--device", "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for rk in (0, 1)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1200:] for o in outs]
    assert open(os.path.join(d2, "targets-selfplay.txt"), "rb").read() == targets
    assert open(os.path.join(d2, "replays.txt"), "rb").read() == replays


def test_a_shard_that_has_to_stop_takes_the_others_with_it(tmp_path):
    """tz_selfplay_run with N shards: before every move the ranks exchange one status word, so a rank whose own look at the
    directory says stop (here: learn's buffer over its cap for longer than --wait-limit, seen by rank 1 only) does not walk out of
    the move's all-gather and leave rank 0 waiting in it for good: both processes end, each with an error that says who stopped."""
    import subprocess

    require_gpu()
    from takzero_amd import formats as F

    exe = _build_example(tmp_path, "selfplay_cli")
    dirs = [str(tmp_path / "run0"), str(tmp_path / "run1")]
    os.makedirs(tmp_path / "xch")
    for d, lengths in zip(dirs, ((0, 0), (40000, 0))):
        os.makedirs(d)
        open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(*lengths))
    procs = [subprocess.Popen([exe, "--directory", dirs[rk], "--arch", "100", "--n", "4", "--blocks", "1", "--games", "32", "--sims", "16",
                               "--sampled-actions", "4", "--search", "gumbel", "--moves", "400", "--wait-limit", "2", "--seed", "3",
                               "--rank", str(rk), "--world", "2", "--comm", "fs", "--comm-dir", str(tmp_path / "xch"), "--device", "0"],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for rk in (0, 1)]
    outs = [p.communicate(timeout=300) for p in procs]          # a stranded rank would sit here until the timeout
    assert all(p.returncode != 0 for p in procs), [o[0][-300:] for o in outs]
    assert "rank 1 stopped before the move" in outs[0][1], outs[0][1][-800:]
    assert "buffer stayed over its cap" in outs[1][1], outs[1][1][-800:]


@pytest.mark.parametrize("kind,sims,k,exploration,agent,n", [(0, 24, 64, 1, 2, 4), (1, 16, 4, 0, 2, 4), (1, 48, 8, 1, 1, 5), (2, 0, 64, 0, 1, 4)])
def test_native_drivers_write_the_same_bytes_over_the_gpu_engine_and_over_the_oracle(tmp_path, kind, sims, k, exploration, agent, n):
    """The strongest statement about the whole loop: csrc/tz_host.cpp driving the HIP engine and the very same driver
    code driving the CPU oracle (tests/host_over_oracle.cpp), same seed — every target line, every replay line, every
    reanalyze target must be identical bytes.  That holds only if every search result the driver ever looked at
    (visit counts, evaluations, policies, UBE targets, terminal results) was bit-identical, move after move."""
    from host_oracle_util import build, run

    A = require_gpu()
    from takzero_amd import reanalyze as RA
    from takzero_amd import selfplay as SP

    B, moves, seed = 24, 50, 9
    want = run(build(tmp_path, sanitize=False), tmp_path / "cpu", n, 4, agent, B, kind, sims, k, exploration, moves, seed)
    mcts = A.BatchedMCTS(B, n, 4, agent_kind=agent, node_capacity=1 << 15)
    sp = SP.NativeSelfPlay(mcts, sims, seed=seed, shard=0, search={0: "puct", 1: "gumbel", 2: "random"}[kind], sampled_actions=k,
                           exploration=bool(exploration))
    got = {"targets": b"", "replays": b"", "exploration": b""}
    for _ in range(moves):
        sp.play_move()
        got["targets"] += sp.take_text(0)
        got["replays"] += sp.take_text(1)
        got["exploration"] += sp.take_text(2)
    for part in ("replays", "exploration", "targets"):
        assert got[part] == want[part], part
    rpath = tmp_path / "gpu.replays"
    rpath.write_bytes(got["replays"])
    ra = RA.NativeReanalyze(mcts, sims if kind == 1 else 32, seed=seed + 1, search="gumbel" if kind == 1 else "puct", sampled_actions=k)
    assert ra.feed(rpath) == want["positions"]
    if want["positions"] >= B:
        ra.iterate()
        ra.iterate()
    assert ra.take_text() == want["reanalyze"]
    # evaluation::compete and the puzzle benchmark, natively, on the 16 openings: same counts, same final positions
    from takzero_amd import evaluation as E
    from takzero_amd import puzzle as P

    other = A.BatchedMCTS(B, n, 4, agent_kind=2 if agent == 1 else 1, node_capacity=1 << 15)
    mcts.new_openings(np.arange(B) % 16)
    games = mcts.get_positions()
    kk = k if k >= 2 else 4
    budget = k * (k.bit_length() - 1) * 2 if k >= 2 else 16
    ev = E.compete_native(mcts, other, games, 0.0, 0.25, seed=seed + 2, sampled_actions=kk, search_budget=budget, max_moves=6)
    puzzles = np.concatenate([games, games[:B // 2]])
    solutions = np.zeros(len(puzzles), np.uint16)
    pw = P.benchmark_native(mcts, puzzles, solutions, True, kk, budget, seed=seed + 3)
    pa = P.benchmark_native(mcts, puzzles, solutions, False, kk, budget, seed=seed + 3)
    lines = ["%d %d %d %d %d %d %d %d %d" % (ev.wins, ev.losses, ev.draws, pw.attempted, pw.solved, pw.proven, pa.attempted, pa.solved,
                                              pa.proven)]
    lines += [A.state_to_tps(st) for st in other.get_positions()]
    assert ("\n".join(lines) + "\n").encode() == want["consumers"]


@pytest.mark.parametrize("kind,sims,k,prec", [(0, 24, 64, 2), (1, 16, 4, 0), (1, 16, 4, 3), (0, 24, 64, 4)])   # fp16, bf16, split precision, fp16 + FP8 corrections
def test_whole_loop_bytes_with_the_real_network(tmp_path, kind, sims, k, prec):
    """The same byte-for-byte statement with the network in the loop: the HIP engine (fused trunk kernel, leaf batches
    compacted on the device) against the oracle search whose Agent is that HIP network called through tz_net_eval."""
    from host_oracle_util import build, run

    A = require_gpu()
    from takzero_amd import selfplay as SP
    from takzero_amd import weights as W

    n, blocks, B, moves, seed = 4, 1, 16, 40, 3
    w = W.init_weights(W.ARCH_TEST, n=n, blocks=blocks, seed=17)
    model = tmp_path / "model.tzw"
    W.save_tzw(model, w)
    want = run(build(tmp_path, sanitize=False, with_net=True), tmp_path / "cpu", n, 4, 0, B, kind, sims, k, 1, moves, seed,
               net_args=(model, A.ARCH_TEST, blocks, prec))
    net = A.Net(arch=A.ARCH_TEST, n=n, precision=prec, blocks=blocks).load_tensors(w)
    mcts = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 14)
    sp = SP.NativeSelfPlay(mcts, sims, seed=seed, shard=0, search="puct" if kind == 0 else "gumbel", sampled_actions=k, exploration=True)
    got = {"targets": b"", "replays": b"", "exploration": b""}
    for _ in range(moves):
        sp.play_move()
        got["targets"] += sp.take_text(0)
        got["replays"] += sp.take_text(1)
        got["exploration"] += sp.take_text(2)
    assert got["replays"] and got["targets"]
    for part in ("replays", "exploration", "targets"):
        assert got[part] == want[part], part
