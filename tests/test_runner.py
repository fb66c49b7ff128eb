"""Directory protocol of the selfplay / reanalyze binaries (buffer_lengths.txt back-pressure, model_latest.ot hot
reload, append-only target / replay files): host logic on CPU, the loops themselves on the GPU."""
import os

import numpy as np
import pytest


class FakeNet:
    def __init__(self):
        self.loaded = []

    def load(self, path):
        data = open(path, "rb").read()
        if data.startswith(b"bad"):
            raise ValueError("unparsable archive")
        self.loaded.append(data)


def test_back_pressure_and_model_watcher(tmp_path):
    from takzero_amd import runner as R

    d = str(tmp_path)
    net = FakeNet()
    w = R.ModelWatcher(net, d)
    # no buffer_lengths.txt: keeps retrying ("Could not read buffer lengths"), here until max_wait
    with pytest.raises(TimeoutError):
        R.wait_until_needed(d, 0, R.MAX_SELFPLAY_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    # wrong checksum is an error too (selfplay/src/main.rs:383-385)
    open(os.path.join(d, "buffer_lengths.txt"), "w").write("10,20,31")
    with pytest.raises(TimeoutError):
        R.wait_until_needed(d, 0, R.MAX_SELFPLAY_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    # over the cap: wait; the two binaries look at different components
    open(os.path.join(d, "buffer_lengths.txt"), "w").write("32001,5,32006")
    with pytest.raises(TimeoutError):
        R.wait_until_needed(d, 0, R.MAX_SELFPLAY_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    # reanalyze's component is under its cap but there is no model yet: retry until max_wait
    with pytest.raises(TimeoutError):
        R.wait_until_needed(d, 1, R.MAX_REANALYZE_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    open(os.path.join(d, "model_latest.ot"), "wb").write(b"model-1")
    R.wait_until_needed(d, 1, R.MAX_REANALYZE_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    assert net.loaded == [b"model-1"]
    R.wait_until_needed(d, 1, R.MAX_REANALYZE_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    assert net.loaded == [b"model-1"], "unchanged file is not re-read"
    open(os.path.join(d, "model_latest.ot"), "wb").write(b"model-22")
    R.wait_until_needed(d, 1, R.MAX_REANALYZE_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    assert net.loaded == [b"model-1", b"model-22"] and w.reloads == 2
    # exactly at the cap is still "needed" (`>` in the reference)
    open(os.path.join(d, "buffer_lengths.txt"), "w").write("32000,0,32000")
    R.wait_until_needed(d, 0, R.MAX_SELFPLAY_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    # an unreadable archive: selfplay keeps the old net and goes on, reanalyze retries
    open(os.path.join(d, "model_latest.ot"), "wb").write(b"bad archive")
    R.wait_until_needed(d, 0, R.MAX_SELFPLAY_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    with pytest.raises(ValueError):
        R.wait_until_needed(d, 1, R.MAX_REANALYZE_BUFFER_LEN, w, sleep=0.01, max_wait=0.05)
    assert len(net.loaded) == 2


def test_append_lines(tmp_path):
    from takzero_amd import runner as R

    p = str(tmp_path / "targets-selfplay.txt")
    R.append_lines(p, [])
    assert not os.path.exists(p)
    R.append_lines(p, ["a\n", "b\n"])
    R.append_lines(p, ["c\n"])
    assert open(p).read() == "a\nb\nc\n"


@pytest.mark.gpu
def test_hot_reload_replaces_the_weights_inside_captured_graphs():
    """Net::load between moves (selfplay/src/main.rs:107-110): after a reload the searches must run the NEW
    weights even though their simulation step was captured as a HIP graph with the old ones."""
    from gpu_util import require_gpu

    A = require_gpu()
    from takzero_amd import weights as W

    n, B = 4, 32
    wa = W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=1)
    wb = W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=2)
    net = A.Net(arch=A.ARCH_TEST, n=n, blocks=1).load_tensors(wa)
    mcts = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 12)
    mcts.new_openings(np.arange(B) % 16)
    start = mcts.get_positions()
    betas = np.zeros(B, np.float32)
    mcts.simulate(betas, 12)  # warm-up sims are eager, the rest replay the graph
    before = mcts.root_children()
    net.load_tensors(wb)      # hot reload
    mcts.set_positions(np.arange(B), start)
    mcts.simulate(betas, 12)
    after = mcts.root_children()
    fresh_net = A.Net(arch=A.ARCH_TEST, n=n, blocks=1).load_tensors(wb)
    fresh = A.BatchedMCTS(B, n, 4, agent=fresh_net, node_capacity=1 << 12)
    fresh.set_positions(np.arange(B), start)
    fresh.simulate(betas, 12)
    want = fresh.root_children()
    for k in want:
        assert np.array_equal(after[k], want[k]), k
    assert not np.array_equal(before["logit"], after["logit"])


@pytest.mark.gpu
def test_run_selfplay_and_reanalyze_on_a_directory(tmp_path):
    from gpu_util import require_gpu

    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import runner as R
    from takzero_amd import weights as W

    d, n, B = str(tmp_path), 4, 48
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(100, 200))
    net = A.Net(arch=A.ARCH_TEST, n=n, blocks=1).load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=7))
    mcts = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 13)
    sp = R.run_selfplay(d, mcts, 16, moves=50, seed=5, search="gumbel", sampled_actions=4, watch_model=False,
                        max_wait=5)
    assert sp.moves_played == 50
    targets = open(os.path.join(d, "targets-selfplay.txt")).read().splitlines(keepends=True)
    replays = open(os.path.join(d, "replays.txt")).read().splitlines(keepends=True)
    assert targets and replays
    for line in targets:
        st, mv, pol, value, ube = F.parse_target(line, n, 4)
        assert abs(float(pol.sum(dtype=np.float64)) - 1.0) < 1e-3 and F.format_target(n, st, mv, pol, value, ube) == line
    positions = sum(len(F.parse_replay(line, n, 4)[1]) for line in replays)
    ra = R.run_reanalyze(d, mcts, 16, iterations=2, seed=5, search="gumbel", sampled_actions=4,
                         min_positions=min(positions, 64), watch_model=False, max_wait=5)
    assert len(ra.buffer.positions) == positions
    lines = open(os.path.join(d, "targets-reanalyze.txt")).read().splitlines(keepends=True)
    assert len(lines) == 2 * B
    for line in lines:
        F.parse_target(line, n, 4)
    # back-pressure: learn says the selfplay buffer is full -> no move is played
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(R.MAX_SELFPLAY_BUFFER_LEN + 1, 0))
    with pytest.raises(TimeoutError):
        R.run_selfplay(d, mcts, 16, moves=1, watch_model=False, sleep=0.01, max_wait=0.1)


@pytest.mark.gpu
def test_exploration_feature_writes_truncated_replays_and_filters_targets(tmp_path):
    """cargo feature "exploration" (selfplay/src/main.rs:79-86, 279-290, 318-320): the first half of the games search
    with beta = 0.25; their openings (first 10 plies) also go to replays-exploration.txt and their positions only
    become targets after ply 10."""
    from gpu_util import require_gpu

    A = require_gpu()
    from takzero_amd import formats as F
    from takzero_amd import runner as R
    from takzero_amd import weights as W

    d, n, B = str(tmp_path), 4, 32
    open(os.path.join(d, "buffer_lengths.txt"), "w").write(F.format_buffer_lengths(0, 0))
    net = A.Net(arch=A.ARCH_TEST, n=n, blocks=1).load_tensors(W.init_weights(W.ARCH_TEST, n=n, blocks=1, seed=7))
    mcts = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 13)
    sp = R.run_selfplay(d, mcts, 16, moves=60, seed=9, search="gumbel", sampled_actions=4, watch_model=False,
                        exploration=True, max_wait=5)
    assert list(sp.betas[:B // 2]) == [0.25] * (B // 2) and not sp.betas[B // 2:].any()
    replays = [F.parse_replay(line, n, 4) for line in open(os.path.join(d, "replays.txt"))]
    expl = [F.parse_replay(line, n, 4) for line in open(os.path.join(d, "replays-exploration.txt"))]
    assert 0 < len(expl) < len(replays)
    full = {(st.tobytes(), tuple(int(m) for m in mv[:10])) for st, mv in replays}
    for st, mv in expl:
        assert len(mv) <= 10 and (st.tobytes(), tuple(int(m) for m in mv)) in full
    # exploitation games contribute every position, exploratory ones only those after ply 10: fewer targets than moves
    n_targets = sum(1 for _ in open(os.path.join(d, "targets-selfplay.txt")))
    assert n_targets < sum(len(mv) for _, mv in replays)


def test_async_appender_keeps_order_and_surfaces_errors(tmp_path):
    from takzero_amd import runner as R

    p = str(tmp_path / "out.txt")
    w = R.AsyncAppender()
    for i in range(50):
        w.submit(lambda i=i: R.append_lines(p, ["%d\n" % i]))
    w.close()
    assert open(p).read() == "".join("%d\n" % i for i in range(50))
    w = R.AsyncAppender()
    w.submit(lambda: R.append_lines(str(tmp_path / "no-such-dir" / "x.txt"), ["a\n"]))
    with pytest.raises(OSError):
        w.close()
