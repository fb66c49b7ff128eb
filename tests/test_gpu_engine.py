"""End-to-end parity: the HIP engine (tree kernels + MFMA net on the device, no host round trip
inside a simulation) against the CPU oracle search whose agent is the *same* HIP network called
through tz_net_eval.  With identical priors on both sides visit counts and chosen moves must be
equal (north_star: 'visit counts and chosen moves matching ... under a fixed RNG seed')."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from gpu_util import require_gpu
from test_gpu_tree import assert_same_roots

pytestmark = pytest.mark.gpu


def _agent_over(net):
    def fn(user, n_envs, states, legal_idx, legal_count, amax, logits_out, value_out, variance_out):
        rc = net.lib.tz_net_eval(net.h, n_envs, C.cast(states, C.c_void_p), C.cast(legal_idx, C.c_void_p),
                                 C.cast(legal_count, C.c_void_p), amax, C.cast(logits_out, C.c_void_p),
                                 C.cast(value_out, C.c_void_p), C.cast(variance_out, C.c_void_p))
        assert rc == 0, net.lib.tz_last_error()
    return fn


# (100,...) small nets; (5,...) full net5 = config 2 at test size; (4,...) config 1: 4x4, 64 games, 100 sims/move,
# net4_simhash; (6,...) config 4 at test size: 6x6, net6_simhash
@pytest.mark.parametrize("arch,n,blocks,prec,B,sims,moves", [(100, 5, 2, 0, 24, 30, 5), (100, 4, 1, 1, 16, 25, 4),
                                                             (5, 5, 20, 0, 12, 24, 3), (5, 5, 20, 2, 12, 24, 2), (4, 4, 16, 0, 64, 100, 2),
                                                             (6, 6, 16, 0, 8, 40, 2)])
def test_engine_matches_oracle_search_with_same_net(oracle, arch, n, blocks, prec, B, sims, moves):
    A = require_gpu()
    from takzero_amd import weights as W

    net = A.Net(arch=arch, n=n, precision=prec, blocks=blocks)
    net.load_tensors(W.init_weights(arch, n=n, blocks=blocks, seed=123))
    gpu = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 14)
    ora = O.OracleSearch(oracle, B, n, 4, agent_kind=0, agent_fn=_agent_over(net))
    rng = np.random.default_rng(2024)
    choice = rng.integers(0, 16, B)
    gpu.new_openings(choice)
    ora.new_openings(choice)
    betas = np.where(np.arange(B) % 2 == 0, 0.25, 0.0).astype(np.float32)
    for mv in range(moves):
        gpu.simulate(betas, 1)
        ora.simulate(betas, 1)
        info = ora.root_info()
        amax = max(1, int(info["n_children"].max()))
        noise = np.zeros((B, amax), np.float32)
        for g in range(B):
            noise[g, :info["n_children"][g]] = rng.dirichlet([0.05] * int(info["n_children"][g])).astype(np.float32)
        gpu.apply_noise(noise, 0.2)
        ora.apply_noise(noise, 0.2)
        gpu.simulate(betas, sims)
        ora.simulate(betas, sims)
        assert_same_roots(gpu, ora, "move %d" % mv)
        acts = gpu.select_best_actions()
        assert np.array_equal(acts, ora.select_best_actions())
        gpu.step(acts)
        ora.step(acts)
        choice = rng.integers(0, 16, B)
        assert np.array_equal(gpu.restart_terminal_envs(choice), ora.restart_terminal(choice))
    assert gpu.counters() == ora.counters()


def test_gumbel_search_at_trained_scale_agrees_with_the_fp32_path():
    """The reference's current self-play search - Gumbel sequential halving, 64 sampled actions, budget 768 - reads the logits
    directly (gumbel + logit picks the candidates, then sigma(q) + logit ranks them), so it is the search most exposed to logit
    error.  Same positions, same Gumbel noise, weights at a trained net's logit scale: the action chosen under each arithmetic against
    the one chosen under the fp32 validation network, and the visit counts at the root.  The two arithmetics that hold the 1e-3
    logit tolerance choose the same action in (nearly) every game; the fp16 default is reported."""
    A = require_gpu()
    from takzero_amd import precision as P

    n, B, k, budget = 5, 128, 64, 768
    states = P.sample_positions(5, 4, B, seed=11, plies=10)
    w = P.trained_scale_weights(A.ARCH_NET5, states[:64], seed=123)
    rng = np.random.default_rng(3)
    gumbel = rng.gumbel(size=(B, 512)).astype(np.float32)      # one draw per child, in possible_moves order
    results = {}
    for name in ("f32", "f16x2", "f16c8", "f16c6", "f16"):
        net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[name]).load_tensors(w)
        mcts = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 14)
        mcts.set_positions(np.arange(B), states)
        top = mcts.gumbel_sequential_halving(np.zeros(B, np.float32), k, budget, gumbel)
        ch = mcts.root_children()
        results[name] = (top.copy(), ch["visits"].astype(np.int64))
        mcts.close()
        net.close()
    ref_top, ref_vis = results["f32"]
    # the fp16 default is gated too, below what round 2 measured for it (0.93 / 0.61), so that a regression of it shows
    for name, min_same, min_identical in (("f16x2", 0.99, 0.97), ("f16c8", 0.98, 0.94), ("f16c6", 0.98, 0.94), ("f16", 0.85, 0.45)):
        top, vis = results[name]
        same = float((top == ref_top).mean())
        identical = float((vis == ref_vis).all(1).mean())
        print("gumbel 64/768 at trained scale, %s vs f32 over %d games: same chosen action %.4f, identical root visit counts %.4f" % (name, B, same, identical))
        assert same >= min_same and identical >= min_identical, name


@pytest.mark.parametrize("scale,B,gates", [
    # (precision, min fraction of games with the same chosen move, min fraction with identical visit counts at every root child,
    #  max mean total-variation distance of the visit distributions)
    ("random-init", 512, (("f16x2", 0.995, 0.98, 0.002), ("f16c8", 0.995, 0.97, 0.003), ("f16c6", 0.995, 0.97, 0.003), ("f16", 0.98, 0.95, 0.02), ("bf16", 0.90, 0.85, 0.10))),
    ("trained", 128, (("f16x2", 0.99, 0.97, 0.005), ("f16c8", 0.99, 0.95, 0.008), ("f16c6", 0.99, 0.95, 0.008), ("f16", 0.90, 0.50, 0.10))),
])
def test_search_with_the_mfma_nets_agrees_with_the_fp32_path_on_moves_and_visits(scale, B, gates):
    """North star: visit counts and chosen moves match the reference's fp32 path.  Bit-exactness of the tree is proven
    against the oracle fed the same network outputs (above); this is the other half: the MFMA precisions against the fp32
    validation network (within 1e-6 of LibTorch) under the same search, same roots, same Dirichlet noise - how far do the
    logit differences move 400-simulation searches?  Random-init weights (|logit| ~ 0.2, flat priors) on 512 games, and weights at
    a trained net's output scale (|logit| = 8, sharp priors: takzero_amd.precision.trained_scale_weights) on 128."""
    A = require_gpu()
    from takzero_amd import precision as P
    from takzero_amd import weights as W

    n, sims = 5, 400
    if scale == "trained":
        w = P.trained_scale_weights(A.ARCH_NET5, P.sample_positions(5, 4, 64, seed=7), seed=123)
    else:
        w = W.init_weights(W.ARCH_NET5, seed=123)
    rng = np.random.default_rng(7)
    choice = rng.integers(0, 16, B)
    results = {}
    noise = None
    for name in ("f32",) + tuple(g[0] for g in gates):
        net = A.Net(arch=A.ARCH_NET5, precision=A.PREC_NAMES[name]).load_tensors(w)
        mcts = A.BatchedMCTS(B, n, 4, agent=net, node_capacity=1 << 15)
        mcts.new_openings(choice)
        betas = np.zeros(B, np.float32)
        mcts.simulate(betas, 1)
        if noise is None:
            info = mcts.root_info()
            amax = int(info["n_children"].max())
            noise = np.zeros((B, amax), np.float32)
            for g in range(B):
                noise[g, :info["n_children"][g]] = rng.dirichlet([0.05] * int(info["n_children"][g])).astype(np.float32)
        mcts.apply_noise(noise, 0.25)
        mcts.simulate(betas, sims)
        ch = mcts.root_children()
        results[name] = (mcts.select_best_actions().copy(), ch["visits"].astype(np.float64), ch["move_idx"].copy())
        mcts.close()
        net.close()
    ref_act, ref_vis, ref_moves = results["f32"]
    for name, min_same, min_identical, max_tv in gates:
        act, vis, moves = results[name]
        assert np.array_equal(moves, ref_moves)
        same = float((act == ref_act).mean())
        tv = 0.5 * np.abs(vis / vis.sum(1, keepdims=True) - ref_vis / ref_vis.sum(1, keepdims=True)).sum(1)
        identical = float((vis == ref_vis).all(1).mean())
        print("%s, %s vs f32 over %d games: same chosen move %.4f of games, identical visit counts %.4f of games, total-variation "
              "distance of the visit distributions mean %.5f max %.4f" % (scale, name, B, same, identical, tv.mean(), tv.max()))
        assert same >= min_same and identical >= min_identical and tv.mean() <= max_tv, name
