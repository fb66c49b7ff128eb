"""`compete` of the reference's `evaluation` binary (evaluation/src/main.rs:224-319) over two BatchedMCTS
handles: two networks, one tree per side and per game, both trees stepped with the action the side to move
chose by Gumbel sequential halving; results counted from White's point of view.

`white_mcts` / `black_mcts` can be any objects with the BatchedMCTS call surface (takzero_amd.api.BatchedMCTS on
the GPU, or the CPU oracle's wrapper in the tests)."""
import numpy as np

from . import api

MAX_MOVES = 200          # evaluation/src/main.rs:41 (moves per side)
SAMPLED_ACTIONS = 64     # :44
SEARCH_BUDGET = 768      # :45


class Evaluation:
    def __init__(self):
        self.wins = self.losses = self.draws = 0

    def win_rate(self):
        total = self.wins + self.losses + self.draws
        return self.wins / total if total else float("nan")

    def __repr__(self):
        return "Evaluation(wins=%d, losses=%d, draws=%d)" % (self.wins, self.losses, self.draws)


def compete(white_mcts, black_mcts, games, white_beta, black_beta, rng, sampled_actions=SAMPLED_ACTIONS,
            search_budget=SEARCH_BUDGET, max_moves=MAX_MOVES, amax=512):
    B = white_mcts.batch
    idx = np.arange(B)
    white_mcts.set_positions(idx, games)      # BatchedMCTS::from_envs(games) x2, :240-241
    black_mcts.set_positions(idx, games)
    betas = {True: np.full(B, white_beta, np.float32), False: np.full(B, black_beta, np.float32)}
    done = np.zeros(B, bool)
    ev = Evaluation()
    for _ in range(max_moves):
        for is_white in (True, False):
            if done.all():
                return ev
            cur, oth = (white_mcts, black_mcts) if is_white else (black_mcts, white_mcts)
            gumbel = rng.gumbel(size=(B, amax)).astype(np.float32)
            top = cur.gumbel_sequential_halving(betas[is_white], sampled_actions, search_budget, gumbel)  # :257-273
            cur.step(top)                                                                               # :276-277
            oth.step(top)
            term = cur.restart_terminal_envs(rng.integers(0, 16, B))                                    # :280-288
            newly = (term != api.TERMINAL_NONE) & ~done
            done |= term != api.TERMINAL_NONE
            if done.any():  # also reset the other side's nodes and envs of finished games, :290-299
                d = np.nonzero(done)[0]
                oth.set_positions(d, cur.get_positions()[d])
            # the terminal is seen after the move: a Loss for the side to move is a win for the mover, :306-313
            for t in term[newly]:
                if t == api.TERMINAL_DRAW:
                    ev.draws += 1
                elif (t == api.TERMINAL_LOSS) == is_white:
                    ev.wins += 1
                else:
                    ev.losses += 1
    return ev


def compete_native(white_mcts, black_mcts, games, white_beta, black_beta, seed=0, sampled_actions=SAMPLED_ACTIONS,
                   search_budget=SEARCH_BUDGET, max_moves=MAX_MOVES):
    """The same match played by native code (tz_compete, csrc/tz_host.cpp); draws come from a generator seeded there."""
    from . import _lib

    st = api._states(games)
    out = np.zeros(3, np.int32)
    _lib.check(_lib.load().tz_compete(white_mcts.h, black_mcts.h, st.ctypes.data, white_beta, black_beta, seed, sampled_actions,
                                      search_budget, max_moves, out.ctypes.data))
    ev = Evaluation()
    ev.wins, ev.losses, ev.draws = (int(x) for x in out)
    return ev
