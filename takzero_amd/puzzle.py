"""Batch puzzle solver on the search ABI — the reference's `puzzle` binary (puzzle/src/main.rs:38-269): tinue
("find the winning road") and tinue-avoidance puzzles from a SQLite database are overwritten into the batch's
positions, searched with Gumbel sequential halving at beta = 0, and scored by the selected move (solved) and by
what the terminal solver proved (proven)."""
import sqlite3
from dataclasses import dataclass

import numpy as np

from . import api

BATCH_SIZE = 64      # puzzle/src/main.rs:33
SEED = 12345         # :34


@dataclass
class PuzzleResult:  # puzzle/src/main.rs:113-129
    attempted: int = 0
    solved: int = 0
    proven: int = 0

    def solve_rate(self):
        return self.solved / self.attempted if self.attempted else float("nan")

    def prove_rate(self):
        return self.proven / self.attempted if self.attempted else float("nan")


_TINUE = """SELECT tps, solution FROM puzzles
    JOIN games ON puzzles.game_id = games.id
    WHERE games.size = :size
        AND instr(tps, '1C') > 0
        AND instr(tps, '2C') > 0
        AND puzzles.tinue_length = :depth
        AND puzzles.tinue_avoidance_length IS NULL
        AND puzzles.tiltak_2komi_second_move_eval < 0.6
    ORDER BY puzzles.game_id ASC"""
_AVOIDANCE = """SELECT tps, solution FROM puzzles
    JOIN games ON puzzles.game_id = games.id
    WHERE games.size = :size
        AND instr(tps, '1C') > 0
        AND instr(tps, '2C') > 0
        AND puzzles.tinue_avoidance_length = :depth
        AND puzzles.tinue_length IS NULL
        AND puzzles.tiltak_2komi_eval < 0.6
    ORDER BY game_id ASC"""


def load_puzzles(db_path, kind, depth, n=6, half_komi=4):
    """The `tinue` / `avoidance` queries (puzzle/src/main.rs:131-166) -> (states, solution move indices)."""
    con = sqlite3.connect(str(db_path))
    try:
        rows = con.execute(_TINUE if kind == "tinue" else _AVOIDANCE, {"size": n, "depth": depth}).fetchall()
    finally:
        con.close()
    states = np.zeros(len(rows), api.STATE_DTYPE)
    solutions = np.zeros(len(rows), np.uint16)
    for i, (tps, solution) in enumerate(rows):
        states[i] = api.state_from_tps(tps, n, half_komi)
        solutions[i] = api.move_from_ptn(n, solution)
    return states, solutions


def benchmark(mcts, puzzles, solutions, win, sampled_actions, search_budget, rng):
    """`benchmark` (puzzle/src/main.rs:168-269).  `puzzles`: tz_state records, `solutions`: move indices."""
    B = mcts.batch
    zero_beta = np.zeros(B, np.float32)
    result = PuzzleResult()
    amax = 512 if mcts.n < 6 else 1024
    for lo in range(0, len(puzzles), B):
        chunk, want = puzzles[lo:lo + B], np.asarray(solutions[lo:lo + B])
        k = len(chunk)
        states = mcts.get_positions()          # the tail of a short last batch keeps its previous positions (:198-204)
        states[:k] = chunk
        mcts.set_positions(np.arange(B), states)  # every node reset (:195-197)
        gumbel = rng.gumbel(size=(B, amax)).astype(np.float32)
        mcts.gumbel_sequential_halving(zero_beta, sampled_actions, search_budget, gumbel)
        selected = mcts.select_best_actions()   # :215
        result.attempted += k
        result.solved += int((selected[:k] == want).sum())
        if win:   # roots solved to a win (:237-243)
            info = mcts.root_info()
            result.proven += int((info["eval_tag"][:k] == api.EVAL_WIN).sum())
        else:     # all but one child solved as a win for the opponent-to-move (:244-258)
            info, ch = mcts.root_info(), mcts.root_children()
            valid = np.arange(ch["eval_tag"].shape[1])[None, :] < info["n_children"][:, None]
            wins = ((ch["eval_tag"] == api.EVAL_WIN) & valid).sum(axis=1)
            result.proven += int((wins[:k] == info["n_children"][:k].astype(np.int64) - 1).sum())
    return result


def run(mcts, db_path, sampled_actions=64, search_budget=768, log=print):
    """real_main (puzzle/src/main.rs:43-111): tinue 3/5/7/9 then avoidance 2/4/6 with one rng."""
    rng = np.random.default_rng(SEED)
    out = {}
    for kind, depths, win in (("tinue", (3, 5, 7, 9), True), ("avoidance", (2, 4, 6), False)):
        for depth in depths:
            states, solutions = load_puzzles(db_path, kind, depth, mcts.n, mcts.half_komi)
            res = benchmark(mcts, states, solutions, win, sampled_actions, search_budget, rng)
            out[(kind, depth)] = res
            if log:
                log("%s %d: %r %s" % (kind, depth, res, res.solve_rate()))
    return out


def benchmark_native(mcts, puzzles, solutions, win, sampled_actions, search_budget, seed=SEED):
    """`benchmark` in native code (tz_puzzle_benchmark, csrc/tz_host.cpp)."""
    from . import _lib

    st = api._states(puzzles)
    sol = np.ascontiguousarray(solutions, np.uint16)
    out = np.zeros(3, np.int32)
    _lib.check(_lib.load().tz_puzzle_benchmark(mcts.h, st.ctypes.data, sol.ctypes.data, len(st), 1 if win else 0, seed,
                                               sampled_actions, search_budget, out.ctypes.data))
    return PuzzleResult(int(out[0]), int(out[1]), int(out[2]))
