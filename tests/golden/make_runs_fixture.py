"""Extracts move-order data from the reference's search dumps (runs/*.txt, data files, not code).

Each line of runs/puct.txt lists the root children of one 5x5 position as
`move:visits:eval:std_dev` in child order = fast-tak `possible_moves` order.
We keep the move names of all 1024 lines (1005 distinct move lists; round 2 kept every 8th): the one in-tree
artefact that pins fast-tak's move ordering.  Run in the build container: python tests/golden/make_runs_fixture.py
"""
import os

SRC = "/root/reference/runs/puct.txt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "runs_puct_move_order.txt")

with open(SRC) as f, open(DST, "w") as out:
    for i, line in enumerate(f):
        moves = [rec.split(":")[0] for rec in line.strip().split(",") if rec]
        out.write(" ".join(moves) + "\n")
print("wrote", DST)
