#!/bin/bash
# Full-width run of examples/selfplay_cli.cpp (no Python in the search process): net5, 4096 games, Gumbel 768 / k 64.
set -e
cd "$(dirname "$0")/.."
D=$(mktemp -d)
python - "$D" <<'PY'
import sys
from takzero_amd import weights as W, formats as F
W.save_tzw(sys.argv[1] + "/start.tzw", W.init_weights(W.ARCH_NET5, seed=123))
open(sys.argv[1] + "/buffer_lengths.txt", "w").write(F.format_buffer_lengths(0, 0))
PY
g++ -std=c++17 -O2 examples/selfplay_cli.cpp -Iinclude -Ltakzero_amd -ltakzero_hip -Wl,-rpath,$PWD/takzero_amd -o "$D/selfplay_cli"
"$D/selfplay_cli" --directory "$D" --model "$D/start.tzw" --arch 5 --games 4096 --sims 768 --search gumbel --moves ${1:-12} --wait-limit 5
ls -la "$D" | tail -5
