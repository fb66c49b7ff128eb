#!/bin/bash
# Soak: examples/selfplay_cli.cpp at full width (net5, GAMES games (default 4096; 128 = the reference's own width, which runs the several-CU net form), Gumbel 768 / k 64, exploration) for $1 moves while a
# second process drops a new model_latest.tzw into the directory every SWAP_S (default 45) seconds (hot reload under load).  $2: extra flags (--f16c8 ...).
set -e
cd "$(dirname "$0")/.."
D=$(mktemp -d)
python - "$D" <<'PY'
import sys
from takzero_amd import weights as W, formats as F
W.save_tzw(sys.argv[1] + "/start.tzw", W.init_weights(W.ARCH_NET5, seed=123))
W.save_tzw(sys.argv[1] + "/other.tzw", W.init_weights(W.ARCH_NET5, seed=124))
open(sys.argv[1] + "/buffer_lengths.txt", "w").write(F.format_buffer_lengths(0, 0))
PY
g++ -std=c++17 -O2 examples/selfplay_cli.cpp -Iinclude -Ltakzero_amd -ltakzero_hip -Wl,-rpath,$PWD/takzero_amd -o "$D/selfplay_cli"
( i=0; while sleep ${SWAP_S:-45}; do i=$((i+1)); if [ $((i % 2)) = 1 ]; then cp "$D/other.tzw" "$D/tmp.tzw"; else cp "$D/start.tzw" "$D/tmp.tzw"; fi; mv "$D/tmp.tzw" "$D/model_latest.tzw"; done ) &
SWAP=$!
mkdir -p gpurun_out
"$D/selfplay_cli" --directory "$D" --model "$D/start.tzw" --arch 5 --games ${GAMES:-4096} --sims 768 --search gumbel --exploration --moves ${1:-100} --wait-limit 5 --watch model_latest.tzw ${2:-} > "$D/result.txt" 2>&1 &
CLI=$!
while kill -0 $CLI 2>/dev/null; do   # a progress line every 30 s (the box kills runs that stay silent)
    sleep 30
    echo "$(date +%T) replays $(wc -l < "$D/replays.txt" 2>/dev/null || echo 0) targets_bytes $(stat -c %s "$D/targets-selfplay.txt" 2>/dev/null || echo 0)" | tee -a gpurun_out/soak.log
done
wait $CLI || true
kill $SWAP 2>/dev/null || true
cat "$D/result.txt" | tee -a gpurun_out/soak.log
ls -la "$D" | grep -v tzw
