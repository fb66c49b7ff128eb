#!/bin/bash
# Three processes, one GPU, one directory: learn + selfplay + reanalyze for $1 seconds (default 240).
cd "$(dirname "$0")/.."
D=$(mktemp -d)
S=${1:-240}
mkdir -p gpurun_out
python tools/concurrent_demo.py learn "$D" $S 2>&1 | tee -a gpurun_out/concurrent.log &
python tools/concurrent_demo.py selfplay "$D" $S 2>&1 | tee -a gpurun_out/concurrent.log &
python tools/concurrent_demo.py reanalyze "$D" $S 2>&1 | tee -a gpurun_out/concurrent.log &
for i in $(seq 1 $((S / 20 + 3))); do   # progress: what the three processes have put into the directory so far
    sleep 20
    echo "[watch] selfplay targets $(wc -l < "$D/targets-selfplay.txt" 2>/dev/null || echo 0) replays $(wc -l < "$D/replays.txt" 2>/dev/null || echo 0) reanalyze targets $(wc -l < "$D/targets-reanalyze.txt" 2>/dev/null || echo 0) buffer_lengths $(cat "$D/buffer_lengths.txt" 2>/dev/null) models $(ls "$D" | grep -c "\.ot$")" | tee -a gpurun_out/concurrent.log
    if ! pgrep -P $$ python > /dev/null; then break; fi
done
pkill -P $$ python 2>/dev/null || true   # only this script's own children
wait
ls -la "$D" | tee -a gpurun_out/concurrent.log
