#!/bin/bash
# Three processes, one GPU, one directory: learn + selfplay + reanalyze for $1 seconds (default 240).
cd "$(dirname "$0")/.."
D=$(mktemp -d)
S=${1:-240}
mkdir -p gpurun_out
LOG=gpurun_out/concurrent.log
PIDS=()
for role in learn selfplay reanalyze; do
    python tools/concurrent_demo.py $role "$D" $S >> $LOG 2>&1 &
    PIDS+=($!)
done
alive() { for p in "${PIDS[@]}"; do kill -0 $p 2>/dev/null && return 0; done; return 1; }
for i in $(seq 1 $((S / 20 + 3))); do   # progress: what the three processes have put into the directory so far
    sleep 20
    echo "[watch] selfplay targets $(wc -l < "$D/targets-selfplay.txt" 2>/dev/null || echo 0) replays $(wc -l < "$D/replays.txt" 2>/dev/null || echo 0) reanalyze targets $(wc -l < "$D/targets-reanalyze.txt" 2>/dev/null || echo 0) buffer_lengths $(cat "$D/buffer_lengths.txt" 2>/dev/null) models $(ls "$D" | grep -c "\.ot$")" | tee -a $LOG
    alive || break
done
for p in "${PIDS[@]}"; do kill $p 2>/dev/null || true; done   # exactly the three processes started above
wait
ls -la "$D" | tee -a $LOG
