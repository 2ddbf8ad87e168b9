#!/usr/bin/env bash
# Same-box A/B of environment settings: tools/ab_env.sh <tag> "<ENV=..>" "<ENV=..>" ...  (alternating bench runs, ms/step per run)
set -e -o pipefail
out=gpurun_out/${1:-ab_env}; shift; mkdir -p "$out"
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    env $setting python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-module-api > "$out/s${i}_$rep.json" 2> "$out/s${i}_$rep.err"
    echo "[$setting] rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/s${i}_$rep.json") $(grep -o '"kernel_ms": [0-9.]*' "$out/s${i}_$rep.json")"
  done
done
