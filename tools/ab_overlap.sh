#!/usr/bin/env bash
# Same-box A/B of FusedTrainer.overlap_level (CAPHN_OVERLAP_LEVEL): alternating bench runs, ms/step per run.
set -e -o pipefail
out=gpurun_out/${1:-ab_overlap}; mkdir -p "$out"
for rep in 1 2; do
  for lv in ${LEVELS:-1 3 4}; do
    CAPHN_OVERLAP_LEVEL=$lv python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-module-api > "$out/l${lv}_$rep.json" 2> "$out/l${lv}_$rep.err"
    echo "level $lv rep $rep: $(grep -o '"ms_per_step": [0-9.]*' "$out/l${lv}_$rep.json")"
  done
done
