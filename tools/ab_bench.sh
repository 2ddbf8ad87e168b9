#!/usr/bin/env bash
# Same-box A/B of bench.py variants (boxes differ by +-4 %, so two variants are only comparable inside ONE gpurun call, and
# alternating).  Usage:  bash tools/ab_bench.sh "" "--tune 16=0"   -> runs each argument string twice, alternating, and prints
# ms/step, the dominant kernel's roofline fraction and its duration for every run.
set -e
run() { echo "== $1"; python3 bench.py --steps 30 --warmup 5 $1 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))
"; }
for rep in 1 2; do for v in "$@"; do run "$v"; done; done
