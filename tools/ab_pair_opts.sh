#!/usr/bin/env bash
# same-box A/B of the pair recurrent kernels' options (caphn_tune key 24: bit 0 no same-XCD hand-off form, bit 1 forward mat-vec
# not split around the score exchange, bit 2 backward transposed mat-vec not split around the d alpha exchange)
set -e
for rep in 1 2; do for v in 0 1 6 7 2 4; do
  echo "== opts $v"; python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-module-api --tune 24=$v 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print(d['ms_per_step'], d['roofline'].get('kernel_ms'))
"; done; done
