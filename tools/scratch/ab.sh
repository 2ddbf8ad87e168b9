set -e
run() { echo "== $EXTRA $*"; env "$@" python3 bench.py --steps 30 --warmup 5 $EXTRA 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms'))
"; }
EXTRA=""
run X=1
EXTRA="--tune 16=0"
run X=1
EXTRA=""
run X=1
EXTRA="--tune 16=0"
run X=1
