#!/usr/bin/env bash
# HBM traffic of the step's kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (they do not fit one pass; no tracing flags beside --pmc), the program itself after `--`.
# Usage (GPU box): bash tools/gpu_pmc.sh <tag>     ->  gpurun_out/<tag>/pmc_{FETCH,WRITE}_SIZE.csv + pmc_traffic.json
set -e -o pipefail
tag=${1:-pmc}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/p_$c" -o pmc -- python3 bench.py --steps 3 --warmup 2 --no-spinup --no-cpu-baseline --no-module-api > "$out/pmc_$c.log" 2>&1
  cp "$(find "$out/p_$c" -name "*counter_collection.csv" | head -1)" "$out/pmc_$c.csv"
  rm -rf "$out/p_$c"
done
python3 tools/pmc_traffic.py "$out/pmc_FETCH_SIZE.csv" "$out/pmc_WRITE_SIZE.csv" > "$out/pmc_traffic.json"
cat "$out/pmc_traffic.json"
