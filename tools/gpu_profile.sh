#!/usr/bin/env bash
# One gpurun call: default bench line, then the same bench under rocprofv3 --kernel-trace --stats, reduced to a one-step
# timeline (tools/timeline.py) and the kernel-stats CSV.  Usage (on the GPU box): bash tools/gpu_profile.sh <tag> [bench args]
set -e -o pipefail
tag=${1:-run}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 50 --warmup 5 "$@" > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o tr -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > "$out/prof_bench.log" 2>&1
f=$(find "$out/prof" -name "*kernel_trace.csv" | head -1)
cp "$f" "$out/kernel_trace.csv"
python3 tools/timeline.py "$f" --marker grad_norm_partials --step 20 > "$out/timeline.txt"
cp "$(find "$out/prof" -name "*kernel_stats.csv" | head -1)" "$out/kernel_stats.csv"
rm -rf "$out/prof"
grep -o '"ms_per_step": [0-9.]*' "$out/bench.json"
