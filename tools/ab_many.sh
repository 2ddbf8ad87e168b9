#!/usr/bin/env bash
# several in-process A/Bs in a row (tools/ab_inproc.py): each argument is "attr A B"
for spec in "$@"; do
  python3 tools/ab_inproc.py $spec 2>&1 | grep -v amdgpu.ids | tail -1
done
