#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> stage times of tools/step_breakdown.py for each value of the env var
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"; env $var=$v python3 tools/step_breakdown.py 2>&1 | tail -2 | head -1
done
