#!/bin/bash
# bench one workload under several values of one environment variable: tools/ab/run_env.sh <workload> <VAR> v1 v2 ...
set -o pipefail
WL="$1"; VAR="$2"; shift 2
for rep in 1 2; do
  for v in "$@"; do
    export $VAR="$v"
    timeout -k 10 200 python bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$VAR=$v', l['config']['workload'], l['ms_per_step'], l['roofline']['frac'])" || exit 1
  done
done
