#!/bin/bash
# bash tools/scan.sh wl1 wl2 ...  -> one line per workload: GB/s achieved, launch ms
cd ${GRAFT_REPO_ROOT:-.}
for wl in "$@"; do
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --workload $wl --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-28s %10.1f %s  launch %.4f ms  frac %.3f  value %.1f %s' % (d['config']['workload'], r['achieved'], r['unit'], r['avg_launch_ms'], r['frac'], d['value'], d['unit']))" || exit 1
done
