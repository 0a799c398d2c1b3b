#!/bin/bash
# A/B of two builds of the library over bench.py workloads: tools/ab/run_ab.sh "<workloads>" libA.so libB.so ...
set -o pipefail
WL="$1"; shift
for rep in 1 2; do
for w in $WL; do
  for lib in default "$@"; do
    if [ "$lib" = default ]; then unset ARMFFT_LIB; else export ARMFFT_LIB="$PWD/$lib"; fi
    timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$lib', l['config']['workload'], l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_ms'])" || exit 1
  done
done
done
