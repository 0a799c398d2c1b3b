#!/bin/bash
# HBM byte counters per workload (run on the GPU box): separate rocprofv3 --pmc passes for FETCH_SIZE
# and WRITE_SIZE, summed over the workload's kernels per step.  bash tools/traffic.sh <tag> wl:filter ...
# -> gpurun_out/traffic_<tag>.txt with one line per workload: FETCH KiB, WRITE KiB per launch
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd "$ROOT" || exit 1
OUT=$ROOT/gpurun_out/traffic_$TAG.txt; : > "$OUT"
for spec in "$@"; do
  WL=${spec%%:*}; FLT=${spec##*:}
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$ROOT/gpurun_out/tr_$C; rm -rf "$D"
    timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d "$D" -- python3 bench.py --no-cpu-baseline --no-extra --workload "$WL" --steps 2 --warmup 1 > "$D.log" 2>&1 || { echo "$WL $C failed"; tail -3 "$D.log"; exit 1; }
    python3 tools/summarize_prof.py "$D" "$D.txt" "$FLT" > /dev/null
    awk -v wl="$WL" -v c="$C" '$0 !~ /^#/ && $0 !~ /^kernel/ { n=NF; printf "%s %s %s calls=%s avg_per_dispatch=%s\n", wl, c, $(n-4), $(n-2), $(n-1) }' "$D.txt" | sed 's/ [^ ]*<\([^>]*\)>[^ ]* / <\1> /' >> "$OUT"
    rm -rf "$D" "$D.log" "$D.txt"
  done
done
cat "$OUT"
