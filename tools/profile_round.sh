#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun): kernel-trace stats and, in separate
# passes, the HBM byte counters for the headline workload; kernel-trace stats for the others.
# Usage: bash tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/*.txt
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT" || exit 1
run() { # name, seconds, rocprof args..., -- program
  local name=$1 secs=$2; shift 2
  timeout -s KILL "$secs" rocprofv3 "$@" > "$OUT/$name.log" 2>&1 || { echo "$name failed, see $OUT/$name.log"; tail -5 "$OUT/$name.log"; return 1; }
}
run kt_cfft 240 --kernel-trace --stats --output-format csv -d "$OUT/kt_cfft" -- python3 bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 || exit 1
run pmc_fetch 120 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 || exit 1
run pmc_write 120 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --no-cpu-baseline --no-extra --steps 3 --warmup 1 || exit 1
python3 tools/summarize_prof.py "$OUT" "$OUT/cfft_f32_4096x65536.txt" cfft_f32_ > /dev/null || exit 1
for wl in cfft_q15_4096x65536 cfft_q31_4096x65536 cfft_q31_16384x16384 rfft_f32_16384x16384 rfft_q15_32x8388608 rfft_q15_1024x262144 rfft_q31_512x524288 cfft_f32_4194304x8 cfft_f32_4194304x64 gnss_acq_gps_l1_x256; do
  run kt_$wl 240 --kernel-trace --stats --output-format csv -d "$OUT/kt_$wl" -- python3 bench.py --no-cpu-baseline --no-extra --workload $wl --steps 20 --warmup 5 || exit 1
  python3 tools/summarize_prof.py "$OUT/kt_$wl" "$OUT/$wl.txt" > /dev/null
done
grep -h "{" "$OUT"/kt_*.log > "$OUT/bench_lines.jsonl"
rm -rf "$OUT"/kt_* "$OUT"/pmc_*   # keep the summaries only
ls -la "$OUT"
