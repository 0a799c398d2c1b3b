#!/bin/bash
# SQ counter passes for one bench workload (run on the GPU box): bash tools/pmc.sh <workload> <kernel-filter> [tag]
set -u
WL=$1; FLT=$2; TAG=${3:-$1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; O=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd "$ROOT" || exit 1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d "$O/p$i" -- python3 bench.py --no-cpu-baseline --no-extra --workload "$WL" --steps 2 --warmup 1 > "$O/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$O/p$i.log"; exit 1; }
done
python3 tools/summarize_prof.py "$O" "$O/sum.txt" "$FLT" | cut -c1-50,112-
rm -rf "$O"/p1 "$O"/p2
