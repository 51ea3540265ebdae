#!/bin/bash
# Stage-A profiles on the GPU box (run through gpurun from the repo root):
#   tools/run_stage_a_profiles.sh <tag>  -> gpurun_out/<tag>/{stage_a_bench.json, stage_a_kernel_stats.csv,
#                                           stage_a_driver.json, stage_a_driver_kernel_stats.csv, diag_scn_step.txt}
set -o pipefail
TAG=${1:-sa}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 python3 bench.py --stage a > $OUT/stage_a_bench.json 2> $OUT/stage_a_bench.err || exit 1
timeout -k 10 200 python3 bench.py --stage a --batch 256 > $OUT/stage_a_bench_b256.json 2>> $OUT/stage_a_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --stage a --steps 200 --warmup 20 > $OUT/stage_a_bench_traced.json 2> $OUT/trace.err || exit 1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/stage_a_kernel_stats.csv \;
rm -rf $OUT/trace
timeout -k 10 300 python3 tools/bench_train_clustering.py > $OUT/stage_a_driver.json 2> $OUT/driver.err || exit 1
timeout -k 10 600 python3 tools/bench_stage_a_visits.py > $OUT/stage_a_visits.json 2>> $OUT/driver.err || exit 1
if [ -f graph-hscn_amd/graph_hscn/lib/libhscn_diag.so ]; then
  HSCN_LIB=graph-hscn_amd/graph_hscn/lib/libhscn_diag.so timeout -k 10 200 python3 tools/diag_scn.py step > $OUT/diag_scn_step.txt 2>&1 || exit 1
fi
ls -la $OUT
