#!/bin/bash
# Profiles of the bench command on the GPU box (run through gpurun from the repo root):
#   tools/run_profiles.sh <tag>     -> gpurun_out/<tag>/{kernel_stats.csv, pmc_fetch.csv, pmc_write.csv, bench_*.json}
# kernel trace + stats of the default bench command (hipGraph replays), then the two PMC passes (separate runs, eager
# issue of the same launches: MI355X_MICROARCH.md prescribes FETCH_SIZE and WRITE_SIZE in passes of their own, with
# --kernel-trace only).
set -o pipefail
TAG=${1:-prof}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-streaming-spmm --no-stage-a --no-other-ids"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 $COMMON > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -- python3 bench.py --mode eager --steps 30 --warmup 5 $COMMON > $OUT/bench_pmc_f.json 2> $OUT/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -- python3 bench.py --mode eager --steps 30 --warmup 5 $COMMON > $OUT/bench_pmc_w.json 2> $OUT/pmc_w.err
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/pmc_f -name "*counter_collection.csv" -exec cp {} $OUT/pmc_fetch.csv \;
find $OUT/pmc_w -name "*counter_collection.csv" -exec cp {} $OUT/pmc_write.csv \;
rm -rf $OUT/trace $OUT/pmc_f $OUT/pmc_w
ls -la $OUT
