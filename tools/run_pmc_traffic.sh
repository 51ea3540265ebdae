#!/bin/bash
# Only the two PMC traffic passes of tools/run_profiles.sh (FETCH_SIZE, WRITE_SIZE; eager issue of the default step):
#   HSCN_COMMIT=<id> bash tools/run_pmc_traffic.sh <tag>  -> gpurun_out/<tag>/pmc_traffic.{json,txt}
set -o pipefail
TAG=${1:-r03}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-stage-a-dense --no-other-ids"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -- python3 bench.py --mode eager --steps 30 --warmup 5 $COMMON > $OUT/bench_pmc_f.json 2> $OUT/pmc_f.err || { tail -5 $OUT/pmc_f.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -- python3 bench.py --mode eager --steps 30 --warmup 5 $COMMON > $OUT/bench_pmc_w.json 2> $OUT/pmc_w.err || { tail -5 $OUT/pmc_w.err; exit 1; }
python3 tools/pmc_traffic.py $(find $OUT/pmc_f -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_w -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json ${HSCN_COMMIT:-unrecorded} > $OUT/pmc_traffic.txt || exit 1
rm -rf $OUT/pmc_f $OUT/pmc_w
ls -la $OUT/pmc_traffic.json
