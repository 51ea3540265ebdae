#!/bin/bash
# The round's committed evidence, regenerated from ONE state of the sources (run through gpurun from the repo root,
# with the commit id of that state: the GPU box has no .git):
#   HSCN_COMMIT=$(git rev-parse --short HEAD) gpurun -- 'HSCN_COMMIT=... bash tools/run_profiles.sh r03'
#   -> gpurun_out/<tag>/: bench.json (the default command), step_kernel_stats.csv (rocprofv3 --kernel-trace --stats of
#      it), pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes, eager issue of the same launches; MI355X_MICROARCH.md
#      prescribes the two counters in passes of their own, with --kernel-trace only), sq_counters.txt (two SQ passes)
# Copy them to profiles/<tag>_* afterwards (tools/final_pass.sh does).
set -o pipefail
TAG=${1:-r03}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo bench done
COMMON="--no-cpu-baseline --no-stage-a-dense --no-other-ids"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 200 --warmup 20 $COMMON --no-streaming-spmm > $OUT/bench_traced.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/step_kernel_stats.csv \;
echo stats done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -- python3 bench.py --mode eager --steps 30 --warmup 5 $COMMON > $OUT/bench_pmc_f.json 2> $OUT/pmc_f.err || { tail -5 $OUT/pmc_f.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -- python3 bench.py --mode eager --steps 30 --warmup 5 $COMMON > $OUT/bench_pmc_w.json 2> $OUT/pmc_w.err || { tail -5 $OUT/pmc_w.err; exit 1; }
python3 tools/pmc_traffic.py $(find $OUT/pmc_f -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_w -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json ${HSCN_COMMIT:-unrecorded} > $OUT/pmc_traffic.txt || { echo "pmc_traffic.py failed"; exit 1; }
echo traffic done
rm -rf $OUT/trace $OUT/pmc_f $OUT/pmc_w
bash tools/run_sq_counters.sh $TAG/sq > /dev/null 2>&1 && cp $OUT/sq/sq_summary.txt $OUT/sq_counters.txt
ls -la $OUT
grep -E "k_hscn_step|k_param_reduce|k_spmm" $OUT/pmc_traffic.txt | head
