#!/bin/bash
# SQ counter passes on the step kernels (run through gpurun from the repo root):
#   tools/run_sq_counters.sh <tag>  -> gpurun_out/<tag>/{sq_pass1.csv, sq_pass2.csv}
# Each pass is its own rocprofv3 run with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots: 8 SQ
# counters per pass); eager issue of the same launches the hipGraph replays.  The stage-A step rides along
# (bench.py's stage_a leg) so k_scn_step is in the same files.
set -o pipefail
TAG=${1:-sq}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
COMMON="--mode eager --steps 30 --warmup 5 --no-cpu-baseline --no-streaming-spmm --no-other-ids --no-stage-a-dense"
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $OUT/p1 -- python3 bench.py $COMMON > $OUT/bench_p1.json 2> $OUT/p1.err || { tail -5 $OUT/p1.err; exit 1; }
echo pass1 done
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/p2 -- python3 bench.py $COMMON > $OUT/bench_p2.json 2> $OUT/p2.err || { tail -5 $OUT/p2.err; exit 1; }
echo pass2 done
find $OUT/p1 -name "*counter_collection.csv" -exec cp {} $OUT/sq_pass1.csv \;
find $OUT/p2 -name "*counter_collection.csv" -exec cp {} $OUT/sq_pass2.csv \;
rm -rf $OUT/p1 $OUT/p2
python3 tools/sq_summary.py $OUT/sq_pass1.csv $OUT/sq_pass2.csv > $OUT/sq_summary.txt
cat $OUT/sq_summary.txt
