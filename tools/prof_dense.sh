#!/bin/bash
# Dense MinCUT route (BASELINE.json configs[3]) on the GPU box: kernel times and MFMA-busy of the step
#   tools/prof_dense.sh <tag> -> gpurun_out/<tag>/{dense_*.json, kernel_stats.csv, pmc_mfma.csv, mfma_summary.txt}
set -o pipefail
TAG=${1:-dense}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
A="--stage a --workload pascalvoc_sp --route dense --steps 100 --warmup 10"
python3 bench.py $A > $OUT/dense_u8.json 2> $OUT/dense_u8.err || { tail -5 $OUT/dense_u8.err; exit 1; }
HSCN_DENSE_ADJ=f32 python3 bench.py $A > $OUT/dense_f32.json 2> $OUT/dense_f32.err || { tail -5 $OUT/dense_f32.err; exit 1; }
HSCN_DENSE_ADJ=f32 HSCN_DENSE_AS=16 python3 bench.py $A > $OUT/dense_f32_16x16.json 2> $OUT/dense_f32_16x16.err || { tail -5 $OUT/dense_f32_16x16.err; exit 1; }
HSCN_DENSE_AS=33 python3 bench.py $A > $OUT/dense_u8_lds.json 2> $OUT/dense_u8_lds.err || { tail -5 $OUT/dense_u8_lds.err; exit 1; }
python3 - <<PY
import json
for n in ("dense_u8", "dense_u8_lds", "dense_f32", "dense_f32_16x16"):
    d = json.load(open("$OUT/" + n + ".json")); r = d["roofline"]
    print(f"{n:18s} step {1e3*d['ms_per_step']:8.1f} us  A S {r['avg_launch_us']:6.1f} us  fwd C call {r.get('fwd_call_us', 0.0):7.1f} us  bwd {r['bwd_launch_us']:7.1f} us  frac {r['frac']:.3f}  ({d['config']['step_issue'][:40]})")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $A > $OUT/traced.json 2> $OUT/trace.err || { tail -5 $OUT/trace.err; exit 1; }
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc -- python3 bench.py $A --mode eager > $OUT/pmc.json 2> $OUT/pmc.err || { tail -5 $OUT/pmc.err; exit 1; }
find $OUT/pmc -name "*counter_collection.csv" -exec cp {} $OUT/pmc_mfma.csv \;
find $OUT/pmc -name "*kernel_trace.csv" -exec cp {} $OUT/pmc_kernel_trace.csv \;
rm -rf $OUT/trace $OUT/pmc
python3 tools/mfma_summary.py $OUT/pmc_mfma.csv $OUT/pmc_kernel_trace.csv > $OUT/mfma_summary.txt
cat $OUT/mfma_summary.txt
head -12 $OUT/kernel_stats.csv | cut -c1-150
