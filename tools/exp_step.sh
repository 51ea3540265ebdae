#!/bin/bash
# one GPU call of the step-kernel loop: parity tests of the step, phase stamps (diagnostic build), bench line
#   tools/exp_step.sh <tag> [pytest selection...]
set -o pipefail
TAG=${1:-exp}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
TESTS=${@:-tests/test_gpu_step.py tests/test_gpu_resident.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py tests/test_gpu_structure.py tests/test_gpu_f16.py tests/test_gpu_pipeline.py tests/test_gpu_allreduce.py}
timeout -k 10 900 python3 -m pytest $TESTS -x -q -m gpu > $OUT/tests.log 2>&1; rc=$?
tail -25 $OUT/tests.log
[ $rc -ne 0 ] && { echo "TESTS FAILED rc=$rc"; exit $rc; }
HSCN_LIB=$PWD/graph-hscn_amd/graph_hscn/lib/libhscn_diag.so timeout -k 10 300 python3 tools/diag_step.py > $OUT/diag_default.txt 2> $OUT/diag_default.err || { tail -5 $OUT/diag_default.err; exit 1; }
HSCN_LIB=$PWD/graph-hscn_amd/graph_hscn/lib/libhscn_diag.so timeout -k 10 300 python3 tools/diag_step.py uniform > $OUT/diag_uniform.txt 2> $OUT/diag_uniform.err || { tail -5 $OUT/diag_uniform.err; exit 1; }
cat $OUT/diag_default.txt
grep -A16 "virtual workgroup slowest" $OUT/diag_uniform.txt
timeout -k 10 400 python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-streaming-spmm > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("ms/step", d["ms_per_step"], "repeats", d["repeats"]["ms_per_step"], "one_step_per_graph", d["one_step_per_graph"]["ms_per_step"],
      "uniform", d["other_cluster_ids"]["ms_per_step"], "stage_a", d["stage_a"]["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"])
PY
