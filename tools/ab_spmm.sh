#!/bin/bash
# streaming SpMM variants on one box: tools/ab_spmm.sh <tag>
OUT=$PWD/gpurun_out/${1:-spmm}; mkdir -p $OUT
for cfg in "HSCN_SPMM_PIPE=1" "HSCN_SPMM_PASSES=1" "HSCN_SPMM_PASSES=2 HSCN_SPMM_XCD=1" "HSCN_SPMM_PASSES=4 HSCN_SPMM_XCD=1" "HSCN_SPMM_PASSES=8 HSCN_SPMM_XCD=1" "HSCN_SPMM_PASSES=16 HSCN_SPMM_XCD=1" "HSCN_SPMM_PASSES=8 HSCN_SPMM_XCD=0"; do
  echo "== $cfg"
  env $cfg python3 tools/bench_spmm.py --hidden 16 128 --iters 40 2>/dev/null | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('   H=%3d fwd %6.1f us %5.0f GB/s  bwd %6.1f us %5.0f GB/s' % (d['hidden'], d['median_us'], d['achieved_GBs'], d['bwd_median_us'], d['bwd_achieved_GBs']))"
done | tee $OUT/ab_spmm.txt
