#!/bin/bash
# MFMA-busy of the A S launch for the row-tile variants (HSCN_DENSE_ROWS = 128 / 256): tools/prof_dense_rows.sh <tag>
set -o pipefail
OUT=$PWD/gpurun_out/${1:-drows}; mkdir -p $OUT
export TMPDIR=/tmp
A="--stage a --workload pascalvoc_sp --route dense --steps 60 --warmup 10 --mode eager"
for rows in 128 256; do
  export HSCN_DENSE_ROWS=$rows
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc$rows -- python3 bench.py $A > $OUT/pmc$rows.json 2> $OUT/pmc$rows.err || { tail -3 $OUT/pmc$rows.err; exit 1; }
  python3 tools/mfma_summary.py $(find $OUT/pmc$rows -name "*counter_collection.csv" | head -1) $(find $OUT/pmc$rows -name "*kernel_trace.csv" | head -1) > $OUT/sum$rows.txt; grep -E "k_adj_s_direct|k_adj_s<" $OUT/sum$rows.txt | sed "s/^/rows $rows: /"
  rm -rf $OUT/pmc$rows
done
