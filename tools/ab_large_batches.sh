#!/bin/bash
# The one-launch step at large batches (2 B workgroups in several rounds of the chip; graph_hscn/step.py: large_b)
# beside the launch pair (HSCN_ONE_LAUNCH_LARGE_B=0: what B > 128 took before): us per step, default and uniform ids.
OUT=$PWD/gpurun_out/${1:-large_b}
mkdir -p $OUT
Q="--no-cpu-baseline --no-streaming-spmm --no-stage-a --no-stage-a-dense"
for b in 192 256 512 1024 2048; do
  HSCN_ONE_LAUNCH_LARGE_B=0 python3 bench.py --batch $b $Q > $OUT/b${b}_pair.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
  python3 bench.py --batch $b $Q > $OUT/b${b}_one.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
done
HSCN_ONE_LAUNCH_LARGE_B=0 python3 bench.py --batch 256 --hidden 32 $Q > $OUT/h32_b256_pair.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
HSCN_ONE_LAUNCH_LARGE_B=all python3 bench.py --batch 256 --hidden 32 $Q > $OUT/h32_b256_one.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(1e3 * d["ms_per_step"], 1), "us", round(d["value"] / 1e6, 2), "M graphs/s | uniform ids",
          round(1e3 * d["other_cluster_ids"]["ms_per_step"], 1), "us")
PY
