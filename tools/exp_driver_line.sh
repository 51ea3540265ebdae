#!/bin/bash
# The driver's command is `bench.py --gpus 1 --steps 20 --warmup 5`: 0.6 ms of timed steps, where the fixed cost of the
# timed region (replay launches, the closing synchronize) shows.  Prints ms_per_step of that command, three runs per
# variant, beside the 200-step default (run through gpurun from the repo root).
OUT=$PWD/gpurun_out/${1:-drv}
mkdir -p $OUT
Q="--no-cpu-baseline --no-streaming-spmm --no-stage-a --no-other-ids --no-stage-a-dense"
for r in 1 2 3; do
  python3 bench.py --steps 20 --warmup 5 $Q > $OUT/s20_$r.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
  HSCN_BENCH_SPIN=1 python3 bench.py --steps 20 --warmup 5 $Q > $OUT/s20_spin_$r.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
done
for r in 1 2; do python3 bench.py $Q > $OUT/s200_$r.json 2>> $OUT/err.txt || exit 1; done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(1e3 * d["ms_per_step"], 2), "us", [round(1e3 * x, 2) for x in d.get("repeats", {}).get("ms_per_step", [])])
PY
