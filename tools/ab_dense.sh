#!/bin/bash
# A/B of the dense stage-A step under environment switches (run through gpurun from the repo root):
#   tools/ab_dense.sh <tag> "NAME=ENV1=V1,ENV2=V2" ...   -> step / forward C call / backward C call per variant, 2 rounds
OUT=$PWD/gpurun_out/${1:-abd}; shift
mkdir -p $OUT
A="--stage a --workload pascalvoc_sp --route dense --steps 100 --warmup 10"
for r in 1 2; do
  for v in "$@"; do
    name=${v%%=*}; envs=${v#*=}; [ "$envs" = "$v" ] && envs=""
    ( for kv in ${envs//,/ }; do export "$kv"; done; python3 bench.py $A > $OUT/${name}_$r.json 2> $OUT/${name}_$r.err ) || { echo "$name failed"; tail -3 $OUT/${name}_$r.err; }
  done
done
python3 - "$OUT" <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f"{os.path.basename(f):28s} step {1e3*d['ms_per_step']:8.1f} us  A S {r['avg_launch_us']:6.1f} us  fwd call {r.get('fwd_call_us', 0.0):7.1f} us  bwd {r['bwd_launch_us']:7.1f} us  frac {r['frac']:.3f}")
    except Exception as e:
        print(f, "unreadable", e)
PY
