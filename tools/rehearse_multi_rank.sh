#!/bin/bash
# What a one-GPU box can say about the N-rank path (run through gpurun from the repo root):
#   tools/rehearse_multi_rank.sh <tag>  -> gpurun_out/<tag>/
#  1. python bench.py --gpus 2 with one GPU visible must REFUSE (exit code != 0, no JSON line)
#  2. the same job as a rehearsal: 2 ranks share the GPU, gloo process group, one-shot all-reduce through hipIpc
#     (self-launch, per-rank shards, exchange in the captured step, strong-scaling leg, A/B child job)
#  3. one rank, collective forced: RCCL vs the one-shot kernel (its default form for this gradient size = granules, and
#     the slab + flag form forced) behind the same captured step
set -o pipefail
TAG=${1:-rehearse}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
COMMON="--steps 200 --warmup 20 --no-cpu-baseline --no-streaming-spmm --no-stage-a --no-other-ids"
python3 bench.py --gpus 2 --steps 5 --warmup 1 > $OUT/refuse.json 2> $OUT/refuse.err
echo "refusal exit code: $? (stdout bytes: $(wc -c < $OUT/refuse.json))" | tee $OUT/refuse.txt
HSCN_BENCH_REHEARSAL=1 timeout -k 10 420 python3 bench.py --gpus 2 --allreduce oneshot $COMMON > $OUT/rehearsal_2ranks.json 2> $OUT/rehearsal_2ranks.err || { echo rehearsal failed; tail -20 $OUT/rehearsal_2ranks.err; exit 1; }
echo rehearsal done
HSCN_BENCH_FORCE_DIST=1 python3 bench.py --allreduce rccl $COMMON > $OUT/forced_rccl.json 2> $OUT/forced_rccl.err || { tail -5 $OUT/forced_rccl.err; exit 1; }
HSCN_BENCH_FORCE_DIST=1 python3 bench.py --allreduce oneshot $COMMON > $OUT/forced_oneshot.json 2> $OUT/forced_oneshot.err || { tail -5 $OUT/forced_oneshot.err; exit 1; }
HSCN_ALLREDUCE_FORM=s HSCN_BENCH_FORCE_DIST=1 python3 bench.py --allreduce oneshot $COMMON > $OUT/forced_oneshot_slabs.json 2> $OUT/forced_oneshot_slabs.err || { tail -5 $OUT/forced_oneshot_slabs.err; exit 1; }
python3 bench.py $COMMON > $OUT/no_collective.json 2> $OUT/no_collective.err
python3 - <<PY
import json
for n in ("rehearsal_2ranks", "forced_rccl", "forced_oneshot", "forced_oneshot_slabs", "no_collective"):
    d = json.load(open("$OUT/" + n + ".json"))
    print(n, "n_gpus", d["n_gpus"], "ms/step", round(d["ms_per_step"], 5), "repeats", d.get("repeats", {}).get("ms_per_step"),
          "allreduce", d["config"].get("allreduce_algorithm"), "strong", d.get("strong_scaling"), "ab", d.get("allreduce_ab"))
PY
