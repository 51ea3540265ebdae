OUT=$PWD/gpurun_out/lb2; mkdir -p $OUT
Q="--no-cpu-baseline --no-streaming-spmm --no-stage-a --no-stage-a-dense"
for w in "pcqm_contact 2048" "pcqm_contact 4096" "peptides_struct 256" "peptides_struct 1024"; do
  set -- $w
  for v in 0 1; do
    HSCN_ONE_LAUNCH_LARGE_B=$v python3 bench.py --workload $1 --batch $2 $Q > $OUT/$1_$2_lb$v.json 2> $OUT/err.txt || { tail -3 $OUT/err.txt; exit 1; }
  done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(1e3 * d["ms_per_step"], 1), "us", round(d["value"] / 1e6, 2), "M graphs/s | uniform ids",
          round(1e3 * d["other_cluster_ids"]["ms_per_step"], 1), "us", d["config"].get("step_issue", "")[:60])
PY
