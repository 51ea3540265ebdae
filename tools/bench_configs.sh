# Stage C step on the other BASELINE configurations (parity-test configurations; not the bench line), the half-storage
# and dataset-resident-structure lines, the batch sweep and the stage-A routes.  Run through gpurun from the repo root:
#   tools/bench_configs.sh <tag>   -> gpurun_out/<tag>/*.json + summary.txt
TAG=${1:-cfg}
OUT=gpurun_out/$TAG
mkdir -p $OUT
COMMON="--no-cpu-baseline --no-streaming-spmm --no-stage-a-dense --steps 200"
run() {  # name, args...
  name=$1; shift
  timeout -k 10 300 python bench.py "$@" $COMMON 2>/dev/null | tail -1 > $OUT/$name.json
  python - "$OUT/$name.json" "$name" <<'PY' >> $OUT/summary.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[2], "FAILED", e); raise SystemExit
c = d["config"]
o = d.get("other_cluster_ids") or {}
s1 = d.get("one_step_per_graph") or {}
sa = d.get("stage_a") or {}
print(f'{sys.argv[2]:28s} B {c.get("graphs_per_gpu")} {d["dtype"][:3]} {round(d["value"]):>9d} graphs/s {d["ms_per_step"]*1e3:7.1f} us'
      f' | 1 step/graph {s1.get("ms_per_step", 0)*1e3:6.1f} us | other ids {o.get("ms_per_step", 0)*1e3:6.1f} us'
      f' | stage A {round(sa.get("graphs_per_s", 0))} | {c.get("step_issue", "")[:40]} V={c.get("virtual_nodes_per_gpu")}')
PY
}
: > $OUT/summary.txt
run peptides_struct --workload peptides_struct
run pascalvoc_sp --workload pascalvoc_sp
run pcqm_contact_f32 --workload pcqm_contact
run pcqm_contact_f16 --workload pcqm_contact --dtype f16
run peptides_func_f16 --dtype f16 --no-stage-a
run peptides_func_h32 --hidden 32 --no-stage-a
run peptides_func_resident_structure --structure dataset-resident --no-stage-a
HSCN_ONE_LAUNCH=0 run peptides_func_pair --no-stage-a
for b in 64 256 512 1024; do run batch_$b --batch $b; done
timeout -k 10 300 python bench.py --workload pascalvoc_sp --stage a --route dense --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/stage_a_dense.json
timeout -k 10 300 python bench.py --workload pascalvoc_sp --stage a --route sparse --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/stage_a_sparse.json
timeout -k 10 300 python bench.py --workload peptides_func --stage a --route sparse --steps 200 2>/dev/null | tail -1 > $OUT/stage_a_peptides.json
python - $OUT <<'PY' >> $OUT/summary.txt
import json, sys
for n in ("stage_a_dense", "stage_a_sparse", "stage_a_peptides"):
    try:
        d = json.load(open(f"{sys.argv[1]}/{n}.json"))
        r = d.get("roofline") or {}
        print(f'{n:28s} {round(d["value"]):>9d} graphs/s {d["ms_per_step"]*1e3:8.1f} us | {d["config"]["step_issue"][:60]} | roofline frac {r.get("frac")} dominant {r.get("avg_launch_us")} us fwd call {r.get("fwd_call_us")} us bwd {r.get("bwd_launch_us")} us')
    except Exception as e:
        print(n, "FAILED", e)
PY
cat $OUT/summary.txt
