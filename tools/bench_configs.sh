# stage C step on the other BASELINE configs (parity-test configurations; not the bench line)
for w in peptides_struct pascalvoc_sp pcqm_contact; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-streaming-spmm --steps 200 2>/dev/null | tail -1 > gpurun_out/cfg_$w.json
  python - <<PY
import json
d=json.load(open("gpurun_out/cfg_$w.json"))
print("$w", "B", d["config"]["graphs_per_gpu"], round(d["value"]), "graphs/s", round(d["ms_per_step"]*1e3,1), "us; engine", d["config"]["engine"], "; stage A", round(d["stage_a"]["graphs_per_s"]) if d.get("stage_a") else None)
PY
done
timeout -k 10 300 python bench.py --workload peptides_func --hidden 32 --no-cpu-baseline --no-streaming-spmm --steps 200 2>/dev/null | tail -1 > gpurun_out/cfg_h32.json
python - <<PY
import json
d=json.load(open("gpurun_out/cfg_h32.json"))
print("peptides_func H=32", round(d["value"]), "graphs/s", round(d["ms_per_step"]*1e3,1), "us")
PY
