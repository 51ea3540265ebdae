for b in 64 256 512 1024; do
  timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline --no-streaming-spmm --steps 200 2>/dev/null | tail -1 > gpurun_out/sweep_$b.json
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_$b.json"))
print($b, round(d["value"]), round(d["ms_per_step"]*1e3,1), "us; stage A", round(d["stage_a"]["graphs_per_s"]), round(d["stage_a"]["ms_per_step"]*1e3,1), "us")
PY
done
