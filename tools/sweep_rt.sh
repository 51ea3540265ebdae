# workgroup size vs batch size for the one-launch kernels (HSCN_RT picks the instantiation)
for b in 128 1024; do
  for rt in 256 512 1024; do
    HSCN_OVERLAP_VIRTUAL=0 HSCN_RT=$rt timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline --no-streaming-spmm --no-stage-a --steps 100 2>/dev/null | tail -1 > gpurun_out/rt_${b}_${rt}.json
    python - <<PY
import json
d=json.load(open("gpurun_out/rt_${b}_${rt}.json"))
print("B", $b, "RT", $rt, round(d["value"]), "graphs/s", round(d["ms_per_step"]*1e3,1), "us")
PY
  done
done
