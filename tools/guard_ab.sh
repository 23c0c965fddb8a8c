for rep in 1 2; do
  for g in 1 0; do
    BTVAE_GUARD=$g timeout -k 10 300 python bench.py --option-steps 0 --imputation-rows 0 --wide-steps 0 --e2e-iterations 0 --scoring-profiles 0 --no-cpu-baseline > gpurun_out/gab_$g$rep.json 2> gpurun_out/gab_$g$rep.err || exit 1
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/gab_$g$rep.json").read().strip().split("\n")[-1])
print("guard=$g rep$rep", round(d["value"]), round(d["ms_per_step"],3), "dom ms", round(d["roofline"]["avg_launch_ms"],4), {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
PY
  done
done
