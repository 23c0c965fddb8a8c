# A/B on one box: the B=4096 headline step eager (default above 32,768 rows) against replayed from a HIP graph
for rep in 1 2; do
  for g in auto 1; do
    BTVAE_STEP_GRAPH=$g timeout -k 10 300 python bench.py --option-steps 0 --imputation-rows 0 --wide-steps 0 --e2e-iterations 0 --scoring-profiles 0 --no-cpu-baseline > gpurun_out/grab_$g$rep.json 2> gpurun_out/grab_$g$rep.err || exit 1
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/grab_$g$rep.json").read().strip().split("\n")[-1])
print("graph=$g rep$rep", round(d["value"]), round(d["ms_per_step"],3), "dom ms", round(d["roofline"]["avg_launch_ms"],4), d["config"]["step_replayed_from_hip_graph"], round(sum(v["ms_per_step"] for v in d["kernels"].values()),3))
PY
  done
done
