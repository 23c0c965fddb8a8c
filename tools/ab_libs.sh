#!/bin/bash
# A/B of two builds of the library on one box (box-to-box spread is larger than most kernel changes): headline step and
# scoring leg, alternating.  tools/libbtvae_old.so = the build to compare with (e.g. `git archive <rev> behavio_temporal_vae_amd/csrc`
# built elsewhere); the in-tree build is "new".  Run from the repository root on the GPU box.
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export BTVAE_HIP_LIB=$PWD/tools/libbtvae_old.so; else unset BTVAE_HIP_LIB; fi
    timeout -k 10 300 python bench.py --option-steps 0 --imputation-rows 0 --wide-steps 0 --e2e-iterations 0 --no-cpu-baseline > gpurun_out/ab_$lib$rep.json 2> gpurun_out/ab_$lib$rep.err || exit 1
    python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab_$lib$rep.json").read().strip().split("\n")[-1])
print("$lib$rep", round(d["value"]), round(d["ms_per_step"],3), "dom ms", round(d["roofline"]["avg_launch_ms"],4), "scoring", round(d["secondary"]["value"]), round(d["secondary"]["roofline"]["avg_launch_ms"],3), {k:round(v["ms_per_step"],3) for k,v in d["kernels"].items()})
PY
  done
done
