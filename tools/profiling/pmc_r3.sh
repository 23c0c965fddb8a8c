#!/bin/bash
# Beyond-L2 traffic of the legs' dominant kernels: one rocprofv3 pass per counter over a short bench.py run (the counters
# do not fit one pass), then tools/profiling/traffic_json.py turns the two CSVs into profiles/traffic.json.
# Run on the GPU box from the repository root: bash tools/profiling/pmc_r3.sh
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_r3_$c -o x -- python3 $R/bench.py --steps 2 --warmup 1 --option-steps 0 --no-cpu-baseline --scoring-profiles 32768 --imputation-rows 25000 --imputation-sweeps 4 --e2e-iterations 0 --wide-steps 3 > $R/gpurun_out/pmc_r3_$c.log 2>&1
  echo done $c
done
cd $R && python3 tools/profiling/traffic_json.py gpurun_out/pmc_r3_FETCH_SIZE gpurun_out/pmc_r3_WRITE_SIZE gpurun_out/traffic.json
