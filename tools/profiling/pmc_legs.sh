set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_legs_$c -o x -- python3 $R/bench.py --steps 2 --warmup 1 --option-steps 0 --no-cpu-baseline --scoring-profiles 32768 --imputation-rows 25000 --imputation-sweeps 4 --e2e-iterations 0 --wide-steps 0 > /dev/null 2>&1
  echo done $c
done
