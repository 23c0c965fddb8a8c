set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BTVAE_BENCH_PREC=fp32_split2
i=0
for cfg in "0 0" "0 1" "1 0" "1 1"; do
  set -- $cfg
  i=$((i+1))
  if [ "$1" = "1" ]; then export BTVAE_SPLIT_NARROW=1; else unset BTVAE_SPLIT_NARROW; fi
  if [ "$2" = "1" ]; then export BTVAE_BENCH_ZERO=1; else unset BTVAE_BENCH_ZERO; fi
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc4_n$1_z$2 -o x -- python3 $R/tools/gemm_bench.py fwd_pw 3 > /dev/null 2>&1
  echo done $cfg
done
