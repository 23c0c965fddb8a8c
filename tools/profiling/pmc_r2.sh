set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BTVAE_BENCH_PREC=fp32_split
for op in wgrad gstats_big bnbwd_big dgrad_pw; do
for c in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc2_${op}_$n -o x -- python3 $R/tools/gemm_bench.py $op 3 > /dev/null 2>&1
  echo done $op $n
done
done
