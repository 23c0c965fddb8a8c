set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BTVAE_BENCH_PREC=fp32_split2
for op in fwd_pw wgrad bnbwd_big; do
for c in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc3_${op}_$n -o x -- python3 $R/tools/gemm_bench.py $op 3 > /dev/null 2>&1
  echo done $op $n
done
done
