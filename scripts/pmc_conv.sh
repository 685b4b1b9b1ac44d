cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python -c "import torch" >/dev/null 2>&1
cd $R
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|LDSBankConflict|MfmaUtil[A-Za-z0-9]*|VALUBusy|MfmaBusy[A-Za-z]*" | sort -u | tr '\n' ' ' > gpurun_out/counters.txt
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  T=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcc_$T -o p -- python scripts/conv_launch_table.py 256 > gpurun_out/pmcc_$T.log 2>&1
  python profiles/summarize.py --pmc $(find /tmp/pmcc_$T -name '*_results.db' | head -1) 2>&1 | grep -E "conv_nhwc|conv_wgrad|^kernel" > gpurun_out/pmcc_$T.txt
done
