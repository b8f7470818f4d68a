#!/bin/bash
# PMC passes over a short command (default: the visual-track microbench at B = 1024); summaries to gpurun_out/$PROF_DIR
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-pmc_micro}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
CMD=${PMC_CMD:-"python $R/scripts/vu_microbench.py 1024 10 1"}
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  rocprofv3 --pmc $C --output-format csv -d /tmp/pm_$i -o p -- $CMD > /tmp/pm_$i.log 2>&1
  python $R/scripts/pmc_summary.py /tmp/pm_$i > $OUT/pmc$i.csv 2>> $OUT/err.log
done
ls -la $OUT
