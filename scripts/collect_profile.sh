#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command plus the
# PMC passes needed for HBM traffic and VALU/MFMA utilisation. PMC passes are separate runs with
# --pmc only (never combined with trace domains). Writes small summaries to gpurun_out/prof_final/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency-mode"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_kt -o kt -- $B > $OUT/bench_under_kernel_trace.log 2>&1
find /tmp/pf_kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR" \
         "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d /tmp/pf_pmc$i -o p -- $B > /tmp/pf_pmc$i.log 2>&1
  python $R/scripts/pmc_summary.py /tmp/pf_pmc$i | grep -E "^kernel|klt_kernel|pyr_level|ekf_|gftt_|ingest_|remap_tile|vu_prepare|rot_ransac" > $OUT/pmc$i.csv
done
tail -1 $OUT/bench_under_kernel_trace.log | cut -c1-400
ls -la $OUT
