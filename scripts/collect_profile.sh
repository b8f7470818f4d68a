#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the headline legs of the bench command (C2 + C3 at the
# default B) plus the PMC passes needed for HBM traffic and VALU / MFMA utilisation. PMC passes are separate runs with --pmc
# only (never combined with trace domains). Writes small summaries to gpurun_out/prof_r06/ (copy what is judged to profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-prof_r06}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
B="python $R/bench.py --steps 10 --warmup 2 --only-headline --repeats 1 --verify 0 --no-graph ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_kt -o kt -- $B > $OUT/bench_under_kernel_trace.log 2>&1
find /tmp/pf_kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d /tmp/pf_pmc$i -o p -- $B > /tmp/pf_pmc$i.log 2>&1
  python $R/scripts/pmc_summary.py /tmp/pf_pmc$i > $OUT/pmc$i.csv
done
python $R/scripts/make_traffic_json.py $OUT $OUT/traffic.json > /dev/null 2> $OUT/make_traffic.err
tail -1 $OUT/bench_under_kernel_trace.log | cut -c1-300
ls -la $OUT
