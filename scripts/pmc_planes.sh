#!/bin/bash
# SQ / GRBM counters of the plane-format attention and GEMM kernels (where does the time go: MFMA, VALU, LDS, waiting; effective clock).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r03/pmc_planes}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_pl_$i -o p -- python $R/scripts/pmc_planes.py > /tmp/pmc_pl_$i.log 2>&1
  f=$(find /tmp/pmc_pl_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_summary.py $f $OUT/set$i.csv
  k=$(find /tmp/pmc_pl_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$k" ] && cp $k $OUT/kernel_trace_$i.csv
  tail -2 /tmp/pmc_pl_$i.log
done
grep -h "flash_attn_pl\|gemm_pl" $OUT/set*.csv | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-70,150-400
