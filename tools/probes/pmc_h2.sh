#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the f16x2 kernels (two --pmc passes of eight counters + one --kernel-trace --stats pass for the
# durations) on the big layer shapes and the thin 24-channel layers.   tools/probes/pmc_h2.sh <out-dir under gpurun_out>
O=$1; mkdir -p $O
R=$(pwd)
export TMPDIR=/tmp GIF_PROBE_MODE=f16x2
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
Bc="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL"
( cd /tmp && rocprofv3 --kernel-trace --pmc $A GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmcA -- python $R/tools/pmc_probe_x3.py > $R/$O/pmcA.log 2>&1 ); echo "A rc=$?"
( cd /tmp && rocprofv3 --kernel-trace --pmc $Bc --output-format csv -d $R/$O/pmcB -- python $R/tools/pmc_probe_x3.py > $R/$O/pmcB.log 2>&1 ); echo "B rc=$?"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kt -- python $R/tools/pmc_probe_x3.py > $R/$O/kt.log 2>&1 ); echo "kt rc=$?"
python tools/pmc_summary.py $O/pmcA > $O/pmcA.txt; python tools/pmc_summary.py $O/pmcB > $O/pmcB.txt
F=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -16 $F > $O/kt_stats.csv
rm -rf $O/pmcA $O/pmcB $O/kt
grep -c "^##" $O/pmcA.txt; cut -c1-160 $O/kt_stats.csv
