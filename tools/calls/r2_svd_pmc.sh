#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "MemUnitStalled MeanOccupancyPerCU VmemLatency MfmaUtil" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_svd$i -o p -- python3 $R/tools_microbench.py svd:10000:1 > $R/gpurun_out/pmc_svd$i.log 2>&1; echo "pass $i rc=$?"
  python3 - $R/gpurun_out/pmc_svd$i/p_results.db <<'PY'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
for kn, cn, cnt, av in con.execute(q):
    k = re.sub(r"\(anonymous namespace\)::", "", kn).split("(")[0]
    if "Panel" in k or "PairEig" in k:
        print("%-34s %-40s n=%5d avg=%.4g" % (k[-34:], cn, cnt, av))
PY
  rm -rf $R/gpurun_out/pmc_svd$i
done
