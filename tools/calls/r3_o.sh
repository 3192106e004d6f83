#!/bin/bash
# Round 3, GPU call O: counters of the split-f16 product (matrix-pipe busy, waits, LDS conflicts, clock)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ring lds; do
  export EPSILON_HIP_GEMM_STAGE=$v
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/r3o_pmc_$v -o pmc --output-format csv -- python3 $R/tools_gemm_pmc.py > $O/r3o_$v.log 2>&1 || { tail -20 $O/r3o_$v.log; exit 1; }
  tail -3 $O/r3o_$v.log
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for v in ("ring", "lds"):
    files = glob.glob("gpurun_out/r3o_pmc_%s/**/*counter_collection.csv" % v, recursive=True)
    kt = glob.glob("gpurun_out/r3o_pmc_%s/**/*kernel_trace.csv" % v, recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = collections.defaultdict(dict)
    name = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if "GemmSplitF16Kernel" not in r["Kernel_Name"]:
                continue
            rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = (r["Kernel_Name"][:60], r["Grid_Size"])
    for d, c in sorted(rows.items(), key=lambda kv: int(kv[0])):
        g = c.get("GRBM_GUI_ACTIVE", 0)
        us = dur.get(d, 0)
        print(v, d, name[d], "us %.0f" % us, "clock GHz %.2f" % (g / 8 / us / 1e3 if us else 0),
              "mfma_busy %.3f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 128) if g else 0),
              {k: "%.3g" % x for k, x in c.items()})
PY
