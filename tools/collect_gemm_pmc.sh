#!/bin/bash
# PMC evidence for a GEMM launch (separate passes, kernel-trace free as the pool requires): MFMA busy cycles vs active cycles, LDS
# bank conflicts, VALU / LDS / wait counters.  Writes gpurun_out/gemm_pmc.txt.
#   tools/collect_gemm_pmc.sh                     the dominant projection GEMM ([15000,4096] x [4096,128], tools/one_shape.py)
#   tools/collect_gemm_pmc.sh tools/probe_eval.py the similarity GEMM of configs[3] (10,500 x 10,500 x 1,200)
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT to the repository root (gpurun exports it)}"
O=gpurun_out/gemm_pmc
rm -rf $O && mkdir -p $O
if [ $# -ge 1 ]; then CMD="$*"; else CMD="tools/one_shape.py 4096 15000 128"; fi
for c in ${PMC_LIST:-SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INST_CYCLES_VMEM}; do
  rocprofv3 --pmc $c --output-format csv -d $O/$c -- python3 $CMD > $O/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/gemm_pmc"
res = collections.defaultdict(dict)
for d in sorted(glob.glob(O + "/*/")):
    name = os.path.basename(d.rstrip("/"))
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            res[k][name] = sum(v) / len(v)
with open("gpurun_out/gemm_pmc.txt", "w") as out:
    for k, d in res.items():
        if "gemm" not in k and "fixup" not in k:
            continue
        out.write(k + "\n")
        for c, v in d.items():
            out.write(f"    {c:28s} {v:16.1f}  (mean per launch, summed over XCDs / SEs as rocprofv3 reports it)\n")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            # 1024 SIMDs each count busy cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            out.write(f"    MFMA busy fraction = BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) = {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}\n")
        if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
            out.write(f"    LDS bank-conflict share of LDS cycles = {d['SQ_LDS_BANK_CONFLICT'] / max(1.0, d['SQ_LDS_IDX_ACTIVE']):.4f}\n")
print(open("gpurun_out/gemm_pmc.txt").read())
PY
rm -rf $O
