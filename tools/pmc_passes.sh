#!/bin/bash
# tools/pmc_passes.sh <tag> -- <program and args...>
# Counter passes of one command on the GPU box (rocprofv3 --pmc, one pass per
# counter group, kernel trace only), summed per kernel into gpurun_out/<tag>_pmc.txt.
# (SQ / SQC / GRBM counters only: a pass with the TA_* counters aborted inside
# rocprofv3 on this image and then sat silent until the run was killed.)
tag=$1; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd "$root" && export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $g -d "$out/p$i" -o p --output-format csv -- "$@" > "$out/p$i.log" 2>&1 || echo "pass $i failed" >> "$out/errors.txt"
  i=$((i+1))
done
python3 - "$out" > "$root/gpurun_out/${tag}_pmc.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen and r["Counter_Name"]:
            seen.add((k, r["Dispatch_Id"]))
    for k, _ in seen:
        n[k] += 0
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:40s} {tot[k][c]:.6g}")
PY
cat "$root/gpurun_out/${tag}_pmc.txt"
