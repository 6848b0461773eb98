#!/bin/bash
# tools/pmc_passes.sh <tag> -- <program and args...>
# Counter passes of one command on the GPU box (rocprofv3 --pmc, one pass per
# counter group, kernel trace only), summed per kernel into gpurun_out/<tag>_pmc.txt.
# Round 2 asked for eight TA_* counters in one pass: rocprofiler_create_counter_config
# refused ("error code 38: Request exceeds the capabilities of the hardware to collect"),
# rocprofv3 aborted and the run sat silent until it was killed.  The memory-path
# counters are therefore collected TWO per pass (the TCP / TCC blocks have few slots,
# and a *_sum counter takes one on every instance), and every pass runs under its
# own timeout so that a refused group costs one pass, not the run.
#   PMC_GROUPS=sq|cache|all (default all)
tag=$1; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p "$out"
cd "$root" && export TMPDIR=/tmp
sq_groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
)
cache_groups=(
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
 "TCC_HIT_sum TCC_MISS_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum"
)
case "${PMC_GROUPS:-all}" in
  sq) groups=("${sq_groups[@]}");;
  cache) groups=("${cache_groups[@]}");;
  *) groups=("${sq_groups[@]}" "${cache_groups[@]}");;
esac
i=0
for g in "${groups[@]}"; do
  echo "[pmc_passes] pass $i: $g"
  timeout -k 10 ${PMC_PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $g -d "$out/p$i" -o p --output-format csv -- "$@" > "$out/p$i.log" 2>&1 || echo "pass $i ($g) failed: $(grep -m1 -i "error\|abort" "$out/p$i.log")" | tee -a "$out/errors.txt"
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
