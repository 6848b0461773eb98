#!/bin/bash
# tools/profile_round.sh <round tag, e.g. r02>
# Everything profiles/<tag>/ holds, produced on the GPU box into
# gpurun_out/profiles_<tag>/ (copy what is to be judged into profiles/<tag>/).
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
mkdir -p "$out"
cd "$root" && export TMPDIR=/tmp
say() { echo "[profile_round] $*"; }
for w in c2 c5 c3 c4 c1; do
  extra=""; [ $w != c2 ] && extra="--no-cpu-baseline"
  say "bench $w"
  python3 bench.py --workload $w --steps 10 --warmup 2 $extra > "$out/bench_$w.json" 2> "$out/bench_$w.err" || say "bench $w FAILED"
done
for w in c2 c5; do
  say "rocprofv3 --kernel-trace --stats $w"
  rocprofv3 --kernel-trace --stats -d "$out/trace_$w" -o t --output-format csv -- python3 bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > "$out/trace_$w.log" 2>&1 || say "trace $w FAILED"
  cp "$out/trace_$w"/t_kernel_stats.csv "$out/${w}_kernel_stats.csv" 2>/dev/null || find "$out/trace_$w" -name "*kernel_stats.csv" -exec cp {} "$out/${w}_kernel_stats.csv" \;
  # the dominant kernel's launches one by one (the --stats average mixes pilots, timed steps and side measurements)
  k="pt::pt_megakernel<false, false, 0>"; [ $w = c5 ] && k="pt::pt_streamkernel<false, false>"
  python3 tools/dispatch_list.py "$(find $out/trace_$w -name '*kernel_trace.csv' | head -1)" "$k" > "$out/${w}_dispatches.csv" 2>&1
done
say "HBM traffic passes (FETCH_SIZE, WRITE_SIZE) c2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/hbm_c2_f" -o p --output-format csv -- python3 tools/one_render.py c2 0x280 256 3 8 > "$out/hbm_c2_f.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/hbm_c2_w" -o p --output-format csv -- python3 tools/one_render.py c2 0x280 256 3 8 > "$out/hbm_c2_w.log" 2>&1
python3 tools/hbm_traffic.py "$(find $out/hbm_c2_f -name '*counter_collection.csv')" "$(find $out/hbm_c2_w -name '*counter_collection.csv')" pt_megakernel 3 > "$out/hbm_c2.json" 2>&1
say "HBM traffic passes c5"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/hbm_c5_f" -o p --output-format csv -- python3 tools/one_render.py c5 0x700 64 2 > "$out/hbm_c5_f.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/hbm_c5_w" -o p --output-format csv -- python3 tools/one_render.py c5 0x700 64 2 > "$out/hbm_c5_w.log" 2>&1
python3 tools/hbm_traffic.py "$(find $out/hbm_c5_f -name '*counter_collection.csv')" "$(find $out/hbm_c5_w -name '*counter_collection.csv')" pt_streamkernel,pt_sum_kernel 2 > "$out/hbm_c5.json" 2>&1
say "done"
ls -la "$out" | head -40
