#!/bin/bash
# Runs the ThreadSanitizer build of the host code (make tsan-driver) under four threads of mixed C-ABI calls.
# GPU box only; device code is not instrumented.  The HIP runtime itself is not instrumented either: reports
# whose every frame lies in libamdhip64 / libhsa are the runtime's business and are suppressed by name.
set -e
out=${1:-gpurun_out/tsan}
mkdir -p "$out"
cat > "$out/suppressions.txt" <<'SUP'
called_from_lib:libamdhip64.so
called_from_lib:libhsa-runtime64.so
race:libamdhip64.so
race:libhsa-runtime64.so
SUP
export TSAN_OPTIONS="suppressions=$out/suppressions.txt:halt_on_error=0:second_deadlock_stack=1:history_size=4"
bin/thread_stress_tsan ${2:-150} > "$out/thread_stress.log" 2>&1 || true
tail -5 "$out/thread_stress.log"
grep -c "WARNING: ThreadSanitizer" "$out/thread_stress.log" || true
