#!/bin/bash
# tools/exp_lib.sh <experimental libptgpu.so> -- <command...>: runs the command with the experimental library
# in place of the built one, then puts the built one back (GPU box; for A/B timing of compile-time variants)
set -e
exp=$1; shift; shift
cp path_tracer_c_amd/lib/libptgpu.so /tmp/libptgpu_real.so
cp "$exp" path_tracer_c_amd/lib/libptgpu.so
"$@" || true
cp /tmp/libptgpu_real.so path_tracer_c_amd/lib/libptgpu.so
