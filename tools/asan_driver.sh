#!/bin/bash
# Runs the host-sanitized driver (make asan-driver) through the legacy path, a BVH scene, the
# progressive/resume path, a two-way device split, the in-HBM multi-device gather and the two opt-in modes.  GPU box only; device code is not instrumented.
set -e
export ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
out=${1:-gpurun_out/asan}
mkdir -p "$out"
exe=bin/ptrender_asan
PTGPU_SPP=8 PTGPU_BOUNCES=4 $exe 0 320 200 "$out/c0.bmp"
PTGPU_SPP=16 PTGPU_BOUNCES=8 $exe 2 640 360 "$out/c2.bmp"
PTGPU_SPP=32 PTGPU_BOUNCES=8 $exe 5 1920 1080 "$out/c5.bmp"          # large enough for the pilots
PTGPU_SPP=4 PTGPU_BOUNCES=8 PTGPU_FLAGS=0x500 $exe 5 320 180 "$out/c5q.bmp"
PTGPU_SPP=8 PTGPU_DEVICES=2 $exe 2 320 200 "$out/c2_2dev.bmp"
PTGPU_SPP=8 PTGPU_MULTI=3 $exe 2 321 200 "$out/c2_multi.bmp"            # frame assembled in HBM (ptgpu_multi_*)
PTGPU_SPP=4 PTGPU_BOUNCES=4 PTGPU_FLAGS=0x1000 $exe 0 320 200 "$out/c0_fast.bmp"   # tolerance build
PTGPU_SPP=4 PTGPU_BOUNCES=4 PTGPU_FLAGS=0x2000 $exe 0 320 200 "$out/c0_glsl.bmp"   # shader renderer's rules
PTGPU_SPP=20 PTGPU_PROGRESSIVE=4 PTGPU_RESUME="$out/acc.bin" PTGPU_STOP_AFTER=8 $exe 2 320 200 "$out/prog.bmp"
PTGPU_SPP=20 PTGPU_PROGRESSIVE=4 PTGPU_RESUME="$out/acc.bin" $exe 2 320 200 "$out/prog.bmp"
echo "asan_driver: ok"
