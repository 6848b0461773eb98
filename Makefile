# Build of the MI355X (gfx950) render path.
#
#   make            -> path_tracer_c_amd/lib/libptgpu.so   (product, C-ABI)
#                      oracle/liboracle.so                  (test oracle)
#                      bin/ptrender                         (host C driver)
#
# Device code is compiled with -ffp-contract=off: the path's contract is
# bit-equality with a sequential IEEE-754 binary32 evaluation.
# Division and square root are correctly rounded (hipcc default, made explicit).
HIPCC    ?= /opt/rocm/bin/hipcc
CC       ?= gcc
ARCH     ?= gfx950

PKG      := path_tracer_c_amd
CSRC     := $(PKG)/csrc
BUILD    := build
LIBDIR   := $(PKG)/lib

HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize $(EXTRA_HIPFLAGS) \
            -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
            -Wall -Wno-unused-function -Iinclude -I$(CSRC)
CFLAGS   := -std=gnu2x -O2 -fPIC -ffp-contract=off -Wall -Wextra -Iinclude

# The tolerance mode (PTGPU_FLAG_FAST_MATH): the same device sources compiled a
# second time into namespace pt_fast with explicit fused multiply-adds and the
# approximate sqrt / rcp / rsq instructions as they come (csrc/device/pt_vec.h).
# Same compiler flags as the exact build: -ffp-contract stays OFF and the `/` and
# sqrtf() operators stay IEEE.  What makes the mode cheaper is written out in
# the source (pt_vec.h, pt_trace.h): explicit fused multiply-adds and the
# v_sqrt / v_rcp / v_rsq builtins.  Left to the compiler (-ffp-contract=fast,
# approximate division) the same expression came out differently in different
# kernels and the mode's frames depended on the schedule.
FASTFLAGS := $(HIPFLAGS) -DPT_FAST_MATH -DPT_NS=pt_fast -DPT_MEGA_WAVES=8 $(EXTRA_FASTFLAGS)

HOST_OBJS := $(BUILD)/scene.o $(BUILD)/vector.o $(BUILD)/bitmap.o $(BUILD)/scenes.o $(BUILD)/threadpool.o
HIP_OBJS  := $(BUILD)/pt_kernels.o $(BUILD)/pt_kernels_fast.o $(BUILD)/pt_glsl.o $(BUILD)/ptgpu_abi.o
DEV_HDRS  := $(wildcard $(CSRC)/device/*.h) $(wildcard $(CSRC)/*.hpp) $(wildcard include/*.h)

all: $(LIBDIR)/libptgpu.so oracle bin/ptrender

$(BUILD) $(LIBDIR) bin:
	mkdir -p $@

$(BUILD)/%.o: $(CSRC)/host/%.c include/path_tracer_api.h include/ptgpu.h $(CSRC)/host/pool_rows.h | $(BUILD)
	$(CC) $(CFLAGS) -c $< -o $@

$(BUILD)/pt_kernels.o: $(CSRC)/device/pt_kernels.hip $(DEV_HDRS) | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(BUILD)/pt_glsl.o: $(CSRC)/device/pt_glsl.hip $(DEV_HDRS) | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(BUILD)/pt_kernels_fast.o: $(CSRC)/device/pt_kernels.hip $(DEV_HDRS) | $(BUILD)
	$(HIPCC) $(FASTFLAGS) -c $< -o $@

$(BUILD)/ptgpu_abi.o: $(CSRC)/ptgpu_abi.hip $(DEV_HDRS) | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libptgpu.so: $(HIP_OBJS) $(HOST_OBJS) | $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS) $(HOST_OBJS) -lpthread -lm

bin/ptrender: $(CSRC)/host/main.c $(LIBDIR)/libptgpu.so | bin
	$(CC) $(CFLAGS) $< -o $@ -L$(LIBDIR) -lptgpu -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -lm -lpthread

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(BUILD) $(LIBDIR) bin
	$(MAKE) -C oracle clean

.PHONY: all oracle clean asan-driver tsan-driver

# Host-side AddressSanitizer + UBSan build of the C driver with the library
# linked in statically: the C-ABI translation unit (all host logic: upload, BVH
# build, schedule choice, scratch, statistics, legacy shims) and the host C are
# instrumented; the kernels' translation unit is built plainly (with
# -Xarch_host -fsanitize on that one the kernels come out as no-ops).  Device
# code is never instrumented (-Xarch_host / -fno-gpu-sanitize).
# Run on a GPU box: tools/asan_driver.sh
ASAN_HOST := -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer
CLANG     := /opt/rocm/lib/llvm/bin/clang
asan-driver: | $(BUILD) bin
	$(HIPCC) $(filter-out -O3,$(HIPFLAGS)) -O1 -g $(ASAN_HOST) -c $(CSRC)/ptgpu_abi.hip -o $(BUILD)/asan_abi.o
	$(HIPCC) $(filter-out -O3,$(HIPFLAGS)) -O1 -g -c $(CSRC)/device/pt_kernels.hip -o $(BUILD)/asan_kernels.o
	$(HIPCC) $(filter-out -O3,$(FASTFLAGS)) -O1 -g -c $(CSRC)/device/pt_kernels.hip -o $(BUILD)/asan_kernels_fast.o
	$(HIPCC) $(filter-out -O3,$(HIPFLAGS)) -O1 -g -c $(CSRC)/device/pt_glsl.hip -o $(BUILD)/asan_glsl.o
	for f in scene vector bitmap scenes threadpool main; do \
	  $(CLANG) -std=gnu2x -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -Iinclude \
	    -c $(CSRC)/host/$$f.c -o $(BUILD)/asan_$$f.o || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) -fno-gpu-sanitize -fsanitize=address,undefined -o bin/ptrender_asan \
	  $(BUILD)/asan_main.o $(BUILD)/asan_threadpool.o $(BUILD)/asan_scene.o $(BUILD)/asan_vector.o $(BUILD)/asan_bitmap.o $(BUILD)/asan_scenes.o \
	  $(BUILD)/asan_abi.o $(BUILD)/asan_kernels.o $(BUILD)/asan_kernels_fast.o $(BUILD)/asan_glsl.o -lpthread -lm

# Host-side ThreadSanitizer build: the C-ABI translation unit and the host C
# instrumented, the kernels plain, linked with a four-thread stress driver
# (tests/c/thread_stress.c).  Run on a GPU box: tools/tsan_driver.sh
TSAN_HOST := -Xarch_host -fsanitize=thread -Xarch_host -fno-omit-frame-pointer
tsan-driver: | $(BUILD) bin
	$(HIPCC) $(filter-out -O3,$(HIPFLAGS)) -O1 -g $(TSAN_HOST) -c $(CSRC)/ptgpu_abi.hip -o $(BUILD)/tsan_abi.o
	$(HIPCC) $(filter-out -O3,$(HIPFLAGS)) -O1 -g -c $(CSRC)/device/pt_kernels.hip -o $(BUILD)/tsan_kernels.o
	$(HIPCC) $(filter-out -O3,$(FASTFLAGS)) -O1 -g -c $(CSRC)/device/pt_kernels.hip -o $(BUILD)/tsan_kernels_fast.o
	$(HIPCC) $(filter-out -O3,$(HIPFLAGS)) -O1 -g -c $(CSRC)/device/pt_glsl.hip -o $(BUILD)/tsan_glsl.o
	for f in scene vector bitmap scenes threadpool; do \
	  $(CLANG) -std=gnu2x -O1 -g -fsanitize=thread -fno-omit-frame-pointer -ffp-contract=off -Iinclude -I$(CSRC)/host \
	    -c $(CSRC)/host/$$f.c -o $(BUILD)/tsan_$$f.o || exit 1; done
	$(CLANG) -std=gnu2x -O1 -g -fsanitize=thread -fno-omit-frame-pointer -Iinclude -c tests/c/thread_stress.c -o $(BUILD)/tsan_stress.o
	$(HIPCC) --offload-arch=$(ARCH) -fno-gpu-sanitize -fsanitize=thread -o bin/thread_stress_tsan \
	  $(BUILD)/tsan_stress.o $(BUILD)/tsan_threadpool.o $(BUILD)/tsan_scene.o $(BUILD)/tsan_vector.o $(BUILD)/tsan_bitmap.o $(BUILD)/tsan_scenes.o \
	  $(BUILD)/tsan_abi.o $(BUILD)/tsan_kernels.o $(BUILD)/tsan_kernels_fast.o $(BUILD)/tsan_glsl.o -lpthread -lm
