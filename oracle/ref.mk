# oracle/ref.mk -- builds the part of the reference that builds from its own
# sources, UNMODIFIED, where they lie under /root/reference, into oracle/_ref/
# (git-ignored; travels to the GPU box with the snapshot like any built .so).
# Test infrastructure only: nothing in the product loads oracle/_ref.
#
#   make -C oracle -f ref.mk            (needs /root/reference; a no-op without)
#
# What the reference lets us build with plain gcc and no stand-in headers:
#   src/bitmap.c      (includes bitmap.h + external/portable_endian.h)
#   src/vector.c      (includes vector.h)
#   src/threadpool.c  (includes threadpool.h, vector.h, <pthread.h>)
# What it does not: src/renderer.c, src/ray.c, src/scene.c and src/main.c all
# include <cglm/...>, and external/cglm/ is an empty, un-vendored submodule
# directory -- building them would need stand-ins for the missing headers, so
# they count as unbuildable here.  The reference's own
# CMake is not run.
#
# -Wl,-Bsymbolic: threadpool.c's calls to vector_push/vector_pop bind inside
# this library even when libptgpu.so (same symbol names) is loaded first.
REF     ?= /root/reference
CC      ?= gcc
OUT     := _ref
# gnu2x as the reference's CMakeLists.txt:3-4 asks (C23 + GNU extensions:
# htole32 needs _DEFAULT_SOURCE); -w: the reference's own warnings are not ours
CFLAGS  := -std=gnu2x -O2 -fPIC -w -I$(REF)/include -I$(REF)/external

ifeq ($(wildcard $(REF)/src/bitmap.c),)
all:
	@echo "oracle/_ref: $(REF) is absent here; keeping whatever is prebuilt"
else
all: $(OUT)/libref_host.so $(OUT)/libref_probe.so

$(OUT):
	mkdir -p $@

$(OUT)/libref_host.so: $(REF)/src/bitmap.c $(REF)/src/vector.c $(REF)/src/threadpool.c | $(OUT)
	$(CC) $(CFLAGS) -shared -Wl,-Bsymbolic -o $@ $^ -lpthread

# our probe file, compiled against the reference's own headers: reports the
# sizes and field offsets those headers compile to
$(OUT)/libref_probe.so: ref_probe.c $(REF)/include/threadpool.h $(REF)/include/vector.h $(REF)/include/bitmap.h | $(OUT)
	$(CC) $(CFLAGS) -shared -o $@ ref_probe.c
endif

clean:
	rm -rf $(OUT)

.PHONY: all clean
