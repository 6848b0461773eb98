"""Host C (scene builder, pointer vector, BMP writer, scene library) and the
CPU oracle under AddressSanitizer + UBSan, and the oracle's row pool under
ThreadSanitizer.  GPU-side sanitizers are not available on this pool, so the
host side is where they run (the reference has none and
several unchecked allocations / a racy pool)."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "path_tracer_c_amd" / "csrc" / "host"
SOURCES = [ROOT / "tests" / "c" / "host_selftest.c", HOST / "scene.c", HOST / "vector.c",
           HOST / "bitmap.c", HOST / "scenes.c", ROOT / "oracle" / "pt_oracle.c"]
COMMON = ["gcc", "-std=gnu2x", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off",
          f"-I{ROOT / 'include'}", f"-I{ROOT / 'oracle'}"]


def build_and_run(tmp_path, name, sanitize, env_extra):
    exe = tmp_path / name
    cmd = COMMON + [f"-fsanitize={sanitize}"] + [str(s) for s in SOURCES] + ["-o", str(exe), "-lm", "-lpthread"]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr and "not supported" in built.stderr:
        pytest.skip(f"-fsanitize={sanitize} not supported by this gcc")
    assert built.returncode == 0, built.stderr
    env = dict(os.environ, **env_extra)
    run = subprocess.run([str(exe), str(tmp_path / "t.bmp")], capture_output=True, text=True,
                         env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "host_selftest: ok" in run.stdout
    assert "runtime error" not in run.stderr and "Sanitizer" not in run.stderr, run.stderr


def test_host_c_and_oracle_under_asan_ubsan(tmp_path):
    build_and_run(tmp_path, "selftest_asan", "address,undefined",
                  {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0",
                   "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})


def test_oracle_row_pool_under_tsan(tmp_path):
    build_and_run(tmp_path, "selftest_tsan", "thread", {"TSAN_OPTIONS": "halt_on_error=1"})


@pytest.mark.parametrize("sanitize", ["thread", "address,undefined"])
def test_pool_shim_under_sanitizers(tmp_path, sanitize):
    """csrc/host/threadpool.c (the legacy pool: worker pthreads for ordinary
    tasks, rows handed to the GPU path at wait) with recording stand-ins for
    render_task and the GPU call, under TSan and under ASan + UBSan."""
    exe = tmp_path / "pool_selftest"
    cmd = ["gcc", "-std=gnu2x", "-O1", "-g", "-fno-omit-frame-pointer", f"-fsanitize={sanitize}",
           f"-I{ROOT / 'include'}", f"-I{HOST}", str(ROOT / "tests" / "c" / "pool_selftest.c"),
           str(HOST / "threadpool.c"), str(HOST / "vector.c"), "-o", str(exe), "-lpthread"]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr and "not supported" in built.stderr:
        pytest.skip(f"-fsanitize={sanitize} not supported by this gcc")
    assert built.returncode == 0, built.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1",
                                  UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "pool_selftest: ok" in run.stdout
    assert "runtime error" not in run.stderr and "Sanitizer" not in run.stderr, run.stderr


def test_bvh_builder_under_asan_ubsan(tmp_path):
    """csrc/bvh_build.hpp is host C++ inside a HIP translation unit; compiled
    alone with g++ it runs under the sanitizers.  The self-test checks the
    invariants the device walk relies on (every object exactly once, children
    inside parents, leaf sizes, depth) on benign and hostile inputs."""
    exe = tmp_path / "bvh_selftest"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer",
           "-fsanitize=address,undefined", f"-I{ROOT / 'path_tracer_c_amd' / 'csrc'}",
           f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c" / "bvh_selftest.cpp"), "-o", str(exe)]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr and "not supported" in built.stderr:
        pytest.skip("-fsanitize not supported by this g++")
    assert built.returncode == 0, built.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "bvh_selftest: ok" in run.stdout
    assert "runtime error" not in run.stderr and "Sanitizer" not in run.stderr, run.stderr
