"""CPU suite, part 3 - sanitizers (GPU AddressSanitizer is not available on this pool, so: CPU builds only).

1. The library's HOST half (csrc/api.hip: argument validation, workspace slots, upload tickets, frame ring; csrc/staging.hip: the
   page-locked staging ring and landing arena) compiled with g++ -fsanitize=address,undefined against a stand-in HIP layer
   (tests/hoststub/: device memory = host memory, kernels = do-little stand-ins) and driven through the product's ctypes
   signatures.  Also proves by execution that no asynchronous runtime copy ever touches pageable memory.
2. The CPU oracle (oracle/*.c) under the same sanitizers, running its own known-answer tests.

Both run in subprocesses with libasan preloaded into an ordinary python.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "hoststub")


def _san_env(**extra):
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no libasan.so")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="2")
    env.update(extra)
    return env


def test_host_half_of_the_library_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", STUB])
    env = _san_env(KARIOS_HIP_RING_CHUNK_KB="64", KARIOS_HIP_UPLOAD_CHECKSUM="1")
    out = subprocess.run([sys.executable, os.path.join(STUB, "driver.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "HOST-ASAN OK" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]
    assert "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-6000:]
    # the default ring geometry (4 x 4 MB) as well
    env = _san_env(KARIOS_HIP_RING_CHUNK_KB="4096")
    out = subprocess.run([sys.executable, os.path.join(STUB, "driver.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "HOST-ASAN OK" in out.stdout, out.stdout[-3000:] + out.stderr[-6000:]


def test_the_sanitizer_build_really_reports():
    """Negative control: a read-back larger than the caller's buffer must abort with an AddressSanitizer report."""
    subprocess.check_call(["make", "-s", "-C", STUB])
    code = f"""
import ctypes as C, numpy as np, sys
lib = C.CDLL({os.path.join(STUB, "_build", "libkarios_host_asan.so")!r})
lib.km_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
lib.km_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
lib.km_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
ctx, d = C.c_void_p(), C.c_void_p()
assert lib.km_ctx_create(0, C.byref(ctx)) == 0 and lib.km_dev_alloc(ctx, 4096, C.byref(d)) == 0
small = np.zeros(100, np.uint8)
lib.km_d2h(ctx, small.ctypes.data_as(C.c_void_p), d, 4096)
print("not detected")
"""
    out = subprocess.run([sys.executable, "-c", code], env=_san_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "AddressSanitizer" in out.stderr and "not detected" not in out.stdout, out.stderr[-3000:]


def test_oracle_known_answer_tests_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = _san_env(KARIOS_ORACLE_SO=os.path.join(ROOT, "oracle", "libkarios_oracle_asan.so"), KARIOS_ORACLE_THREADS="2")
    # the oracle's own known-answer / golden-vector tests (everything in test_oracle_golden.py that is not the slow config-1 pipeline)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider",
                          "-k", "not config1 and not full"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-6000:]
    assert "Sanitizer" not in out.stderr and "runtime error" not in out.stderr and "runtime error" not in out.stdout, out.stderr[-6000:]
