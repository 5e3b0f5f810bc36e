"""The host side of the C ABI under AddressSanitizer + UndefinedBehaviorSanitizer ("Race detection /
sanitizers"): `make -C language-transformer-vae_amd/csrc asan` builds liblvae_asan.so with the HOST pass of every file
instrumented (device code untouched: GPU sanitizers are not available on this pool), and the host-only C-ABI tests of
tests/test_cabi_host.py -- argument validation, workspace queries, the 4,000-shape layout fuzz, the split plans -- run
against it in a child process with the sanitizer runtime preloaded.  No GPU is touched."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "language-transformer-vae_amd", "csrc")
ASAN_LIB = os.path.join(ROOT, "language-transformer-vae_amd", "liblvae_asan.so")


def _runtime():
    found = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not found:
        pytest.skip("the clang AddressSanitizer runtime is not in this image")
    return found[-1]


def _env(rt):
    env = dict(os.environ, LVAE_LIB=ASAN_LIB, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env["LD_PRELOAD"] = rt + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    return env


def test_host_side_of_the_c_abi_is_clean_under_asan_and_ubsan():
    rt = _runtime()
    if not os.path.exists(ASAN_LIB) or os.path.getmtime(ASAN_LIB) < os.path.getmtime(os.path.join(CSRC, "lvae_api.hip")):
        subprocess.run(["make", "-C", CSRC, "asan", "-j", str(min(8, os.cpu_count() or 1))], check=True,
                       capture_output=True, timeout=1500)
    env = _env(rt)
    # the child must really be on the instrumented library
    probe = subprocess.run([sys.executable, "-c",
                            "import sys; sys.path.insert(0, %r)\n"
                            "from lvae import _native as N\n"
                            "assert N.LIB_PATH.endswith('liblvae_asan.so'), N.LIB_PATH\n"
                            "N.lib(); maps = open('/proc/self/maps').read()\n"
                            "assert 'liblvae_asan.so' in maps and 'libclang_rt.asan' in maps\n" % ROOT],
                           env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0, probe.stderr[-2000:]
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_cabi_host.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in p.stdout
