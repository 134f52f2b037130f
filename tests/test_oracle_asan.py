"""Sanitizer run of the CPU oracle ("-fsanitize=address,undefined on the CPU oracle"): `make -C oracle asan`
builds the same source with ASan + UBSan (no recovery), and a child interpreter with libasan preloaded runs the oracle's own CPU
tests against it -- RNG known answers, the math spec, a Monte-Carlo frame, gen_rays, train-ray generation with the ring buffer,
compositing, the metrics, and the NN arithmetic (encodings, forward, backward, optimizer; fused, generic and HashGrid models)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_address_and_undefined_behaviour_sanitizers():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.exists(libasan)
    env = dict(os.environ, NRC_ORACLE_ASAN="1", LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_rng.py"), os.path.join(ROOT, "tests", "asan_cases.py")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
    assert " passed" in out          # (asan_cases.py::test_sanitizer_build_is_loaded asserts that the child loaded the sanitizer build)
