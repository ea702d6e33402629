"""SURVEY.md 5.2 / VERDICT round 2 item 8: the oracle is the parity authority -- 1 500 lines of index arithmetic -- so its golden
and known-answer suites are re-run against an AddressSanitizer + UndefinedBehaviorSanitizer build of the same source
(`make -C oracle asan`).  The sanitizer runtime has to be in the process before Python starts: a child pytest with libasan
preloaded loads libbez_oracle_*_asan.so through BEZ_ORACLE_VARIANT."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_suites_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), libasan
    env = dict(os.environ, BEZ_ORACLE_VARIANT="asan", LD_PRELOAD=libasan + ":" + libubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_golden.py", "tests/test_oracle_physics.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout
