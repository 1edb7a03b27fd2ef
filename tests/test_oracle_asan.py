"""SURVEY.md 5 / VERDICT r3 item 9: the oracle is the checker, so an out-of-bounds read or an undefined shift inside it would silently
define the expected bytes.  `make -C oracle asan-test` builds rcv_oracle.c with -fsanitize=address,undefined and runs the oracle's CPU
tests (reference vectors, golden fixtures, the independent numpy / scipy implementations, bench.py's oracle legs) against that build in
a child process with the sanitizer runtimes preloaded; any report aborts the child."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    if os.environ.get("RCV_ORACLE_LIB"):
        pytest.skip("already running inside the sanitizer child")
    cc = shutil.which(os.environ.get("CC", "gcc"))
    asan = subprocess.run([cc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("this gcc has no libasan")
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan-test"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert " passed" in p.stdout and "failed" not in p.stdout
