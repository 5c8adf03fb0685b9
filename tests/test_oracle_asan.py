"""The CPU oracle under AddressSanitizer + UBSan (tools/oracle_asan.sh): the checker itself must be memory-clean on
every fixture the reference's tests hold.  CPU only."""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not shutil.which("gcc") or not os.path.isabs(asan):
        pytest.skip("no libasan in this toolchain")
    proc = subprocess.run(["bash", os.path.join(ROOT, "tools", "oracle_asan.sh")], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "clean" in proc.stdout and "ERROR: AddressSanitizer" not in proc.stderr
