"""The oracle suites once more against an AddressSanitizer + UBSan build of oracle/icp_oracle.c (tests/tools/oracle_sanitize.sh;
SURVEY.md section 5: sanitizer builds -- the reference has none).  A restatement that reads past an array would otherwise pass as parity."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_suites_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    out = subprocess.run(["bash", os.path.join(ROOT, "tests", "tools", "oracle_sanitize.sh")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert " passed" in out.stdout and "failed" not in out.stdout
