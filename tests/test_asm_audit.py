"""Static audit of the hand-pipelined LDS fragment reads (inline-asm ds_read_b128 with counted s_waitcnt lgkmcnt): in the
compiled ISA of every split-bf16 kernel no instruction may touch an asm load's destination before a sufficient wait, and
no control flow / scalar memory load may sit inside a counted window (tools/audit_asm_loads.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc (cross-compiles gfx950 without a GPU)")
def test_asm_fragment_reads_are_waited_for_before_any_use():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_loads.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "problems 0" in r.stdout.strip().splitlines()[-1]
