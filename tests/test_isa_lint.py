"""Static check on the gfx950 assembly of the LDS-DMA kernels (no GPU needed, hipcc cross-compiles): their marches wait for
staged planes with hand-counted `s_waitcnt vmcnt(n)`, which is only valid while the compiler adds no vector-memory operation
of its own to those loops -- a register spill (scratch_load / scratch_store) is one.  scripts/isa_loop_check.py exits 1 if any
loop with packed FMAs in the lga_apply_pp* / lga_filter_grad_pp* kernels contains scratch traffic."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_spills_inside_the_hand_counted_vmcnt_loops():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_loop_check.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "lga_apply_pp<2, false, false>" in r.stdout and "lga_filter_grad_pp<2, 3, 0>" in r.stdout
    assert "UNSAFE" not in r.stdout
