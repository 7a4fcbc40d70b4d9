"""ISA-level guard (no GPU needed: hipcc cross-compiles): no s_barrier of the persistent one-pass gate can be reached with an LDS
write of the same wave still un-waited-for.  Round 6: the compiler left the wait out on one path of that kernel (an LDS store of
the next ticket, `continue`, barrier at the loop head) and the kernel lost hand-offs next to other streams' kernels; the wait is
explicit in the source now, and tools/audit_barrier_waits.py finds the hole again if it ever comes back."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "noisereduce_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

TU = """#include <hip/hip_runtime.h>
#include <cstdint>
%s
#include "kernels.hpp"
#include "thresh.hpp"
#include "fastpath.hpp"
#include "fused.hpp"
#include "onepass.hpp"
template __global__ void sg::fast::k_gate_onepass<4, false, false, false, true>(sg::fast::OnePassArgs);
template __global__ void sg::fast::k_gate_onepass<4, true, false, false, true>(sg::fast::OnePassArgs);
template __global__ void sg::fast::k_gate_onepass<4, false, false, false, false>(sg::fast::OnePassArgs);
"""


def _audit(tmp_path, defines):
    src = tmp_path / "tu.hip"
    src.write_text(TU % defines)
    asm = tmp_path / "tu.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", CSRC, "--cuda-device-only", "-S",
                    str(src), "-o", str(asm)], check=True, capture_output=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_barrier_waits.py"), str(asm)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_no_barrier_of_the_one_pass_gates_is_reached_with_an_unwaited_lds_write(tmp_path):
    out = _audit(tmp_path, "")
    assert "barriers flagged: 0" in out, out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_the_audit_finds_the_hole_without_the_explicit_wait(tmp_path):
    """(the tool's own check: -DOP_NO_LOOPTOP_WAIT=1 is the source as it was when the persistent gate lost hand-offs.  If a later
    compiler emits the wait by itself this finds 0 and the test is skipped -- the explicit wait stays either way)"""
    out = _audit(tmp_path, "#define OP_NO_LOOPTOP_WAIT 1")
    if "barriers flagged: 0" in out:
        pytest.skip("this compiler emits the wait itself")
    assert "k_gate_onepass" in out and "barriers flagged: 2" in out, out
