"""Host side of libcaphn under AddressSanitizer + UBSan ("ASAN host build"): tools/run_san.sh builds the
`san` target of csrc/Makefile and runs the torch-free ABI exercise (symbol table, workspace-size queries, argument validation --
no GPU call) against it.  GPU-side sanitizers are not available on the pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_host_side_abi_paths_are_clean_under_asan_and_ubsan():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "run_san.sh")], capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host-side validation paths ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
