"""Boundary contract against the reference's own caller: see tests/boundary_contract_worker.py, run here in a process of
its own (it imports the reference through tests/golden/ref_shim.py and replaces neuradar_amd.ops with the CPU oracle, neither
of which may leak into the other tests).  Skipped where /root/reference is absent (the GPU box)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/nerfstudio"), reason="needs the reference checkout (build container)")
def test_reference_model_runs_on_the_dropin_classes():
    r = subprocess.run([sys.executable, os.path.join(HERE, "boundary_contract_worker.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "boundary contract: OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
