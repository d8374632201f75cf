"""The real quimb running on ``quimb_amd`` arrays (drop-in boundary B1): see tests/golden/dropin_check.py.
Needs the reference sources, which exist in the build container only -- skipped elsewhere (the GPU box)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/quimb"), reason="reference sources not present")
def test_real_quimb_runs_on_quimb_amd_arrays():
    """quimb's own ``TensorNetwork.contract``, ``Tensor`` layout ops and ``@``, structured MPS contraction,
    ``contract_boundary`` (QR / truncated-SVD sweeps through the registered split drivers), ``Tensor.split`` and
    ``Circuit.amplitude`` with the data held in ``quimb_amd.Array`` objects -- a subprocess, so the stand-in
    modules for quimb's third-party imports never leak into this test session."""
    res = subprocess.run([sys.executable, os.path.join(HERE, "golden", "dropin_check.py")], capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0 and "DROPIN OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
