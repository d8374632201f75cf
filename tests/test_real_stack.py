"""The boundary checks against the REAL quimb / autoray / cotengra (scripts/verify_real_stack.py): run wherever the
three import, skipped -- with the script's own message -- everywhere else (this build's container and GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "verify_real_stack.py")], capture_output=True,
                         text=True, timeout=1800)
    if res.stdout.startswith("SKIPPED"):
        pytest.skip(res.stdout.strip().splitlines()[0])
    assert res.returncode == 0 and "REAL STACK OK" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]


def test_real_stack_on_the_interpreter():
    _run()


@pytest.mark.gpu
def test_real_stack_on_the_device():
    _run()
