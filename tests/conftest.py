import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (it is git-ignored): build it once, as __graft_entry__.build() does
    so = os.path.join(ROOT, "quimb_amd", "libquimb_amd.so")
    if not os.path.exists(so) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess

        subprocess.run(["make", "-s", "-j4", "-C", os.path.join(ROOT, "quimb_amd", "csrc")], check=False)


def pytest_sessionstart(session):
    """Host-memory watchdog.  A GPU box is a container with a memory limit (300 GiB on the round-4 pool): a checker
    that runs away on the host -- rounds 3 / 4: an oracle call on a path with a 627 GB intermediate -- gets the whole
    container OOM-killed, which the harness can only report as "the GPU box was lost".  A daemon thread samples this
    process's resident set and ends the run with a message and exit code 99 long before that
    (``QAMD_TEST_MAX_RSS_GB``, default 64; 0 disables)."""
    import threading
    import time

    limit_gb = float(os.environ.get("QAMD_TEST_MAX_RSS_GB", "64"))
    if limit_gb <= 0:
        return
    try:
        import psutil
    except ImportError:
        return
    proc = psutil.Process()

    def watch():
        while True:
            try:
                rss = proc.memory_info().rss
            except Exception:
                return
            if rss > limit_gb * 2**30:
                sys.stderr.write(f"\n[conftest] host memory watchdog: resident set {rss / 2**30:.1f} GiB > "
                                 f"{limit_gb:.0f} GiB -- aborting the test run (QAMD_TEST_MAX_RSS_GB)\n")
                sys.stderr.flush()
                os._exit(99)
            time.sleep(0.2)

    threading.Thread(target=watch, name="qamd-rss-watchdog", daemon=True).start()


def pytest_collection_modifyitems(config, items):
    """No test of this suite may hang a run: with pytest-timeout present (it is in this image) every test gets a
    15-minute ceiling unless the command line set its own -- a GPU test stuck in a device call ends the run with a
    failure instead of occupying the box until an outer limit kills it."""
    if not config.pluginmanager.hasplugin("timeout") or getattr(config.option, "timeout", None):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900, method="thread"))


@pytest.fixture
def emu():
    """Install the numpy plan interpreter as the default device (CPU tests)."""
    import quimb_amd.device as qd
    from emu_device import EmuDevice

    old = qd._DEFAULT
    dev = EmuDevice()
    qd.set_default_device(dev)
    try:
        yield dev
    finally:
        qd.set_default_device(old)


@pytest.fixture
def hip():
    """The real device; fails loudly if the HIP library / GPU is missing."""
    import quimb_amd.device as qd

    old = qd._DEFAULT
    dev = qd.HipDevice()
    qd.set_default_device(dev)
    try:
        yield dev
    finally:
        qd.set_default_device(old)


def pytest_terminal_summary(terminalreporter):
    """The achieved errors next to the bars: the worst max-norm relative error ``checks.assert_close`` saw per dtype."""
    try:
        import checks
    except Exception:
        return
    if checks.WORST:
        terminalreporter.write_line("achieved parity (worst relative error seen, bar in tests/checks.py:RTOL):")
        for k, (err, where) in sorted(checks.WORST.items()):
            terminalreporter.write_line(f"  {k:>12}: {err:.3e}  ({where})")
    if getattr(checks, "WORST_ELEM", None):
        terminalreporter.write_line(f"element-wise (worst |got - want| / (|want| + {checks.ELEM_FLOOR} max|want|) seen, bar "
                                    f"{checks.ELEM_FACTOR:g} x RTOL):")
        for k, (err, where) in sorted(checks.WORST_ELEM.items()):
            terminalreporter.write_line(f"  {k:>12}: {err:.3e}  ({where})")
