"""tests/golden/verify_against_upstream.py is the script that pins the oracle to the real reference
once its source is available.  Here: its plumbing (name map, goldens, per-item checks) must hold
with the oracle standing in for upstream, and it must say so loudly when there is no source."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "golden", "verify_against_upstream.py")


def test_self_test_passes_every_item():
    r = subprocess.run([sys.executable, SCRIPT, "--self-test"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l[:2] in ("C.", "MO", "LO")]
    assert len(lines) == 11 and not any(" FAIL " in l for l in lines), r.stdout


def test_without_source_it_reports_unpinned():
    env = dict(os.environ, COPERCEPTION_SRC="")
    r = subprocess.run([sys.executable, SCRIPT], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "not available" in r.stdout
