"""The asset readers' seed-fuzz soak (tests/tools/soak_assets.py) on seeds of its own: host code, no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_asset_readers_on_unseen_seeds_and_damaged_files():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "soak_assets.py"), "80", "36000"], text=True, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    tail = [l for l in p.stdout.splitlines() if l.startswith(("FAIL", "soak"))]
    assert p.returncode == 0 and tail and " 0 failures" in tail[-1], "\n".join(tail[-12:]) or p.stdout[-2000:]
