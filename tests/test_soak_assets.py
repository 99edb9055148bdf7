"""The host-code soaks (asset readers, host octree builder: tests/tools/) on seeds of their own - no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_asset_readers_on_unseen_seeds_and_damaged_files():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "soak_assets.py"), "80", "36000"], text=True, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    tail = [l for l in p.stdout.splitlines() if l.startswith(("FAIL", "soak"))]
    assert p.returncode == 0 and tail and " 0 failures" in tail[-1], "\n".join(tail[-12:]) or p.stdout[-2000:]


def test_host_octree_builder_on_unseen_seeds():
    """tests/tools/soak_tree_host.py: the host builder against the restatement of the reference's worker, no GPU."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "soak_tree_host.py"), "16", "37000", "12000"], text=True,
                       timeout=900, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    tail = [l for l in p.stdout.splitlines() if l.startswith(("FAIL", "soak"))]
    assert p.returncode == 0 and tail and " 0 failures" in tail[-1], "\n".join(tail[-12:]) or p.stdout[-2000:]
