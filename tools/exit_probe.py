"""Which teardown order lets a python process that used libgsplat_hip exit cleanly? (diagnostic)"""
import subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {
    "plain": "import gaussiansplats3d_amd as g\nc=g.Context(0)\nc.close()\n",
    "no_close": "import gaussiansplats3d_amd as g\nc=g.Context(0)\n",
    "torch_first": "import torch\nimport gaussiansplats3d_amd as g\nc=g.Context(0)\nc.close()\n",
    "load_only": "import gaussiansplats3d_amd as g\ng.load()\n",
    "count_only": "import gaussiansplats3d_amd as g\nprint(g.load().gs_device_count())\n",
}
for name, code in VARIANTS.items():
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], capture_output=True, text=True)
    print(name, "rc=", r.returncode, (r.stderr.strip().splitlines() or [""])[-1][:200])
