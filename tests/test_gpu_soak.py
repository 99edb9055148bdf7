"""-m gpu: short runs of the seed-fuzz soaks (tests/tools/soak*.py) on seeds of their own - the long runs are evidence
(profiles/r04z_soak*.txt), these keep the soaks themselves alive and add a few hundred unseen cases to every GPU tier run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _soak(script, *args, env=None):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", script)] + [str(a) for a in args], text=True, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, **(env or {})))
    tail = [l for l in p.stdout.splitlines() if l.startswith(("FAIL", "soak"))]
    assert p.returncode == 0 and tail and " 0 failures" in tail[-1], "\n".join(tail[-12:]) or p.stdout[-2000:]
    return tail[-1]


def test_whole_path_on_unseen_seeds():
    """sort bit-exact, frame vs the raster oracle, strips == full frame, culled sorts draw the same frame, both context kinds"""
    print(_soak("soak.py", 24, 31000, 80000))


@pytest.mark.parametrize("ballot", [False, True], ids=["lds_atomic_rank", "ballot_rank"])
def test_sort_modes_on_unseen_seeds(ballot):
    """precision 10-24, int / float, dynamic, precomputed, partial, permuted, duplicates, wrap-around - both ranking paths"""
    print(_soak("soak_sort.py", 120, 32000 + (500 if ballot else 0), 200000, env={"GSPLAT_NO_LDS_ATOMIC_RANK": "1"} if ballot else None))


def test_sort_at_the_kernels_size_boundaries():
    """tile (4096), chunk (3 tiles) and 2^24 boundaries of radix.hpp (the table's 25.2 M boundary is in the long run only)"""
    print(_soak("soak_sort.py", "sizes", "1,2,65,4095,4096,4097,12287,12288,12289,24577,16777215,16777216,16777217"))


def test_octree_rows_on_unseen_seeds():
    """device tree == host tree == oracle tree, gathers, asynchronous gather + fused sort (+ frustum cull), partial sorts"""
    print(_soak("soak_tree.py", 30, 33000, 100000))


def test_shader_permutations_on_unseen_seeds():
    print(_soak("soak_options.py", 40, 34000, 30000))


def test_a_long_lived_viewer_on_unseen_seeds():
    """one mesh + sorter + octree over random event sequences (re-uploads, culls, strips, asynchronous frames, overflow)"""
    print(_soak("soak_stateful.py", 4, 35000, 20, 40000))


@pytest.mark.parametrize("front", ["stream", "compact"])
def test_whole_path_with_either_front_end_of_the_culled_sort(front):
    """Round 5: the visibility-culled sort has a streaming front end (full frames) and the compact-then-gather one (strips); the
    soak forces each of them onto every culled sort it draws, strips and full frames alike."""
    print(_soak("soak.py", 10, 36000 + (700 if front == "compact" else 0), 60000, env={"GSPLAT_VIS_FRONT": front}))
    print(_soak("soak_stateful.py", 2, 37000 + (700 if front == "compact" else 0), 16, 30000, env={"GSPLAT_VIS_FRONT": front}))
