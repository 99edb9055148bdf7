"""CPU tier: the C-ABI library loads and exports exactly what include/gsplat_hip.h declares; without a GPU the
product fails loudly (no CPU fallback)."""
import os
import re

import pytest

import gaussiansplats3d_amd as g
from gaussiansplats3d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "gsplat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = g.load()
    declared = header_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gsplat_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "ctypes table and header disagree"
    assert lib.gs_abi_version() == 5


def test_struct_layouts_match_the_header():
    import ctypes as C
    assert C.sizeof(_lib.SortStats) == 24
    assert C.sizeof(_lib.Camera) == 272
    assert C.sizeof(_lib.SceneParams) == 8 + 32 * (64 + 16 + 4 + 4 + 4 + 4)
    assert C.sizeof(_lib.GatherParams) == 160 and C.sizeof(_lib.TreeInfo) == 64
    assert C.sizeof(_lib.RenderStats) == 80
    assert C.sizeof(_lib.Destination) == 48 and _lib.Destination.width.offset == 32
    assert _lib.RenderStats.tile_entries.offset == 24


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the loud-failure path cannot be exercised")
    with pytest.raises(g.GsError) as e:
        g.Context(0)
    assert e.value.status == _lib.GS_ERR_HIP


def test_product_never_imports_the_oracle():
    # the product (package, Node seam, headers) and the measurement / profiling tools: the oracles are the tests' checkers only
    # (tests/, tests/tools/, __graft_entry__.smoke() and bench.py's cpu_baseline leg)
    for top in ("gaussiansplats3d_amd", "node", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            if "node_modules" in dirpath:
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".js", ".mjs", ".c", ".sh")):
                    src = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, os.path.join(top, f)


def test_graft_entry_checks_the_header_abi_version():
    """The driver's build() asserts the ABI version: it must follow include/gsplat_hip.h (round 6: the header went to 5 and the
    entry point still said 4 until the smoke run on a GPU box found it)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "gsplat_hip.h")).read()
    version = int(re.search(r"#define GS_ABI_VERSION (\d+)", header).group(1))
    entry = open(os.path.join(root, "__graft_entry__.py")).read()
    assert f"gs_abi_version() == {version}" in entry
    assert g.load().gs_abi_version() == version
