"""CPU tier: the arithmetic behind bench.py's JSON line (SURVEY.md 8d formulas), the PMC look-up and the CPU baseline leg."""
import importlib.util
import sys
import time
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_frame_formula_matches_the_survey():
    # B = 56 Rs + (84 + S) R + 40 D + 4 P, S = 48 for SH-2 (C3 numbers)
    R = Rs = 5_800_000
    D, P = 14_551_988, 1920 * 1080
    assert bench.frame_algorithmic_bytes(R, Rs, D, P, 2, False) == 56 * Rs + (84 + 48) * R + 40 * D + 4 * P == 1_680_773_920
    # fp16 covariances (C4) and a third radix pass (precision 20)
    assert bench.frame_algorithmic_bytes(10, 10, 0, 0, 0, True) == 56 * 10 + 72 * 10
    assert bench.frame_algorithmic_bytes(10, 10, 0, 0, 0, False, precision=20) == (56 + 16) * 10 + 84 * 10


def test_project_bytes():
    assert bench.project_algorithmic_bytes(5_800_000, 1_444_147, 2, False) == 12 * 5_800_000 + (24 + 4 + 48 + 40) * 1_444_147 + 725_000
    assert bench.project_algorithmic_bytes(1000, 0, 0, True) == 12_000 + 125


def test_pmc_lookup_reads_the_committed_summary():
    traffic, src = bench.pmc_traffic("k_project")
    assert src is not None and src.endswith("pmc_traffic.json")
    d = json.load(open(os.path.join(ROOT, "profiles", src)))
    name = next(k for k in d["kernels"] if k.split("<")[0] == "k_project")
    assert traffic == d["kernels"][name]["hbm_bytes_per_launch"] > 100e6
    assert bench.pmc_traffic("no_such_kernel")[0] is None
    # a summary recorded on one config is never quoted for another
    d4, _ = bench._newest_profile("*pmc_traffic.json", "no-such-config")
    assert d4 is None


def test_cpu_baseline_leg_times_the_reference_sorter():
    from gaussiansplats3d_amd import camera, scenes
    scene = scenes.scene_like(20000, 0, 5, name="t")
    cam = camera.demo_camera("garden", 320, 180)
    r = bench.cpu_baseline(scene, cam.sort_mvp(), 0.2)
    assert r["cores"] == 1 and r["kind"] in ("reference", "port") and r["value"] > 1.0 and r["ms_per_sort"] > 0
    assert "sorts" in r["sample"]
    if r.get("wasm"):                      # the reference's prebuilt WASM sorter under Node, timed beside the native build
        assert r["wasm"]["kind"] == "reference" and r["wasm"]["ms_per_sort"] > 0 and r["wasm"]["cores"] == 1


def test_self_spawn_sets_up_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks with the torch.distributed environment."""
    started = []

    class P:
        pid = -1

        def __init__(self, cmd, env):
            started.append((cmd, env))

        def poll(self):
            return 0

        def wait(self, timeout=None):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", lambda cmd, env=None, **kw: P(cmd, env))
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.spawn_ranks(4) == 0
    assert [e["RANK"] for _, e in started] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["LOCAL_RANK"] == e["RANK"] for _, e in started)
    assert len({e["MASTER_PORT"] for _, e in started}) == 1
    assert all(c[-4:] == ["--gpus", "4", "--steps", "2"] for c, _ in started)


def test_a_failing_rank_takes_the_others_down(monkeypatch, tmp_path):
    """One rank exits non-zero while the others would run for ever (a peer stuck in ncclCommInitRank): spawn_ranks stops
    them all, waits for them and returns the failing rank's code; a deadline does the same with 124."""
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
                      "time.sleep(600)\n")
    procs = []
    real_popen = bench.subprocess.Popen

    def popen(cmd, env=None, **kw):
        p = real_popen([sys.executable, str(script)], env=env, **kw)
        procs.append(p)
        return p

    monkeypatch.setattr(bench.subprocess, "Popen", popen)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "3"])
    t0 = time.monotonic()
    assert bench.spawn_ranks(3, poll_s=0.05) == 7
    assert time.monotonic() - t0 < 30 and len(procs) == 3 and all(p.poll() is not None for p in procs)
    # the deadline
    script.write_text("import time\ntime.sleep(600)\n")
    procs.clear()
    assert bench.spawn_ranks(2, timeout_s=1.0, poll_s=0.05) == 124
    assert all(p.poll() is not None for p in procs)


def test_gather_floor_is_the_largest_peer_strip_over_one_link(monkeypatch):
    """bench --gpus N prints what the strip gather alone allows beside each speed-up.  C5 (7680 x 4320) on
    eight ranks: seven peers send ~1/8 of a 133 MB frame each; at 76.8 GB/s per link and direction the largest strip is the floor."""
    monkeypatch.delenv("GS_LINK_GBPS", raising=False)
    rows = 4320 // 16                                    # 270 tile rows
    strips = [(k * rows // 8, (k + 1) * rows // 8) for k in range(8)]
    g = bench.gather_floor(strips, 7680, 4320, 8, 0.845)
    largest = max(min(r1 * 16, 4320) - r0 * 16 for r0, r1 in strips[1:]) * 7680 * 4
    assert g["largest_strip_bytes"] == largest and g["links"] == 7
    assert abs(g["gather_floor_ms"] - largest / 76.8e9 * 1e3) < 1e-3
    assert g["bytes_into_rank0"] == sum(min(r1 * 16, 4320) - r0 * 16 for r0, r1 in strips[1:]) * 7680 * 4
    assert g["speedup_ceiling_from_gather"] == round(0.845 / g["gather_floor_ms"], 2)
    monkeypatch.setenv("GS_LINK_GBPS", "38.4")
    assert abs(bench.gather_floor(strips, 7680, 4320, 8, None)["gather_floor_ms"] - 2 * g["gather_floor_ms"]) < 2e-4
    assert bench.gather_floor(strips, 7680, 4320, 1, 1.0) is None and bench.gather_floor(None, 7680, 4320, 8, 1.0) is None
