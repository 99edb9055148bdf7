"""-m gpu: `python bench.py --gpus 2` as the driver runs it (no launcher: bench.py starts the ranks itself).  On a 1-GPU box
the two ranks share the device and exchange their strips over gloo; the gathered frame must equal the N = 1 frame."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, gpus, config="C3"):
    dump = str(tmp_path / f"frame_{gpus}_{config}.npy")
    env = dict(os.environ, GS_BENCH_DUMP=dump)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3",
                                   "--warmup", "1", "--no-cpu", "--no-cull", "--splats", "300000", "--median-frames", "5"] +
                                  (["--config", config] if config else []), env=env, text=True, timeout=600)
    line = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, out
    return json.loads(line[0]), np.load(dump)


def test_two_ranks_gather_the_single_gpu_frame(tmp_path):
    one, frame1 = _run(tmp_path, 1)
    two, frame2 = _run(tmp_path, 2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["parallelism"] == "tile-row strips x2"
    assert len(two["config"]["strips"]) == 2 and two["config"]["strips"][0][1] == two["config"]["strips"][1][0]
    assert frame1.shape == frame2.shape == (1080, 1920, 4) and frame1.any()
    np.testing.assert_array_equal(frame1, frame2)
    assert two["value"] > 0 and two["ms_per_step"] > 0


def test_a_multi_gpu_run_keeps_the_metric_configuration_and_adds_the_8k_one(tmp_path):
    """`bench.py --gpus N` as the driver runs it (no --config): BASELINE.json's metric configuration (C3, 1920x1080) stays the
    headline - the driver computes the scaling from the per-N values, so they must be the same workload - with rank 0 alone as
    the line's own N = 1 reference; configs[4] (garden at 7680x4320) rides along on the same ranks with ITS one-GPU reference.
    And the 8K configuration asked for by name: the gathered 8K frame equals the one-GPU 8K frame."""
    two, frame2 = _run(tmp_path, 2, config=None)
    assert two["n_gpus"] == 2 and two["config"]["width"] == 1920 and two["config"]["workload"].startswith("C3")
    assert two["metric"].startswith("Msplats/s sorted+rasterized at 1920x1080")
    assert frame2.shape == (1080, 1920, 4) and frame2.any()
    assert two["same_config_1gpu"]["n_gpus"] == 1 and two["same_config_1gpu"]["width"] == 1920
    assert two["speedup_vs_same_config_1gpu"] is not None and two["median_ms_per_step"] > 0 and two["median_frames"] == 5
    assert two["c5"]["width"] == 7680 and two["c5"]["n_gpus"] == 2 and two["c5"]["ms_per_step"] > 0
    assert two["c5_1gpu"]["width"] == 7680 and two["c5_1gpu"]["n_gpus"] == 1
    # (two ranks sharing one GPU and gathering an 8K frame over gloo: the ratio itself means nothing here, only that it is there)
    assert two["c5_speedup_vs_1gpu"] is not None
    two8, frame8_2 = _run(tmp_path, 2, config="C5")
    one8, frame8_1 = _run(tmp_path, 1, config="C5")
    assert two8["config"]["width"] == 7680 and two8["c5"] is None
    assert frame8_1.shape == frame8_2.shape == (4320, 7680, 4) and frame8_1.any()
    np.testing.assert_array_equal(frame8_1, frame8_2)
