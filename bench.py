#!/usr/bin/env python
"""bench.py — headline benchmark: Msplats/s sorted + rasterized (BASELINE.json metric).

One "step" = one frame of the hot path on the garden.ply stand-in (configs[2]: 5.8 M splats, SH-2, 1920x1080):
depth-key + device-wide stable radix sort of ALL splats (cull off, R = N, the reference's own no-tree path,
src/Viewer.js:2061-2073), then project -> bin -> entry sort -> blend into an RGBA8 framebuffer.  Inputs are resident in
HBM before the timed region.

`value` is the §8d metric: R / (t_sort + t_raster) with the whole frame on ONE stream (a single-stream context:
sort -> draw, nothing overlaps).  The engine's default shape — the sort on a stream of its own, like the reference's Web
Worker, so that the sort of frame k+1 overlaps the tail of frame k's draw — is reported next to it as `pipelined`.

--gpus N: every rank holds the scene, rasterises a strip of tile rows (and sorts only what can reach it); strips are
gathered to rank 0 over RCCL inside the timed region (strong scaling).  Launched by torch.distributed.run, or — when
WORLD_SIZE is not set — bench.py starts the N ranks itself.

--config C1: configs[0], the reference's own CPU-runnable case: the bonsai stand-in through createSortWorker under Node,
the reference's WASM sorter on the host beside the HIP sorter, outputs compared bit for bit.

Prints ONE JSON line on rank 0 (see the task contract) with `roofline`, `blend`, `frame` and `cpu_baseline` objects.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s is the measured copy ceiling)
FP32_PEAK_TFLOPS = 157.3      # MI355X fp32 vector peak (same guide)
SH_BYTES = {0: 0, 1: 18, 2: 48}
# fp32 operations the blend spends per (pixel, walked splat): dy, two projected offsets (2 fma), power (mul + fma), exp2,
# discard mask (fma), alpha (2 mul), weight, three colour fma, transmittance update (sub, fma, mul); fma = 2
BLEND_FLOPS_PER_PIXEL_SPLAT = 24


def frame_algorithmic_bytes(R, Rs, D16, P, sh_degree, cov_half, precision=16):
    """B = 56*Rs + (84+S)*R + 40*D + 4*P with D = 16x16-px tiles touched (72+S instead of 84+S for
    fp16 covariances; +16*Rs per radix pass beyond two)."""
    sort = 56 + 16 * max(0, (precision + 7) // 8 - 2)
    proj = (72 if cov_half else 84) + SH_BYTES[sh_degree]
    return sort * Rs + proj * R + 40 * D16 + 4 * P


def project_algorithmic_bytes(N, visible, sh_degree, cov_half):
    """k_project (the largest HBM-bound kernel), bytes it HAS to move per launch: every splat's centre (12 B) is read to
    decide visibility; covariance 24 (12 as fp16) + rgba 4 + SH S are needed only for the splats that survive, each of
    which writes a 32-byte record + 8-byte tile rect; 1 mask bit per splat.  Bytes fetched for culled
    splats that share a wave with a survivor are waste, not algorithmic bytes."""
    per_visible = (12 if cov_half else 24) + 4 + SH_BYTES[sh_degree] + 40
    return 12 * N + per_visible * visible + N // 8


def _newest_profile(pattern, config):
    """Newest committed profiles/<pattern> summary that was recorded on `config` (files carry a "config" key; older
    files without one were all recorded on C3)."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("config", "C3") == config:
            return d, os.path.basename(path)
    return None, None


def pmc_traffic(kernel, config="C3"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary of THIS config
    (profiles/*pmc_traffic.json, written by tools/pmc_traffic.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes);
    (None, None) when no summary of this config is committed."""
    d, src = _newest_profile("*pmc_traffic.json", config)
    if d is None:
        return None, None
    k = d["kernels"].get(kernel)
    if k is None:                                           # template instances: "k_project<false>"
        k = next((v for name, v in sorted(d["kernels"].items()) if name.split("<")[0] == kernel), None)
    return (int(k["hbm_bytes_per_launch"]) if k else None), src


def pmc_frame_traffic(config="C3"):
    """Counter-measured HBM bytes of one whole frame (sum over the frame's kernels, tools/pmc_traffic.py), or None."""
    d, src = _newest_profile("*pmc_traffic.json", config)
    if d is None or "frame_hbm_bytes" not in d:
        return None, None
    return int(d["frame_hbm_bytes"]), src


SORT_KERNELS = ("k_depth_key", "k_minmax_count", "k_mask_compact", "k_radix_hist<DepthLoader", "k_radix_hist<ArrayLoader<unsigned int>",
                "k_radix_hist<ArrayLoader<unsigned short>", "k_radix_scatter_chunk", "k_radix_scatter<DepthLoader", "k_radix_scatter<PackedLoader")


def sort_profile(config="C3"):
    """The depth sort's kernels in the newest committed profiles of THIS config: per-kernel microseconds per frame from the
    rocprofv3 kernel table (profiles/*_kstats.json, tools/kstats.py) and counter bytes per frame (profiles/*pmc_traffic.json).
    The tile-entry sort's 16-bit-key kernels are the draw's, not the sort's; the 32-bit-key instances run at upload only."""
    out = {"per_kernel_us": None, "kernel_us_sum": None, "kernels_source": None, "counter_bytes": None, "counter_source": None}
    d, src = _newest_profile("*_kstats.json", config)
    if d is not None:
        ks = {n: v["us_per_frame"] for n, v in d["kernels"].items()
              if n.startswith(SORT_KERNELS) and "unsigned short" not in n and not (n.startswith("k_radix_scatter<ArrayLoader"))}
        out["per_kernel_us"], out["kernel_us_sum"], out["kernels_source"] = ks, round(sum(ks.values()), 2), src
    d, src = _newest_profile("*pmc_traffic.json", config)
    if d is not None and "frame_kernels_bytes_per_frame" in d:
        b = sum(v for n, v in d["frame_kernels_bytes_per_frame"].items()
                if n.startswith(SORT_KERNELS) and "unsigned short" not in n)
        out["counter_bytes"], out["counter_source"] = int(b), src
    return out


def pmc_valu(kernel, config="C3"):
    """VALU-busy fraction of `kernel` from the newest committed SQ counter pass (profiles/*pmc_valu.json)."""
    d, src = _newest_profile("*pmc_valu.json", config)
    if d is None:
        return None, None
    k = next((v for name, v in sorted(d["kernels"].items()) if name.split("<")[0] == kernel), None)
    return k, src


def cpu_baseline(scene, mvp, budget_s):
    """The reference's own sorter timed on this host, 1 thread (one Web Worker in the reference): (1) its prebuilt
    sorter_no_simd_non_shared.wasm under Node, timed as src/worker/SortWorker.js:53-60 runs it, and (2)
    sorter_no_simd.cpp compiled natively -O2 (oracle/_ref).  The reference has no CPU rasteriser, so the baseline covers
    the sort half of the frame only."""
    import oracle
    from gaussiansplats3d_amd import util
    n = scene.count
    ci = util.integer_centers(scene.centers)
    idx = np.arange(n, dtype=np.uint32)
    kind = "reference" if oracle.have_ref() else "port"
    fn = oracle.ref_sort_indexes if kind == "reference" else oracle.sort_indexes
    fn(idx[:1000], ci, mvp)                                  # warm the library
    t_total, reps = 0.0, 0
    while t_total < budget_s and reps < 2000:
        t0 = time.perf_counter()
        fn(idx, ci, mvp)
        t_total += time.perf_counter() - t0
        reps += 1
    per = t_total / reps
    out = {"value": round(n / per / 1e6, 2), "unit": "Msplats/s (sort only)", "cores": 1, "kind": kind,
           "ms_per_sort": round(per * 1e3, 2), "host_cpus": os.cpu_count(),
           "sample": f"{reps} full sorts ({t_total:.1f} s) of the same {n} splats / same MVP, precision 16, integer "
                     "static path, sorter_no_simd.cpp built -O2; the reference has no CPU rasteriser, so raster has "
                     "no CPU leg"}
    wasm = oracle.wasm_sort_timing(idx, ci, mvp, repeat=max(3, min(40, int(budget_s / max(per * 1.6, 1e-3)))))
    if wasm is not None:
        out["wasm"] = {"value": round(n / (wasm["ms_mean"] * 1e-3) / 1e6, 2), "unit": "Msplats/s (sort only)",
                       "ms_per_sort": round(wasm["ms_mean"], 2), "ms_per_sort_min": round(wasm["ms_min"], 2), "cores": 1,
                       "kind": "reference", "node": wasm["node"],
                       "sample": f"{wasm['repeat']} sorts of the same input by the reference's prebuilt "
                                 "sorter_no_simd_non_shared.wasm under Node, frequencies zeroed and process.hrtime "
                                 "around exports.sortIndexes as src/worker/SortWorker.js:53-60 does"}
    return out


# ------------------------------------------------------------------------------------------------------------------
# launching
# ------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, timeout_s=None, poll_s=0.2):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, each the leader of a process
    group of its own) and wait for ALL of them.  The first rank that fails - or the deadline ($GS_BENCH_RANK_TIMEOUT, default
    1500 s: a rank that died inside ncclCommInitRank leaves the others waiting for ever) - takes the others down with it, so no
    orphan keeps a GPU; returns the first non-zero exit code (124 on a timeout)."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    if timeout_s is None:
        timeout_s = float(os.environ.get("GS_BENCH_RANK_TIMEOUT", "1500"))
    procs = []
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      start_new_session=True))

    def stop(p, sig):
        try:
            os.killpg(p.pid, sig)                         # the rank and anything it started
        except (ProcessLookupError, PermissionError, AttributeError):
            try:
                p.send_signal(sig)
            except Exception:
                pass

    import signal
    deadline = time.monotonic() + timeout_s
    rc, codes = 0, [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        failed = [c for c in codes if c not in (None, 0)]
        timed_out = time.monotonic() > deadline
        if failed or timed_out:
            rc = failed[0] if failed else 124
            live = [p for i, p in enumerate(procs) if codes[i] is None]
            for p in live:
                stop(p, signal.SIGTERM)
            t_kill = time.monotonic() + 5.0
            while live and time.monotonic() < t_kill:
                live = [p for p in live if p.poll() is None]
                time.sleep(poll_s)
            for p in live:
                stop(p, signal.SIGKILL)
            for p in procs:
                try:
                    p.wait(timeout=10)
                except Exception:
                    pass
            print(f"bench.py: {'a rank failed' if failed else 'timeout'} (exit codes {codes}); the other ranks were stopped",
                  file=sys.stderr)
            return rc
        time.sleep(poll_s)
    return rc


class Watchdog:
    """Ends THIS rank if a collective set-up does not return (ncclCommInitRank blocks until every rank has called it: a peer
    that died before it would hang the rest until the driver's own limit)."""

    def __init__(self, seconds, what):
        import threading
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.what, self.seconds = what, seconds

    def _fire(self):
        print(f"bench.py: {self.what} did not finish within {self.seconds:.0f} s - giving up", file=sys.stderr, flush=True)
        os._exit(86)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=["C1", "C2", "C3", "C4", "C5", "C3T", "C3S"],
                    help="default: C3 (BASELINE.json's metric configuration) at every N; with --gpus N > 1 a second object `c5` "
                         "(BASELINE.json configs[4]: the 8-GPU configuration, garden at 7680x4320) on the same ranks")
    ap.add_argument("--splats", type=int, default=0, help="override the splat count (debug only; invalid as a result)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cull", action="store_true",
                    help="skip the secondary columns (orbit, octree cull, fused frustum cull, translucent / capture-like scenes)")
    ap.add_argument("--only-headline", action="store_true",
                    help="probe + warm-up + the timed frames and nothing else (profiler runs: every frame is a headline frame)")
    ap.add_argument("--median-frames", type=int, default=50,
                    help="frames of the hipEvent-bracketed median (SURVEY.md 8d: >= 50); 0 = skip")
    return ap.parse_args()


class Rig:
    """Scene resident on one context: sort worker + mesh + the frame function."""

    def __init__(self, ctx, scene, cam, device, torch):
        from gaussiansplats3d_amd import SplatMesh, create_sort_worker, util
        self.ctx, self.scene, self.cam, self.torch = ctx, scene, cam, torch
        self.N = scene.count
        self.mvp = cam.sort_mvp()
        self.worker = create_sort_worker(ctx, self.N)             # integerBasedSort, precision 16: Viewer defaults
        self.worker.post_message({"centers": util.integer_centers(scene.centers),
                                  "range": {"from": 0, "to": self.N - 1, "count": self.N}})
        self.mesh = SplatMesh(ctx, self.N, scene.sh_degree, scene.cov_half)
        self.mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
        self.mesh.set_camera(cam)
        self.frames = 0

    def set_view(self, cam):
        self.cam, self.mvp = cam, cam.sort_mvp()
        self.mesh.set_camera(cam)

    def probe(self, out_ptr):
        """Untimed full-frame draws: grow the entry buffer if needed, settle the list-bin size, return the statistics."""
        st = None
        for _ in range(2):
            self.worker.sort_on_device(self.mvp, self.N)
            self.mesh.use_sorter_result(self.worker, self.N)
            _, st = self.mesh.render(out_device_ptr=out_ptr, to_host=False, want_stats=True)
            self.frames += 1
        return st

    def frame(self, out_ptr, tile_rows=None):
        if self.worker.visibility_cull:            # vertex stage first: its mask (this rank's strip) is the sort's keep test
            self.mesh.project(tile_rows)
        self.worker.sort_on_device(self.mvp, self.N)
        self.mesh.render(tile_rows=tile_rows, out_device_ptr=out_ptr, to_host=False, want_stats=False)
        self.frames += 1

    def timed_sort_stats(self, out_ptr):
        """One more frame with the stage events on (they are off in timed regions): the sort's device time and result size."""
        self.ctx.set_stage_timing(True)
        self.frame(out_ptr)
        self.torch.cuda.synchronize()
        st, _ = self.worker.last_stats()
        self.ctx.set_stage_timing(False)
        return st

    def timed(self, steps, out_ptr, tile_rows=None, after=None):
        """out_ptr: a device pointer, or a callable that names the target of the next frame (alternating strip buffers of an
        overlapped multi-GPU gather); after: called behind every frame (the gather)."""
        torch = self.torch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.frame(out_ptr() if callable(out_ptr) else out_ptr, tile_rows)
            if after:
                after()
        enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        return time.perf_counter() - t0, enq

    def event_frames(self, frames, stream, out_ptr, tile_rows=None, after=None):
        """SURVEY.md 8d's form of the metric: every frame bracketed by a pair of HIP events on the frame's own stream (the
        context was created on `stream`, so torch's events see its kernels); returns the per-frame device milliseconds.
        The two event records per frame are barrier packets (~3 us each), which is why `value` is taken from the
        unbracketed region and this median is reported beside it."""
        torch = self.torch
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(frames)]
        torch.cuda.synchronize()
        for a, b in evs:
            a.record(stream)
            self.frame(out_ptr() if callable(out_ptr) else out_ptr, tile_rows)
            if after:
                after()
            b.record(stream)
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    def close(self):
        self.worker.terminate()
        self.mesh.dispose()


def bench_c1(args):
    """configs[0]: bonsai stand-in (1.2 M splats, SH-0), sort only: the reference's WASM sorter on the host and the HIP
    sorter behind the same createSortWorker message protocol, both driven from Node, outputs compared bit for bit."""
    from gaussiansplats3d_amd import camera, scenes, util
    cfg = scenes.CONFIGS["C1"]
    scene = scenes.make_config_scene("C1", args.splats or None)
    cam = camera.demo_camera(cfg["pose"], cfg["width"], cfg["height"])
    n = scene.count
    ci = util.integer_centers(scene.centers)
    mvp = np.asarray(cam.sort_mvp(), dtype=np.float64)
    import oracle
    wasm = oracle.wasm_path()
    with tempfile.TemporaryDirectory() as d:
        ci.tofile(os.path.join(d, "centers.bin"))
        mvp.tofile(os.path.join(d, "mvp.bin"))
        cmd = ["node", os.path.join(ROOT, "node", "bench_c1.js"), os.path.join(d, "centers.bin"), os.path.join(d, "mvp.bin"),
               str(n), str(args.steps), str(args.warmup), wasm or "-"]
        line = subprocess.check_output(cmd, cwd=os.path.join(ROOT, "node"), text=True).strip().splitlines()[-1]
    r = json.loads(line)
    out = {"metric": "Msplats/s sorted (configs[0]: bonsai stand-in, integer sort through createSortWorker)",
           "value": round(n / (r["hip_ms"] * 1e-3) / 1e6, 2), "unit": "Msplats/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(r["hip_ms"], 4), "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "int32 keys", "data": "synthetic" if scene.name == "C1" else f"file:{scene.name}",
           "config": {"workload": f"C1: {cfg['label']}, sort only (the reference's CPU-runnable configuration)",
                      "splats": n, "sort_precision_bits": 16, "path": "node/gsplat.js createSortWorker -> N-API -> gs_sorter_sort",
                      "includes": "PCIe both ways: the sorted indexes return to a JS Uint32Array like the reference's sortDone"},
           "device_sort_ms": round(r["hip_device_ms"], 4),
           "cpu_baseline": None if r.get("wasm_ms") is None else
           {"value": round(n / (r["wasm_ms"] * 1e-3) / 1e6, 2), "unit": "Msplats/s (sort only)", "cores": 1, "kind": "reference",
            "ms_per_sort": round(r["wasm_ms"], 3), "host_cpus": os.cpu_count(),
            "sample": f"{r['wasm_reps']} sorts by the reference's sorter_no_simd_non_shared.wasm in the same Node process"},
           "identical_to_reference": r.get("identical")}
    print(json.dumps(out), flush=True)
    return 0 if r.get("identical") in (True, None) else 1


class Env:
    """What every measurement of this process shares: torch, the process group, one context on one stream, one rig."""
    pass


SETTLE_FRAMES = 64


def measure(env, cfg_name, steps, warmup, world, rank, stages=True, median_frames=50, dump=None):
    """One workload (a BASELINE.json configuration = viewport + pose of the scene the rig holds) on `world` ranks: probe,
    strips, warm-up, the timed region (barrier + synchronize on both sides, MAX over ranks), the hipEvent-bracketed median,
    the k_project clock and (stages) the per-stage times.  world = 1 on a multi-rank job = rank 0 alone, the others wait."""
    torch, dist, rig, stream, device = env.torch, env.dist, env.rig, env.stream, env.device
    from gaussiansplats3d_amd import camera, scenes
    from gaussiansplats3d_amd import dist as gdist
    cfg = scenes.CONFIGS[cfg_name]
    W, H = cfg["width"], cfg["height"]
    cam = camera.demo_camera(cfg["pose"], W, H)
    rig.set_view(cam)
    worker, mesh = rig.worker, rig.mesh
    rows_total = (H + 15) // 16
    m = {"cfg": cfg_name, "W": W, "H": H, "cam": cam, "world": world}
    with torch.cuda.stream(stream):
        worker.set_visibility_cull(False)
        full = torch.zeros((H, W, 4), dtype=torch.uint8, device=device) if rank == 0 else None
        probe = torch.empty((H, W, 4), dtype=torch.uint8, device=device)
        m["st_probe"] = rig.probe(probe.data_ptr())
        row_cost = mesh.tile_row_costs()
        strips = [(0, rows_total)]
        m["strip_balance"] = None
        if world > 1:
            # Strips balanced for drawing AND for the transfer to the root (gdist.transfer_balanced_row_strips): with 2 or 4 ranks
            # a peer's strip takes longer to cross its xGMI link than to draw, so the root takes more rows.  The draw model comes
            # from rank 0's probe frame (the blend's time per unit of row cost, the other stages as a fixed part) and is
            # broadcast: every rank must cut the frame alike.  $GS_LINK_GBPS = the per-link, per-direction rate assumed
            # (default 77); $GS_STRIP_BALANCE=cost restores the cost-only cut.
            st0 = m["st_probe"]
            total_cost = float(np.sum(row_cost)) + len(row_cost)
            model = [float(st0.blend_ms) / max(total_cost, 1.0), max(float(st0.device_ms) - float(st0.blend_ms), 0.0) + 0.07]
            obj = [model]
            dist.broadcast_object_list(obj, src=0)
            model = obj[0]
            link = float(os.environ.get("GS_LINK_GBPS", "77"))
            if os.environ.get("GS_STRIP_BALANCE", "transfer") == "cost" or model[0] <= 0.0:
                strips = gdist.balanced_row_strips(row_cost, world)
                m["strip_balance"] = {"kind": "cost"}
            else:
                strips = gdist.transfer_balanced_row_strips(row_cost, world, W, model[0], model[1], link_GBps=link)
                d_ms, x_ms = gdist.strip_frame_model(strips, row_cost, W, model[0], model[1], link_GBps=link)
                m["strip_balance"] = {"kind": "transfer-aware", "link_GBps_assumed": link, "model_draw_ms": round(d_ms, 4),
                                      "model_transfer_ms": round(x_ms, 4), "ms_per_cost": model[0], "fixed_ms": round(model[1], 4)}
        my = strips[rank]
        y0, y1 = gdist.strip_pixel_rows(my, H)
        strip = full if (world == 1) else torch.empty((max(y1 - y0, 0), W, 4), dtype=torch.uint8, device=device)
        del probe
        tile_rows = my if world > 1 else None
        gather, gather_kind = None, None
        target = strip.data_ptr()                             # or, with an overlapped gather, a callable: the next frame's strip
        overlap = world > 1 and not os.environ.get("GS_BENCH_NO_OVERLAP")
        m["gather_overlapped"] = bool(overlap)
        if world > 1:
            # every rank keys all splats but sorts / bins / blends only what reaches its strip
            worker.set_visibility_cull(True)
            # The strips of an 8K frame are 133 MB - 16.6 MB from each of 7 peers over one xGMI link each, about as long as a
            # rank's whole frame (with 2 ranks: 66 MB over ONE link, longer than the frame).  The transfer of frame k therefore
            # runs on a stream of its own beside the draw of frame k + 1, out of / into alternating buffers; the draw of frame
            # k + 2 waits for it.  Every frame is still drawn and gathered inside the timed region.
            strips_buf = [strip, torch.empty_like(strip)] if overlap else [strip]
            fulls_buf = ([full, torch.zeros_like(full)] if overlap else [full]) if rank == 0 else [None, None]
            turn = {"k": 0}
            nbuf = len(strips_buf)
            if overlap:
                target = lambda: strips_buf[turn["k"] % nbuf].data_ptr()                                # noqa: E731
            if env.group is not None:
                env.group.set_overlap(overlap)
                gather_kind = "gs_group_gather_strips (RCCL grouped send/recv behind the C ABI)" + (", overlapped with the next frame" if overlap else "")

                def gather():
                    i = turn["k"] % nbuf
                    env.group.gather_strips(strips_buf[i].data_ptr(), fulls_buf[i].data_ptr() if rank == 0 else 0, W, strips, H)
                    turn["k"] += 1
            else:
                gather_kind = f"torch.distributed batch_isend_irecv ({env.backend})" + (", overlapped with the next frame" if overlap else "")
                side = torch.cuda.Stream(device=device) if overlap else None
                last_done = [None]

                def gather():
                    i = turn["k"] % nbuf
                    if overlap:
                        side.wait_stream(stream)                  # the transfer starts when this frame's draw has finished
                        if last_done[0] is not None:
                            stream.wait_event(last_done[0])       # the next draw waits for the previous transfer
                        with torch.cuda.stream(side):
                            gdist.gather_strips(strips_buf[i], strips, fulls_buf[i], rank, world, dist)
                            done = torch.cuda.Event()
                            done.record(side)
                        last_done[0] = done
                    else:
                        gdist.gather_strips(strips_buf[i], strips, fulls_buf[i], rank, world, dist)
                    turn["k"] += 1
        # Untimed frames in front of the W warm-up steps until the device has drawn SETTLE_FRAMES of them: the first ~10 ms of work
        # after the set-up run 3 % slower than what follows (20 timed steps behind 3 warm-up steps: 0.2453-0.2476 ms per step; behind
        # 50 or 200: 0.2381-0.2391, same box - the clocks of a device that has just been woken), and the driver's W is a handful.
        # Reported as `settle_frames`; $GS_BENCH_SETTLE=0 turns them off.
        # (not in the gloo dry run of the N > 1 path on one GPU: its host-side gather takes ~0.15 s per frame)
        m["settle_frames"] = 0 if (world > 1 and env.backend == "gloo") else max(0, int(os.environ.get("GS_BENCH_SETTLE", str(SETTLE_FRAMES))) - warmup)
        for _ in range(m["settle_frames"] + warmup):
            rig.frame(target() if callable(target) else target, tile_rows)
            if gather:
                gather()
        stream.synchronize()
        torch.cuda.synchronize()
        mesh.kernel_time(0, reset=True)                       # start the per-launch k_project clock
        if world > 1:
            dist.barrier()
        elapsed, t_enqueued = rig.timed(steps, target, tile_rows, gather)
        if world > 1:
            dist.barrier()
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if env.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        if dump and rank == 0:                                # tests: the frame the timed steps left on rank 0
            torch.cuda.synchronize()
            last_full = fulls_buf[(turn["k"] - 1) % nbuf] if world > 1 else full
            np.save(dump, last_full.cpu().numpy())
        m["proj_ms_sum"], m["proj_launches"] = mesh.kernel_time(0, reset=True)    # HIP events on the kernel's own stream
        st_timed = mesh.last_stats()                          # the list-bin size the timed frames used, and their entries
        m["list_px"], m["D32"] = int(st_timed.list_bin_px), int(st_timed.tile_entries)
        m["ms_per_step"] = elapsed / steps * 1e3
        m["enqueue_ms"] = t_enqueued / steps * 1e3
        m["strips"], m["gather_kind"] = strips, gather_kind

        # median of >= 50 frames, hipEvents around the whole frame on its one stream (MAX over ranks of
        # the ranks' medians; each frame includes the gather)
        m["median_ms"], m["median_frames"], m["frame_ms_min"], m["frame_ms_p90"] = None, 0, None, None
        if median_frames > 0:
            if world > 1:
                dist.barrier()
            ev_ms = rig.event_frames(median_frames, stream, target, tile_rows, gather)
            med = float(np.median(ev_ms))
            if world > 1:
                t = torch.tensor([med], dtype=torch.float64, device=device if env.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                med = float(t.item())
            m["median_ms"], m["median_frames"] = med, median_frames
            m["frame_ms_min"], m["frame_ms_p90"] = float(np.min(ev_ms)), float(np.percentile(ev_ms, 90))

        m["stage_ms"], m["latency"] = {}, []
        if stages:
            # per-stage device times (HIP events recorded by the library), one synchronised frame at a time
            stage = {"sort": [], "project": [], "bin": [], "entry_sort": [], "blend": []}
            env.ctx.set_stage_timing(True)    # off in the timed region: an event record per stage costs ~10% of the frame
            for _ in range(min(steps, 10)):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rig.frame(strip.data_ptr(), tile_rows)
                torch.cuda.synchronize()
                m["latency"].append((time.perf_counter() - t1) * 1e3)
                rs = mesh.last_stats()
                ss, _ = worker.last_stats()
                stage["sort"].append(ss.device_ms); stage["project"].append(rs.project_ms); stage["bin"].append(rs.bin_ms)
                stage["entry_sort"].append(rs.tile_sort_ms); stage["blend"].append(rs.blend_ms)
            m["stage_ms"] = {k: float(np.median(v)) for k, v in stage.items()}
            env.ctx.set_stage_timing(False)
        m["strip_tensor"], m["full"] = strip, full
    return m


def gather_floor(strips, width, height, world, one_gpu_ms):
    """north_star's gather puts every other rank's strip into rank 0 over point-to-point xGMI links: the time those bytes need at
    the link rate is a floor under a frame however well the strips overlap, and one-GPU frame / floor a ceiling over the speed-up
    (at 8K the gather, not the strips, bounds the rate near 3.9x).  $GS_LINK_GBPS: GB/s per link and direction
    (default 76.8 = half of the 153.6 GB/s bidirectional figure in MI355X_MICROARCH.md); one link per peer, at most seven."""
    if not strips or world < 2:
        return None
    gbps = float(os.environ.get("GS_LINK_GBPS", "76.8"))
    rows = [min(r1 * 16, height) - min(r0 * 16, height) for r0, r1 in strips]
    into_rank0 = sum(rows[1:]) * width * 4
    per_link = max(rows[1:]) * width * 4                     # the peers send concurrently, each over its own link
    floor_ms = per_link / (gbps * 1e9) * 1e3
    out = {"bytes_into_rank0": int(into_rank0), "largest_strip_bytes": int(per_link), "links": min(world - 1, 7), "link_GBps": gbps,
           "gather_floor_ms": round(floor_ms, 4),
           "note": "largest peer strip / one link's rate: rank 0 receives over world - 1 links at once; HBM write of the gathered frame not included"}
    # rank 0 has ONE set of links: the seven strips arrive in parallel only while the aggregate stays under what its ports take
    out["gather_floor_serial_ms"] = round(into_rank0 / (gbps * 1e9 * min(world - 1, 7)) * 1e3, 4)
    if one_gpu_ms:
        out["speedup_ceiling_from_gather"] = round(one_gpu_ms / max(floor_ms, 1e-9), 2)
    return out


def brief(m, N):
    """The short form of one measurement (secondary objects of the line)."""
    return {"workload": m["cfg"], "width": m["W"], "height": m["H"], "n_gpus": m["world"],
            "ms_per_step": round(m["ms_per_step"], 4), "value": round(N / (m["ms_per_step"] * 1e-3) / 1e6, 2), "unit": "Msplats/s",
            "median_ms_per_step": round(m["median_ms"], 4) if m["median_ms"] else None, "median_frames": m["median_frames"],
            "strips": m["strips"] if m["world"] > 1 else None, "strip_balance": m.get("strip_balance"),
            "visible_splats": int(m["st_probe"].visible_splats), "tiles16_D": int(m["st_probe"].tiles16),
            "list_entries": m["D32"], "list_bin_px": m["list_px"],
            "stage_ms_isolated_frame": {k: round(v, 4) for k, v in m["stage_ms"].items()} or None}


def blend_lane_fractions(cfg_name, timeout_s=240):
    """What fraction of the blend's evaluated pixel lanes pass the fragment shader's A <= 8 test: measured in this run by
    tools/blend_lanes.py on the lane-counting build of the same kernel (csrc/libgsplat_hip_blendprof.so, built by
    __graft_entry__.build()), in a process of its own.  None if that build is not there."""
    from gaussiansplats3d_amd import _lib
    if not os.path.exists(_lib.BLENDPROF_LIB_PATH):
        return None
    try:
        out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "blend_lanes.py"), cfg_name],
                                      env=dict(os.environ, GSPLAT_HIP_LIB=_lib.BLENDPROF_LIB_PATH), text=True,
                                      stderr=subprocess.DEVNULL, timeout=timeout_s)
        return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    except Exception as e:                                   # a measurement aid must not take the bench line down
        print(f"bench.py: blend_lanes.py failed: {e}", file=sys.stderr)
        return None


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    if args.config == "C1":
        sys.exit(bench_c1(args))

    # the live k_project clock (gs_mesh_kernel_time) samples every n-th launch: at least 8 samples in the timed region
    os.environ.setdefault("GSPLAT_KERNEL_SAMPLE", str(max(1, min(8, args.steps // 8))))

    import torch
    import torch.distributed as dist

    from gaussiansplats3d_amd import Context, camera, scenes
    from gaussiansplats3d_amd import dist as gdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: WORLD_SIZE={world} overrides --gpus {args.gpus}", file=sys.stderr)
    # fewer GPUs than ranks (a 1-GPU box): dry run of the N > 1 path, the ranks share the device and talk over gloo
    n_dev = max(torch.cuda.device_count(), 1)
    dry_run = world > n_dev
    backend = os.environ.get("GS_BENCH_BACKEND", "gloo" if dry_run else "nccl")
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with Watchdog(float(os.environ.get("GS_BENCH_INIT_TIMEOUT", "600")), "init_process_group"):
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)

    # BASELINE.json's metric configuration (C3: garden at 1920x1080) is the headline at EVERY N - north_star asks for "Msplats/s
    # at 1920x1080 reported at 1, 2, 4 and 8 GPUs", and the driver computes the scaling itself from the per-N `value`s, which
    # only means something when they are the same workload (rounds 2-3 headlined the 8K configuration at N > 1: a different
    # metric under the same key).  BASELINE's 8-GPU configuration (configs[4] = C5, the same scene at 7680x4320: the one that
    # was meant to be sharded - strips of a 0.24 ms 1080p frame cannot scale) rides along on the same ranks with its
    # own one-GPU reference: `c5`, `c5_1gpu`, `c5_speedup_vs_1gpu`.
    headline = args.config or "C3"
    second_cfg = "C5" if (world > 1 and args.config is None) else None
    cfg = scenes.CONFIGS[headline]
    W, H = cfg["width"], cfg["height"]
    t_gen = time.perf_counter()
    scene = scenes.make_config_scene(headline, args.splats or None)
    cam = camera.demo_camera(cfg["pose"], W, H)
    N = scene.count
    t_gen = time.perf_counter() - t_gen
    extras = world == 1 and not args.no_cull and not args.only_headline

    env = Env()
    env.torch, env.dist, env.device, env.backend, env.group = torch, dist, device, backend, None
    env.stream = stream = torch.cuda.Stream(device=device)
    env.ctx = ctx = Context(local_rank, stream.cuda_stream, single_stream=True)     # §8d: the whole frame on one stream
    env.rig = rig = Rig(ctx, scene, cam, device, torch)
    worker, mesh, mvp = rig.worker, rig.mesh, rig.mvp
    if world > 1 and backend == "nccl" and os.environ.get("GS_BENCH_GATHER", "capi") == "capi":
        try:
            with Watchdog(float(os.environ.get("GS_BENCH_INIT_TIMEOUT", "600")), "gs_group_create (ncclCommInitRank)"):
                env.group = gdist.StripGroup(ctx, rank, world, dist)          # RCCL behind the C ABI (gs_group_*)
        except Exception as e:                                       # fall back to torch.distributed P2P
            print(f"bench.py rank {rank}: gs_group unavailable ({e}); gathering through torch.distributed", file=sys.stderr)
        # every rank must take the same path: one rank without its communicator sends all of them to the fallback
        ok = torch.tensor([1 if env.group is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0 and env.group is not None:
            env.group.close()
            env.group = None

    detailed = not args.only_headline
    M = measure(env, headline, args.steps, args.warmup, world, rank, stages=detailed, median_frames=args.median_frames,
                dump=os.environ.get("GS_BENCH_DUMP"))
    st_probe, strips, strip = M["st_probe"], M["strips"], M["strip_tensor"]
    frames_headline = rig.frames
    second, solo, c5_one = None, None, None
    second_solo = None
    if world > 1 and detailed:
        if second_cfg:
            second = measure(env, second_cfg, args.steps, args.warmup, world, rank, stages=True, median_frames=args.median_frames)
        # the same configuration(s) on ONE GPU of this node, in this run: rank 0 draws the whole frame alone while the others
        # wait, so the line carries its own N = 1 references for the strong-scaling ratios
        dist.barrier()
        if rank == 0:
            solo = measure(env, headline, min(args.steps, 20), 3, 1, 0, stages=True, median_frames=min(args.median_frames, 50))
            if second_cfg:
                second_solo = measure(env, second_cfg, min(args.steps, 20), 3, 1, 0, stages=True, median_frames=min(args.median_frames, 50))
        dist.barrier()
    pipelined, orbit, cull, fused, vis_fused, translucent, capture, lanes = None, None, None, None, None, None, None, None
    rop8 = None
    if world == 1:
        with torch.cuda.stream(stream):
            worker.set_visibility_cull(False)
            if detailed:
                # the engine's default shape: sorter and vertex stage on streams of their own (the reference sorts in a Web
                # Worker concurrently with drawing); every frame still waits for ITS OWN sort before it bins
                ctx2 = Context(local_rank, stream.cuda_stream, single_stream=False)
                rig2 = Rig(ctx2, scene, cam, device, torch)
                rig2.probe(strip.data_ptr())
                for _ in range(args.warmup):
                    rig2.frame(strip.data_ptr())
                p_el, p_enq = rig2.timed(args.steps, strip.data_ptr())
                pipelined = {"ms_per_step": round(p_el / args.steps * 1e3, 4),
                             "Msplats_per_s": round(N / (p_el / args.steps) / 1e6, 1),
                             "host_enqueue_ms_per_frame": round(p_enq / args.steps * 1e3, 4),
                             "note": "sorter + vertex stage on their own HIP streams: the sort of frame k+1 overlaps the tail "
                                     "of frame k's draw (same camera); throughput of whole frames, not the §8d metric"}
                rig2.close()
                ctx2.close()
                # ... and the same streams with SERIAL frames (GS_CTX_FORK_JOIN): the sort and the vertex stage of a frame run side
                # by side and join at the binner, but no sort starts before the previous frame has finished - the 8d frame
                # (t_sort + t_raster of one frame at a time), shortened to max(t_sort, t_vertex) + t_bin + t_blend
                ctx3 = Context(local_rank, stream.cuda_stream, single_stream=False, fork_join=True)
                rig3 = Rig(ctx3, scene, cam, device, torch)
                rig3.probe(strip.data_ptr())
                for _ in range(args.warmup):
                    rig3.frame(strip.data_ptr())
                f_el, _ = rig3.timed(args.steps, strip.data_ptr())
                fj_ev = rig3.event_frames(max(min(args.median_frames, 50), 1), stream, strip.data_ptr())
                pipelined["fork_join"] = {"ms_per_step": round(f_el / args.steps * 1e3, 4),
                                          "median_ms_per_step": round(float(np.median(fj_ev)), 4),
                                          "Msplats_per_s": round(N / (f_el / args.steps) / 1e6, 1),
                                          "note": "GS_CTX_FORK_JOIN: sort || vertex stage inside ONE frame, frames strictly serial "
                                                  "(every frame bracketed by events on the caller's stream for the median)"}
                rig3.close()
                ctx3.close()

            if extras:
                # a 60-pose orbit about the look-at point, one synchronised frame per pose, for medians (the
                # headline stays the fixed demo pose so rounds remain comparable)
                ms, vis = [], []
                for oc in camera.orbit_cameras(cfg["pose"], W, H, 60):
                    mesh.set_camera(oc)
                    o_mvp = oc.sort_mvp()
                    worker.sort_on_device(o_mvp, N)                # untimed: this draw may grow the entry buffer
                    _, o_st = mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=True)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    worker.sort_on_device(o_mvp, N)
                    mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=False)
                    torch.cuda.synchronize()
                    ms.append((time.perf_counter() - t1) * 1e3)
                    vis.append(int(o_st.visible_splats))
                # ... and the same orbit as a MOVING camera sees it: a new pose every frame, sort + draw enqueued back to back, no
                # host synchronisation between the frames (two laps; the first pass above has grown every buffer)
                o_cams = camera.orbit_cameras(cfg["pose"], W, H, 60)
                o_mvps = [oc.sort_mvp() for oc in o_cams]
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for lap in range(2):
                    for oc, o_mvp in zip(o_cams, o_mvps):
                        mesh.set_camera(oc)
                        worker.sort_on_device(o_mvp, N)
                        mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=False)
                moving_enq_ms = (time.perf_counter() - t1) / (2 * len(o_cams)) * 1e3      # the host's share: all 120 frames enqueued
                torch.cuda.synchronize()
                moving_ms = (time.perf_counter() - t1) / (2 * len(o_cams)) * 1e3
                # ... and at a viewer's pace: the first 120 poses of a 360-pose orbit (1 degree per frame - still 60 degrees per second at
                # 60 fps), then of a 1440-pose one.  Below ~0.06 screen heights of motion per frame the blend takes the previous frame's
                # bin order (tile_bin.hip); at the 60-pose orbit's 6 degrees it takes none
                slow = {}
                for label, poses in (("1_deg_per_frame", 360), ("quarter_deg_per_frame", 1440)):
                    s_cams = camera.orbit_cameras(cfg["pose"], W, H, poses)[:120]
                    s_mvps = [oc.sort_mvp() for oc in s_cams]
                    for oc, o_mvp in list(zip(s_cams, s_mvps))[:4]:
                        mesh.set_camera(oc)
                        worker.sort_on_device(o_mvp, N)
                        mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=False)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for oc, o_mvp in zip(s_cams, s_mvps):
                        mesh.set_camera(oc)
                        worker.sort_on_device(o_mvp, N)
                        mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=False)
                    torch.cuda.synchronize()
                    slow[label] = round((time.perf_counter() - t1) / len(s_cams) * 1e3, 4)
                mesh.set_camera(cam)
                orbit = {"poses": 60, "frame_latency_ms_median": round(float(np.median(ms)), 4),
                         "moving_camera_ms_per_frame_slow_orbit": slow,
                         "frame_latency_ms_min": round(float(np.min(ms)), 4), "frame_latency_ms_max": round(float(np.max(ms)), 4),
                         "visible_splats_median": int(np.median(vis)), "visible_splats_max": int(np.max(vis)),
                         "frame_latency_ms_mean": round(float(np.mean(ms)), 4),
                         "moving_camera_ms_per_frame": round(moving_ms, 4),
                         "moving_camera_host_enqueue_ms_per_frame": round(moving_enq_ms, 4),
                         "moving_camera_Msplats_per_s": round(N / (moving_ms * 1e-3) / 1e6, 1),
                         "note": "frame_latency_*: isolated (synchronised) frames, so compare with frame_latency_ms, not ms_per_step; "
                                 "moving_camera_*: 120 frames, a new pose each, enqueued back to back like the headline's (the orbit's "
                                 "poses see 1.8 M splats at the median, the demo pose 1.44 M)"}

                # GS_DRAW_ROP8 (round 6): the headline frame in the reference's own blend state - RGBA8, rounded after every splat,
                # back to front (SplatMaterial3D.js:65-75) - over the splats in front of each quadrant's saturation depth, and
                # GS_DRAW_ROP8_FULL over every list to its end; same sort, same vertex stage, same lists
                rop8 = {}
                for label, full_walk in (("bounded", False), ("full", True)):
                    mesh.set_draw_mode(rop8=True, full=full_walk)
                    for _ in range(2):
                        rig.frame(strip.data_ptr())
                    r_steps = min(args.steps, 20 if not full_walk else 8)
                    r_el, _ = rig.timed(r_steps, strip.data_ptr())
                    _, r_st = mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=True)
                    rop8[label] = {"ms_per_frame": round(r_el / r_steps * 1e3, 4), "Msplats_per_s": round(N / (r_el / r_steps) / 1e6, 1),
                                   "blend_ms": round(float(r_st.blend_ms), 4), "splats_walked": int(r_st.splats_walked)}
                mesh.set_draw_mode(rop8=False)
                for _ in range(2):                              # (the fp32 statistics back in place for whatever follows)
                    rig.frame(strip.data_ptr())
                rop8["note"] = ("gs_mesh_set_draw_mode: GS_DRAW_ROP8 = colour to the ROP-emulating oracle's gate (>= 99.5 % of channel values "
                                "equal, <= 1 apart; alpha <= 2 where it stalls below 255), GS_DRAW_ROP8_FULL = all four channels "
                                "(tests/test_gpu_crops.py); the fp32 draw is the headline")
                rig.frame(strip.data_ptr())                       # the frame every culled variant must reproduce bit for bit
                torch.cuda.synchronize()
                ref_img = strip.clone()
                # second column: cull ON = the reference's octree + gatherSceneNodesForSort in front of the sort
                from gaussiansplats3d_amd import SplatTree
                t_tree = time.perf_counter()
                tree = SplatTree(ctx, 8, 1000).process_splat_mesh(scene.centers, alphas=scene.rgba[:, 3])
                t_tree = time.perf_counter() - t_tree

                # the whole frame without a host round trip: the gather leaves splatRenderCount on the device, the sort takes
                # the list's length from there and drops, splat by splat, what the leaf test kept but the frustum cannot show
                # (gs_sorter_set_frustum_cull composes with the tree), the draw takes the sorted list's length from the sort
                tree_splats = int(tree.info().splats)
                worker.set_frustum_cull(True)

                def cull_frame(c=cam, c_mvp=mvp):
                    tree.gather_scene_nodes_for_sort(c, sort_worker=worker, to_host=False, asynchronous=True)
                    worker.sort_gathered(c_mvp, keep_on_device=True)
                    mesh.use_sorter_result(worker, tree_splats)
                    mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=False)

                for _ in range(3):
                    cull_frame()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    cull_frame()
                torch.cuda.synchronize()
                cull_ms = (time.perf_counter() - t1) / args.steps * 1e3
                # (not expected to be identical: the reference takes min / max - hence the buckets and the tie order - over the
                # gathered list only, and its leaf test may drop leaves with a splat at the very edge of the frame)
                cull_diff = int((ref_img.to(torch.int16) - strip.to(torch.int16)).abs().max().item())
                cs = rig.worker.last_stats()[0]
                worker.set_frustum_cull(False)
                Rc = tree.gather_scene_nodes_for_sort(cam, sort_worker=worker, to_host=False)["splatRenderCount"]
                cull = {"render_count": int(Rc), "kept_after_frustum_cull": int(cs.result_count), "leaves": int(tree.info().leaves),
                        "ms_per_frame": round(cull_ms, 4), "max_abs_diff_vs_cull_off_frame_u8": cull_diff,
                        "Msplats_per_s_scene": round(N / (cull_ms * 1e-3) / 1e6, 1),
                        "Msplats_per_s_rendered": round(Rc / (cull_ms * 1e-3) / 1e6, 1), "tree_build_s": round(t_tree, 2),
                        "note": "asynchronous gather (four small plan kernels; the copy of the kept leaves' lists is fused with the sort's key kernel and streams leaf-ordered centres; no host round trip) + sort with the per-splat frustum cull on top "
                                "+ draw, EVERY frame; render_count = R kept by the reference's leaf test, scene = all N splats per frame.  "
                                "The reference gathers and sorts only when the camera has turned or moved enough (Viewer.runSplatSort, "
                                "src/Viewer.js:1858-1961) and keeps drawing with the last order in between"}
                # ... and a pose where an octree cull has something to remove: the camera INSIDE the scene, at the
                # demo's look-at point, looking on along the demo's view direction - what a viewer sees after walking in.  Cull off
                # (full sort + draw) against cull on (gather + fused copy / keys / frustum cull + draw) at that pose.
                up_v, pos_v, look_v = (np.asarray(v, dtype=np.float64) for v in camera.DEMO_POSES[cfg["pose"]])
                cam_in = camera.PerspectiveCamera(W, H, tuple(look_v), tuple(look_v + (look_v - pos_v)), tuple(up_v))
                mvp_in = cam_in.sort_mvp()
                mesh.set_camera(cam_in)
                worker.set_frustum_cull(False)

                def plain_frame():
                    worker.sort_on_device(mvp_in, N)
                    mesh.use_sorter_result(worker, N)
                    mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=False)

                def time_frames(fn, steps):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize()
                    t = time.perf_counter()
                    for _ in range(steps):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t) / steps * 1e3

                in_off = time_frames(plain_frame, args.steps)
                _, in_st = mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=True)
                off_img = strip.clone()
                worker.set_frustum_cull(True)
                in_on = time_frames(lambda: cull_frame(cam_in, mvp_in), args.steps)
                in_diff = int((off_img.to(torch.int16) - strip.to(torch.int16)).abs().max().item())
                in_kept = int(rig.worker.last_stats()[0].result_count)
                worker.set_frustum_cull(False)
                in_R = tree.gather_scene_nodes_for_sort(cam_in, sort_worker=worker, to_host=False)["splatRenderCount"]
                cull["inside_pose"] = {"camera": "at the demo's look-at point, looking on along the demo's view direction",
                                       "render_count": int(in_R), "removed_by_the_leaf_test_frac": round(1.0 - in_R / N, 4),
                                       "kept_after_frustum_cull": in_kept, "visible_splats": int(in_st.visible_splats),
                                       "cull_off_ms_per_frame": round(in_off, 4), "cull_on_ms_per_frame": round(in_on, 4),
                                       "cull_on_over_cull_off": round(in_on / in_off, 4),
                                       "max_abs_diff_vs_cull_off_frame_u8": in_diff}
                del off_img
                mesh.set_camera(cam)
                tree.dispose()
                mesh.use_sorter_result(worker, N)

                # third column: the per-splat frustum cull fused into pass 0 of the sort (gs_sorter_set_frustum_cull).  Keys,
                # range and buckets still span all N splats, so the frame must be bit-identical to the headline path's
                worker.set_frustum_cull(True)
                for _ in range(3):
                    rig.frame(strip.data_ptr())
                f_el, f_enq = rig.timed(args.steps, strip.data_ptr())
                f_ms = f_el / args.steps * 1e3
                fs = rig.timed_sort_stats(strip.data_ptr())
                fused = {"kept": int(fs.result_count), "ms_per_frame": round(f_ms, 4),
                         "Msplats_per_s_scene": round(N / (f_ms * 1e-3) / 1e6, 1),
                         "host_enqueue_ms_per_frame": round(f_enq / args.steps * 1e3, 4), "sort_ms_last": round(float(fs.device_ms), 4),
                         "frame_identical_to_full_sort": bool(torch.equal(ref_img, strip)),
                         "note": "keys + min/max over all N, then pass 0 of the radix sort drops the splats whose centre is "
                                 "outside 1.25x the clip volume; the list is the full sort's list minus those splats"}
                worker.set_frustum_cull(False)

                # the exact version, and what every rank of a multi-GPU run does with its strip: vertex stage first, then the
                # sort keeps only the splats that survived it (gs_sorter_set_visibility_cull)
                worker.set_visibility_cull(True)
                for _ in range(3):
                    rig.frame(strip.data_ptr())
                v_el, v_enq = rig.timed(args.steps, strip.data_ptr())
                v_ms = v_el / args.steps * 1e3
                vs = rig.timed_sort_stats(strip.data_ptr())
                vis_fused = {"kept": int(vs.result_count), "ms_per_frame": round(v_ms, 4),
                             "Msplats_per_s_scene": round(N / (v_ms * 1e-3) / 1e6, 1), "sort_ms_last": round(float(vs.device_ms), 4),
                             "frame_identical_to_full_sort": bool(torch.equal(ref_img, strip)),
                             "note": "project -> sort (keys + min/max over all N, radix passes over the visible splats only) -> "
                                     "bin -> blend: the per-rank frame of a multi-GPU run, here with the whole screen as the strip"}
                worker.set_visibility_cull(False)
                del ref_img

            if extras and headline == "C3":
                # BASELINE.json configs[4]'s viewport on this one GPU: the N = 1 reference of a scaling run's headline
                c5_one = brief(measure(env, "C5", min(args.steps, 20), 3, 1, 0, stages=True, median_frames=0), N)
                rig.set_view(cam)
                # secondary scenes, same pose and viewport: (a) the same geometry with translucent splats - the stand-in's
                # pixels saturate after ~90 splats (the blend reads ~2 % of its lists), here they do not, so the blend walks
                # its lists; (b) a stand-in that resembles a trained capture: surfels on a ground disc / an object / a backdrop,
                # bimodal opacity, the camera outside the object (scenes.capture_like) - it brackets the headline from the
                # honest side: most splats are inside the frustum and low-alpha splats lengthen every pixel's list
                rig.close()
                rig = env.rig = None
                out_ptr = strip.data_ptr()
                for key in ("C3T", "C3S"):
                    sc = scenes.make_config_scene(key)
                    rg = Rig(ctx, sc, cam, device, torch)
                    rg.probe(out_ptr)
                    for _ in range(3):
                        rg.frame(out_ptr)
                    k_steps = min(args.steps, 20)
                    t_el, _ = rg.timed(k_steps, out_ptr)
                    t_ms = t_el / k_steps * 1e3
                    ev = rg.event_frames(max(min(args.median_frames, 50), 1), stream, out_ptr)
                    _, st_t = rg.mesh.render(out_device_ptr=out_ptr, to_host=False, want_stats=True)
                    ss_t = rg.timed_sort_stats(out_ptr)
                    obj = {"workload": scenes.CONFIGS[key]["label"], "ms_per_frame": round(t_ms, 4),
                           "median_ms_per_frame": round(float(np.median(ev)), 4),
                           "Msplats_per_s": round(sc.count / (t_ms * 1e-3) / 1e6, 1),
                           "sort_ms": round(float(ss_t.device_ms), 4), "project_ms": round(float(st_t.project_ms), 4),
                           "bin_ms": round(float(st_t.bin_ms), 4), "entry_sort_ms": round(float(st_t.tile_sort_ms), 4),
                           "blend_ms": round(float(st_t.blend_ms), 4),
                           "visible_splats": int(st_t.visible_splats), "V_over_N": round(int(st_t.visible_splats) / sc.count, 4),
                           "tiles16_D": int(st_t.tiles16), "D_over_R": round(int(st_t.tiles16) / sc.count, 3),
                           "list_entries": int(st_t.tile_entries), "list_bin_px": int(st_t.list_bin_px),
                           "entries_scanned": int(st_t.entries_scanned), "splats_walked": int(st_t.splats_walked),
                           "halves_evaluated": int(st_t.halves_evaluated)}
                    info = rg.mesh.deep_pass_info()
                    obj["deep_pass"] = {"bins": int(len(info["bins"])), "bins_over_threshold": info["candidates"],
                                        "chunks_closed_by_bin_workgroups": info["chunks_closed_by_bins"],
                                        "pool_exhausted": info["pool_exhausted"]}
                    if key == "C3S":
                        # the same frames with every bin on one workgroup (the deep pass off): same pixels, the tail of a few very
                        # deep bins back on four waves each
                        rg.mesh.set_deep_pass(False)
                        for _ in range(3):
                            rg.frame(out_ptr)
                        s_el, _ = rg.timed(k_steps, out_ptr)
                        _, st_s = rg.mesh.render(out_device_ptr=out_ptr, to_host=False, want_stats=True)
                        obj["deep_pass"]["off"] = {"ms_per_frame": round(s_el / k_steps * 1e3, 4), "blend_ms": round(float(st_s.blend_ms), 4)}
                        rg.mesh.set_deep_pass(True)
                        # Under MOTION: the deep pass picks its bins from the PREVIOUS frame's statistics, so a fixed pose flatters
                        # it.  Two laps of the 60-pose orbit, ONE frame per pose (the selection always comes from the neighbouring
                        # pose); the first lap only grows the buffers, the second is timed (synchronised frames: latencies).
                        o_ms, o_bins = [], []
                        for lap in range(2):
                            for oc in camera.orbit_cameras(cfg["pose"], W, H, 60):
                                rg.set_view(oc)
                                torch.cuda.synchronize()
                                t1 = time.perf_counter()
                                rg.frame(out_ptr)
                                torch.cuda.synchronize()
                                if lap:
                                    o_ms.append((time.perf_counter() - t1) * 1e3)
                                    o_bins.append(int(len(rg.mesh.deep_pass_info()["bins"])))
                        torch.cuda.synchronize()                     # ... and as a MOVING camera: a new pose every frame, nothing synchronised
                        t1 = time.perf_counter()
                        for lap in range(2):
                            for oc in camera.orbit_cameras(cfg["pose"], W, H, 60):
                                rg.set_view(oc)
                                rg.frame(out_ptr)
                        torch.cuda.synchronize()
                        o_moving = (time.perf_counter() - t1) / 120 * 1e3
                        rg.set_view(cam)

                        def cold_loop():
                            """A render loop that never asks for statistics, from a cold start: fresh worker and mesh, one frame per
                            pose, the first lap synchronised at every frame boundary (a viewer presents its frames), the second free."""
                            rc = Rig(ctx, sc, cam, device, torch)
                            rc.worker.sort_on_device(rc.mvp, rc.N)
                            rc.mesh.use_sorter_result(rc.worker, rc.N)
                            lat = []
                            cams60 = camera.orbit_cameras(cfg["pose"], W, H, 60)
                            for oc in cams60:
                                rc.set_view(oc)
                                torch.cuda.synchronize()
                                t2 = time.perf_counter()
                                rc.frame(out_ptr)
                                torch.cuda.synchronize()
                                lat.append((time.perf_counter() - t2) * 1e3)
                            t2 = time.perf_counter()
                            for oc in cams60:
                                rc.set_view(oc)
                                rc.frame(out_ptr)
                            torch.cuda.synchronize()
                            mv = (time.perf_counter() - t2) / 60 * 1e3
                            px = int(rc.mesh.last_stats().list_bin_px)
                            rc.close()
                            return {"first_lap_frame_latency_ms_median": round(float(np.median(lat)), 4),
                                    "first_lap_frame_latency_ms_max": round(float(np.max(lat)), 4),
                                    "second_lap_moving_camera_ms_per_frame": round(mv, 4), "list_bin_px": px}

                        loop = cold_loop()
                        os.environ["GSPLAT_NO_ASYNC_LIST_BINS"] = "1"    # rounds 1-5: the list-bin size only followed gs_mesh_last_stats
                        loop["with_GSPLAT_NO_ASYNC_LIST_BINS"] = cold_loop()
                        del os.environ["GSPLAT_NO_ASYNC_LIST_BINS"]
                        loop["note"] = ("every draw leaves {visible splats, 16-px tiles} in mapped host words and the following draws size their "
                                        "list bins from them (mesh.hip, mesh_adapt_list_bins); before round 6 a loop that never called "
                                        "gs_mesh_last_stats kept the first guess (128-px lists) for ever")
                        obj["render_loop_without_statistics"] = loop
                        obj["orbit"] = {"poses": 60, "moving_camera_ms_per_frame": round(o_moving, 4),
                                        "frame_latency_ms_median": round(float(np.median(o_ms)), 4),
                                        "frame_latency_ms_min": round(float(np.min(o_ms)), 4),
                                        "frame_latency_ms_max": round(float(np.max(o_ms)), 4),
                                        "frame_latency_ms_p90": round(float(np.percentile(o_ms, 90)), 4),
                                        "deep_bins_median": int(np.median(o_bins)), "deep_bins_max": int(np.max(o_bins)),
                                        "note": "one synchronised frame per pose, the deep pass's bins chosen from the previous POSE's "
                                                "statistics; compare with the fixed pose's frame latency, not with ms_per_frame"}
                    if key == "C3T":
                        obj["note"] = ("opacity ~ sigmoid(N(-2,1)): pixels do not saturate early, the blend scans and walks "
                                       "its entry lists (compare entries_scanned / splats_walked with the `blend` object)")
                        translucent = obj
                    else:
                        obj["note"] = ("not a BASELINE.json configuration: surfels on 2-D manifolds, one axis 5-20x thinner, 40 % "
                                       "of the splats with alpha < 0.1, the demo camera outside the object looking in "
                                       "(demo/garden.html:38-43 for the pose)")
                        capture = obj
                    rg.close()
                    del sc, rg
            if extras:
                lanes = blend_lane_fractions(headline)       # own process, the lane-counting build of k_tile_blend

    ms_per_step = M["ms_per_step"]
    D16 = int(st_probe.tiles16)
    visible = int(st_probe.visible_splats)
    if rank == 0:
        R = Rs = N
        P = W * H
        B = frame_algorithmic_bytes(R, Rs, D16, P, scene.sh_degree, scene.cov_half)
        frame_gbs = B / (ms_per_step * 1e-3) / 1e9
        kb = project_algorithmic_bytes(N, visible, scene.sh_degree, scene.cov_half)
        proj_ms_sum, proj_launches = M["proj_ms_sum"], M["proj_launches"]
        if not proj_launches:
            raise SystemExit("bench.py: no k_project launch was timed (GSPLAT_KERNEL_SAMPLE=0?)")
        k_ms = proj_ms_sum / proj_launches
        k_gbs = kb / (k_ms * 1e-3) / 1e9
        # (since round 5 the per-block frustum test of the vertex stage is a kernel of its own, k_block_test, ~5 us and ~2 MB in front of
        # k_project: the live clock and the algorithmic bytes here are k_project's alone, as rocprofv3 lists it; the whole vertex stage
        # is frame.stage_ms_isolated_frame.project)
        roof_note = ("k_project alone (events right around its launch; rocprofv3's average for the same kernel is a few us lower: the "
                     "two event packets); k_block_test (~5 us, ~2 MB) runs in front of it and is part of "
                     "frame.stage_ms_isolated_frame.project")
        if world > 1:
            # a rank projects all N centres but only its strip's survivors: the kernel's algorithmic bytes are the rank's
            # own; reported for rank 0
            roof_note += "; rank 0's launch: all N centres, the survivors of its own strip (visible_splats is the full frame's)"
        traffic, traffic_src = pmc_traffic("k_project", headline) if world == 1 else (None, None)
        frame_traffic, frame_traffic_src = pmc_frame_traffic(headline) if world == 1 else (None, None)
        walked, scanned, halves = int(st_probe.splats_walked), int(st_probe.entries_scanned), int(st_probe.halves_evaluated)
        stage_ms = M["stage_ms"]
        blend_ms = stage_ms.get("blend")
        # fp32 operations the blend really executed: 128 pixel lanes per evaluated half tile
        blend_tflops = (halves * 128 * BLEND_FLOPS_PER_PIXEL_SPLAT / (blend_ms * 1e-3) / 1e12) if (blend_ms and world == 1) else None
        valu, valu_src = pmc_valu("k_tile_blend", headline)
        proj_valu, _ = pmc_valu("k_project", headline)
        out = {
            "metric": "Msplats/s sorted+rasterized at 1920x1080 SH-2" if headline == "C3"
                      else f"Msplats/s sorted+rasterized ({cfg['label']})",
            "value": round(N / (ms_per_step * 1e-3) / 1e6, 2), "unit": "Msplats/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # untimed frames drawn in front of the warm-up steps (a freshly woken device: measure())
            "settle_frames": M.get("settle_frames", 0),
            # SURVEY.md 8d's own form: median of >= 50 frames, each bracketed by HIP events on the frame's stream
            "median_ms_per_step": round(M["median_ms"], 4) if M["median_ms"] else None, "median_frames": M["median_frames"],
            "median_value": round(N / (M["median_ms"] * 1e-3) / 1e6, 2) if M["median_ms"] else None,
            "frame_ms_min": round(M["frame_ms_min"], 4) if M["frame_ms_min"] else None,
            "frame_ms_p90": round(M["frame_ms_p90"], 4) if M["frame_ms_p90"] else None,
            "fps": round(1e3 / ms_per_step, 2),
            "frame_latency_ms": round(float(np.median(M["latency"])), 4) if M["latency"] else None,
            "host_enqueue_ms_per_frame": round(M["enqueue_ms"], 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32 keys / f32 raster",
            "data": "synthetic" if scene.name in (headline, "C3") else f"file:{scene.name}",
            "config": {"workload": f"{headline}: {cfg['label']}", "splats": N, "sh_degree": scene.sh_degree,
                       "width": W, "height": H, "cull": "off (R=N)", "sort_precision_bits": 16,
                       # what the stand-in scene looks like to the engine at this pose, next to the number it produces (VERDICT
                       # r04: capture-like content at the same N / SH / resolution is 5x slower): the fraction of the splats that
                       # survives the vertex stage and the 16-px tiles a rendered splat touches
                       "V_over_N": round(visible / N, 4), "D_over_R": round(D16 / N, 3),
                       "scene_statistics_note": "synthetic stand-in (no garden.ply in the image); capture_like / translucent "
                                                "below are the same N, SH and resolution with other content",
                       "streams": "one (SURVEY.md 8d: sort -> draw on a single stream)" if not M.get("gather_overlapped") else
                                  "sort -> draw on one stream per rank; the strip gather of frame k on a second stream beside frame k + 1",
                       "parallelism": f"tile-row strips x{world}" if world > 1 else "1 GPU",
                       "strips": strips if world > 1 else None, "strip_balance": M.get("strip_balance"),
                       "backend": backend if world > 1 else None,
                       "sort": "full list (R = N)" if world == 1 else "per rank: keys over all N, radix passes over the splats "
                               "its strip draws (gs_sorter_set_visibility_cull)",
                       "gather": M["gather_kind"],
                       "dry_run_shared_gpu": dry_run if world > 1 else None},
            # the largest HBM-bound kernel: the vertex stage (the largest kernel overall is the blend, which is VALU-bound:
            # see `blend`)
            "roofline": {"bound": "hbm", "kernel": "k_project", "achieved": round(k_gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(k_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(kb),
                         "avg_launch_ms": round(k_ms, 5), "launches_timed": proj_launches,
                         # the whole vertex stage (k_block_test + mask reset + k_project) against the same bytes: rounds 1-4
                         # bracketed this, round 5 moved the bracket to k_project alone - both stay in the line
                         "stage_ms": round(stage_ms.get("project"), 5) if stage_ms.get("project") else None,
                         "stage_frac": round(kb / (stage_ms["project"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if stage_ms.get("project") else None,
                         "launches_in_timed_region": args.steps,
                         # the vertex stage is co-limited: its SIMDs issue vector instructions most of the launch as well
                         "valu_busy_frac": proj_valu.get("valu_busy_frac") if proj_valu else None,
                         "visible_splats": visible, "note": roof_note},
            # the depth sort (the north star's first half): all of its kernels together.  `achieved` / `frac` price SURVEY 8d's
            # FORMULA (56 bytes per sorted splat for two radix passes) against the sort's device time in the frames (the library's
            # HIP events around the sort, median of the synchronised frames below); `traffic` is what the COUNTERS saw the sort's
            # kernels move per frame ((2 x FETCH_SIZE + WRITE_SIZE) of a separate rocprofv3 --pmc pass), `engine_bytes` what the
            # passes have to move by construction (keys 12 r + 4 w, two histograms 4 r each, pass 0: 4 r key + 4 r payload map + 4 w
            # packed word, pass 1: 4 r + 4 w)
            "roofline_sort": (lambda sp, sms: {
                "bound": "hbm", "kernels": "k_depth_key + 2 x (k_radix_hist + k_radix_scatter_chunk)",
                "sort_ms": round(sms, 4) if sms else None, "sort_ms_source": "gs_sort_stats.device_ms, median of the synchronised frames",
                "algorithmic_bytes": 56 * Rs, "algorithmic_bytes_source": "SURVEY 8d formula: 56 B per sorted splat",
                "achieved": round(56 * Rs / (sms * 1e-3) / 1e9, 1) if sms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(56 * Rs / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sms else None,
                "engine_bytes": 48 * Rs,
                "traffic": sp["counter_bytes"], "traffic_source": sp["counter_source"],
                "traffic_frac": round(sp["counter_bytes"] / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (sp["counter_bytes"] and sms) else None,
                "per_kernel_us": sp["per_kernel_us"], "kernel_us_sum": sp["kernel_us_sum"], "kernels_source": sp["kernels_source"],
                "verdict": "frac is the formula's; traffic_frac the counters' - the sort moves fewer bytes than the formula charges"
            })(sort_profile(headline), stage_ms.get("sort")) if world == 1 else None,
            # the largest kernel of the frame
            "blend": {"kernel": "k_tile_blend", "bound": "valu", "ms": round(blend_ms, 4) if blend_ms else None,
                      "splats_walked": walked, "halves_evaluated": halves, "entries_scanned": scanned, "list_entries": M["D32"],
                      "flops_per_pixel_splat": BLEND_FLOPS_PER_PIXEL_SPLAT,
                      "tflops": round(blend_tflops, 2) if blend_tflops else None,
                      "frac_of_157TF": round(blend_tflops / FP32_PEAK_TFLOPS, 4) if blend_tflops else None,
                      # of the pixel lanes the kernel evaluates (128 per half tile), the fraction that passes A <= 8
                      "lanes_kept_frac": lanes.get("lanes_kept_frac") if lanes else None,
                      "lanes_useful_frac": lanes.get("lanes_useful_frac") if lanes else None,
                      "lanes_kept_frac_of_whole_quadrants": lanes.get("lanes_kept_frac_of_whole_quadrants") if lanes else None,
                      "lanes_source": "tools/blend_lanes.py on libgsplat_hip_blendprof.so, this run" if lanes else None,
                      "valu_busy_frac": valu.get("valu_busy_frac") if valu else None,
                      "valu_insts_per_launch": valu.get("SQ_INSTS_VALU") if valu else None, "counter_source": valu_src},
            # whole frame: SURVEY.md §8d's formula, and what the counters say the engine really moves
            "frame": {"algorithmic_bytes": int(B), "bytes_per_splat": round(B / R, 1), "GBps": round(frame_gbs, 1),
                      "frac_of_hbm_peak": round(frame_gbs / HBM_PEAK_GBS, 4),
                      "counter_bytes": frame_traffic, "counter_source": frame_traffic_src,
                      "counter_GBps": round(frame_traffic / (ms_per_step * 1e-3) / 1e9, 1) if frame_traffic else None,
                      "counter_frac_of_hbm_peak": round(frame_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                      if frame_traffic else None,
                      "tiles16_D": D16, "D_per_splat": round(D16 / R, 3), "list_entries": M["D32"], "list_bin_px": M["list_px"],
                      "stage_ms_isolated_frame": {k: round(v, 4) for k, v in stage_ms.items()}},
            # N > 1: the headline configuration on one GPU of this node (rank 0 alone) - the line's own strong-scaling reference -
            # and BASELINE.json's 8-GPU configuration (C5: the same scene at 7680x4320) on the same ranks, with ITS one-GPU reference
            "same_config_1gpu": brief(solo, N) if solo else None,
            "speedup_vs_same_config_1gpu": round(solo["ms_per_step"] / ms_per_step, 4) if solo else None,
            "c5": brief(second, N) if second else None,
            # (N = 1: configs[4]'s viewport on this one GPU)
            "c5_1gpu": brief(second_solo, N) if second_solo else c5_one,
            "c5_speedup_vs_1gpu": round(second_solo["ms_per_step"] / second["ms_per_step"], 4) if (second and second_solo) else None,
            # what the strip gather alone allows (bytes into rank 0 over its links), beside each measured speed-up
            "gather_floor": gather_floor(strips, W, H, world, solo["ms_per_step"] if solo else None) if world > 1 else None,
            "c5_gather_floor": gather_floor(second["strips"], scenes.CONFIGS[second_cfg]["width"], scenes.CONFIGS[second_cfg]["height"], world,
                                            second_solo["ms_per_step"] if second_solo else None) if second else None,
            "pipelined": pipelined,
            "orbit": orbit,
            "cull_on": cull,
            "frustum_cull_fused": fused,
            "visibility_cull_fused": vis_fused,
            "translucent": translucent,
            "capture_like": capture,
            "rop8_draw_mode": rop8,
            "cpu_baseline": None,
            "frames_drawn_before_timing_ended": frames_headline,
            "scene_gen_s": round(t_gen, 1),
        }
        if world == 1 and not args.no_cpu and not args.only_headline:
            out["cpu_baseline"] = cpu_baseline(scene, mvp, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if env.group is not None:
        env.group.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if env.rig is not None:
        env.rig.close()
    ctx.close()


if __name__ == "__main__":
    main()
