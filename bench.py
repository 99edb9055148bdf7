#!/usr/bin/env python
"""bench.py — headline benchmark: Msplats/s sorted + rasterized (BASELINE.json metric).

One "step" = one frame of the hot path on the garden.ply stand-in (configs[2]: 5.8 M splats, SH-2, 1920x1080):
depth-key + device-wide stable radix sort of ALL splats (cull off, R = N, the reference's own no-tree path,
src/Viewer.js:2061-2073), then project -> bin -> entry sort -> blend into an RGBA8 framebuffer.  Inputs are
resident in HBM before the timed region.  With --gpus N every rank sorts + projects the replicated scene and
rasterises a strip of tile rows; strips are gathered to rank 0 over RCCL inside the timed region (strong scaling).

Like the reference (sort in a Web Worker, draw on the main thread) the sorter owns a HIP stream of its own: the
sort of frame k+1 may overlap the tail of frame k's draw.  Every frame still waits for ITS OWN sort before it
bins (use_sorter_result joins the streams), so `value` is whole frames per second times R; `frame_latency_ms`
is one isolated frame (sort -> draw, synchronised on both sides).

Prints ONE JSON line on rank 0 (see the task contract) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s is the measured copy ceiling)
SH_BYTES = {0: 0, 1: 18, 2: 48}


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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*pmc_traffic.json,
    written by tools/pmc_traffic.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        k = d["kernels"].get(kernel)
        if k is None:                                           # template instances: "k_project<false>"
            k = next((v for name, v in sorted(d["kernels"].items()) if name.split("<")[0] == kernel), None)
        return (int(k["hbm_bytes_per_launch"]) if k else None), os.path.basename(files[-1])
    except Exception:
        return None, None


def cpu_baseline(scene, mvp, budget_s):
    """Time the reference's own sorter (oracle/_ref, compiled from the reference's sorter_no_simd.cpp) on this
    host: 1 thread, the faithful configuration (one Web Worker).  The reference has no CPU rasteriser, so the
    baseline covers the sort half of the frame only."""
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
    return {"value": round(n / per / 1e6, 2), "unit": "Msplats/s (sort only)", "cores": 1, "kind": kind,
            "ms_per_sort": round(per * 1e3, 2), "host_cpus": os.cpu_count(),
            "sample": f"{reps} full sorts ({t_total:.1f} s) of the same {n} splats / same MVP, precision 16, integer "
                      "static path, sorter_no_simd.cpp built -O2; the reference has no CPU rasteriser, so raster has "
                      "no CPU leg"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--splats", type=int, default=0, help="override the splat count (debug only; invalid as a result)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cull", action="store_true", help="skip the secondary cull-on measurement (octree build + gather)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from gaussiansplats3d_amd import Context, SplatMesh, camera, create_sort_worker, scenes, util
    from gaussiansplats3d_amd import dist as gdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # GS_BENCH_BACKEND=gloo + fewer GPUs than ranks: dry run of the N > 1 path on a 1-GPU box (ranks share the device)
    backend = os.environ.get("GS_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = scenes.CONFIGS[args.config]
    W, H = cfg["width"], cfg["height"]
    t_gen = time.perf_counter()
    scene = scenes.make_config_scene(args.config, args.splats or None)
    cam = camera.demo_camera(cfg["pose"], W, H)
    mvp = cam.sort_mvp()
    N = scene.count
    t_gen = time.perf_counter() - t_gen

    stream = torch.cuda.Stream(device=device)
    ctx = Context(local_rank, stream.cuda_stream)
    worker = create_sort_worker(ctx, N)                       # integerBasedSort, precision 16: Viewer defaults
    worker.post_message({"centers": util.integer_centers(scene.centers), "range": {"from": 0, "to": N - 1, "count": N}})
    mesh = SplatMesh(ctx, N, scene.sh_degree, scene.cov_half)
    mesh.build(scene.centers, scene.cov, scene.rgba, scene.sh if scene.sh_degree else None)
    mesh.set_camera(cam)

    rows_total = (H + 15) // 16
    with torch.cuda.stream(stream):
        full = torch.zeros((H, W, 4), dtype=torch.uint8, device=device) if rank == 0 else None

        # probe frame (untimed): full-frame draw gives per-tile-row entry counts to balance the strips, and grows
        # the entry buffer if the first guess was too small
        worker.sort_on_device(mvp, N)
        mesh.use_sorter_result(worker, N)
        probe = torch.empty((H, W, 4), dtype=torch.uint8, device=device)
        _, st_probe = mesh.render(out_device_ptr=probe.data_ptr(), to_host=False, want_stats=True)
        row_cost = mesh.tile_row_costs()
        strips = gdist.balanced_row_strips(row_cost, world) if world > 1 else [(0, rows_total)]
        my = strips[rank]
        y0, y1 = gdist.strip_pixel_rows(my, H)
        strip = full if (world == 1) else torch.empty((max(y1 - y0, 0), W, 4), dtype=torch.uint8, device=device)
        del probe

        def frame():
            worker.sort_on_device(mvp, N)
            mesh.render(tile_rows=my if world > 1 else None, out_device_ptr=strip.data_ptr(), to_host=False,
                        want_stats=False)
            if world > 1:
                gdist.gather_strips(strip, strips, full, rank, world, dist)

        for _ in range(args.warmup):
            frame()
        stream.synchronize()
        torch.cuda.synchronize()
        mesh.kernel_time(0, reset=True)                       # start the per-launch k_project clock
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            frame()
        t_enqueued = time.perf_counter() - t0                # host time to enqueue the K frames (no device wait in it)
        stream.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        proj_ms_sum, proj_launches = mesh.kernel_time(0, reset=True)    # HIP events on the kernel's own stream
        st_timed = mesh.last_stats()                          # the list-bin size the timed frames used, and their entries
        list_px_timed, D32 = int(st_timed.list_bin_px), int(st_timed.tile_entries)

        # per-stage device times (HIP events recorded by the library), one synchronised frame at a time; also the
        # latency of an isolated frame
        stage = {"sort": [], "project": [], "bin": [], "entry_sort": [], "blend": []}
        latency = []
        for _ in range(min(args.steps, 10)):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            worker.sort_on_device(mvp, N)
            mesh.render(tile_rows=my if world > 1 else None, out_device_ptr=strip.data_ptr(), to_host=False,
                        want_stats=False)
            torch.cuda.synchronize()
            latency.append((time.perf_counter() - t1) * 1e3)
            rs = mesh.last_stats()
            ss, _ = worker.last_stats()
            stage["sort"].append(ss.device_ms); stage["project"].append(rs.project_ms); stage["bin"].append(rs.bin_ms)
            stage["entry_sort"].append(rs.tile_sort_ms); stage["blend"].append(rs.blend_ms)
        stage_ms = {k: float(np.median(v)) for k, v in stage.items()}

        # a 60-pose orbit about the look-at point, one synchronised frame per pose, for medians (the
        # headline stays the fixed demo pose so rounds remain comparable)
        orbit = None
        if world == 1 and not args.no_cull:
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
            mesh.set_camera(cam)
            orbit = {"poses": 60, "frame_latency_ms_median": round(float(np.median(ms)), 4),
                     "frame_latency_ms_min": round(float(np.min(ms)), 4), "frame_latency_ms_max": round(float(np.max(ms)), 4),
                     "visible_splats_median": int(np.median(vis)), "visible_splats_max": int(np.max(vis)),
                     "note": "isolated (synchronised) frames, so compare with frame_latency_ms, not ms_per_step"}

        # second column: cull ON = the reference's octree + gatherSceneNodesForSort in front of the sort
        cull = None
        if world == 1 and not args.no_cull:
            from gaussiansplats3d_amd import SplatTree
            t_tree = time.perf_counter()
            tree = SplatTree(ctx, 8, 1000).process_splat_mesh(scene.centers, alphas=scene.rgba[:, 3])
            t_tree = time.perf_counter() - t_tree

            def cull_frame():
                r = tree.gather_scene_nodes_for_sort(cam, sort_worker=worker, to_host=False)   # 4-byte read-back inside
                worker.sort_gathered(mvp, keep_on_device=True)
                mesh.use_sorter_result(worker, r["splatRenderCount"])
                mesh.render(out_device_ptr=strip.data_ptr(), to_host=False, want_stats=False)
                return r["splatRenderCount"]

            for _ in range(3):
                Rc = cull_frame()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                cull_frame()
            torch.cuda.synchronize()
            cull_ms = (time.perf_counter() - t1) / 20 * 1e3
            cull = {"render_count": int(Rc), "leaves": int(tree.info().leaves), "ms_per_frame": round(cull_ms, 4),
                    "Msplats_per_s_scene": round(N / (cull_ms * 1e-3) / 1e6, 1),
                    "Msplats_per_s_rendered": round(Rc / (cull_ms * 1e-3) / 1e6, 1), "tree_build_s": round(t_tree, 2),
                    "note": "gather + sort + draw of the frustum-culled list; scene = all N splats per frame, "
                            "rendered = R kept by the cull"}
            tree.dispose()
            mesh.use_sorter_result(worker, N)

        # third column: the per-splat frustum cull fused into pass 0 of the sort (gs_sorter_set_frustum_cull).  Keys,
        # range and buckets still span all N splats, so the frame must be bit-identical to the headline path's
        fused = None
        if world == 1 and not args.no_cull:
            frame()
            torch.cuda.synchronize()
            ref_img = strip.clone()
            worker.set_frustum_cull(True)
            for _ in range(3):
                frame()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                frame()
            f_enq = (time.perf_counter() - t1) / args.steps * 1e3
            torch.cuda.synchronize()
            f_ms = (time.perf_counter() - t1) / args.steps * 1e3
            fs, _ = worker.last_stats()
            fused = {"kept": int(fs.result_count), "ms_per_frame": round(f_ms, 4),
                     "Msplats_per_s_scene": round(N / (f_ms * 1e-3) / 1e6, 1),
                     "host_enqueue_ms_per_frame": round(f_enq, 4), "sort_ms_last": round(float(fs.device_ms), 4),
                     "frame_identical_to_full_sort": bool(torch.equal(ref_img, strip)),
                     "note": "keys + min/max over all N, then pass 0 of the radix sort drops the splats whose centre is "
                             "outside 1.25x the clip volume; the list is the full sort's list minus those splats"}
            worker.set_frustum_cull(False)
            del ref_img

    ms_per_step = elapsed / args.steps * 1e3
    D16 = int(st_probe.tiles16)
    visible = int(st_probe.visible_splats)
    if rank == 0:
        R = Rs = N
        P = W * H
        B = frame_algorithmic_bytes(R, Rs, D16, P, scene.sh_degree, scene.cov_half)
        frame_gbs = B / (ms_per_step * 1e-3) / 1e9
        kb = project_algorithmic_bytes(N, visible, scene.sh_degree, scene.cov_half)
        k_ms = proj_ms_sum / max(proj_launches, 1)
        k_gbs = kb / (k_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic("k_project")
        out = {
            "metric": "Msplats/s sorted+rasterized at 1920x1080 SH-2" if args.config == "C3"
                      else f"Msplats/s sorted+rasterized ({cfg['label']})",
            "value": round(N / (ms_per_step * 1e-3) / 1e6, 2), "unit": "Msplats/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "fps": round(1e3 / ms_per_step, 2), "frame_latency_ms": round(float(np.median(latency)), 4),
            "host_enqueue_ms_per_frame": round(t_enqueued / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int32 keys / f32 raster",
            "data": "synthetic" if scene.name == args.config else f"file:{scene.name}",
            "config": {"workload": f"{args.config}: {cfg['label']}", "splats": N, "sh_degree": scene.sh_degree,
                       "width": W, "height": H, "cull": "off (R=N)", "sort_precision_bits": 16,
                       "parallelism": f"tile-row strips x{world}" if world > 1 else "1 GPU",
                       "strips": strips if world > 1 else None},
            # dominant kernel (largest single kernel of the frame): the vertex stage
            "roofline": {"bound": "hbm", "kernel": "k_project", "achieved": round(k_gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(k_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(kb),
                         "avg_launch_ms": round(k_ms, 5), "launches_timed": proj_launches,
                         "visible_splats": visible},
            # whole frame against SURVEY.md §8d's formula
            "frame": {"algorithmic_bytes": int(B), "bytes_per_splat": round(B / R, 1), "GBps": round(frame_gbs, 1),
                      "frac_of_hbm_peak": round(frame_gbs / HBM_PEAK_GBS, 4), "tiles16_D": D16,
                      "D_per_splat": round(D16 / R, 3), "list_entries": D32, "list_bin_px": list_px_timed,
                      "stage_ms_isolated_frame": {k: round(v, 4) for k, v in stage_ms.items()}},
            "orbit": orbit,
            "cull_on": cull,
            "frustum_cull_fused": fused,
            "cpu_baseline": None,
            "scene_gen_s": round(t_gen, 1),
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(scene, mvp, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    worker.terminate()
    mesh.dispose()
    ctx.close()


if __name__ == "__main__":
    main()
