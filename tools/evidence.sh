#!/bin/bash
# One-box evidence set for a round tag: GPU tests, the bench line (and the lines of the other configs), rocprofv3 kernel
# stats of the headline frames, the HBM and VALU counter passes (C3 and, HBM only, C4), the blend's lane counters, the deep pass
# A/B and unit timeline, the
# vertex stage's floors, the cull-on kernel table, the per-rank cost of strip-sharded frames and a 2-rank
# dry run.  Everything lands under gpurun_out/<tag>/; copy what should be judged into profiles/.   usage: tools/evidence.sh <tag>
TAG=${1:-rXX}
cd /root/repo; mkdir -p gpurun_out/$TAG
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6) > gpurun_out/$TAG/pytest_gpu.txt
cp gpurun_out/crops_C*.json gpurun_out/whole_frame_*.json gpurun_out/$TAG/ 2>/dev/null    # (before tools/probes/rop8_arith_ab.sh redraws three of them with variant builds)
bash tools/pmc.sh $TAG hbm > /dev/null 2>&1
bash tools/pmc.sh $TAG valu > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_$TAG gpurun_out/$TAG/pmc_traffic.json > gpurun_out/$TAG/pmc_traffic.txt 2>&1
python tools/pmc_valu.py gpurun_out/pmc_$TAG gpurun_out/$TAG/pmc_valu.json > gpurun_out/$TAG/pmc_valu.txt 2>&1
# the bench reads the counter summaries from profiles/: stage this run's so that the line it prints quotes them
cp gpurun_out/$TAG/pmc_traffic.json profiles/${TAG}_pmc_traffic.json; cp gpurun_out/$TAG/pmc_valu.json profiles/${TAG}_pmc_valu.json
bash tools/prof.sh ${TAG}_kstats > gpurun_out/$TAG/kstats.txt 2>&1
cp gpurun_out/${TAG}_kstats/kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv
cp gpurun_out/${TAG}_kstats/bench_under_rocprof.json gpurun_out/$TAG/bench_under_rocprof.json
cp gpurun_out/${TAG}_kstats/kstats.json gpurun_out/$TAG/kstats.json; cp gpurun_out/$TAG/kstats.json profiles/${TAG}_kstats.json   # (the line's per-kernel sort times)
(timeout 700 python bench.py 2>gpurun_out/$TAG/bench.err | tail -1) > gpurun_out/$TAG/bench.json
for C in C2 C4 C5; do (timeout 300 python bench.py --config $C --no-cpu --no-cull --steps 30 2>/dev/null | tail -1) > gpurun_out/$TAG/bench_$C.json; done
(timeout 200 python bench.py --config C1 --steps 30 2>/dev/null | tail -1) > gpurun_out/$TAG/bench_C1.json
# C4: the configuration where the binner and the entry sort weigh most
bash tools/prof.sh ${TAG}_C4_kstats --config C4 > gpurun_out/$TAG/kstats_C4.txt 2>&1
bash tools/pmc.sh ${TAG}_C4 hbm --config C4 > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_${TAG}_C4 gpurun_out/$TAG/pmc_traffic_C4.json > gpurun_out/$TAG/pmc_traffic_C4.txt 2>&1
(GSPLAT_HIP_LIB=gaussiansplats3d_amd/csrc/libgsplat_hip_blendprof.so timeout 400 python tools/blend_lanes.py C3 C3T C2 C5 C3S 2>&1 | grep -v amdgpu.ids) > gpurun_out/$TAG/blend_lanes.txt
(python tools/project_floor.py C3; echo "-- GSPLAT_NO_BLOCK_LIST=1 (every workgroup tests its own block: round 4)"; GSPLAT_NO_BLOCK_LIST=1 python tools/project_floor.py C3; echo "-- GSPLAT_NO_BLOCK_CULL=1"; GSPLAT_NO_BLOCK_CULL=1 python tools/project_floor.py C3) 2>&1 | grep "k_project\|^--" > gpurun_out/$TAG/project_floor.txt
(GSPLAT_SERIAL=1 bash tools/prof_script.sh ${TAG}_cull 44 /root/repo/tools/cull_prof.py; grep "cull-on" gpurun_out/${TAG}_cull/log.txt) > gpurun_out/$TAG/cull_on_serial_kstats.txt 2>&1
# the deep pass on / off (same pixels), its unit timeline on the capture-like scene, and this tree against the library of the
# previous evidence set (gpurun_ab/lib_r03x.so, built by tools/build_variant.sh from that commit) in one process
(timeout 300 python tools/deep_ab.py "C3S C3T" 10 2>&1 | grep -v amdgpu.ids) > gpurun_out/$TAG/deep_ab.txt
(GSPLAT_HIP_LIB=gaussiansplats3d_amd/csrc/libgsplat_hip_blendprof.so timeout 200 python tools/blend_profile.py C3S 2>&1 | grep -v amdgpu.ids) > gpurun_out/$TAG/blend_profile_C3S.txt
# this tree against the library of the previous round (gpurun_ab/lib_r03.so: tools/build_variant.sh with GS_VARIANT_SRC = a checkout
# of round 3's last commit) in one process: whole frames, and the depth sort alone (bit-identical lists, checked against the oracle)
# (round 5: the previous round's library is gpurun_ab/lib_r04.so; $GS_EVIDENCE_PREV names it)
PREV=${GS_EVIDENCE_PREV:-gpurun_ab/lib_r03.so}
if [ -f $PREV ]; then
  cp gaussiansplats3d_amd/csrc/libgsplat_hip.so gpurun_ab/lib_this_tree.so
  (timeout 500 python tools/ab_libs.py "C3 C3T C2 C5 C4 C3S" $PREV gpurun_ab/lib_this_tree.so --frames 30 --rounds 2 2>&1 | grep -v amdgpu.ids) > gpurun_out/$TAG/ab_r03_vs_this_tree.txt
  (timeout 500 python tools/sort_ab.py "C3 C4 C2" $PREV gpurun_ab/lib_this_tree.so --check --sorts 30 --rounds 2 2>&1 | grep -v amdgpu.ids) > gpurun_out/$TAG/sort_ab_r03_vs_this_tree.txt
fi
# the depth sort alone: kernel table and counters (FETCH_SIZE / WRITE_SIZE / SQ / LDS, one set per pass)
cp gaussiansplats3d_amd/csrc/libgsplat_hip.so gpurun_ab/lib_this_tree.so
bash tools/sort_prof.sh ${TAG}_sort_k C3 gpurun_ab/lib_this_tree.so 30 > gpurun_out/$TAG/sort_kstats_C3.txt 2>&1
bash tools/sort_prof.sh ${TAG}_sort_k4 C4 gpurun_ab/lib_this_tree.so 15 > gpurun_out/$TAG/sort_kstats_C4.txt 2>&1
bash tools/sort_pmc.sh ${TAG}_sort_pmc C3 gpurun_ab/lib_this_tree.so > gpurun_out/$TAG/sort_pmc_C3.txt 2>&1
[ -z "$GS_EVIDENCE_LIGHT" ] && (tools/probes/gather_rate.bin 2>&1 | grep -v "^start") > gpurun_out/$TAG/gather_rate.txt
(python tools/cull_prof.py 2>&1 | grep cull-on) > gpurun_out/$TAG/cull_on_frame.txt
# round 6: the blend's schedule under camera motion (this tree against the previous round's library, one process), and the kernel
# table of the N = 1 visibility-culled frame (the mask derived by the sorter)
[ -f $PREV ] && (timeout 500 python tools/orbit_ab.py "C3 C2 C3S" $PREV gpurun_ab/lib_this_tree.so --rounds 2 2>&1 | grep -v amdgpu.ids) > gpurun_out/$TAG/orbit_ab.txt
bash tools/rank_prof.sh $TAG C3 1:0 > gpurun_out/$TAG/vis_cull_n1_kstats.txt 2>&1
# second half of round 6: how far the camera may move before the previous frame's bin order stops paying (orbit and turning in place:
# the gate / always the previous order / never an order), and the ROP8 draw modes with and without their own bin order
(timeout 400 python tools/motion_ab.py "C3 C2" --rounds 1 --steps "0.5 1 2 3 6"; timeout 400 python tools/motion_ab.py "C3 C2" --pan --rounds 1 --steps "0.5 1 2 4 8"; timeout 300 python tools/motion_ab.py C3S --rounds 1 --steps "1 6") 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/motion_ab.txt
(timeout 300 python tools/rop8_ab.py "C3 C2" 2>&1 | grep -v amdgpu.ids) > gpurun_out/$TAG/rop8_order_ab.txt
# end of round 6: the rounding arithmetic of the ROP8 modes (three builds: tools/probes/rop8_arith_ab.sh needs gpurun_ab/lib_rop8f{0,1,2}.so,
# built by tools/build_variant.sh rop8f<k> -DGS_ROP8_FUSED=<k>), the kernel tables of the bounded ROP8 frame and of the capture-like frame, and
# where the deep pass's workgroups sit in the blend's launch (the adopted rule against GSPLAT_DEEP_UNITS_LAST=1)
[ -f gpurun_ab/lib_rop8f0.so ] && bash tools/probes/rop8_arith_ab.sh > gpurun_out/$TAG/rop8_arith_ab.txt 2>&1
(bash tools/prof_script.sh ${TAG}_rop8 44 /root/repo/tools/rop8_prof.py C3 bounded 40; grep frame gpurun_out/${TAG}_rop8/log.txt) > gpurun_out/$TAG/kstats_rop8_bounded.txt 2>&1
(bash tools/prof_script.sh ${TAG}_c3s 47 /root/repo/tools/rop8_prof.py C3S fp32 40; grep frame gpurun_out/${TAG}_c3s/log.txt) > gpurun_out/$TAG/kstats_C3S.txt 2>&1
(L=gpurun_ab/lib_this_tree.so; timeout 500 python tools/orbit_ab.py "C3S" "GSPLAT_DEEP_UNITS_LAST=1|$L" $L --rounds 2; for e in GSPLAT_DEEP_UNITS_LAST=1 GS_NOTHING=1; do echo "== $e"; env $e timeout 300 python tools/motion_ab.py C3S --rounds 1 --steps "0.25 1 6"; done; timeout 500 python tools/ab_libs.py "C3 C3T C2 C4" "GSPLAT_DEEP_UNITS_LAST=1|$L" $L --frames 30 --rounds 2) 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/deep_unit_ab.txt
[ -n "$GS_EVIDENCE_SORT_MIDDLE" ] && (timeout 600 python tools/strip_scaling.py C5 15 sm; timeout 600 python tools/strip_scaling.py C3 20 sm) 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/sort_middle_parts.txt
(python tools/strip_scaling.py C3 20; python tools/strip_scaling.py C5 15) 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/strip_scaling.txt
# the same rank frames on the default context (streams of its own + two sets of vertex-stage outputs: what bench.py --gpus N runs)
(GS_STRIP_STREAMS=1 python tools/strip_scaling.py C3 20; GS_STRIP_STREAMS=1 python tools/strip_scaling.py C5 15) 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/strip_scaling_streams.txt
(timeout 400 python bench.py --gpus 2 --steps 10 --no-cpu --no-cull 2>/dev/null | tail -1) > gpurun_out/$TAG/bench_2ranks_dry_run.json
# forced list-bin sizes against the per-mesh rule
[ -z "$GS_EVIDENCE_LIGHT" ] && (timeout 500 python tools/list_shift_ab.py "C3 C2 C3T C5" "1 3 4 5 auto" 2>&1 | grep -v amdgpu.ids) > gpurun_out/$TAG/list_shift_ab.txt
# FILE -> native reader -> sort -> draw at BASELINE size: a 5.8 M-splat INRIA .ply staged on the box (synthetic content, real format)
[ -z "$GS_EVIDENCE_LIGHT" ] && (timeout 600 python tools/stage_ply.py C3 /tmp/gsdata 2>&1 | grep -v amdgpu.ids; GS_DATA_DIR=/tmp/gsdata timeout 500 python bench.py --only-headline --no-cpu 2>/dev/null | tail -1; rm -rf /tmp/gsdata) > gpurun_out/$TAG/bench_from_file.txt
(timeout 60 tools/probes/lookback_probe.bin 5800000 20; timeout 60 tools/probes/lookback_probe.bin 270000 20) > gpurun_out/$TAG/lookback_probe.txt 2>&1
cat gpurun_out/$TAG/pytest_gpu.txt; head -c 700 gpurun_out/$TAG/bench.json; echo; head -14 gpurun_out/$TAG/kstats.txt; cat gpurun_out/$TAG/pmc_traffic.txt | tail -3; cat gpurun_out/$TAG/strip_scaling.txt
