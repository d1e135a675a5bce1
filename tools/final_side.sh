#!/bin/bash
# The round's side measurements in one go (run on the GPU box from the repo root); everything lands under gpurun_out/final/ and
# tools/collect_final.py copies what is to be judged into profiles/.
F=$PWD/gpurun_out/final
mkdir -p "$F"
make -s -C sfm-gms_amd/csrc libgms_hip_diag.so > "$F/diag_build.log" 2>&1
python3 tools/measure_misc.py > "$F/misc.json" 2> "$F/misc.err" && echo misc ok
python3 tools/config4_bench.py > "$F/config4_bench.txt" 2>&1 && cp gpurun_out/config4_bench.json "$F/config4_bench.json" && echo config4 ok
bash tools/config4_prof.sh > "$F/config4_prof.txt" 2>&1 && cp gpurun_out/prof4/kernel_stats_rot_scale_64.csv gpurun_out/prof4/kernel_stats_default_256.csv "$F/" && echo prof4 ok
python3 tools/stream_phase_timing.py > "$F/stream_phase.json" 2> "$F/stream_phase.err" && echo stream phase ok
python3 tools/phase_timing.py --pairs 1024 > "$F/phase.json" 2> "$F/phase.err" && echo phase ok
python3 tools/phase_timing.py --pairs 512 --rot 1 --scale 1 > "$F/phase_rs.json" 2>> "$F/phase.err" && echo phase rs ok
GMS_SCALE_PROBE=0 python3 tools/phase_timing.py --pairs 512 --rot 1 --scale 1 > "$F/phase_rs_noprobe.json" 2>> "$F/phase.err" && echo phase rs noprobe ok
for k in orb sift; do bash tools/bf_pmc.sh $k > "$F/bf_pmc_$k.txt" 2>&1 && cp gpurun_out/bf_pmc_$k/summary.csv "$F/matcher_counters_$k.csv" && echo bf_pmc $k ok; done
{ echo "== tools/bf_overhead.py"; python3 tools/bf_overhead.py orb 2>/dev/null; python3 tools/bf_overhead.py sift 2>/dev/null;
  echo "== tools/ubench/mfma_fp4_rate.hip"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/mfma_fp4_rate.hip -o /tmp/mfma_fp4_rate 2>/dev/null && /tmp/mfma_fp4_rate;
  echo "== BF_ZERO=1 tools/bf_bench.py orb 1024 (all-zero descriptors)"; BF_ZERO=1 python3 tools/bf_bench.py orb 1024 2>/dev/null;
  echo "== tools/bf_bench.py orb 1024"; python3 tools/bf_bench.py orb 1024 2>/dev/null; } > "$F/matcher_limits.txt" 2>&1 && echo matcher limits ok
python3 tools/gms_image_pair.py --check > "$F/image_pair_sparse.json" 2>/dev/null && python3 tools/gms_image_pair.py --dense --check > "$F/image_pair_dense.json" 2>/dev/null && echo image pair ok
python3 tools/crowded_bench.py > "$F/crowded.json" 2>/dev/null && echo crowded ok
# round 4: the host-pointer batch through the C ABI from C++ with the box's copy ceilings beside it, the real-pixel scenario, the scale
# probes per scale hypothesis
{ /opt/rocm/bin/hipcc -O2 -pthread tools/ubench/host_copy_rate.cpp -o /tmp/host_copy_rate 2>/dev/null && /tmp/host_copy_rate > "$F/host_copy_rate.json"; } && echo host copy rate ok
{ /opt/rocm/bin/hipcc -O2 -pthread -Iinclude tools/ubench/host_batch_rate.cpp -Lsfm-gms_amd/csrc -lgms_hip -Wl,-rpath,$PWD/sfm-gms_amd/csrc -o /tmp/host_batch_rate 2>/dev/null && /tmp/host_batch_rate > "$F/host_batch_rate.json"; } && echo host batch rate ok
python3 tools/host_batch_bench.py > "$F/host_batch_python.json" 2>/dev/null && echo host batch python ok
python3 tools/real_pixels_bench.py > "$F/real_pixels.json" 2>/dev/null && echo real pixels ok
GMS_DEAL=0 python3 tools/real_pixels_bench.py > "$F/real_pixels_list_order.json" 2>/dev/null && echo real pixels undealt ok
python3 tools/ab_scales.py --zoom --rounds 1 default:libgms_hip.so byte_probes:libgms_hip.so:GMS_PROBE_NIBBLE=0 > "$F/ab_scales.txt" 2>&1 && echo ab scales ok
