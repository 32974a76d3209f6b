#!/bin/bash
# GPU box: regenerate everything profiles/ cites -> gpurun_out/collect/   (then copy the summaries into profiles/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/collect; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-graph --no-3d --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err
python scripts/bench_hbm.py > $O/bench_hbm.txt 2>&1
python scripts/bench_conv.py 32 > $O/bench_conv.txt 2>&1
python scripts/bench_conv3d.py > $O/bench_conv3d.txt 2>&1
DFMIR_CONV3D_FP32=1 python scripts/bench_conv3d.py > $O/bench_conv3d_fp32.txt 2>&1
python scripts/bench_warp_roofline.py > $O/bench_warp_roofline.json 2>&1
python scripts/bench_upconv3d.py > $O/bench_upconv3d.txt 2>&1
DFMIR_CONV3D_NO_UPPHASE=1 python scripts/bench_upconv3d.py >> $O/bench_upconv3d.txt 2>&1
python scripts/bench_wgrad3d.py > $O/bench_wgrad3d.txt 2>&1
DFMIR_CONV3D_WGRAD_COPIES=1 python scripts/bench_wgrad3d.py >> $O/bench_wgrad3d.txt 2>&1
DFMIR_CONV3D_WGRAD_NO_PAIR=1 python scripts/bench_wgrad3d.py >> $O/bench_wgrad3d.txt 2>&1
python scripts/launch_census.py capture_step=False > $O/launch_census.txt 2>&1
python scripts/bench_in_blurdown.py > $O/bench_in_blurdown.txt 2>&1
DFMIR_IN_BLUR_BANDED=1 python scripts/bench_in_blurdown.py >> $O/bench_in_blurdown.txt 2>&1
python scripts/conv2d_layer_census.py > $O/conv2d_layer_census.txt 2>&1
python scripts/conv3d_step_census.py > $O/conv3d_step_census.txt 2>&1
for sw in DFMIR_NO_OVERLAP_R DFMIR_CONV_CS_PLAIN NONE; do env $sw=1 python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step')"; done > $O/ab_switches.txt 2>&1
for u in mfma_peak lds_unaligned valu_under_mfma; do   # built from source on the box (binaries are not tracked)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/$u scripts/ubench/$u.hip && /tmp/$u > $O/ubench_$u.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
# the bench command itself under rocprofv3 (graph replays are traced kernel by kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-3d --roofline-steps 0 > $O/kt.log 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
python $R/scripts/step_trace.py $(ls $O/kt/*/*kernel_trace.csv | head -1) 70 > $O/step_trace.txt 2>&1
python $R/scripts/overlap_trace.py $(ls $O/kt/*/*kernel_trace.csv | head -1) > $O/overlap_trace.txt 2>&1
rm -rf $O/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python $R/scripts/bench_3d.py > $O/bench_3d.txt 2>&1
cp $(ls $O/kt3/*/*kernel_stats.csv | head -1) $O/bench_3d_kernel_stats.csv
rm -rf $O/kt3
cd $R
bash scripts/prof_conv.sh fwd 256 256 64 32 > $O/pmc_conv_fwd.txt 2>&1
bash scripts/prof_conv.sh wgrad 256 256 64 32 > $O/pmc_conv_wgrad.txt 2>&1
CELL=32 AMP=1.0 bash scripts/prof_warp.sh > $O/pmc_warp.txt 2>&1
bash scripts/prof_conv3d.sh 34-32 > $O/pmc_conv3d_34_32.txt 2>&1
bash scripts/prof_conv3d.sh 32-16 > $O/pmc_conv3d_32_16.txt 2>&1
bash scripts/prof_conv3d.sh 16-16,16-32 > $O/pmc_conv3d_march.txt 2>&1
python scripts/bench_march.py > $O/bench_march.txt 2>&1
if [ -f build/ko/libdfmir_hip_m3trace.so ]; then DFMIR_HIP_LIB=$PWD/build/ko/libdfmir_hip_m3trace.so python scripts/march_trace.py 2>&1 | grep -v amdgpu.ids > $O/march_trace.txt; fi
DFMIR_CONV3D_NO_MARCH=1 TAG=tiled python scripts/bench_march.py >> $O/bench_march.txt 2>&1
bash scripts/prof_3d_step.sh 60 > $O/step_trace_3d.txt 2>&1
bash scripts/prof_step.sh scripts/bench_3d_128.py 3d128 50 > $O/step_trace_3d_128.txt 2>&1
bash scripts/prof_upconv3d.sh > $O/pmc_upconv3d.txt 2>&1
for sw in 0 1 0 1; do if [ $sw = 1 ]; then export DFMIR_CONV3D_NO_MARCH=1; else unset DFMIR_CONV3D_NO_MARCH; fi; echo "DFMIR_CONV3D_NO_MARCH=$sw"; python scripts/bench_3d.py 2>/dev/null | cut -c1-72; done > $O/ab_march_3d.txt 2>&1; unset DFMIR_CONV3D_NO_MARCH
python scripts/bench_upwgrad.py 2>&1 | grep -v amdgpu.ids > $O/bench_upwgrad.txt
DFMIR_UPWGRAD_MIN_VOX=0 TAG=every-level python scripts/bench_upwgrad.py 2>&1 | grep -v amdgpu.ids >> $O/bench_upwgrad.txt
DFMIR_UPWGRAD_MIN_VOX=0 DFMIR_UPWGRAD_NO_FUSEB=1 TAG=skip-share-on-the-direct-kernel python scripts/bench_upwgrad.py 2>&1 | grep -v amdgpu.ids >> $O/bench_upwgrad.txt
DFMIR_UPWGRAD_MIN_VOX=0 DFMIR_UPWGRAD_NO_FUSEB=1 DFMIR_UPWGRAD_8WAVE=1 TAG=two-waves-per-SIMD python scripts/bench_upwgrad.py 2>&1 | grep -v amdgpu.ids >> $O/bench_upwgrad.txt
if [ -f build/ko/libdfmir_hip_uwko14.so ]; then rm -f gpurun_out/upwgrad_ko.txt; bash scripts/gpu_upwgrad_ko.sh - 1 2 32 8 14 15 16 > /dev/null 2>&1; cp gpurun_out/upwgrad_ko.txt $O/upwgrad_ko.txt; fi
for sw in 0 1 0 1; do if [ $sw = 1 ]; then export DFMIR_UPWGRAD_DIRECT=1; else unset DFMIR_UPWGRAD_DIRECT; fi; echo "DFMIR_UPWGRAD_DIRECT=$sw"; python scripts/bench_3d.py 2>/dev/null | cut -c1-72; done > $O/ab_upwgrad_3d.txt 2>&1; unset DFMIR_UPWGRAD_DIRECT
for sw in 0 1 0 1; do if [ $sw = 1 ]; then export DFMIR_CONV3D_NO_WGRAD_MARCH=1; else unset DFMIR_CONV3D_NO_WGRAD_MARCH; fi; echo "DFMIR_CONV3D_NO_WGRAD_MARCH=$sw"; python scripts/bench_3d.py 2>/dev/null | cut -c1-72; ONLY=32-16 python scripts/bench_conv3d.py 2>/dev/null | tail -1; done > $O/ab_wgrad_march_3d.txt 2>&1; unset DFMIR_CONV3D_NO_WGRAD_MARCH
for v in 0 2; do DFMIR_CS_XCD_PAIR=$v python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('DFMIR_CS_XCD_PAIR=$v', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step issued_frac', round(r['roofline']['issued_frac'],4))"; done > $O/ab_xcd_order.txt 2>&1
rm -rf $R/gpurun_out/conv_prof $R/gpurun_out/warp_prof $R/gpurun_out/conv3d_prof $R/gpurun_out/kt3d $R/gpurun_out/upconv_prof
TAG=${TAG:-r06}; python scripts/pmc_json.py $TAG > $O/pmc_json.log 2>&1; cp profiles/${TAG}_pmc.json $O/pmc.json
# ---- round 6: the A/Bs of this round's switches, the parity margins, the staged-step probe
for sw in NONE DFMIR_NO_STAGED NONE DFMIR_NO_STAGED; do env $sw=1 python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step  host enqueue', round(r['host_enqueue_ms_per_step'],2), 'ms  loader-fed', round(r['value_pil_loader'] or 0, 1), r['step_submission'])"; done > $O/ab_staged.txt 2>&1
for sw in NONE DFMIR_NO_1X1_WGRAD NONE DFMIR_NO_1X1_WGRAD; do env $sw=1 python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step')"; done > $O/ab_1x1_wgrad.txt 2>&1
for sw in NONE DFMIR_DETERMINISTIC_WGRAD; do env $sw=1 python bench.py --steps 20 --warmup 5 --no-3d --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', round(r['value'],1), 'pairs/s', round(r['ms_per_step'],2), 'ms/step')"; echo $sw; env $sw=1 python scripts/bench_3d.py 2>/dev/null | cut -c1-100; done > $O/ab_deterministic.txt 2>&1
for sw in 0 1 0 1; do if [ $sw = 1 ]; then export DFMIR_CONV3D_NO_FLOW_WGRAD=1; else unset DFMIR_CONV3D_NO_FLOW_WGRAD; fi; echo "DFMIR_CONV3D_NO_FLOW_WGRAD=$sw"; python scripts/bench_3d.py 2>/dev/null | cut -c1-100; ONLY=16-3 python scripts/bench_conv3d.py 2>/dev/null | tail -n 1; done > $O/ab_flow_wgrad.txt 2>&1; unset DFMIR_CONV3D_NO_FLOW_WGRAD
python scripts/graph_split_probe.py 2>&1 | grep -v amdgpu.ids > $O/graph_split_probe.txt
DFMIR_MARGINS_OUT=$O/parity_margins.txt timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_margins.txt 2>&1; tail -n 3 $O/pytest_margins.txt
python scripts/bench_conv1x1.py 2>&1 | grep -v amdgpu.ids > $O/bench_conv1x1.txt; DFMIR_NO_1X1_WGRAD=1 python scripts/bench_conv1x1.py 2>&1 | grep -v amdgpu.ids | sed 's/^/generic kernel: /' >> $O/bench_conv1x1.txt
bash scripts/prof_cmd.sh "python $R/scripts/bench_conv1x1.py" conv1x1_wgrad_k conv1x1 > $O/pmc_conv1x1.txt 2>&1
ONLY=16-3 bash scripts/prof_cmd.sh "python $R/scripts/bench_conv3d.py" conv3d_flow_wgrad_k flow > $O/pmc_flow_wgrad.txt 2>&1
rm -rf $R/gpurun_out/prof_conv1x1 $R/gpurun_out/prof_flow

# ---- round 6, second half: the 3-D step's switches (flow head on the march kernel, stride-2 levels, resize by rows) under the
# graph (bench.py's also_3d / also_3d_128), the flow head / helper micro-benchmarks, the HBM-side bytes of every kernel of the step
for sw in NONE DFMIR_CONV3D_NO_S2 DFMIR_CONV3D_NO_FLOW_MARCH DFMIR_RESIZE_NO_ROWS NONE DFMIR_CONV3D_NO_S2 DFMIR_CONV3D_NO_FLOW_MARCH DFMIR_RESIZE_NO_ROWS; do
  env $sw=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pil-workers 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw=1', '3-D 160x192x224 %.3f ms | 128^3 %.3f ms | 2-D %.2f ms' % (r['also_3d']['ms_per_step'], r['also_3d_128']['ms_per_step'], r['ms_per_step']))"
done > $O/ab_3d_round6.txt 2>&1
python scripts/bench_flow_head.py 2>&1 | grep -v amdgpu.ids > $O/bench_flow_head.txt
for sw in NONE DFMIR_SMOOTH_NO_MARCH NONE DFMIR_SMOOTH_NO_MARCH; do env $sw=1 python scripts/bench_3d.py 2>/dev/null | cut -c1-64 | sed "s/^/$sw  /"; done > $O/ab_smooth.txt; python scripts/bench_flow_smooth.py 2>&1 | tail -n 1 >> $O/ab_smooth.txt; DFMIR_SMOOTH_NO_MARCH=1 python scripts/bench_flow_smooth.py 2>&1 | tail -n 1 | sed 's/^/DFMIR_SMOOTH_NO_MARCH=1: /' >> $O/ab_smooth.txt
ONLY=160x192 bash scripts/prof_cmd.sh "python $R/scripts/bench_3d.py" "" step3d > $O/pmc_step3d.txt 2>&1; rm -rf $R/gpurun_out/prof_step3d
