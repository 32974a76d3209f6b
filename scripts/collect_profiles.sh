#!/bin/bash
# GPU box: regenerate everything profiles/ cites -> gpurun_out/collect/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/collect; rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python scripts/bench_hbm.py > $O/bench_hbm.txt 2>&1
python scripts/bench_conv.py 32 > $O/bench_conv.txt 2>&1
for u in mfma_peak lds_unaligned valu_under_mfma; do   # built from source on the box (binaries are not tracked)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/$u scripts/ubench/$u.hip && /tmp/$u > $O/ubench_$u.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-3d > $O/kt.log 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
cd $R
bash scripts/prof_conv.sh fwd 256 256 64 32 > $O/pmc_conv_fwd.txt 2>&1
bash scripts/prof_conv.sh wgrad 256 256 64 32 > $O/pmc_conv_wgrad.txt 2>&1
CELL=32 AMP=1.0 bash scripts/prof_warp.sh > $O/pmc_warp.txt 2>&1
rm -rf $O/kt
