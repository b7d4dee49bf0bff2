#!/bin/bash
# Regenerate the evidence under profiles/ on the GPU box (one gpurun call); outputs land in gpurun_out/refresh/ and are
# copied into profiles/ afterwards by hand (gpurun_out/ is scratch).
# usage: tools/refresh_profiles.sh <round-tag>      e.g. r02
set -e -o pipefail
R=${1:-r02}
O=gpurun_out/refresh
mkdir -p $O
export TMPDIR=/tmp
python tools/layer_profile.py resnet101 2048 3 > $O/${R}_layers_resnet101_b2048.txt 2>&1
python tools/layer_profile.py resnet18 2048 3 > $O/${R}_layers_resnet18_b2048.txt 2>&1
echo "layers done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --images 32 --steps 1 --warmup 1 --cpu-masks 0 > $O/${R}_bench_images32.json 2> $O/rocprof_stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | sed -n 1p) $O/${R}_rocprofv3_kernel_stats_bench_images32_steps1.csv
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/layer_profile.py resnet101 2048 1 > $O/pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/layer_profile.py resnet101 2048 1 > $O/pmc_write.log 2>&1
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write resnet101 2048 3 > $O/${R}_pmc_traffic.json
echo "pmc done"
cp $O/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json      # bench.py reads roofline.traffic from profiles/
python bench.py > $O/${R}_bench_n1.json 2> $O/bench.err
cat $O/${R}_bench_n1.json
python bench.py --arch resnet18 --images 32 --masks 256 --images-per-forward 8 --steps 8 --warmup 2 --cpu-masks 0 > $O/${R}_bench_cfg2_resnet18.json 2> $O/bench18.err
cat $O/${R}_bench_cfg2_resnet18.json
python bench.py --force-dist --images 16 --steps 1 --warmup 1 --cpu-masks 0 > $O/${R}_force_dist_rehearsal.txt 2>&1
tail -2 $O/${R}_force_dist_rehearsal.txt
rm -rf $O/stats $O/pmc_fetch $O/pmc_write
