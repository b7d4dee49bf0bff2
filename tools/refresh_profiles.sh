#!/bin/bash
# Regenerate the evidence under profiles/ on the GPU box (one gpurun call); outputs land in gpurun_out/refresh/ and are
# copied into profiles/ afterwards by hand (gpurun_out/ is scratch).
# usage: tools/refresh_profiles.sh <round-tag>      e.g. r03
set -e -o pipefail
R=${1:-r03}
B=${2:-2340}          # forward batch of the per-layer tables and the PMC passes = what bench.py packs at (engine.whole_round_batch(2400))
O=gpurun_out/refresh
mkdir -p $O
export TMPDIR=/tmp
python tools/layer_profile.py resnet101 $B 3 > $O/${R}_layers_resnet101_b$B.txt 2>&1
MPX_FUSION_MASK=1 python tools/layer_profile.py resnet101 $B 3 > $O/${R}_layers_resnet101_b${B}_no_block_tails.txt 2>&1
python tools/layer_profile.py resnet18 $B 3 > $O/${R}_layers_resnet18_b$B.txt 2>&1
echo "layers done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --images 32 --steps 1 --warmup 1 --cpu-masks 0 > $O/${R}_bench_images32.json 2> $O/rocprof_stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | sed -n 1p) $O/${R}_rocprofv3_kernel_stats_bench_images32_steps1.csv
echo "stats done"
for A in resnet101 resnet18; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$A -- python3 tools/layer_profile.py $A $B 1 > $O/pmc_fetch_$A.log 2>&1
  echo "fetch pass $A done"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$A -- python3 tools/layer_profile.py $A $B 1 > $O/pmc_write_$A.log 2>&1
  echo "write pass $A done"
done
python tools/pmc_traffic.py $O/pmc_fetch_resnet101 $O/pmc_write_resnet101 resnet101 $B 3 > $O/${R}_pmc_traffic_b$B.json
python tools/pmc_traffic.py $O/pmc_fetch_resnet18 $O/pmc_write_resnet18 resnet18 $B 3 > $O/${R}_pmc_traffic_resnet18_b$B.json
echo "pmc done"
cp $O/${R}_pmc_traffic_b$B.json $O/${R}_pmc_traffic_resnet18_b$B.json profiles/      # bench.py reads roofline.traffic from profiles/
# power and clock next to the bench (one sample per second while it runs)
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|mclk|Power \(W\)" | tr '\n' ';'; echo; sleep 1; done ) > $O/${R}_rocm_smi_during_bench.txt &
SMI=$!
python bench.py > $O/${R}_bench_n1.json 2> $O/bench.err
kill $SMI
cat $O/${R}_bench_n1.json
python bench.py --arch resnet18 --images 32 --masks 256 --steps 8 --warmup 2 --cpu-masks 0 > $O/${R}_bench_cfg2_resnet18.json 2> $O/bench18.err
cat $O/${R}_bench_cfg2_resnet18.json
python bench.py --force-dist --images 16 --steps 1 --warmup 1 --cpu-masks 0 > $O/${R}_force_dist_rehearsal.txt 2>&1
tail -2 $O/${R}_force_dist_rehearsal.txt
python tools/latency_bench.py resnet101 10 2>&1 | grep -v amdgpu.ids > $O/${R}_latency_cfg5.txt
cat $O/${R}_latency_cfg5.txt
( python tools/api_throughput.py resnet101 32 $B noise; python tools/api_throughput.py resnet101 32 $B blobs ) 2>&1 | grep -v amdgpu.ids > $O/${R}_api_throughput.txt
python tools/probes/btail_phases.py $B 2>&1 | grep -v amdgpu.ids > $O/${R}_btail_phases.txt
rm -rf $O/stats $O/pmc_fetch_* $O/pmc_write_*
