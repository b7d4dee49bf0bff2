#!/bin/bash
# in-network tile-rule sweep (MPX_TILE_RULES of tools/layer_profile.py), one gpurun call; prints conv ms per batch per rule
mkdir -p gpurun_out/sweep
run() { MPX_TILE_RULES="$1" timeout -k 10 120 python tools/layer_profile.py resnet101 2048 2 > gpurun_out/sweep/$2.txt 2>&1; echo "$1 -> $(tail -n 1 gpurun_out/sweep/$2.txt | cut -c1-40)   [$(grep '^rule ' gpurun_out/sweep/$2.txt | cut -d'>' -f2 | tr '\n' ';')]"; }
timeout -k 10 120 python tools/layer_profile.py resnet101 2048 2 > gpurun_out/sweep/default.txt 2>&1; echo "default -> $(tail -n 1 gpurun_out/sweep/default.txt | cut -c1-40)"
run k1exp:2 k1exp2
run k1exp:10 k1exp10
run k1red:2 k1red2
run k1red:7 k1red7
run k3s1:0 k3s1_0
run c64k3:6 c64k3_6
run c64k3:4 c64k3_4
run c64k1:1 c64k1_1
run k3s2:2 k3s2_2
run stem:4 stem4
timeout -k 10 120 python tools/layer_profile.py resnet101 2048 2 > gpurun_out/sweep/default2.txt 2>&1; echo "default -> $(tail -n 1 gpurun_out/sweep/default2.txt | cut -c1-40)"
