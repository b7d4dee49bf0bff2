#!/bin/bash
# Time every distinct ResNet-101 conv shape with the kernel tile variants 0,1,2,4,7 (run on the GPU box).
B=${1:-2048}
for L in conv1 layer1.0.conv1 layer1.0.conv2 layer1.0.conv3 layer1.0.downsample.0 layer1.1.conv1 layer2.0.conv1 layer2.0.conv2 layer2.0.conv3 layer2.0.downsample.0 layer2.1.conv1 layer2.1.conv2 layer3.0.conv1 layer3.0.conv2 layer3.0.downsample.0 layer3.5.conv1 layer3.5.conv2 layer3.5.conv3 layer4.0.conv1 layer4.0.conv2 layer4.0.downsample.0 layer4.1.conv1 layer4.1.conv2 layer4.1.conv3; do
  python tools/conv_bench.py resnet101 $L $B 10 0,1,2,4,7 2>&1 | grep -v amdgpu.ids
done
