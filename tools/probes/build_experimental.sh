#!/bin/bash
# Probe build of the library WITH the kernels that never became a default (tile ids 3, 5, 8, 11; tools/probes/experimental/):
#   tools/probes/build_experimental.sh  ->  tools/probes/libmpx_experimental.so   (never the product .so)
set -e
cd "$(dirname "$0")/../../network_interpretation_imagenet_amd/csrc"
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMPX_EXPERIMENTAL -o ../../tools/probes/libmpx_experimental.so mpx_api.hip
echo built tools/probes/libmpx_experimental.so
