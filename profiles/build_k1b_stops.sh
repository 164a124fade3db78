#!/bin/bash
# Experiment builds of libmpe_hip.so whose blob kernel ends a frame's work after phase n (K1B_STOP_AFTER): timed one
# against the other on the GPU they give the phases' shares.  Output: build_variants/libmpe_hip_stop<n>.so (ignored by git,
# travels with gpurun).
set -e
cd "$(dirname "$0")/../rpg_monocular_pose_estimator_amd/csrc"
mkdir -p ../../build_variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
for n in ${@:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc $FLAGS -DK1B_STOP_AFTER=$n -c mpe_k1.hip -o /tmp/mpe_k1_stop$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libmpe_hip_stop$n.so /tmp/mpe_k1_stop$n.o mpe_k2.o mpe_k3.o mpe_abi.o mpe_tracker.o -ldl
done
ls -la ../../build_variants
