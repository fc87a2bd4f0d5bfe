#!/bin/bash
# builds the microbenchmarks in place (hipcc cross-compiles gfx950 without a GPU)
set -e
cd "$(dirname "$0")"
for f in diag_stream fetch_calib lse_latency valu_rate far_step lse_fast; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w -o $f $f.hip
done
