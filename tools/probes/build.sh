#!/bin/bash
# builds the standalone hardware probes (gfx950) next to their sources; the binaries travel to the GPU box with the snapshot
cd "$(dirname "$0")" && for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o "${f%.hip}" "$f" || exit 1; done
