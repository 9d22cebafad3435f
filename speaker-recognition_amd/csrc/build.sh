#!/bin/bash
# Build libskm.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libskm.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
SRCS=("${HERE}"/skm_*.hip)
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"${HERE}/../../include" \
    "${SRCS[@]}" -o "${OUT}"
echo "built ${OUT}"
