#!/bin/bash
# Build libskm.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# One object per source under csrc/obj/, recompiled only when the source or a header is newer.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libskm.so"
OBJ="${HERE}/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"${HERE}/../../include")
mkdir -p "${OBJ}"
newest_hdr="$(ls -t "${HERE}"/*.h "${HERE}/../../include"/*.h | head -n 1)"
objs=()
pids=()
for src in "${HERE}"/skm_*.hip; do
    o="${OBJ}/$(basename "${src%.hip}").o"
    objs+=("${o}")
    if [ ! -f "${o}" ] || [ "${src}" -nt "${o}" ] || [ "${newest_hdr}" -nt "${o}" ] || [ "${BASH_SOURCE[0]}" -nt "${o}" ]; then
        "${HIPCC}" "${FLAGS[@]}" -c "${src}" -o "${o}" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do
    [ -n "${p}" ] && wait "${p}"
done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "${OUT}"
echo "built ${OUT}"
# static check of the generated ISA for the wait states around the asm MFMAs and the LDS-direct loads
# (tools/isa_lint.py; about a minute).  SKM_SKIP_LINT=1 skips it, e.g. while iterating on a timing variant.
if [ "${SKM_SKIP_LINT:-0}" != "1" ]; then
    python3 "${HERE}/../../tools/isa_lint.py" "${OUT}"
fi
