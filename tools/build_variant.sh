#!/bin/bash
# Development aid: builds libtilemotion with extra -D flags into tiler_amd/lib/variants/libtilemotion_<name>.so (loaded with
# TM_LIB_VARIANT=<name>), so that kernel-shape experiments (TM_KNN_NW, TM_KNN_NQ, TM_KNN_NBUF, TM_KNN_OCC ...) can be A/B-ed in one GPU call.
set -euo pipefail
NAME="$1"; shift
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../tiler_amd/csrc" && pwd)"
OUT="$HERE/../lib/variants"; OBJ="$HERE/.obj_$NAME"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall -Wno-unused-function $*"
pids=(); objs=()
for src in "$HERE"/*.hip; do
  obj="$OBJ/$(basename "${src%.hip}").o"; objs+=("$obj")
  case "$(basename "$src")" in
    tm_knn*.hip|tm_kmeans.hip|tm_features.hip|tm_kmodes.hip) $HIPCC $FLAGS -c "$src" -o "$obj" & pids+=($!) ;;
    *) cp "$HERE/.obj/$(basename "${src%.hip}").o" "$obj" ;;   # unaffected by the knn shape macros
  esac
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtilemotion_$NAME.so" "${objs[@]}" -L/opt/rocm/lib -lrccl
rm -rf "$OBJ"
echo "built $OUT/libtilemotion_$NAME.so"
