#!/bin/bash
# Builds a variant of libfrx.so into ab_<name>/ (a package root that scripts/r03/ab_libs.py can load next to others: builds are compared in
# alternating processes on ONE box, because boxes differ by 3-5 %).
#   scripts/r04/make_variant.sh <name> [git-ref | "." for the working tree] [extra hipcc flags, e.g. -DFRX_HS_CHUNK=8]
set -e
NAME=$1; REF=${2:-.}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
DST=$ROOT/ab_$NAME
rm -rf "$DST"; mkdir -p "$DST"
if [ "$REF" = "." ]; then SRC=$ROOT; else SRC=/tmp/frx_variant_$NAME; rm -rf "$SRC"; git -C "$ROOT" worktree prune; git -C "$ROOT" worktree add -f --detach "$SRC" "$REF" > /dev/null 2>&1; fi
BUILD=/tmp/frx_build_$NAME; rm -rf "$BUILD"; mkdir -p "$BUILD/fast-racing_amd" "$BUILD/include"
cp -r "$SRC/fast-racing_amd/csrc" "$BUILD/fast-racing_amd/csrc"; cp "$SRC"/include/*.h* "$BUILD/include/"
rm -f "$BUILD"/fast-racing_amd/csrc/*.o
make -C "$BUILD/fast-racing_amd/csrc" DEVFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC $*" > "$BUILD/make.log" 2>&1 || { tail -20 "$BUILD/make.log"; exit 1; }
mkdir -p "$DST/fast-racing_amd"
cp "$BUILD/fast-racing_amd/libfrx.so" "$DST/fast-racing_amd/"
cp "$SRC"/fast-racing_amd/*.py "$DST/fast-racing_amd/"; cp "$SRC/frx_import.py" "$DST/"
[ "$REF" = "." ] || git -C "$ROOT" worktree remove --force "$SRC"
echo "built $DST ($(stat -c %s "$DST/fast-racing_amd/libfrx.so") bytes) from $REF flags: $*"
