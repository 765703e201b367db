#!/bin/bash
# Build a variant of libfhelin_amd.so for A/B measurements: tools/build_variant.sh <name> [git-rev] [-DFLAG ...]
#   copies fhe-linformer_amd/csrc (from <git-rev> if given, else the working tree) to tmp_variants/<name>/csrc and builds
#   tmp_variants/<name>/libfhelin_amd.so; run with FHELIN_LIB=$PWD/tmp_variants/<name>/libfhelin_amd.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
rev=""
if [ -n "$1" ] && [[ "$1" != -* ]]; then rev="$1"; shift; fi
dst="$ROOT/tmp_variants/$name"
rm -rf "$dst"; mkdir -p "$dst/csrc" "$dst/include"
if [ -n "$rev" ]; then
  git -C "$ROOT" archive "$rev" fhe-linformer_amd/csrc include | tar -x -C "$dst" --strip-components=0
  mv "$dst/fhe-linformer_amd/csrc/"* "$dst/csrc/"; rm -rf "$dst/fhe-linformer_amd"
else
  cp "$ROOT"/fhe-linformer_amd/csrc/*.{hip,cpp,h} "$ROOT"/fhe-linformer_amd/csrc/Makefile "$dst/csrc/"
  cp "$ROOT"/include/*.h "$dst/include/"
fi
# the Makefile refers to ../../include and writes ../libfhelin_amd.so
mkdir -p "$dst/x"; mv "$dst/csrc" "$dst/x/csrc"; mv "$dst/include" "$dst/include_tmp"
mkdir -p "$dst/x"; ln -sfn "$dst/include_tmp" "$dst/include"
( cd "$dst/x/csrc" && sed -i 's#\.\./\.\./include#../../include#' Makefile && make -s -j8 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*" )
mv "$dst/x/libfhelin_amd.so" "$dst/libfhelin_amd.so"
rm -rf "$dst/x/csrc"/*.o
echo "built $dst/libfhelin_amd.so"
