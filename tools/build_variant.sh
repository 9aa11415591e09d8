#!/bin/bash
# Build the current csrc/ into variants/libcorrif_<name>.so for same-call A/B runs (CORRIF_LIB=...).
set -e
cd "$(dirname "$0")/.."
PKG=$(ls -d corrifnet*_amd)
mkdir -p variants/obj_$1
for f in $PKG/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $EXTRA -c $f -o variants/obj_$1/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libcorrif_$1.so variants/obj_$1/*.o
rm -rf variants/obj_$1
echo built variants/libcorrif_$1.so
