#!/bin/bash
# Build an experimental variant of the library: tools/build_variant.sh NAME "-DFOO -DBAR" -> vae_tagger_amd/csrc/exp/libvt_NAME.so
set -e
cd "$(dirname "$0")/../vae_tagger_amd/csrc"
name=$1; defs=$2
mkdir -p exp/obj_$name
for f in $(sed -n 's/^SRCS = //p' Makefile | sed 's/\.hip//g'); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $defs -c $f.hip -o exp/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libvt_$name.so exp/obj_$name/*.o
rm -rf exp/obj_$name
echo built exp/libvt_$name.so
