#!/bin/bash
# Build an experimental variant of the library: tools/build_variant.sh NAME "-DFOO -DBAR" -> vae_tagger_amd/csrc/exp/libvt_NAME.so
# (select it with VAE_TAGGER_HIP_LIB=$PWD/vae_tagger_amd/csrc/exp/libvt_NAME.so; exp/ and *.so are git-ignored but travel with gpurun)
set -e
cd "$(dirname "$0")/../vae_tagger_amd/csrc"
name=$1; defs=$2
mkdir -p exp/obj_$name
pids=()
for f in $(sed -n 's/^SRCS = //p' Makefile | sed 's/\.hip//g'); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $defs -c $f.hip -o exp/obj_$name/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "build_variant: a translation unit failed to compile" >&2; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o exp/libvt_$name.so exp/obj_$name/*.o
rm -rf exp/obj_$name
echo built exp/libvt_$name.so
