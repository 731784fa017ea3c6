#!/bin/bash
# Builds a variant of the library for tools/ab.sh: am_fft.hip recompiled with extra -D flags, the other
# objects taken from the last regular build.  Usage: tools/build_variant.sh <name> [-DAM_...=...]...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
am=$root/audio-matcher_amd
mkdir -p $am/build/variants /tmp/am_variants
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -c $am/csrc/am_fft.hip -o /tmp/am_variants/$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $am/build/variants/$name.so /tmp/am_variants/$name.o $am/build/am_peaks.hip.o $am/build/am_api.hip.o
echo built $am/build/variants/$name.so
