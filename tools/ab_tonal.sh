#!/bin/bash
lib=audio-matcher_amd/libaudiomatch_amd.so
cp $lib /tmp/keep.so
for r in 1 2; do
  for v in audio-matcher_amd/build/variants/*.so; do
    cp $v $lib
    echo -n "$(basename $v) "; timeout -k 5 120 python3 tools/tonal_probe.py 2>/dev/null | tail -1
  done
done
cp /tmp/keep.so $lib
