#!/bin/bash
# A/B of prebuilt library variants on BASELINE configs[3]'s shape (tools/multi_bench.py: NN needles x
# one resident 1 h haystack), alternating on one box.  Usage: tools/ab_multi.sh [rounds] [needles] [groups]
rounds=${1:-2}; nn=${2:-32}; groups=${3:-8}
lib=audio-matcher_amd/libaudiomatch_amd.so
cp $lib /tmp/keep.so
for r in $(seq $rounds); do
  for v in audio-matcher_amd/build/variants/*.so; do
    cp $v $lib
    python3 tools/multi_bench.py $nn $groups 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for g,v in d.items(): print('$v'.split('/')[-1], g, 'ms/needle-hour', round(v['ms_per_needle_hour'],4), v['kernel_ms_per_call'])"
  done
done
cp /tmp/keep.so $lib
