#!/bin/bash
# A/B on one box: bench every prebuilt library variant under audio-matcher_amd/build/variants/,
# alternating, and print the per-kernel times.  Usage: tools/ab.sh [rounds] [bench args...]
rounds=${1:-2}; shift
lib=audio-matcher_amd/libaudiomatch_amd.so
cp $lib /tmp/keep.so
for r in $(seq $rounds); do
  for v in audio-matcher_amd/build/variants/*.so; do
    cp $v $lib
    python bench.py --no-cpu-baseline --no-extra-legs --no-batch-1000 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline_pipeline']['kernel_ms_per_haystack']
print('$v'.split('/')[-1], 'ms/haystack', round(d['ms_per_step']/d['config']['haystacks_per_rank_per_step'],4), {a:round(b,4) for a,b in k.items()})"
  done
done
cp /tmp/keep.so $lib
