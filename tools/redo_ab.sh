#!/bin/bash
for r in 1 2 3; do for v in 1 0; do python3 bench.py --no-cpu-baseline --no-extra-legs --no-batch-1000 --opt device_redo=$v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline_pipeline']['kernel_ms_per_haystack']
print('device_redo=$v', 'ms/haystack', round(d['ms_per_step']/d['config']['haystacks_per_rank_per_step'],4), {a:round(b,4) for a,b in k.items()})"; done; done
