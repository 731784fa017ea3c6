#!/bin/bash
set -o pipefail
out=gpurun_out/pmc_one; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
lean="--no-cpu-baseline --no-extra-legs --no-batch-1000"
for cfg in ${1:-3}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d "$out/pmc$cfg/$c" -o run -- python3 bench.py --config $cfg $lean --steps 1 --warmup 1 --ramp-steps 0 --haystacks-per-step 3 > "$out/pmc${cfg}_$c.log" 2>&1 || exit 1
  done
  python3 tools/pmc_summary.py "$out/pmc$cfg" "$out/pmc_traffic_config$cfg.json" > "$out/pmc_summary$cfg.log" 2>&1
  rm -rf "$out/pmc$cfg"
done
cat $out/pmc_traffic_config*.json
