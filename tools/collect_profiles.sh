#!/bin/bash
# Collects the evidence kept under profiles/: bench line, rocprofv3 kernel-trace stats of the same
# command, and the two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace domains).
# Run on the GPU box from the repo root:  tools/collect_profiles.sh gpurun_out/<tag>
set -o pipefail
out=${1:-gpurun_out/prof}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || exit 1
lean="--no-cpu-baseline --no-extra-legs --no-batch-1000"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 bench.py $lean > "$out/stats.log" 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc/$c" -o run -- python3 bench.py $lean --steps 1 --warmup 1 --ramp-steps 0 --haystacks-per-step 3 > "$out/pmc_$c.log" 2>&1 || exit 1
done
python3 tools/pmc_summary.py "$out/pmc" "$out/pmc_traffic.json" > "$out/pmc_summary.log" 2>&1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
rm -rf "$out/stats" "$out/pmc"
for level in 1 2; do
  python3 bench.py $lean --half-pipeline $level > "$out/bench_half$level.json" 2> "$out/bench_half$level.err" || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_h2" -o run -- python3 bench.py $lean --half-pipeline 2 > "$out/stats_h2.log" 2>&1 || exit 1
find "$out/stats_h2" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_half_level2.csv" \;
rm -rf "$out/stats_h2"
tools/sq_counters.sh "$out/sq" > /dev/null 2>&1 && cp "$out/sq/sq_counters.json" "$out/sq_counters.json"
python3 tools/extra_bench.py > "$out/extra.json" 2> "$out/extra.err"
python3 tools/multi_bench.py 32 8 > "$out/multi32.json" 2> "$out/multi32.err"
echo collected
