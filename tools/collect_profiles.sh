#!/bin/bash
# Collects the evidence kept under profiles/<round>/: the bench line of every BASELINE config, the rocprofv3
# kernel-trace stats of the same commands, and the two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate
# runs, no trace domains).  Run on the GPU box from the repo root:  tools/collect_profiles.sh gpurun_out/<tag>
set -o pipefail
out=${1:-gpurun_out/prof}
part=${2:-all}     # "a": bench lines, kernel stats, PMC passes; "b": everything else (two gpurun calls of at most 20 minutes each)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$part" != "b" ]; then
python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || exit 1
lean="--no-cpu-baseline --no-extra-legs --no-batch-1000"
python3 bench.py --config 3 $lean > "$out/bench_config3.json" 2> "$out/bench_config3.err" || exit 1
python3 bench.py --config 4 $lean > "$out/bench_config4.json" 2> "$out/bench_config4.err" || exit 1
for cfg in 2 3 4; do
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats$cfg" -o run -- python3 bench.py --config $cfg $lean > "$out/stats$cfg.log" 2>&1 || exit 1
  find "$out/stats$cfg" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_config$cfg.csv" \;
  rm -rf "$out/stats$cfg"
done
for cfg in 2 3 4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d "$out/pmc$cfg/$c" -o run -- python3 bench.py --config $cfg $lean --steps 1 --warmup 1 --ramp-steps 0 --haystacks-per-step 3 > "$out/pmc${cfg}_$c.log" 2>&1 || exit 1
  done
  python3 tools/pmc_summary.py "$out/pmc$cfg" "$out/pmc_traffic_config$cfg.json" > "$out/pmc_summary$cfg.log" 2>&1
  rm -rf "$out/pmc$cfg"
done
fi
[ "$part" = "a" ] && { echo collected part a; exit 0; }
lean="--no-cpu-baseline --no-extra-legs --no-batch-1000"
for level in 1 2; do
  python3 bench.py $lean --half-pipeline $level > "$out/bench_half$level.json" 2> "$out/bench_half$level.err" || exit 1
done
tools/sq_counters.sh "$out/sq" > /dev/null 2>&1 && cp "$out/sq/sq_counters.json" "$out/sq_counters.json"
[ -x tools/ntbench ] && tools/ntbench > "$out/ntbench_cache_policy.txt" 2>&1
# two ranks sharing the one GPU of this box: a rehearsal of the rank plumbing for every config (flagged in the line)
for cfg in 2 3 4; do
  python3 bench.py --gpus 2 --config $cfg $lean --steps 3 --warmup 1 --ramp-steps 2 --haystacks-per-step 2 > "$out/rehearsal_2_ranks_config$cfg.json" 2> "$out/rehearsal$cfg.err"
done
python3 bench.py --gpus 2 --config 3 $lean --total-haystacks 4 --steps 2 --warmup 1 > "$out/rehearsal_2_ranks_config3_strong.json" 2>> "$out/rehearsal3.err"
# BASELINE configs[3] and [4] at their stated batch (1000 haystacks) on this one GPU, strong-scaling form, one pass
python3 bench.py --config 3 $lean --total-haystacks 1000 --steps 1 --warmup 0 > "$out/strong_1000_config3_1_gpu.json" 2> "$out/strong3.err"
python3 bench.py --config 4 $lean --total-haystacks 1000 --steps 1 --warmup 0 > "$out/strong_1000_config4_1_gpu.json" 2> "$out/strong4.err"
# ONE long haystack split by window ranges: 8 h on this GPU, and the two-rank rehearsal of the rank plumbing
python3 bench.py --long-haystack 8 --steps 5 --warmup 2 > "$out/long_haystack_8h_1_gpu.json" 2> "$out/long1.err"
python3 bench.py --gpus 2 --long-haystack 8 --steps 5 --warmup 2 > "$out/rehearsal_2_ranks_long_haystack_8h.json" 2> "$out/long2.err"
[ -x audio-matcher_amd/bin/pushbench ] && audio-matcher_amd/bin/pushbench > "$out/host_feed.json" 2> "$out/host_feed.err"
echo collected
