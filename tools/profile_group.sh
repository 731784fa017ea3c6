#!/bin/bash
# Evidence for the needle-group row kernel (k2_rows_r16_group, BASELINE configs[3]): rocprofv3
# kernel-trace stats, FETCH_SIZE / WRITE_SIZE and SQ counters of tools/multi_bench.py (separate --pmc
# passes, no trace domains).  Run on the GPU box from the repo root: tools/profile_group.sh gpurun_out/<tag>
set -o pipefail
out=${1:-gpurun_out/group}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/multi_bench.py 32 8 > "$out/multi_needle_32.json" 2> "$out/multi32.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 tools/multi_bench.py 32 8 > "$out/stats.log" 2>&1 || exit 1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_multi_needle_32.csv" \;
rm -rf "$out/stats"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc/$c" -o run -- python3 tools/multi_bench.py 32 8 --lean > "$out/pmc_$c.log" 2>&1 || exit 1
done
python3 tools/pmc_summary.py "$out/pmc" "$out/pmc_traffic_multi_needle_32.json" > "$out/pmc_summary.log" 2>&1
rm -rf "$out/pmc"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/sq/pass$i" -o run -- python3 tools/multi_bench.py 32 8 --lean > "$out/sq_pass$i.log" 2>&1 || exit 1
done
python3 tools/sq_summary.py "$out/sq" "$out/sq_counters_multi_needle_32.json" > "$out/sq_summary.log" 2>&1
rm -rf "$out/sq"
echo collected
