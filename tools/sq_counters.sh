#!/bin/bash
# SQ counters of the pipeline kernels (three separate --pmc passes of a short bench run, no trace
# domains), summarised per full-size launch by tools/sq_summary.py.  Run on the GPU box from the
# repo root:  tools/sq_counters.sh gpurun_out/<tag>
set -o pipefail
out=${1:-gpurun_out/sq}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
lean="--no-cpu-baseline --no-extra-legs --no-batch-1000 --steps 1 --warmup 1 --ramp-steps 0 --haystacks-per-step 3"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$out/pass$i" -o run -- python3 bench.py $lean $EXTRA > "$out/pass$i.log" 2>&1 || exit 1
done
python3 tools/sq_summary.py "$out" "$out/sq_counters.json" > "$out/sq_summary.log" 2>&1
rm -rf "$out"/pass[0-9]
echo collected
