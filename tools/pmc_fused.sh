#!/bin/bash
# Diagnostic (GPU box): SQ counters + phase stamps of the level-0 fused kernels.  usage: tools/pmc_fused.sh <tag>
set -e
tag=${1:-x}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $out/a -o a --output-format csv -- python3 bench.py --no-cpu-baseline --no-profile --steps 1 --warmup 1 > $out/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE -d $out/b -o b --output-format csv -- python3 bench.py --no-cpu-baseline --no-profile --steps 1 --warmup 1 > $out/b.log 2>&1
python3 tools/pmc_summary.py $out/a kernel > $out/summary_a.txt
python3 tools/pmc_summary.py $out/b kernel > $out/summary_b.txt
grep -A12 "attn_front_kernel\|ffn_fused_kernel" $out/summary_a.txt $out/summary_b.txt | head -120
