#!/bin/bash
# GPU box: everything the judged numbers come from.  usage: tools/profile_round.sh <tag>   (writes gpurun_out/<tag>/)
tag=${1:-r02}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
B="python3 bench.py --no-cpu-baseline --no-profile --steps 2 --warmup 1"
# 1. kernel-trace statistics of the bench command itself (default flags) and of cfg4
rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline > $out/bench_cfg2_profiled.json 2> $out/stats.log
rocprofv3 --kernel-trace --stats -d $out/stats4 -o s --output-format csv -- python3 bench.py --workload cfg4 --no-cpu-baseline --steps 5 > $out/bench_cfg4_profiled.json 2> $out/stats4.log
# 2. HBM traffic: separate PMC passes (no other trace domain)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o f --output-format csv -- $B > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o w --output-format csv -- $B > $out/write.log 2>&1
# 3. matrix-pipe utilisation (SQ counters)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $out/sq -o q --output-format csv -- $B > $out/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $out/sq2 -o q --output-format csv -- $B > $out/sq2.log 2>&1
python3 tools/traffic.py $(ls $out/fetch/*/*_counter_collection.csv $out/fetch/*_counter_collection.csv 2>/dev/null | head -1) $(ls $out/write/*/*_counter_collection.csv $out/write/*_counter_collection.csv 2>/dev/null | head -1) $out/traffic.json > $out/traffic.txt 2>&1
python3 tools/mfma_util.py $out/sq $out/mfma_util_a.json > $out/mfma_util.txt 2>&1
python3 tools/mfma_util.py $out/sq2 $out/mfma_util_b.json >> $out/mfma_util.txt 2>&1
# 4. bench lines (un-profiled) for every workload, the default one with the CPU baseline
python3 bench.py > $out/bench_cfg2.json 2> $out/bench_cfg2.err
for w in cfg1 cfg3 cfg4 frame1 cfg5; do python3 bench.py --workload $w --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err; done
# 4b. kernel-trace statistics of the training step
rocprofv3 --kernel-trace --stats -d $out/stats5 -o s --output-format csv -- python3 bench.py --workload cfg5 --no-cpu-baseline --steps 5 --warmup 2 > $out/bench_cfg5_profiled.json 2> $out/stats5.log
# 5. microbenchmarks
tools/ubench/mfma_valu > $out/ubench_mfma_valu.txt 2>&1
tools/ubench/bf16x3 > $out/ubench_bf16x3.txt 2>&1
find $out -name "*_kernel_stats.csv" | head; ls $out
