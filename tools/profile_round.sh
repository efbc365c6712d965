#!/bin/bash
# GPU box: everything the judged numbers come from.
#   usage: tools/profile_round.sh <tag> [workloads...]      (default workloads: cfg2 cfg3 cfg4 cfg5; writes gpurun_out/<tag>/)
# Per workload: rocprofv3 kernel-trace statistics of the bench command, HBM traffic (FETCH_SIZE and WRITE_SIZE in SEPARATE
# PMC passes, no other trace domain), matrix-pipe / issue counters (two SQ passes), and the un-profiled bench line.
# Under rocprofv3 the program itself follows `--` (python3 bench.py ...), never a launcher.
tag=${1:-r03}
shift
wl=${@:-cfg2 cfg3 cfg4 cfg5}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
out=gpurun_out/$tag
mkdir -p $out
for w in $wl; do
  steps=5; [ "$w" = cfg2 ] && steps=20
  B="python3 bench.py --workload $w --no-cpu-baseline --no-profile --steps 2 --warmup 1"
  echo "[profile_round] $w: kernel statistics"
  rocprofv3 --kernel-trace --stats -d $out/stats_$w -o s --output-format csv -- python3 bench.py --workload $w --no-cpu-baseline --steps $steps > $out/bench_${w}_under_rocprof.json 2> $out/stats_$w.log
  echo "[profile_round] $w: FETCH_SIZE"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch_$w -o f --output-format csv -- $B > $out/fetch_$w.log 2>&1
  echo "[profile_round] $w: WRITE_SIZE"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write_$w -o w --output-format csv -- $B > $out/write_$w.log 2>&1
  echo "[profile_round] $w: SQ counters"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $out/sq_$w -o q --output-format csv -- $B > $out/sq_$w.log 2>&1
  python3 tools/traffic.py $(find $out/fetch_$w -name "*_counter_collection.csv" | head -1) $(find $out/write_$w -name "*_counter_collection.csv" | head -1) $out/traffic_$w.json $w > $out/traffic_$w.txt 2>&1
  python3 tools/mfma_util.py $out/sq_$w $out/mfma_util_$w.json $w > $out/mfma_util_$w.txt 2>&1
  cp $(find $out/stats_$w -name "*_kernel_stats.csv" | head -1) $out/bench_${w}_kernel_stats.csv 2>/dev/null
  echo "[profile_round] $w: bench line"
  if [ "$w" = cfg2 ]; then python3 bench.py > $out/bench_$w.json 2> $out/bench_$w.err; else python3 bench.py --workload $w --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err; fi
  rm -rf $out/fetch_$w $out/write_$w $out/sq_$w $out/stats_$w
done
for w in cfg1 frame1; do python3 bench.py --workload $w --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err; done
ls $out
