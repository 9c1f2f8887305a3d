#!/bin/bash
# rocprofv3 kernel summaries of the multi-GPU modes at world size 1 on the GPU box: tools/collect_mode_profiles.sh <tag>
set -o pipefail
tag=${1:-x}
R=/root/repo
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in planes rows; do
  extra=""; [ $mode = rows ] && extra="--exchange peer"
  timeout -k 10 200 python3 $R/bench.py --mode $mode $extra --steps 100 --warmup 10 > $out/${tag}_${mode}.json 2> $out/${tag}_${mode}.err || { echo "$mode bench failed"; tail -5 $out/${tag}_${mode}.err; exit 1; }
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof_${tag}_$mode -o k -- python3 $R/bench.py --mode $mode $extra --steps 100 --warmup 10 > /dev/null 2> $out/${tag}_${mode}_prof.err || { echo "$mode trace failed"; tail -5 $out/${tag}_${mode}_prof.err; exit 1; }
  db=$(ls $out/prof_${tag}_$mode/*/k_results.db $out/prof_${tag}_$mode/k_results.db 2>/dev/null | head -1)
  python3 $R/tools/rocpd_summary.py $db > $out/${tag}_${mode}_kernel_stats.txt || exit 1
  rm -rf $out/prof_${tag}_$mode
  head -8 $out/${tag}_${mode}_kernel_stats.txt
done
