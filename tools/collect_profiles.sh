#!/bin/bash
# Round artefacts on the GPU box: bench line, rocprofv3 kernel summary, PMC traffic passes.  usage: tools/collect_profiles.sh <tag>
set -o pipefail
tag=${1:-x}
R=/root/repo
out=$R/gpurun_out
mkdir -p $out
timeout -k 10 400 python $R/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { echo "bench failed"; tail -5 $out/${tag}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o k -- python3 $R/bench.py --cpu-frames 0 --no-large-kernels > $out/${tag}_prof_bench.json 2> $out/${tag}_prof.err || { echo "kernel trace failed"; tail -5 $out/${tag}_prof.err; exit 1; }
db=$(ls $out/prof_$tag/*/k_results.db $out/prof_$tag/k_results.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_summary.py $db > $out/${tag}_kernel_stats.txt || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_f_$tag -o f -- python3 $R/bench.py --cpu-frames 0 --steps 40 --warmup 5 --no-large-kernels --no-h2d > /dev/null 2> $out/${tag}_pmc_f.err || { echo "pmc fetch failed"; tail -5 $out/${tag}_pmc_f.err; exit 1; }
db=$(ls $out/pmc_f_$tag/*/f_results.db $out/pmc_f_$tag/f_results.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_pmc.py $db > $out/${tag}_pmc_fetch_size.txt || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_w_$tag -o w -- python3 $R/bench.py --cpu-frames 0 --steps 40 --warmup 5 --no-large-kernels --no-h2d > /dev/null 2> $out/${tag}_pmc_w.err || { echo "pmc write failed"; tail -5 $out/${tag}_pmc_w.err; exit 1; }
db=$(ls $out/pmc_w_$tag/*/w_results.db $out/pmc_w_$tag/w_results.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_pmc.py $db > $out/${tag}_pmc_write_size.txt || exit 1
rm -rf $out/prof_$tag $out/pmc_f_$tag $out/pmc_w_$tag
head -14 $out/${tag}_kernel_stats.txt
cut -c1-400 $out/${tag}_bench.json
