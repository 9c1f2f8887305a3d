#!/bin/bash
# rocprofv3 kernel summary + PMC traffic passes of the config-3 kernels.  usage: tools/collect_decode_profiles.sh <tag>
set -o pipefail
tag=${1:-x}
R=/root/repo
out=$R/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/dprof_$tag -o k -- python3 $R/tools/bench_decode_kernel.py 16 > $out/${tag}_decode_run.txt 2> $out/${tag}_dprof.err || { echo "kernel trace failed"; tail -5 $out/${tag}_dprof.err; exit 1; }
db=$(ls $out/dprof_$tag/*/k_results.db $out/dprof_$tag/k_results.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_summary.py $db > $out/${tag}_decode_kernel_stats.txt || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $out/dpmc_f_$tag -o f -- python3 $R/tools/bench_decode_kernel.py 3 > /dev/null 2> $out/${tag}_dpmc_f.err || { echo "pmc fetch failed"; tail -5 $out/${tag}_dpmc_f.err; exit 1; }
db=$(ls $out/dpmc_f_$tag/*/f_results.db $out/dpmc_f_$tag/f_results.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_pmc.py $db > $out/${tag}_decode_pmc_fetch_size.txt || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $out/dpmc_w_$tag -o w -- python3 $R/tools/bench_decode_kernel.py 3 > /dev/null 2> $out/${tag}_dpmc_w.err || { echo "pmc write failed"; tail -5 $out/${tag}_dpmc_w.err; exit 1; }
db=$(ls $out/dpmc_w_$tag/*/w_results.db $out/dpmc_w_$tag/w_results.db 2>/dev/null | head -1)
python3 $R/tools/rocpd_pmc.py $db > $out/${tag}_decode_pmc_write_size.txt || exit 1
rm -rf $out/dprof_$tag $out/dpmc_f_$tag $out/dpmc_w_$tag
head -8 $out/${tag}_decode_kernel_stats.txt
grep -E "bump|counter" $out/${tag}_decode_pmc_fetch_size.txt $out/${tag}_decode_pmc_write_size.txt
