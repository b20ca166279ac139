#!/bin/bash
# rocprofv3 kernel statistics of the 256x256x16 configuration (whole step as one hipGraph) and of --loss wgan-gp (VERDICT r3:
# those two bench lines had no kernel-trace CSV beside them)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_c2c4; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o bench -- python3 $R/bench.py --height 256 --width 256 --steps 10 --warmup 6 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/c2.json 2> $O/c2.err || { tail -5 $O/c2.err; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -o bench -- python3 $R/bench.py --loss wgan-gp --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-host-floor > $O/c4.json 2> $O/c4.err || { tail -5 $O/c4.err; exit 1; }
cd $R
cp $(find $O/c2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c2.csv
cp $(find $O/c4 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c4.csv
rm -rf $O/c2 $O/c4
head -8 $O/kernel_stats_c2.csv | cut -c1-160; cat $O/c2.json | cut -c1-200
