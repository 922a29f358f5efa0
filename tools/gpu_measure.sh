#!/bin/bash
# Measurement-only GPU session: bench line, rocprofv3 kernel stats, ABN microbench + HBM counters (no tests).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R; export MIOPEN_LOG_LEVEL=3
(timeout 500 python bench.py) > $O/bench.json 2> $O/bench.err
timeout 200 python tools/abn_microbench.py 20 > $O/abn_microbench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-pairwise-sweep > $O/prof_bench.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_abn -o abn -- python $R/tools/abn_microbench.py 10 > $O/prof_abn.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o abn -- python $R/tools/abn_microbench.py 3 > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o abn -- python $R/tools/abn_microbench.py 3 > $O/pmc_write.log 2>&1
cd $R; cat $O/bench.json; tail -2 $O/bench.err
