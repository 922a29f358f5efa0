#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel stats and HBM counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=line --timeout 600 > $O/pytest_gpu.log 2>&1
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1
(timeout 500 python bench.py) > $O/bench.json 2> $O/bench.err
timeout 200 python tools/abn_microbench.py 20 > $O/abn_microbench.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-pairwise-sweep > $O/prof_bench.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_abn -o abn -- python $R/tools/abn_microbench.py 10 > $O/prof_abn.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o abn -- python $R/tools/abn_microbench.py 3 > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o abn -- python $R/tools/abn_microbench.py 3 > $O/pmc_write.log 2>&1
cd $R
grep -E "Error|FAILED|passed|failed" $O/pytest_gpu.log | cut -c1-400 | tail -30; tail -7 $O/smoke.log; cat $O/bench.json; tail -3 $O/bench.err; cat $O/abn_microbench.log | head -12; ls -la $O/prof_bench $O/prof_abn 2>/dev/null | head -20
