#!/bin/bash
# What a SUSTAINED run does to the chip: one long bench.py run (default 900 steps) with rocm-smi sampled beside it (clocks, power,
# temperature), then a 20-step run on the warm chip.   tools/sustained_probe.sh <outdir> [steps]
O=${1:-gpurun_out/sustained}; N=${2:-900}; mkdir -p $O
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > $O/cool_20.json 2>/dev/null
(timeout 900 python bench.py --steps $N --warmup 5 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > $O/long.json 2>/dev/null &
PID=$!
t0=$(date +%s)
while kill -0 $PID 2>/dev/null; do
  echo "t=$(( $(date +%s) - t0 )) s" >> $O/smi.log
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature" >> $O/smi.log
  sleep 5
done
(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > $O/warm_20.json 2>/dev/null
for f in cool_20 long warm_20; do python - $O/$f.json $f <<'EOF'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("%s: %d steps %.3f ms/step %.2f images/s" % (sys.argv[2], d["steps"], d["ms_per_step"], d["value"]))
except Exception as e:
    print(sys.argv[2], "no line", e)
EOF
done
