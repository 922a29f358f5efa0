#!/bin/bash
# SKD_TEACHER_STREAM=0/1 on ONE box, cool and warm: 20-step A/B (3 alternations), one 500-step run per flag with rocm-smi sampled beside
# it, the 20-step A/B again.   tools/teacher_stream_probe.sh <outdir>
O=${1:-gpurun_out/tstream}; mkdir -p $O
one() { # flag steps tag
  (SKD_TEACHER_STREAM=$1 timeout 900 python bench.py --steps $2 --warmup 5 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$3 SKD_TEACHER_STREAM=$1: %d steps %.3f ms/step %.2f images/s' % (d['steps'], d['ms_per_step'], d['value']))" | tee -a $O/summary.txt
}
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|junction" | sed 's/^GPU\[0\]\t\t: //' | tr '\n' ';' | sed "s/^/$1: /" | tee -a $O/summary.txt; echo | tee -a $O/summary.txt; }
smi "idle"
for i in 1 2 3; do one 0 20 cool; one 1 20 cool; done
for f in 0 1; do
  (SKD_TEACHER_STREAM=$f timeout 900 python bench.py --steps 500 --warmup 5 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > $O/long_$f.json 2>/dev/null &
  PID=$!; sleep 20; smi "20 s into the 500-step run, flag $f"; wait $PID
  python -c "
import json
d = json.loads([l for l in open('$O/long_$f.json') if l.startswith('{')][-1]); print('long SKD_TEACHER_STREAM=$f: %d steps %.3f ms/step %.2f images/s' % (d['steps'], d['ms_per_step'], d['value']))" | tee -a $O/summary.txt
done
for i in 1 2 3; do one 0 20 warm; one 1 20 warm; done
