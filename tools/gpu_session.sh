#!/bin/bash
# One GPU-box session (gpurun): every section writes under gpurun_out/<tag>/ and is bounded by its own timeout.
#   tools/gpu_session.sh <tag> <section> [<section> ...]
# sections: tests_all | tests_r5a | tests_r5b | tests_dist | world8 | smoke | bench | bench_quick | bench_prof | timeline | micro | micro_prof |
#           pmc | pmc2 | conv | lab | lab_pmc | det | dstep | fused_ab | dist | dist_ab | scale8 | solo   (A/B sections of switches that
#           no longer exist were removed in round 5; their verdicts are under profiles/)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export TMPDIR=/tmp
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $1" | tee -a $O/session.log; }
for sec in "$@"; do
  case $sec in
    tests_all)
      timeout 1500 python -m pytest tests -m gpu -q --tb=short -s --durations=12 > $O/pytest_gpu.log 2>&1
      stamp "tests_all rc=$?"; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3 | tee -a $O/session.log
      grep -E "worst|B=8 |config1 " $O/pytest_gpu.log | tee -a $O/session.log ;;
    world8)
      # the two eight-rank tests alone, the whole-step one WITHOUT its non-strict xfail marker (round 4 ended before its re-seeded
      # checks (3)-(5) ran on hardware; drop the marker in tests/test_distributed_gpu.py once this section is green)
      timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -q --tb=short -s --runxfail --durations=3 -k "eight_ranks" > $O/world8.log 2>&1
      stamp "world8 rc=$?"; grep -E "passed|failed|error|world 8" $O/world8.log | grep -v Gloo | tail -12 | tee -a $O/session.log ;;
    smoke)
      (timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1
      stamp "smoke rc=$?"; tail -7 $O/smoke.log | tee -a $O/session.log ;;
    bench)
      (timeout 600 python bench.py) > $O/bench.json 2> $O/bench.err
      stamp "bench rc=$?"; cut -c1-1500 $O/bench.json | tee -a $O/session.log ;;
    bench_quick)
      (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pairwise-sweep) > $O/bench_quick.json 2> $O/bench_quick.err
      stamp "bench_quick rc=$?"; cut -c1-1200 $O/bench_quick.json | tee -a $O/session.log ;;
    bench_prof)
      (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- \
        python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-pairwise-sweep > $O/prof_bench.log 2>&1)
      stamp "bench_prof rc=$?"
      # the roofline kernel per hardware queue: its launches beside the student's forward (teacher stream, timed steps) against its
      # launches on the main stream (the steps bench.py brackets with HIP events)
      T=$(find $O/prof_bench -name "*kernel_trace.csv" | head -1)
      [ -n "$T" ] && python tools/kernel_by_queue.py $T conv1x1_abn_kernel $O/tail_gemm_by_queue.md | tee -a $O/session.log ;;
    micro)
      timeout 300 python tools/kernel_microbench.py time 20 > $O/micro.jsonl 2> $O/micro.err
      stamp "micro rc=$?"; grep -v manifest $O/micro.jsonl | cut -c1-230 | tee -a $O/session.log ;;
    micro_prof)
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_micro -o micro -- \
        python $R/tools/kernel_microbench.py time 5 > $O/prof_micro.log 2>&1)
      stamp "micro_prof rc=$?" ;;
    pmc)
      # counters in their own passes, --kernel-trace only (no --stats / sys-trace next to --pmc)
      i=0
      for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
                 "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
                 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
        i=$((i+1))
        (cd /tmp && timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc$i -o k -- \
          python $R/tools/kernel_microbench.py pmc > $O/pmc$i.log 2>&1)
        stamp "pmc pass $i ($set) rc=$?"
        # keep only what the summariser needs from the (large) counter CSV: drop the long torch kernel names
        for f in $(find $O/pmc$i -name "*counter_collection.csv"); do
          python - "$f" <<'EOF'
import csv, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f, newline="")))
keep = [r for r in rows if "skd::" in r["Kernel_Name"]]
for r in keep:
    r["Kernel_Name"] = r["Kernel_Name"][:200]
w = csv.DictWriter(open(f, "w", newline=""), fieldnames=list(rows[0].keys()) if rows else [])
w.writeheader(); w.writerows(keep)
EOF
        done
        find $O/pmc$i -name "*kernel_trace.csv" -delete
      done
      python tools/summarise_pmc.py $O/pmc_summary.json $O/pmc1.log $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5 >> $O/session.log 2>&1
      stamp "pmc summarised" ;;
    conv)
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_conv -o conv -- \
        python $R/tools/conv_table.py 10 > $O/conv_table.jsonl 2> $O/conv_table.err)
      stamp "conv rc=$?"; tail -1 $O/conv_table.jsonl | tee -a $O/session.log
      python tools/summarise_conv_table.py $O/conv_table.jsonl $O/prof_conv $O/conv_shapes.md >> $O/session.log 2>&1 ;;
    lab)
      timeout 240 tools/gemm_lab time 10 > $O/gemm_lab.jsonl 2> $O/gemm_lab.err
      stamp "lab rc=$?"; cut -c1-330 $O/gemm_lab.jsonl | tee -a $O/session.log ;;
    lab_pmc)
      i=0
      for set in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
                 "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
                 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1))
        (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/labpmc$i -o k -- $R/tools/gemm_lab pmc 2 > $O/labpmc$i.log 2>&1)
        stamp "lab_pmc pass $i ($set) rc=$?"
        find $O/labpmc$i -name "*kernel_trace.csv" -delete
      done
      python tools/summarise_lab_pmc.py $O $O/gemm_lab_pmc.json > $O/gemm_lab_pmc.log 2>&1
      stamp "lab_pmc summarised" ;;
    tests_dist)
      timeout 900 python -m pytest tests/test_distributed_gpu.py tests/test_kernels_gpu.py -m gpu -q --tb=short -s -k "distributed or one_call or mailbox or two_ranks or torchrun" > $O/pytest_dist.log 2>&1
      stamp "tests_dist rc=$?"; grep -E "passed|failed|error" $O/pytest_dist.log | tail -3 | tee -a $O/session.log
      grep -E "^E  |^FAILED|SyncABN exchange" $O/pytest_dist.log | cut -c1-300 | head -30 | tee -a $O/session.log ;;
    dstep)
      timeout 300 python tools/d_step_error_probe.py 8 > $O/d_step_error.jsonl 2> $O/d_step_error.err
      stamp "dstep rc=$?"; cut -c1-420 $O/d_step_error.jsonl | tee -a $O/session.log ;;
    fused_ab)
      for v in "SKD_ABN_FUSED=1" "SKD_ABN_FUSED=0"; do
        (env $v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pairwise-sweep) > "$O/bench_ab_$(echo $v | tr ' =' '__').json" 2>> $O/bench_ab.err
        stamp "fused_ab $v rc=$?"; cut -c1-260 "$O/bench_ab_$(echo $v | tr ' =' '__').json" | tee -a $O/session.log
      done ;;
    det)
      timeout 400 python tools/determinism_probe.py 8 > $O/determinism.jsonl 2> $O/determinism.err
      stamp "det rc=$?"; cut -c1-700 $O/determinism.jsonl | tee -a $O/session.log ;;
    dist_ab)
      # two ranks on ONE MI355X over gloo (what a 1-GPU box can show of N > 1): SyncABN exchange inside the one-launch kernels vs
      # the three-launch form
      for v in "SKD_ABN_SYNC_FUSED=1" "SKD_ABN_SYNC_FUSED=0"; do
        (env $v SKD_DIST_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
          --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --batch 4 --no-cpu-baseline) > "$O/bench_2ranks_$(echo $v | tr ' =' '__').json" 2>> $O/bench_2ranks.err
        stamp "dist_ab $v rc=$?"; cut -c1-300 "$O/bench_2ranks_$(echo $v | tr ' =' '__').json" | tee -a $O/session.log
        python - "$O/bench_2ranks_$(echo $v | tr ' =' '__').json" <<'PYEOF' | tee -a $O/session.log
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    print("   comm:", json.dumps(d.get("comm")))
except Exception as e:
    print("   (no line)", e)
PYEOF
      done ;;
    pmc2)
      # counters in their own passes (no --stats / sys-trace next to --pmc): HBM bytes and the MFMA pipe, over the microbench
      i=0
      for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
                 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
        i=$((i+1))
        (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc$i -o k -- \
          python $R/tools/kernel_microbench.py pmc > $O/pmc$i.log 2>&1)
        stamp "pmc2 pass $i ($set) rc=$?"
        for f in $(find $O/pmc$i -name "*counter_collection.csv"); do
          python - "$f" <<'PYEOF'
import csv, sys
f = sys.argv[1]
rows = list(csv.DictReader(open(f, newline="")))
keep = [r for r in rows if "skd::" in r["Kernel_Name"]]
for r in keep:
    r["Kernel_Name"] = r["Kernel_Name"][:200]
w = csv.DictWriter(open(f, "w", newline=""), fieldnames=list(rows[0].keys()) if rows else [])
w.writeheader(); w.writerows(keep)
PYEOF
        done
        find $O/pmc$i -name "*kernel_trace.csv" -delete
      done
      python tools/summarise_pmc.py $O/pmc_summary.json $O/pmc1.log $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 >> $O/session.log 2>&1
      stamp "pmc2 summarised" ;;
    timeline)
      # (TL_ENV="SKD_DIST_SOLO=1" TL_NAME=_solo: the same trace of the N > 1 form on a one-rank communicator)
      # ONE kernel trace of the default step (teacher = hipGraph replay, D step on its stream; no HIP-event bracketing) -> per-stream
      # busy / idle, main-stream gaps, exposed D tail (tools/timeline.py); the slimmed per-dispatch trace travels back for re-analysis
      (cd /tmp && env $TL_ENV timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_timeline -o bench -- \
        python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing > $O/prof_timeline.log 2>&1)
      stamp "timeline trace rc=$?"
      T=$(find $O/prof_timeline -name "*kernel_trace.csv" | head -1)
      python tools/timeline.py $T $O/timeline${TL_NAME}.md >> $O/session.log 2>&1
      stamp "timeline rc=$?"
      python - "$T" "$O/timeline${TL_NAME}_trace_slim.csv" <<'PYEOF'
import csv, sys
rd = csv.DictReader(open(sys.argv[1], newline=""))
cols = [c for c in ("Queue_Id", "Stream_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Workgroup_Size_X", "Grid_Size", "Workgroup_Size") if c in rd.fieldnames]
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=cols)
w.writeheader()
for r in rd:
    r = {c: r[c] for c in cols}
    r["Kernel_Name"] = r["Kernel_Name"][:140]
    w.writerow(r)
PYEOF
      head -2 $T | cut -c1-600 >> $O/session.log
      find $O/prof_timeline -name "*kernel_trace.csv" -delete ;;
    tests_r5a)
      # round 5, first call: the kink-aware bounds (tests/kinks.py) where round 4 had widened them, and the hipGraph cases after the
      # capture-mode change
      timeout 900 python -m pytest tests/test_step_gpu.py tests/test_distributed_gpu.py -m gpu -q --tb=short -s --durations=5 \
        -k "b8_vs_golden or hipgraph or eight_ranks_vs or discriminator_step" > $O/pytest_r5a.log 2>&1
      stamp "tests_r5a rc=$?"; grep -E "passed|failed|error" $O/pytest_r5a.log | tail -3 | tee -a $O/session.log
      grep -E "^E  |^FAILED|LeakyReLU decisions|worst rank|movement|im2col" $O/pytest_r5a.log | cut -c1-300 | head -60 | tee -a $O/session.log ;;
    scale8)
      # UNATTENDED multi-GPU recipe: runs on any box with >= 2 GPUs, every leg under its own timeout, every leg
      # produces a JSON line or a logged reason.  (i) RCCL + cross-device HIP IPC tests; (ii) the scaling curve N = 1, 2, 4, 8 in the
      # default (safe) form -- bench.py itself falls back to a safer SyncABN form if a warm-up step raises a device status word and says
      # so in comm.form / comm.fallback_reason; (iii) at the largest N the A/B the defaults are waiting for: in-kernel exchange
      # (SKD_ABN_SYNC_FUSED=1, with the compute-unit reserve), collectives only (SKD_SYNC_IPC=0), teacher hipGraph at N > 1.
      NG=$(python -c "import torch; print(torch.cuda.device_count())")
      stamp "scale8: $NG GPUs visible"
      if [ "$NG" -ge 2 ]; then
        # first of all: does RCCL come up on this node, and does every collective kind the step issues return the right answer
        n=$NG; [ "$n" -gt 8 ] && n=8
        (timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29599 \
          tools/rccl_probe.py) > $O/rccl_probe_n$n.json 2> $O/rccl_probe_n$n.err
        stamp "scale8 rccl_probe N=$n rc=$?"; grep '^{' $O/rccl_probe_n$n.json | tee -a $O/session.log
        timeout 900 python -m pytest tests/test_distributed_gpu.py -m gpu -q --tb=short -s -k multi_gpu > $O/pytest_multi_gpu.log 2>&1
        stamp "scale8 multi_gpu tests rc=$?"; grep -E "passed|failed|error|skipped" $O/pytest_multi_gpu.log | tail -3 | tee -a $O/session.log
      fi
      port=29600
      for n in 1 2 4 8; do
        [ "$n" -le "$NG" ] || continue
        port=$((port+1))
        if [ "$n" -eq 1 ]; then
          (timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-pairwise-sweep) > $O/scale_n$n.json 2> $O/scale_n$n.err
        else
          (timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
            bench.py --gpus $n --steps 20 --warmup 5) > $O/scale_n$n.json 2> $O/scale_n$n.err
        fi
        stamp "scale8 N=$n rc=$?"; grep '^{' $O/scale_n$n.json | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l); print('   N=%d %.1f img/s %.2f ms/step comm=%s' % (d['n_gpus'], d['value'], d['ms_per_step'], json.dumps(d.get('comm'))[:600]))" | tee -a $O/session.log
      done
      if [ "$NG" -ge 2 ]; then
        n=$NG; [ "$n" -gt 8 ] && n=8
        for v in "SKD_ABN_SYNC_FUSED=1" "SKD_ABN_SYNC_FUSED=1 SKD_ABN_RCCL_RESERVE_CUS=0" "SKD_SYNC_IPC=0" "SKD_TEACHER_GRAPH=force" "SKD_D_GRAPH=1" "SKD_TEACHER_STREAM=1"; do
          port=$((port+1)); f="$O/scale_ab_n${n}_$(echo $v | tr ' =' '__').json"
          (env $v timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
            bench.py --gpus $n --steps 10 --warmup 3) > "$f" 2>> $O/scale_ab.err
          stamp "scale8 A/B $v N=$n rc=$?"; grep '^{' "$f" | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l); print('   %.1f img/s %.2f ms/step comm=%s' % (d['value'], d['ms_per_step'], json.dumps(d.get('comm'))[:600]))" | tee -a $O/session.log
        done
      fi ;;
    solo)
      # RCCL on a 1-GPU box: the probe (communicator + every collective the step uses), then the N > 1 FORM of the step on a
      # communicator of ONE rank (SKD_DIST_SOLO=1: hooks, buckets, synchronised ABN over torch.distributed, eager teacher) under
      # torchrun AND self-launched, with a kernel trace that shows which RCCL kernels ran
      (timeout 200 python tools/rccl_probe.py) > $O/rccl_probe.json 2> $O/rccl_probe.err
      stamp "rccl_probe rc=$?"; cat $O/rccl_probe.json | tee -a $O/session.log
      (SKD_DIST_SOLO=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-pairwise-sweep) > $O/bench_solo_rccl.json 2> $O/bench_solo_rccl.err
      stamp "bench solo (torchrun, RCCL, one rank) rc=$?"; cut -c1-900 $O/bench_solo_rccl.json | tee -a $O/session.log
      python - $O/bench_solo_rccl.json <<'EOF' | tee -a $O/session.log
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("   comm:", json.dumps(d.get("comm")))
    print("   rehearsal:", d.get("rehearsal"))
except Exception as e:
    print("   no line:", e)
EOF
      (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > $O/bench_solo_ref_n1.json 2> $O/bench_solo_ref_n1.err
      stamp "bench N = 1 (plain, same box) rc=$?"; cut -c1-300 $O/bench_solo_ref_n1.json | tee -a $O/session.log
      (SKD_DIST_SOLO=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > $O/bench_solo_direct.json 2> $O/bench_solo_direct.err
      stamp "bench solo (started directly, no torchrun) rc=$?"; cut -c1-300 $O/bench_solo_direct.json | tee -a $O/session.log
      (cd /tmp && SKD_DIST_SOLO=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_solo -o solo -- \
        python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing > $O/prof_solo.log 2>&1)
      stamp "solo kernel trace rc=$?"
      for f in $(find $O/prof_solo -name "*kernel_stats.csv"); do grep -i "nccl\|rccl" $f | cut -c1-200 | tee -a $O/session.log; cp $f $O/solo_kernel_stats.csv; done
      rm -rf $O/prof_solo
      timeout 600 python -m pytest tests/test_distributed_gpu.py -m gpu -q --tb=short -s -k "solo" > $O/pytest_solo.log 2>&1
      stamp "tests solo rc=$?"; tail -5 $O/pytest_solo.log | tee -a $O/session.log ;;
    solo_ab)
      # what the N > 1 FORM of the step costs on ONE GPU with nothing on the wire (SKD_DIST_SOLO=1, RCCL communicator of one rank):
      # the software overhead data parallelism adds to every rank before any link is involved.  Interleaved, two rounds.
      for rep in 1 2; do
        i=0
        for v in "SKD_DIST_SOLO=0" "SKD_DIST_SOLO=1" "SKD_DIST_SOLO=1 SKD_ABN_SYNC_FUSED=1" "SKD_DIST_SOLO=1 SKD_SYNC_IPC=0" \
                 "SKD_DIST_SOLO=1 SKD_TEACHER_GRAPH=force" "SKD_DIST_SOLO=1 SKD_ABN_SYNC_FUSED=1 SKD_TEACHER_GRAPH=force" \
                 "SKD_DIST_SOLO=1 SKD_D_GRAPH=1" "SKD_DIST_SOLO=1 SKD_D_GRAPH=1 SKD_TEACHER_GRAPH=force" "SKD_DIST_SOLO=1 SKD_D_STREAM=0" \
                 "SKD_DIST_SOLO=0 SKD_D_STREAM=0" "SKD_DIST_SOLO=1 SKD_TEACHER_STREAM=1" "SKD_DIST_SOLO=1 SKD_TEACHER_STREAM=1 SKD_ABN_SYNC_FUSED=1" \
                 "SKD_DIST_SOLO=0 SKD_TEACHER_STREAM=0"; do
          i=$((i+1))
          f=$O/solo_ab_${i}_$rep.json
          (env $v timeout 300 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > $f 2>> $O/solo_ab.err
          python - "$f" "$v" <<'EOF' | tee -a $O/session.log
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("   [%s] %.3f ms/step  %.2f img/s  form: %s" % (sys.argv[2], d["ms_per_step"], d["value"], (d.get("comm") or {}).get("form", "single rank")[:70]))
except Exception as e:
    print("   [%s] no line: %s" % (sys.argv[2], e))
EOF
        done
      done
      stamp "solo_ab done" ;;
    selflaunch)
      # round 6: `python bench.py --gpus 2` WITHOUT torchrun (the driver's command form): bench.self_launch starts the ranks itself.
      # On a 1-GPU box the two ranks share the device over gloo (SKD_DIST_BACKEND=gloo); without that switch the same command must
      # print the device-count error line and exit 2.
      (SKD_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --batch 4 --no-cpu-baseline) > $O/bench_selflaunch.json 2> $O/bench_selflaunch.err
      stamp "selflaunch (gloo, shared device) rc=$?"; cut -c1-700 $O/bench_selflaunch.json | tee -a $O/session.log
      (timeout 100 python bench.py --gpus 8 --steps 20 --warmup 5) > $O/bench_selflaunch_8.json 2> $O/bench_selflaunch_8.err
      stamp "selflaunch --gpus 8 on this box rc=$? (2 expected on a 1-GPU box)"; cut -c1-400 $O/bench_selflaunch_8.json | tee -a $O/session.log ;;
    dstream2)
      # round 6: is the D step hidden when TWO ranks share the chip?  SKD_D_STREAM=1/0, SKD_D_GRAPH=0/1
      for v in "SKD_D_STREAM=1" "SKD_D_STREAM=0" "SKD_D_GRAPH=1"; do
        f="$O/bench_2ranks_$(echo $v | tr ' =' '__').json"
        (env $v SKD_DIST_BACKEND=gloo timeout 500 python bench.py --gpus 2 --steps 5 --warmup 3 --batch 4 --no-cpu-baseline --no-kernel-timing) > "$f" 2>> $O/bench_2ranks.err
        stamp "dstream2 $v rc=$?"; cut -c1-260 "$f" | tee -a $O/session.log
      done ;;
    ab_env)
      # generic same-box A/B of environment switches: AB_ENVS="A=1;A=0;B=1 C=2" (one leg per ';'), three alternations, 20 timed steps
      IFS=';' read -ra LEGS <<< "${AB_ENVS:-SKD_NOOP=0}"
      for i in 1 2 3; do
        for v in "${LEGS[@]}"; do
          f="$O/ab_$(echo $v | tr ' =' '__')_$i.json"
          (env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > "$f" 2>> $O/ab_env.err
          stamp "ab_env [$v] $i rc=$?"; python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][0]); print('   [$v] $i: %.2f img/s %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $O/session.log
        done
      done ;;
    ab_const)
      # same-box A/B of module constants (tools/ab_patch.py): AB_CONSTS="functional.PAD_WEIGHTS=True;functional.PAD_WEIGHTS=False"
      IFS=';' read -ra LEGS <<< "${AB_CONSTS:-functional.PAD_WEIGHTS=True;functional.PAD_WEIGHTS=False}"
      for i in 1 2 3; do
        for v in "${LEGS[@]}"; do
          f="$O/abc_$(echo $v | tr ' =.' '___')_$i.json"
          (timeout 300 python tools/ab_patch.py $v -- bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > "$f" 2>> $O/ab_const.err
          stamp "ab_const [$v] $i rc=$?"; python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][0]); print('   [$v] $i: %.2f img/s %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $O/session.log
        done
      done ;;
    tests_k)
      # a -k selection of the GPU tests: TESTS_K="conv1x1 or bottleneck"
      timeout 1200 python -m pytest tests -m gpu -q --tb=short -s --durations=8 -k "${TESTS_K:-conv1x1}" > $O/pytest_k.log 2>&1
      stamp "tests_k [${TESTS_K:-conv1x1}] rc=$?"; grep -E "passed|failed|error" $O/pytest_k.log | tail -3 | tee -a $O/session.log
      grep -E "^E  |^FAILED" $O/pytest_k.log | cut -c1-300 | head -40 | tee -a $O/session.log ;;
    pmc_only)
      # FETCH_SIZE / WRITE_SIZE passes over a -k style selection of the microbench (PMC_ONLY="conv1x1"): HBM traffic of a few kernels
      i=0
      for set in "FETCH_SIZE" "WRITE_SIZE"; do
        i=$((i+1))
        (cd /tmp && timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc$i -o k -- \
          python $R/tools/kernel_microbench.py pmc 2 "${PMC_ONLY:-conv1x1}" > $O/pmc$i.log 2>&1)
        stamp "pmc_only pass $i ($set) rc=$?"
        find $O/pmc$i -name "*kernel_trace.csv" -delete
      done
      python tools/summarise_pmc.py $O/pmc_summary.json $O/pmc1.log $O/pmc1 $O/pmc2 >> $O/session.log 2>&1
      stamp "pmc_only summarised" ;;
    ab_r04)
      # same-box A/B against the round-4 tree (git archive e78944f into _r04_tree/, built here, git-ignored): box-to-box variation is
      # +-1 %, the round's gains are of that size, so the two trees alternate on ONE box
      for i in 1 2 3; do
        for t in new old; do
          d=$R; [ $t = old ] && d=$R/_r04_tree
          (cd $d && timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pairwise-sweep --no-kernel-timing) > $O/ab_${t}_$i.json 2>> $O/ab.err
          stamp "ab_r04 $t $i rc=$?"; python -c "
import json,sys
d=json.loads([l for l in open('$O/ab_${t}_$i.json') if l.startswith('{')][0]); print('   $t $i: %.2f img/s %.3f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $O/session.log
        done
      done ;;
    ce_lab)
      timeout 200 python tools/ce_lab.py > $O/ce_lab.jsonl 2> $O/ce_lab.err
      stamp "ce_lab rc=$?"; cat $O/ce_lab.jsonl | tee -a $O/session.log ;;
    dist)
      SKD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 2 --batch 4 --no-cpu-baseline > $O/bench_dist.json 2> $O/bench_dist.err
      stamp "dist rc=$?"; cut -c1-600 $O/bench_dist.json | tee -a $O/session.log ;;
    tests_r5b)
      # round 5, kernels rewritten this round: fused CE + upsample (cell formulation), channels-last pair-wise pooling, stem max-pool
      # backward over 2 x 2 blocks, pair-wise backward through the node-major copy, packed scalar read-back, the b8 golden step
      timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -q --tb=short -s --durations=8 \
        -k "ce_dsn or maxpool or pairwise or criteria or operational_switches or b8_vs_golden or config1 or full_step_vs_oracle or stem or networks_forward or bottleneck or psp" > $O/pytest_r5b.log 2>&1
      stamp "tests_r5b rc=$?"; grep -E "passed|failed|error" $O/pytest_r5b.log | tail -3 | tee -a $O/session.log
      grep -E "^E  |^FAILED|LeakyReLU decisions|im2col" $O/pytest_r5b.log | cut -c1-300 | head -60 | tee -a $O/session.log ;;
  esac
done
# large raw traces do not travel back (64 MiB cap): keep stats, drop per-dispatch traces except the conv one (names needed)
find $O -name "*kernel_trace.csv" -size +20M -delete
du -sh $O | tee -a $O/session.log
