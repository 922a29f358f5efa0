O=gpurun_out/r04h; mkdir -p $O; export TMPDIR=/tmp; R=$(pwd)
# (1) what makes two ranks on ONE device slow this round?  (B=2, 3 timed steps, no kernel-timing extras)
for v in "X=1" "SKD_TEACHER_GRAPH=0" "SKD_SN_TOGETHER=0" "SKD_ABN_SYNC_FUSED=0" "SKD_D_STREAM=0"; do
  s0=$(date +%s)
  (env $v SKD_DIST_BACKEND=gloo timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 2 --batch 2 --no-cpu-baseline --no-kernel-timing) > "$O/b2_$(echo $v | tr ' =' '__').json" 2> "$O/b2_$(echo $v | tr ' =' '__').err"
  echo "b2 $v rc=$? wall $(( $(date +%s) - s0 )) s: $(cut -c1-210 $O/b2_$(echo $v | tr ' =' '__').json | grep -o '"value.*ms_per_step": [0-9.]*')"
done
# (2) super-tile geometry sweep on the K = 512 / N = 2048 tail GEMM: L2 -> fabric bytes per launch
export SKD_MICRO_TILE_ORDERS="1,16;2,8;2,16;4,4;4,16;8,4;8,8;16,2;2,32;1,33"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$O/pmc$i -o k -- python $R/tools/kernel_microbench.py pmc 2 "tail GEMM" > $R/$O/pmc$i.log 2>&1)
  echo "pmc pass $i rc=$?"
  find $O/pmc$i -name "*kernel_trace.csv" -delete
done
python tools/summarise_pmc.py $O/pmc_summary.json $O/pmc1.log $O/pmc1 $O/pmc2 > $O/pmc_sum.log 2>&1
timeout 120 python tools/kernel_microbench.py time 10 "tail GEMM" > $O/micro_tail.jsonl 2> $O/micro_tail.err
python - <<'PY'
import json
p = json.load(open("gpurun_out/r04h/pmc_summary.json"))
t = {}
for l in open("gpurun_out/r04h/micro_tail.jsonl"):
    try:
        r = json.loads(l)
        if "us" in r: t[r["id"]] = r["us"]
    except Exception: pass
for c in p["cases"]:
    if c.get("hbm_read_MB") is not None and c["shape"][2] == 2048:
        print(c["name"], "read MB", c["hbm_read_MB"], "write MB", c["hbm_write_MB"], "x algo", c["hbm_over_algorithmic"], "us", t.get(c["id"]))
PY
