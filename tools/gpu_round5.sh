#!/bin/bash
# Round-5 GPU box pass: the -m gpu tier, the driver-format bench line (+ per-launch breakdown), and a kernel trace of synchronous calls
# (gaps between launches = per-call overhead).  Everything lands in gpurun_out/$TAG/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); TAG=${1:-r05a}; OUT=$ROOTDIR/gpurun_out/$TAG; mkdir -p $OUT
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 900 python bench.py --steps 16 --warmup 3 --breakdown-out $OUT/bench_breakdown_stage1_256.json > $OUT/bench_cascade64_256_n1.json 2> $OUT/bench.err; echo "bench rc=$?"
  python - <<PY
import json
try:
    j = json.loads([l for l in open("$OUT/bench_cascade64_256_n1.json") if l.startswith("{")][-1])
    print({k: j.get(k) for k in ("value", "value_no_pipeline", "value_one_lane", "ms_per_step", "ms_per_step_no_pipeline", "pipelined_equals_synchronous")})
    print("roofline", {k: v for k, v in j.get("roofline", {}).items() if k != "executed"})
    ue = j.get("unet_eval", {}); print("unet_eval", {k: ue.get(k) for k in ("graph_step_ms", "hbm_frac_graph_step", "sum_kernel_ms")}, ue.get("conv_only"))
    print("t5", j.get("t5_encode", {}).get("ms"), "cpu", j.get("cpu_baseline", {}).get("value"))
    for k, v in j.get("secondary", {}).items(): print(k, {a: v.get(a) for a in ("denoising_steps_per_s", "denoising_steps_per_s_no_pipeline", "error", "hip_kernels", "speedup_vs_torch_ops")})
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/bench.err").read()[-2000:])
PY
fi
if [ "${SKIP_TRACE:-0}" != "1" ]; then
  cd /tmp; export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sync -o sync -- python $ROOTDIR/bench.py --steps 2 --warmup 1 --no-pipeline --no-cpu-baseline --no-secondary --no-t5 --no-breakdown > $OUT/trace_sync.log 2>&1; echo "trace rc=$?"
  cd $ROOTDIR; python tools/trace_gaps.py $OUT/trace_sync/sync_kernel_trace.csv > $OUT/trace_sync_gaps.txt 2>&1; tail -25 $OUT/trace_sync_gaps.txt
  rm -f $OUT/trace_sync/sync_kernel_trace.csv.keep
fi
