#!/bin/bash
# A/B of two builds of the library on the GPU box: usage  bash tools/gpu_ab.sh <variant .so name under minimagen_amd/> [pytest -k expression]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
VAR=$1; KEXPR=${2:-cross_att}
timeout 900 python -m pytest tests/test_kernels.py tests/test_unet.py tests/test_sampler.py -m gpu -q -x -k "$KEXPR" --timeout 600 > $OUT/ab_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/ab_pytest.log
for v in default $VAR; do
  if [ $v = default ]; then unset MINIMAGEN_HIP_LIB; else export MINIMAGEN_HIP_LIB=$ROOTDIR/minimagen_amd/$v; fi
  timeout 400 python bench.py --no-cpu-baseline --no-secondary --no-t5 --breakdown-out $OUT/ab_bd_$v.json > $OUT/ab_bench_$v.log 2>&1
  python - <<PY
import json
d=json.loads(open("$OUT/ab_bench_$v.log").read().strip().splitlines()[-1])
print("$v", round(d["value"]), "one lane", round(d.get("value_one_lane",0)), "sync", round(d["value_no_pipeline"]), "roofline", d["roofline"])
PY
done
