#!/bin/bash
# A/B of a VARIANT BUILD of the library against the default build on ONE box, twice each, back to back (round 6: every kept / dropped change of the
# stripe conv kernel, profiles/r06_stripe_late_ab.txt).  Build the variant first, e.g.
#   make -C minimagen_amd/csrc BUILD=build_la0 LIBNAME=libminimagen_hip_la0.so VARIANT_FLAGS=-DST_LOOKAHEAD=0
# usage: bash tools/gpu_ab_lib.sh TAG VARIANT_NAME          (variant = minimagen_amd/libminimagen_hip_VARIANT_NAME.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tag=${1:-ab_lib}; var=${2:?variant name}; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_conv_stripe.py tests/test_unet.py -x -q -m gpu > $out/pytest.log 2>&1; tail -2 $out/pytest.log
run() { name=$1; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --breakdown-out $out/breakdown_$name.json > $out/bench_$name.json 2> $out/bench_$name.err
python - <<P
import json
try:
    d=json.loads([l for l in open("$out/bench_$name.json") if l.startswith("{")][-1])
    u=d["unet_eval"]
    print("$name value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)), "graph_step_ms", round(u["graph_step_ms"],4), "conv_only", {k:round(v["ms"]*1e3,1) for k,v in u["conv_only"]["by_level"].items()})
except Exception as e: print("$name failed", e, open("$out/bench_$name.err").read()[-300:])
P
}
run default X=1
run $var MINIMAGEN_HIP_LIB=$PWD/minimagen_amd/libminimagen_hip_$var.so
run default_b X=1
run ${var}_b MINIMAGEN_HIP_LIB=$PWD/minimagen_amd/libminimagen_hip_$var.so
