#!/bin/bash
# round 6: A/B of a variant library against the default build on one box (here: LDS operand lookahead of the MFMA waves, libminimagen_hip_la0.so = none)
out=gpurun_out/r06z; mkdir -p $out
python -m pytest tests/test_conv_stripe.py tests/test_unet.py -x -q -m gpu > $out/pytest.log 2>&1; tail -2 $out/pytest.log
run() { name=$1; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --breakdown-out $out/breakdown_$name.json > $out/bench_$name.json 2> $out/bench_$name.err
python - <<P
import json
try:
    d=json.loads([l for l in open("$out/bench_$name.json") if l.startswith("{")][-1])
    u=d["unet_eval"]
    print("$name value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)), "graph_step_ms", round(u["graph_step_ms"],4), "conv_only", {k:round(v["ms"]*1e3,1) for k,v in u["conv_only"]["by_level"].items()})
except Exception as e: print("$name failed", open("$out/bench_$name.err").read()[-300:])
P
}
run ncw2 X=1
run ncw4 MINIMAGEN_HIP_LIB=$PWD/minimagen_amd/libminimagen_hip_ncw4.so
run ncw2_b X=1
run ncw4_b MINIMAGEN_HIP_LIB=$PWD/minimagen_amd/libminimagen_hip_ncw4.so
