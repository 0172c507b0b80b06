#!/bin/bash
# round 6: statistics blocks per workgroup re-swept after the loader-group / lookahead changes (same box)
out=gpurun_out/r06aa; mkdir -p $out
run() { name=$1; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --no-breakdown > $out/bench_$name.json 2> $out/bench_$name.err
python - <<P
import json
try:
    d=json.loads([l for l in open("$out/bench_$name.json") if l.startswith("{")][-1])
    u=d["unet_eval"]
    print("$name value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)), "graph_step_ms", round(u["graph_step_ms"],4), "conv_only", {k:round(v["ms"]*1e3,1) for k,v in u["conv_only"]["by_level"].items()})
except Exception as e: print("$name failed", open("$out/bench_$name.err").read()[-300:])
P
}
run base X=1
run S1 MINIMAGEN_ST_NBLK_S=1
run S4 MINIMAGEN_ST_NBLK_S=4
run M2 MINIMAGEN_ST_NBLK_M=2
run L2 MINIMAGEN_ST_NBLK_L=2
run pipeS2 MINIMAGEN_ST_NBLK_PIPE_S=2
run pipeS8 MINIMAGEN_ST_NBLK_PIPE_S=8
run base_b X=1
