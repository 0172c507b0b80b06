out=gpurun_out/r06m; mkdir -p $out
run() { name=$1; shift; env "$@" python bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --breakdown-out $out/bd_$name.json > $out/bench_$name.json 2> $out/bench_$name.err
python - <<P
import json
d=json.loads([l for l in open("$out/bench_$name.json") if l.startswith("{")][-1])
u=d["unet_eval"]
print("$name value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)), "graph_step_ms", round(u["graph_step_ms"],4), "conv_only", {k:round(v["ms"]*1e3,1) for k,v in u["conv_only"]["by_level"].items()})
P
}
run LM MINIMAGEN_CONV_STRIPE=LM
run LMS_S2 MINIMAGEN_CONV_STRIPE=LMS MINIMAGEN_ST_NBLK_S=2
run tile MINIMAGEN_CONV_STRIPE=0
run LM_b MINIMAGEN_CONV_STRIPE=LM
for w in base64; do
for v in LMS S0; do
 MINIMAGEN_CONV_STRIPE=${v/S0/0} MINIMAGEN_ST_NBLK_S=2 python bench.py --workload base64 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 > $out/base_$v.json 2>$out/base_$v.err
 python - <<P
import json
d=json.loads([l for l in open("$out/base_$v.json") if l.startswith("{")][-1])
u=d["unet_eval"]
print("base64 stripe=$v value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "graph_step_ms", round(u["graph_step_ms"],4), "conv_only", {k:round(v["ms"]*1e3,1) for k,v in u["conv_only"]["by_level"].items()})
P
done; done
