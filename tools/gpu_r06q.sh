out=gpurun_out/r06u; mkdir -p $out
python -m pytest tests/test_conv_stripe.py tests/test_unet.py tests/test_sampler.py -x -q -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
run() { name=$1; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --breakdown-out $out/bd_$name.json > $out/bench_$name.json 2> $out/bench_$name.err
python - <<P
import json
d=json.loads([l for l in open("$out/bench_$name.json") if l.startswith("{")][-1])
u=d["unet_eval"]
print("$name value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)), "graph_step_ms", round(u["graph_step_ms"],4), "conv_only", {k:round(v["ms"]*1e3,1) for k,v in u["conv_only"]["by_level"].items()})
P
}
run stripe X=1
run tile MINIMAGEN_CONV_STRIPE=0
run stripe_b X=1
python - <<P
import json
a=json.load(open("$out/bd_stripe.json")); b=json.load(open("$out/bd_tile.json"))
for x,y in zip(a,b):
    if x['kernel']=='conv' and 'u ' in x['op'].replace('k3s1u','u '): print(f"{x['op']:50s} stripe {x['ms']*1e3:7.1f}   tile {y['ms']*1e3:7.1f}")
P
