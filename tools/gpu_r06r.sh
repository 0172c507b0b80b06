out=gpurun_out/r06r; mkdir -p $out
python -m pytest tests -x -q -m gpu --deselect tests/test_quoted_configs.py > $out/pytest.log 2>&1; tail -3 $out/pytest.log
python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --breakdown-out $out/bd.json > $out/bench.json 2> $out/bench.err
python - <<P
import json
d=json.loads([l for l in open("$out/bench.json") if l.startswith("{")][-1])
u=d["unet_eval"]
print("value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)), "graph_step_ms", round(u["graph_step_ms"],4), "conv_only", {k:round(v["ms"]*1e3,1) for k,v in u["conv_only"]["by_level"].items()})
for x in json.load(open("$out/bd.json")):
    if "res" in x["op"] and x["kernel"]=="conv" and ("16->16 @64" in x["op"] or "8->8 @128" in x["op"]): print(x["op"], round(x["ms"]*1e3,1))
P
python bench.py --workload base64 --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 > $out/base.json 2>$out/base.err
python - <<P
import json
d=json.loads([l for l in open("$out/base.json") if l.startswith("{")][-1])
print("base64 value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "graph_step_ms", round(d["unet_eval"]["graph_step_ms"],4))
P
