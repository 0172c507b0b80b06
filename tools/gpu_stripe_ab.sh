#!/bin/bash
# round 6: the stripe conv kernel (tile_cfg 12) -- parity on the device, isolated launches next to the tile kernel, bench.py A/B on the same box
# usage: tools/gpu_stripe_ab.sh <outdir> [variants of MINIMAGEN_CONV_STRIPE, default "LMS 0"] ; FULL=1 adds the sampler / benched-config tests
out=gpurun_out/${1:-r06b}; shift
variants=${*:-LMS 0}
mkdir -p $out
tests="tests/test_conv_stripe.py tests/test_unet.py"
[ -n "$FULL" ] && tests="$tests tests/test_sampler.py tests/test_benched_configs.py"
python -m pytest $tests -x -q -m gpu > $out/pytest.log 2>&1
tail -3 $out/pytest.log
{
for a in "64 8 8 256 256 1 id" "64 8 8 256 256 1 none" "64 8 3 256 256 0 none" "64 8 8 128 128 1 id" "64 8 8 128 128 1 none" "64 16 8 128 128 1 none" \
         "64 16 16 64 64 1 id" "64 16 16 64 64 1 none" "64 32 16 64 64 1 none" "64 8 8 64 64 1 id" "64 16 8 64 64 1 none" "64 16 16 32 32 1 id" "64 8 8 32 32 1 id"; do
  set -- $a
  tile=rp6; [ "$5" = "32" ] && tile=rp7
  python tools/bench_conv.py $a rp12 2>&1 | grep "us  ("
  python tools/bench_conv.py $a $tile 2>&1 | grep "us  ("
done
} > $out/conv_isolated.txt 2>&1
cat $out/conv_isolated.txt
for v in $variants; do
  MINIMAGEN_CONV_STRIPE=$v python bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --breakdown-out $out/breakdown_stripe_$v.json > $out/bench_stripe_$v.json 2> $out/bench_stripe_$v.err
  python - <<P
import json
d=json.loads([l for l in open("$out/bench_stripe_$v.json") if l.startswith("{")][-1])
u=d["unet_eval"]
print("stripe=$v value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)), "graph_step_ms", round(u["graph_step_ms"],4), "conv_only", {k:round(v["ms"]*1e3,1) for k,v in u["conv_only"]["by_level"].items()}, "conv frac", round(u["conv_only"]["hbm_frac"],3), "hbm_frac", round(u["hbm_frac_graph_step"],3))
P
done
