out=gpurun_out/r06s; mkdir -p $out
run() { name=$1; wl=$2; shift; shift; env "$@" python bench.py --workload $wl --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --no-breakdown > $out/bench_$name.json 2> $out/bench_$name.err
python - <<P
import json
d=json.loads([l for l in open("$out/bench_$name.json") if l.startswith("{")][-1])
print("$name value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)))
P
}
run casc_sr cascade64_256 MINIMAGEN_ST_PIPE_SCOPE=sr
run casc_all cascade64_256 MINIMAGEN_ST_PIPE_SCOPE=all
run base_sr base64 MINIMAGEN_ST_PIPE_SCOPE=sr
run base_all base64 MINIMAGEN_ST_PIPE_SCOPE=all
run casc_sr_b cascade64_256 MINIMAGEN_ST_PIPE_SCOPE=sr
run casc_all_b cascade64_256 MINIMAGEN_ST_PIPE_SCOPE=all
