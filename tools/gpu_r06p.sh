out=gpurun_out/r06p; mkdir -p $out
run() { name=$1; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --no-breakdown > $out/bench_$name.json 2> $out/bench_$name.err
python - <<P
import json
d=json.loads([l for l in open("$out/bench_$name.json") if l.startswith("{")][-1])
print("$name value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)))
P
}
run base X=1
run pipe_S4 MINIMAGEN_ST_NBLK_PIPE_S=4
run pipe_M2 MINIMAGEN_ST_NBLK_PIPE_M=2
run pipe_L2 MINIMAGEN_ST_NBLK_PIPE_L=2
run pipe_M2S4 MINIMAGEN_ST_NBLK_PIPE_M=2 MINIMAGEN_ST_NBLK_PIPE_S=4
run pipe_L2M2S4 MINIMAGEN_ST_NBLK_PIPE_L=2 MINIMAGEN_ST_NBLK_PIPE_M=2 MINIMAGEN_ST_NBLK_PIPE_S=4
run pipe_S1 MINIMAGEN_ST_NBLK_PIPE_S=1
run base_b X=1
