out=gpurun_out/r06t; mkdir -p $out
run() { name=$1; shift; env "$@" python bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline --no-t5 --no-breakdown > $out/bench_$name.json 2> $out/bench_$name.err
python - <<P
import json
try:
    d=json.loads([l for l in open("$out/bench_$name.json") if l.startswith("{")][-1])
    print("$name value", round(d["value"]), "sync", round(d["value_no_pipeline"]), "one lane", round(d.get("value_one_lane",0)))
except Exception as e: print("$name failed", open("$out/bench_$name.err").read()[-300:])
P
}
run base X=1
run lanes3 MINIMAGEN_SAMPLE_LANES=3
run spg10 MINIMAGEN_STEPS_PER_GRAPH=10
run rp_pipe_S4 MINIMAGEN_RP_NTILE_PIPE_S=4
run rp_pipe_M2 MINIMAGEN_RP_NTILE_PIPE_M=2
run rp_pipe_off MINIMAGEN_RP_NTILE_PIPE_S=0 MINIMAGEN_RP_NTILE_PIPE_M=0
run st_pipe_S8 MINIMAGEN_ST_NBLK_PIPE_S=8
run st_pipe_M2L2 MINIMAGEN_ST_NBLK_PIPE_M=2 MINIMAGEN_ST_NBLK_PIPE_L=2
run base_b X=1
