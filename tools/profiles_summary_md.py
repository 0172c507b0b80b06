"""profiles/<tag>_summary.md from the committed CSVs / bench lines of a round (after tools/summarize_profiles.py <tag>):
python tools/profiles_summary_md.py r03"""
import csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = lambda n: os.path.join("profiles", f"{tag}_{n}")
j = json.loads(open(P("bench_cascade64_256_n1.json")).read())
h = json.loads(open(P("bench_cascade64_256_half_n1.json")).read())
b = json.loads(open(P("bench_base64_n1.json")).read())
pmc = {(r["kernel"], r["grid"], r["wg"]): r for r in csv.DictReader(open(P("cascade_T25_pmc_by_launch_shape.csv")))}
tr = list(csv.DictReader(open(P("cascade_T25_kernel_trace_by_launch_shape.csv"))))
tot = sum(float(r["total_ms"]) for r in tr)
rf, ue = j["roofline"], j["unet_eval"]
out = [f"# Round {int(tag[1:])} profile summary (1 x MI355X)", "",
       f"Source: `tools/gpu_final.sh` on the GPU box -> `tools/summarize_profiles.py {tag}` -> `tools/profiles_summary_md.py {tag}`.  The rocprofv3 runs profile the bench command",
       "`bench.py --steps 1 --warmup 0 --timesteps 25 --no-cpu-baseline --no-secondary --no-breakdown --no-t5 --no-pipeline` (one cascade `sample()`, B=32, 25 steps per stage):",
       "kernel trace + stats in one run, SQ counters, FETCH_SIZE and WRITE_SIZE in separate `--pmc` runs.", "",
       f"Bench line of the same GPU call (`{tag}_bench_cascade64_256_n1.json`): **{j['value']:.0f} denoising-steps/s** pipelined over {j.get('lanes', 1)} call lanes "
       f"({j['ms_per_step']:.1f} ms per `sample()`, {j['steps']} timed calls; the second of two back-to-back pipelined calls is checked bit for bit against the synchronous call in the same run), "
       + (f"**{j['value_one_lane']:.0f}** with one lane (stage overlap of successive calls only), " if "value_one_lane" in j else "")
       + f"**{j['value_no_pipeline']:.0f}** with synchronous calls ({j['ms_per_step_no_pipeline']:.1f} ms); reduced precision (`{tag}_bench_cascade64_256_half_n1.json`): {h['value']:.0f}; "
       f"base 64^2 alone (`{tag}_bench_base64_n1.json`): {b['value']:.0f} / {b['value_no_pipeline']:.0f} synchronous;",
       f"one SR denoising step = {ue['graph_step_ms']:.3f} ms = {100 * ue['hbm_frac_graph_step']:.1f} % of the HBM roofline on {ue['alg_bytes_MB_per_image_forward']:.1f} MB per image-forward;",
       f"dominant launch {rf['kernel']}: {rf['kernel_ms'] * 1e3:.1f} us in program order = {rf['achieved']:.0f} TFLOP/s algorithmic = {100 * rf['frac']:.1f} % of the dense f16 MFMA peak.  "
       "Every kernel is built with -fno-slp-vectorize (profiles/r03_pk_f32_hazard.txt).", "",
       "| kernel (launch shape) | launches | avg us | share of GPU time | VALU busy (of wave cycles) | MFMA pipe busy | LDS bank conflicts | HBM read / written per launch (PMC) |",
       "|---|---|---|---|---|---|---|---|"]
for r in sorted(tr, key=lambda r: -float(r["total_ms"]))[:24]:
    c = pmc.get((r["kernel"], r["grid"], r["wg"]), {})
    g = lambda k, f="{:.1f}": (f.format(float(c[k])) if c.get(k) not in (None, "") else "-")
    out.append(f"| `{r['kernel']}` grid {r['grid']} x {r['wg']} | {r['launches']} | {float(r['avg_us']):.1f} | {100 * float(r['total_ms']) / tot:.1f} % | {g('active_valu_pct')} % | "
               f"{g('mfma_pipe_util_pct')} % | {g('lds_bank_conflict_pct')} % | {g('fetch_MB_x2')} / {g('write_MB')} MB |")
open(P("summary.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
