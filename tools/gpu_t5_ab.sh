# T5 encode timing (bench.py's t5 leg) with 64-wide (default) and 32-wide (MI_GEMM_KS1) K slices of the f16x3 GEMM; the wide-preset step; tests
for v in "" "MI_GEMM_KS4=1"; do
env $v python - <<'P'
import sys, os, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
r = bench.t5_leg(dev, 32); r = bench.t5_leg(dev, 32)
print("KS4" if os.environ.get("MI_GEMM_KS4") else "KS2", "t5 encode B=32 L=64: %.3f ms  %.1f TFLOP/s algorithmic" % (r["ms"], r["tflops_algorithmic"]))
P
env $v python bench.py --wide-step-only 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('wide Unet() step ms', round(d['ms_per_denoising_step'],2))"
env $v python tools/bench_gemm.py 2>/dev/null | tail -8
done
python -m pytest tests/test_t5.py tests/test_unet.py -x -q -m gpu -k "t5 or gemm or default_unet or wide or preset" 2>&1 | tail -2
