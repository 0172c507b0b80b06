# T5 encode timing (bench.py's t5 leg), several repeats
python - <<'P'
import sys, torch, time
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
for k in range(3):
    r = bench.t5_leg(dev, 32)
    print("t5 encode B=32 L=64: %.3f ms  %.1f TFLOP/s algorithmic" % (r["ms"], r["tflops_algorithmic"]))
P
python -m pytest tests/test_t5.py -x -q -m gpu 2>&1 | tail -2
