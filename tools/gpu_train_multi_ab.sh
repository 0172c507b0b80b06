#!/bin/bash
# one box: GPU tests of the training path and the host-side ABI checks, the SR training step (bench.py --train-step-only) twice, then the host profile of
# the step (tools/gpu_train_hostprof.py: time to issue ten steps against the time with the GPU drained, cProfile of the calling thread)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/train_multi; out=gpurun_out/train_multi
timeout 1200 python -m pytest tests/test_training.py tests/test_training_loop.py tests/test_host.py -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for tag in a b; do
  timeout 600 python bench.py --train-step-only > $out/train_$tag.json 2> $out/train_$tag.err
  python - <<PY
import json
d = json.loads([l for l in open("$out/train_$tag.json") if l.startswith("{")][-1])
h = d["hip_kernels"]
print("$tag", "fwd+bwd", round(h["ms_per_fwd_bwd"], 2), "with clip + Adam", round(h["ms_per_step_with_clip_and_adam"], 2), "loss", h["loss"], "torch ops", round(d["torch_ops_miopen"]["ms_per_fwd_bwd"], 2))
PY
done
timeout 600 python tools/gpu_train_hostprof.py > $out/prof.txt 2>&1; head -30 $out/prof.txt | cut -c1-150
