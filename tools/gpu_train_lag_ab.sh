#!/bin/bash
# A/B on one box: the SR training step (bench.py --train-step-only) with begin_step's lagged scales (default) against the synchronous
# fingerprint path (MINIMAGEN_TRAIN_LAGGED_SCALES=0), twice each; before that the GPU tests that cover the mode and the one-rank RCCL tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/train_lag; mkdir -p $out
timeout 1200 python -m pytest tests/test_training.py tests/test_training_loop.py tests/test_generate.py -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
for tag in lag sync lag_b sync_b; do
  case $tag in lag*) v=1;; *) v=0;; esac
  MINIMAGEN_TRAIN_LAGGED_SCALES=$v timeout 600 python bench.py --train-step-only > $out/train_$tag.json 2> $out/train_$tag.err
  python - <<PY
import json
d = json.loads([l for l in open("$out/train_$tag.json") if l.startswith("{")][-1])
h = d["hip_kernels"]
print("$tag", "fwd+bwd", round(h["ms_per_fwd_bwd"], 2), "with clip + Adam", round(h["ms_per_step_with_clip_and_adam"], 2), "loss", h["loss"], "torch ops", round(d["torch_ops_miopen"]["ms_per_fwd_bwd"], 2))
PY
done
