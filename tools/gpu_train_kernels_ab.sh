#!/bin/bash
# one box: GPU tests of the training kernels, then the SR training step (bench.py --train-step-only) twice per setting of the CrossEmbed
# weight-gradient workgroup count, then the tail of the step's kernel profile (tools/gpu_train_profile.sh, HIP path)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/train_kernels; mkdir -p $out
timeout 1200 python -m pytest tests/test_training.py tests/test_kernels.py -x -q -m gpu -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
for tag in nwg512 nwg1024 nwg512_b nwg1024_b; do
  case $tag in nwg512*) v=512;; *) v=1024;; esac
  MINIMAGEN_CE_WGRAD_NWG=$v timeout 600 python bench.py --train-step-only > $out/train_$tag.json 2> $out/train_$tag.err
  python - <<PY
import json
d = json.loads([l for l in open("$out/train_$tag.json") if l.startswith("{")][-1])
h = d["hip_kernels"]
print("$tag", "fwd+bwd", round(h["ms_per_fwd_bwd"], 2), "with clip + Adam", round(h["ms_per_step_with_clip_and_adam"], 2), "loss", h["loss"], "torch ops", round(d["torch_ops_miopen"]["ms_per_fwd_bwd"], 2))
PY
done
MODES=1 bash tools/gpu_train_profile.sh 2>&1 | tail -30
