#!/bin/bash
# training path iteration on the GPU box: the GPU tier of the training tests, then the kernel profile of the SR training step (HIP path)
out=gpurun_out/train_iter; mkdir -p $out
python -m pytest tests/test_training.py tests/test_training_loop.py -x -q -m gpu > $out/pytest.log 2>&1; tail -2 $out/pytest.log
MODES=1 bash tools/gpu_train_profile.sh 2>&1 | tail -45
