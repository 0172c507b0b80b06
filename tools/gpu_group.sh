#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels.py tests/test_sampler.py -m gpu -q -x -k "group" --timeout 500 > $OUT/group_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/group_pytest.log
timeout 200 python tools/bench_sampler_tail.py 256 32 > $OUT/group_ubench.txt 2>&1; timeout 100 python tools/bench_sampler_tail.py 1024 4 >> $OUT/group_ubench.txt 2>&1; cat $OUT/group_ubench.txt
for g in 1 0; do MINIMAGEN_SAMPLER_GROUP=$g timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-t5 --no-breakdown > $OUT/bench_group$g.log 2>&1; tail -1 $OUT/bench_group$g.log | cut -c1-420; done
