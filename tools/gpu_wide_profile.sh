#!/bin/bash
# rocprofv3 kernel stats of sampling with the default Unet() (tools/gpu_wide_sample.py) -> gpurun_out/wide_sample_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/wide_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wide_prof -o wide -- python $R/tools/gpu_wide_sample.py ${B:-16} 25 > $R/gpurun_out/wide_prof.log 2>&1
f=$(find /tmp/wide_prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/wide_sample_kernel_stats.csv; head -22 "$f" | cut -c1-200; fi
tail -1 $R/gpurun_out/wide_prof.log
