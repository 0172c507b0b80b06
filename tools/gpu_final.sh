#!/bin/bash
# Round deliverables on the GPU box: the -m gpu tier, smoke, the driver-format bench line (+ per-launch breakdown), the reduced-precision and
# base-stage lines, ubenches, and the rocprofv3 passes of the SAME command (kernel trace + stats, then every PMC set in its own pass).
# Everything lands in gpurun_out/$TAG; tools/summarize_profiles.py <tag> turns the rocprofv3 outputs into the tracked summaries.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); TAG=${1:-final}; OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 900 python bench.py --breakdown-out $OUT/bd_cascade.json > $OUT/bench_cascade.log 2> $OUT/bench_cascade.err; echo "bench rc=$?"; tail -1 $OUT/bench_cascade.log | cut -c1-300
timeout 300 python bench.py --precision half --no-cpu-baseline --no-secondary --no-t5 --breakdown-out $OUT/bd_half.json > $OUT/bench_half.log 2>&1; tail -1 $OUT/bench_half.log | cut -c1-200
timeout 300 python bench.py --workload base64 --no-cpu-baseline --no-secondary --no-t5 --breakdown-out $OUT/bd_base.json > $OUT/bench_base.log 2>&1; tail -1 $OUT/bench_base.log | cut -c1-200
timeout 200 python tools/bench_sampler_tail.py 256 32 > $OUT/ubench_sampler_tail.txt 2>&1
timeout 100 python tools/bench_ce.py > $OUT/ubench_crossembed.txt 2>&1
CMD="python $ROOTDIR/bench.py --steps 1 --warmup 0 --timesteps 25 --no-cpu-baseline --no-secondary --no-breakdown --no-t5 --no-pipeline"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o cascade -- $CMD > $OUT/rocprof_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/prof_sq -o cascade -- $CMD > $OUT/rocprof_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof_sq2 -o cascade -- $CMD > $OUT/rocprof_sq2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/prof_fetch -o cascade -- $CMD > $OUT/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/prof_write -o cascade -- $CMD > $OUT/rocprof_write.log 2>&1
cd $ROOTDIR; python tools/summarize_profiles.py $TAG > $OUT/summarize.log 2>&1; tail -2 $OUT/summarize.log
# keep the merge small: the raw traces stay on the box, the summaries (profiles/${TAG}_*) travel through gpurun_out/profiles_$TAG
mkdir -p $OUT/profiles_$TAG; cp profiles/${TAG}_* $OUT/profiles_$TAG/ 2>/dev/null; rm -rf $OUT/prof_trace $OUT/prof_sq $OUT/prof_sq2 $OUT/prof_fetch $OUT/prof_write
ls $OUT/profiles_$TAG
