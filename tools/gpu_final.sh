#!/bin/bash
# Round deliverables on the GPU box: GPU test-suite, smoke, default bench (with cpu_baseline), rocprofv3 trace + PMC passes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOTDIR=$(pwd); OUT=$ROOTDIR/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py --breakdown-out $OUT/bd_cascade.json > $OUT/bench_cascade.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench_cascade.log
timeout 300 python bench.py --precision half --no-cpu-baseline --no-secondary --no-t5 --breakdown-out $OUT/bd_half.json > $OUT/bench_half.log 2>&1; tail -1 $OUT/bench_half.log | cut -c1-200
timeout 300 python bench.py --workload base64 --no-cpu-baseline --no-secondary --no-t5 --breakdown-out $OUT/bd_base.json > $OUT/bench_base.log 2>&1; tail -1 $OUT/bench_base.log | cut -c1-400
timeout 600 python tools/gpu_full_parity.py > $OUT/full_parity.txt 2>&1; tail -3 $OUT/full_parity.txt
# resident conv chains (opt-in): A/B lines and the phase trace of the chain kernel (library built with -DRS_TRACE)
for r in 1 2; do MINIMAGEN_RESIDENT=$r timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-t5 --breakdown-out $OUT/bd_resident$r.json > $OUT/bench_resident$r.log 2>&1; tail -1 $OUT/bench_resident$r.log | cut -c1-160; done
MINIMAGEN_RESIDENT=1 timeout 300 python bench.py --workload base64 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-t5 --no-breakdown > $OUT/bench_base_resident1.log 2>&1; tail -1 $OUT/bench_base_resident1.log | cut -c1-160
# A/B lines of this round's default-path changes (each knob back to the previous behaviour) and of the opt-in fused tail
for e in "MINIMAGEN_SAMPLER_GROUP=0" "MINIMAGEN_CONV_REVERSE=0" "MINIMAGEN_TAIL_FUSE=128"; do
  env $e timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-t5 > $OUT/bench_ab_${e%%=*}.log 2>&1; echo "$e $(tail -1 $OUT/bench_ab_${e%%=*}.log | cut -c1-130)"; done
timeout 200 python tools/bench_sampler_tail.py 256 32 > $OUT/ubench_sampler_tail.txt 2>&1; timeout 100 python tools/bench_sampler_tail.py 1024 4 >> $OUT/ubench_sampler_tail.txt 2>&1
timeout 200 python tools/bench_tail.py 64 256 256 0 > $OUT/ubench_conv_tail.txt 2>&1
timeout 100 python tools/bench_ce.py > $OUT/ubench_crossembed.txt 2>&1
if [ -f minimagen_amd/libminimagen_hip_trace.so ]; then MINIMAGEN_HIP_LIB=$ROOTDIR/minimagen_amd/libminimagen_hip_trace.so timeout 200 python tools/bench_resident.py 64 > $OUT/resident_trace.txt 2>&1; fi
timeout 200 python tools/bench_resident.py 64 > $OUT/resident_notrace.txt 2>&1; tail -1 $OUT/resident_notrace.txt
CMD="python $ROOTDIR/bench.py --steps 1 --warmup 0 --timesteps 25 --no-cpu-baseline --no-secondary --no-breakdown --no-t5 --no-pipeline"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -o cascade -- $CMD > $OUT/rocprof_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/prof_sq -o cascade -- $CMD > $OUT/rocprof_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/prof_sq2 -o cascade -- $CMD > $OUT/rocprof_sq2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/prof_fetch -o cascade -- $CMD > $OUT/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/prof_write -o cascade -- $CMD > $OUT/rocprof_write.log 2>&1
ls $OUT/prof_*/ | head -30
