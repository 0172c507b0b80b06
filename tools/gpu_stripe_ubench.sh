#!/bin/bash
# round 6: the stripe-pipeline skeleton (tools/ubench/stripe_pipe.hip) next to today's conv_rp_kernel on the same shapes, same box, isolated launches
mkdir -p gpurun_out/r06a
( cd tools/ubench && ./stripe_pipe ) > gpurun_out/r06a/stripe_pipe.txt 2>&1
{
for a in "64 8 8 256 256 1 id rp6" "64 8 8 256 256 1 none rp6" "64 8 8 128 128 1 id rp6" "64 8 8 128 128 1 none rp6" "64 16 8 128 128 1 none rp6" \
         "64 16 16 64 64 1 id rp6" "64 16 16 64 64 1 none rp6" "64 32 16 64 64 1 none rp6" "64 8 8 64 64 1 id rp6" "64 16 16 32 32 1 id rp7"; do
  python tools/bench_conv.py $a 2>&1 | grep -v "^ "
done
} > gpurun_out/r06a/conv_rp_isolated.txt 2>&1
tail -80 gpurun_out/r06a/stripe_pipe.txt; cat gpurun_out/r06a/conv_rp_isolated.txt
