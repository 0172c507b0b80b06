#!/bin/bash
# matrix-pipe utilisation of the training path's folded-attention kernels (rocprofv3 --pmc, its own pass: kernel trace + counters only) in the SR
# training step at B = 32: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), as tools/summarize_profiles.py computes it
R="${GRAFT_REPO_ROOT:-/root/repo}"; out=$R/gpurun_out/train_attn; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/attn_pmc
PROFILE_ONLY=1 B1=2 B2=32 timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d /tmp/attn_pmc -o t -- python $R/tools/gpu_train_step.py > $out/pmc.log 2>&1
f=$(find /tmp/attn_pmc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $out/attn_pmc.txt
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "folded_attn" not in k and "crossembed_wgrad_partial" not in k and "channel_stats" not in k:
        continue
    k = re.search(r"(\w+_kernel(<[^>]*>)?)", k).group(1)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[k] += 1
print("kernel, launches, mfma_pipe_util_pct, valu_insts_per_wave, active_valu_pct")
for k, m in sorted(acc.items()):
    util = 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024, 1)
    print(f"{k}, {n[k]}, {util:.1f}, {m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_WAVES', 1), 1):.0f}, {100 * m.get('SQ_ACTIVE_INST_VALU', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1) * 4:.1f}")
PY
