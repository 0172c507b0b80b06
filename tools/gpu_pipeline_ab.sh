cd "${GRAFT_REPO_ROOT:-/root/repo}"
python - <<'PY'
import torch, sys, time
sys.path.insert(0, ".")
import bench
from minimagen_amd import _lib as L
L.use_library(L.DEFAULT_LIB)
dev = torch.device("cuda:0")
im, sizes = bench.build_imagen("cascade64_256", 100, dev)
emb, mask = bench.synthetic_text(32)
emb, mask = emb.to(dev), mask.to(dev)
for k in range(3):
    im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=k, _async=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(3):
    im.sample(text_embeds=emb, text_masks=mask, cond_scale=3., _seed=99 + k, _async=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
PY
