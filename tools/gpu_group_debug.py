"""Debug aid: the 3-stage cascade (config 5 shape), synchronous vs pipelined calls, with the grouped sampler tail; prints where they differ and
the kernels' error words."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                            # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 25
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
im, sizes = bench.build_imagen("cascade64_256_1024", T, dev)
emb, mask = bench.synthetic_text(B)
emb, mask = emb.to(dev), mask.to(dev)
kw = dict(text_embeds=emb, text_masks=mask, cond_scale=3.0, _precision="half")


def diff(a, b, name):
    ne = a != b
    print(f"{name}: {int(ne.sum())} differing elements, per image {ne.flatten(1).sum(1).tolist()}", flush=True)


a = im.sample(**kw, _seed=7).clone(); torch.cuda.synchronize()
a2 = im.sample(**kw, _seed=7).clone(); torch.cuda.synchronize()
diff(a, a2, "sync vs sync (lane 0)")
im.check_device_status()
x = im.sample(**kw, _seed=6, _async=True)
c = im.sample(**kw, _seed=7, _async=True)
torch.cuda.synchronize()
diff(a, c, "sync vs second of two pipelined calls")
try:
    im.check_device_status()
    print("error words clear")
except Exception as e:
    print("device status:", e)
x = im.sample(**kw, _seed=7, _async=True)
c = im.sample(**kw, _seed=7, _async=True)
torch.cuda.synchronize()
diff(a, x, "sync vs pipelined call A (same seed)")
diff(a, c, "sync vs pipelined call B (same seed)")
for unet in im.unets:
    for key, ws in unet.engine()._ws.items():
        for st in ws.__dict__.get("sampler_state", {}).values():
            gs = getattr(st, "group_sync", None)
            if gs is not None:
                w = gs[:16].cpu().view(torch.int64)
                cnt = gs[64:64 + 8 * ws.B].cpu().view(torch.int64)
                print(f"  ws {key[:4]} lane {key[7] if len(key) > 7 else 0}: tickets {int(w[0])}, error {int(w[1]) & 0xffffffff:#x}, counters {cnt.tolist()}")
