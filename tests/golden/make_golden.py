"""Generate the golden fixtures in tests/golden/ by running the UNMODIFIED reference
(/root/reference, through oracle/ref_loader.py) on CPU.  Build-container only.

    python tests/golden/make_golden.py

Inputs are regenerated from the recorded seeds by tests (torch CPU generators are
deterministic for a given torch build; the GPU box runs the same image), so only
weights, outputs and metadata are stored.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ref_loader import load_reference, injected_noise, REFERENCE_ROOT  # noqa: E402
from oracle import restated as R  # noqa: E402  (only for synthetic_text / seeds helpers)

torch.set_num_threads(8)
ref = load_reference()


def perturb_norms(module, seed):
    """Default init leaves every norm at gamma=1/beta=0; perturb so affine handling is actually tested."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in list(module.named_parameters()) + list(module.named_buffers()):
            leaf = name.split(".")[-1]
            is_norm = ("groupnorm" in name or "norm" in name.split(".")[-2:-1] or leaf in ("gamma", "beta", "g")
                       or name.startswith("norm_cond") or ".to_out.1." in name or "to_text_non_attn_cond.0" in name)
            if not is_norm or "_temp" in name:
                continue
            if leaf in ("weight", "gamma", "g"):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif leaf in ("bias", "beta"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))


def seeded(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def main():
    p0 = json.load(open(os.path.join(REFERENCE_ROOT, "parameters/unet_0_params_20220816_165729.json")))
    p1 = json.load(open(os.path.join(REFERENCE_ROOT, "parameters/unet_1_params_20220816_165729.json")))
    torch.manual_seed(0)
    u0, u1 = ref.Unet(**p0), ref.Unet(**p1)
    T = 25
    im = ref.Imagen((u0, u1), text_encoder_name="t5_small", image_sizes=(64, 128), timesteps=T, cond_drop_prob=0.15)
    u0, u1 = im.unets[0], im.unets[1]          # u1 was re-instantiated with lowres_cond=True (Unet.py:340-353)
    perturb_norms(u0, 11)
    perturb_norms(u1, 12)
    im.eval()
    sd0 = {k: v.clone() for k, v in u0.state_dict().items()}
    sd1 = {k: v.clone() for k, v in u1.state_dict().items()}
    torch.save(sd0, os.path.join(HERE, "unet0_sd.pt"))
    torch.save(sd1, os.path.join(HERE, "unet1_sd.pt"))
    json.dump(dict(unet0=p0, unet1={**p1, "lowres_cond": True}), open(os.path.join(HERE, "unet_params.json"), "w"), indent=1)

    with torch.no_grad():
        # ---- forward A: unet_0 @64, B=2, ragged masks, distinct times
        emb, mask = R.synthetic_text(2, length=11, seed=7)
        mask[1, 6:] = False
        emb = emb.masked_fill(~mask[:, :, None], 0.)
        x = seeded((2, 3, 64, 64), 21)
        time = torch.tensor([17, 3])
        outA_c = u0(x, time, text_embeds=emb, text_mask=mask, cond_drop_prob=0.)
        outA_n = u0(x, time, text_embeds=emb, text_mask=mask, cond_drop_prob=1.)
        torch.save(dict(meta=dict(B=2, S=64, L=11, text_seed=7, x_seed=21, time=[17, 3], mask_cut=(1, 6)),
                        out_cond=outA_c, out_null=outA_n), os.path.join(HERE, "fwdA.pt"))
        # ---- forward B: unet_1 (lowres) @128, B=2
        xb = seeded((2, 3, 128, 128), 22)
        lr = seeded((2, 3, 128, 128), 23)
        timeb = torch.tensor([24, 0])
        ltime = torch.tensor([5, 5])
        outB_c = u1(xb, timeb, lowres_cond_img=lr, lowres_noise_times=ltime, text_embeds=emb, text_mask=mask, cond_drop_prob=0.)
        outB_n = u1(xb, timeb, lowres_cond_img=lr, lowres_noise_times=ltime, text_embeds=emb, text_mask=mask, cond_drop_prob=1.)
        torch.save(dict(meta=dict(B=2, S=128, L=11, text_seed=7, x_seed=22, lr_seed=23, time=[24, 0], ltime=[5, 5], mask_cut=(1, 6)),
                        out_cond=outB_c, out_null=outB_n), os.path.join(HERE, "fwdB.pt"))
        # ---- forward B at full 256, B=1 (cond only)
        xc = seeded((1, 3, 256, 256), 24)
        lrc = seeded((1, 3, 256, 256), 25)
        outC = u1(xc, torch.tensor([9]), lowres_cond_img=lrc, lowres_noise_times=torch.tensor([5]),
                  text_embeds=emb[:1], text_mask=mask[:1], cond_drop_prob=0.)
        torch.save(dict(meta=dict(B=1, S=256, L=11, text_seed=7, x_seed=24, lr_seed=25, time=[9], ltime=[5]),
                        out_cond=outC), os.path.join(HERE, "fwdC.pt"))
        # ---- one _p_sample step (unet_0), t=13 and t=0, cond_scale 3
        steps = {}
        for t in (13, 0):
            xt = seeded((2, 3, 64, 64), 30 + t)
            with injected_noise(40 + t):
                xprev = im._p_sample(u0, xt, torch.full((2,), t, dtype=torch.long), noise_scheduler=im.noise_schedulers[0],
                                     text_embeds=emb, text_mask=mask, cond_scale=3.)
            pred = u0.forward_with_cond_scale(xt, torch.full((2,), t, dtype=torch.long), text_embeds=emb, text_mask=mask, cond_scale=3.)
            x0 = im.noise_schedulers[0].predict_start_from_noise(xt, t=torch.full((2,), t, dtype=torch.long), noise=pred)
            s = torch.quantile(x0.reshape(2, -1).abs(), 0.9, dim=-1)
            steps[t] = dict(x_seed=30 + t, noise_seed=40 + t, x_prev=xprev, pred=pred, s_quantile=s)
        torch.save(dict(meta=dict(B=2, S=64, T=T, cond_scale=3., L=11, text_seed=7, mask_cut=(1, 6)), steps=steps),
                   os.path.join(HERE, "step.pt"))
        # ---- full sample(): base only and the 2-stage cascade
        im0 = ref.Imagen((u0,), text_encoder_name="t5_small", image_sizes=(64,), timesteps=T, cond_drop_prob=0.15)
        im0.unets[0].load_state_dict(sd0)
        for cs in (1., 3.):
            with injected_noise(1234):
                o = im0.sample(text_embeds=emb, text_masks=mask, cond_scale=cs)
            torch.save(dict(meta=dict(B=2, T=T, sizes=[64], cond_scale=cs, noise_seed=1234, L=11, text_seed=7, mask_cut=(1, 6)), out=o),
                       os.path.join(HERE, f"sample_base_cs{int(cs)}.pt"))
        with injected_noise(1234):
            o = im.sample(text_embeds=emb, text_masks=mask, cond_scale=3.)
        torch.save(dict(meta=dict(B=2, T=T, sizes=[64, 128], cond_scale=3., noise_seed=1234, L=11, text_seed=7, mask_cut=(1, 6),
                                  lowres_sample_noise_level=0.2), out=o), os.path.join(HERE, "sample_cascade.pt"))
        # ---- quantile known-answer tests (torch.quantile is what Imagen.py:313 calls)
        kats = []
        for n, seed in ((12288, 51), (196608, 52), (12288, 53), (48, 54), (3145728, 55)):
            v = seeded((3, n), seed).abs()
            if seed == 53:
                v = (v * 4).round() / 4          # heavy ties
            kats.append(dict(n=n, seed=seed, ties=(seed == 53), q=0.9, out=torch.quantile(v, 0.9, dim=-1)))
        torch.save(kats, os.path.join(HERE, "quantile.pt"))
        # ---- schedule tables
        tabs = {}
        for TT in (25, 100, 1000):
            gd = ref.GaussianDiffusion(timesteps=TT)
            tabs[TT] = {k: v.clone() for k, v in gd.named_buffers()}
        torch.save(tabs, os.path.join(HERE, "schedule.pt"))
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
