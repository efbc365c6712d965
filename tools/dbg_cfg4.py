import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import cases
from bayer_low_light_image_enhancement_amd import RawFormer, synth
from oracle import rawformer_ref as R
dev = torch.device('cuda:0')
def build(dim, seed):
    m = RawFormer(dim=dim); sd = cases.model_state(dim, seed); m.load_state_dict(sd, strict=False); return m.to(dev).eval(), sd
# 1. mid-size L frame vs oracle
m, sd = build(64, 164)
for hw in ((352, 528), (704, 1056)):
    x = torch.from_numpy(synth.bayer_mosaic(10, 1, *hw))
    with torch.no_grad():
        out = m(x.to(dev)).cpu(); ref = R.rawformer_forward(sd, x, R.RawFormerConfig(dim=64))
    d = (out - ref).abs()
    print('L', hw, 'max', float(d.max()), 'mean', float(d.mean()), 'ref absmax', float(ref.abs().max()))
# 2. config 4 samples
gm = cases.golden('model_cfg4_L_1x1424x2128')
x = torch.from_numpy(synth.bayer_mosaic(10, 1, 2848, 4256)).to(dev)
with torch.no_grad():
    out = m(x)
idx = torch.from_numpy(gm['idx']).to(dev)
d = (out.reshape(-1)[idx].cpu() - torch.from_numpy(gm['samples'])).abs()
print('cfg4 samples: max', float(d.max()), 'mean', float(d.mean()), 'median', float(d.median()), 'frac>1e-4', float((d > 1e-4).float().mean()))
# where are the big errors?
bad = torch.nonzero(d > 1e-3).flatten()[:10]
H, W = 2848, 4256
for i in bad.tolist():
    f = int(gm['idx'][i]); c, r = divmod(f, H * W); y, xx = divmod(r, W)
    print('  bad sample c', c, 'y', y, 'x', xx, 'err', float(d[i]))
print('chan_mean diff', (out.mean(dim=(0, 2, 3)).cpu() - torch.from_numpy(gm['chan_mean'])).abs().tolist())
