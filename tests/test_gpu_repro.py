"""GPU: what round 2's miscompare came down to.

* the smallest operator-level input that separated right from wrong: the level-0 TransformerBlock at 2 x 32 x 512 x 512 -- 512
  workgroups of attn_front_kernel<32>, i.e. TWO co-resident workgroups on every CU (one image = 256 workgroups = one per CU never
  failed) -- and the headline shape 8 x 32 x 512 x 512, against the pinned oracle and for run-to-run bit stability;
* the hardware behaviour itself: packed-f32 operand-select forms beside bf16 MFMA bursts (tools/ubench/pk_opsel_probe.hip); the
  forms the build lets through must be exact.
"""
import os
import subprocess

import pytest
import torch

import cases
from cases import params, rnd
from bayer_low_light_image_enhancement_amd import ops
from oracle import rawformer_ref as R

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shape", [(2, 32, 512, 512), (8, 32, 512, 512)])
def test_level0_transformer_block_with_two_workgroups_per_cu(device, shape):
    c, heads = 32, 8
    p = params(cases.transformer_spec(c))
    x = rnd("tb.big.x", shape)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = R.transformer_block(x, p, "", heads)
    pd = {k: v.to(device) for k, v in p.items()}
    xd = x.to(device)
    first = ops.transformer_block(xd, pd, heads=heads)
    err = float((first.cpu() - ref).abs().max())
    assert err <= 2e-5, err                                   # the failing builds were at 0.24-0.52 here
    for _ in range(60):                                       # they also differed from run to run (1-60 % of the pixels)
        assert torch.equal(ops.transformer_block(xd, pd, heads=heads), first)


def test_packed_f32_forms_the_build_allows_are_exact_beside_bf16_mfma_bursts():
    exe = os.path.join(REPO, "tools", "ubench", "pk_opsel_probe")
    src = exe + ".hip"
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", src, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "forms the build allows with wrong results: 0" in r.stdout
