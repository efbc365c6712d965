"""a17: gated ``FeedForward`` (+ ``Conv2d_BN`` fold) and ``Illumination_Estimator`` of
RawFomer_WFB_FFAB/model.py:17-87, 174-200.  Golden = the reference classes executed from their source text
(oracle/make_golden.py --only-attenblock).  CPU: oracle vs golden; GPU: HIP path vs golden."""
import pytest
import torch

import cases
from cases import WFB_FF_CASES, WFB_IE_CASES, golden, params, rnd, wfb_ff_spec, wfb_ie_spec
from oracle import rawformer_ref as R

TOL = 2e-5


def _ff(tag, dim, fac, b, h, w):
    return rnd(f"wfb.{tag}.x", (b, dim, h, w), seed=51), params(wfb_ff_spec(dim, fac), seed=800 + dim)


def _ie(tag, mid, b, h, w):
    return rnd(f"wfb.{tag}.img", (b, 3, h, w), 0.0, 1.0, seed=52), params(wfb_ie_spec(mid), seed=900 + mid)


def test_oracle_matches_reference_wfb_extras():
    g = golden("wfb_extras")
    for case in WFB_FF_CASES:
        x, p = _ff(*case)
        assert float((R.wfb_feed_forward(x, p, "") - torch.from_numpy(g[f"{case[0]}.out"])).abs().max()) < 1e-6
    for case in WFB_IE_CASES:
        img, p = _ie(*case)
        fea, imap = R.illumination_estimator(img, p, "")
        assert float((fea - torch.from_numpy(g[f"{case[0]}.fea"])).abs().max()) < 1e-6
        assert float((imap - torch.from_numpy(g[f"{case[0]}.map"])).abs().max()) < 1e-6


def test_rep_conv_fold_equals_eval_mode_branches():
    """FeedForward.fuse(): x + BN(dw3(x)) + BN(dw1(x)) == dw3(x; folded) (host-side weight preparation)."""
    from bayer_low_light_image_enhancement_amd import ops
    tag, dim, fac, b, h, w = WFB_FF_CASES[0]
    _, p = _ff(tag, dim, fac, b, h, w)
    hid = rnd("fold.hid", (1, int(dim * fac), 9, 11))
    wa, ba = ops.fuse_rep_convs(p)
    ref = hid + R._conv_bn(hid, p, "rep_conv1.", 1) + R._conv_bn(hid, p, "rep_conv2.", 0)
    got = torch.nn.functional.conv2d(hid, wa, ba, padding=1, groups=hid.shape[1])
    assert float((got - ref).abs().max()) < 2e-6


@pytest.mark.gpu
def test_wfb_feed_forward_matches_reference(device):
    from bayer_low_light_image_enhancement_amd import ops
    g = golden("wfb_extras")
    for case in WFB_FF_CASES:
        x, p = _ff(*case)
        y = ops.wfb_feed_forward(x.to(device), {k: v.to(device) for k, v in p.items()}).cpu()
        assert float((y - torch.from_numpy(g[f"{case[0]}.out"])).abs().max()) < TOL, case[0]


@pytest.mark.gpu
def test_illumination_estimator_matches_reference(device):
    from bayer_low_light_image_enhancement_amd import ops
    g = golden("wfb_extras")
    for case in WFB_IE_CASES:
        img, p = _ie(*case)
        fea, imap = ops.illumination_estimator(img.to(device), {k: v.to(device) for k, v in p.items()})
        assert float((fea.cpu() - torch.from_numpy(g[f"{case[0]}.fea"])).abs().max()) < TOL, case[0]
        assert float((imap.cpu() - torch.from_numpy(g[f"{case[0]}.map"])).abs().max()) < TOL, case[0]
