"""OTF degradation primitives (first slice of SURVEY.md 8f-1): `filter2d` against the REFERENCE function's own outputs
(tests/golden/filter2d.pt from utils/img_process_util.py), the 8-bit clamp/round, and the training pair pool's bookkeeping."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_filter2d_matches_reference_function(cuda):
    from trainner_redux_amd.data import otf

    fix = torch.load(os.path.join(os.path.dirname(__file__), "golden", "filter2d.pt"), weights_only=True)
    for tag, e in fix.items():
        out = otf.filter2d(e["img"].to(cuda), e["kernel"].to(cuda))
        err = rel_l2(out.cpu(), e["out"])
        assert out.shape == e["out"].shape and err < 2e-6, (tag, err)
    with pytest.raises(ValueError, match="Wrong kernel size"):
        otf.filter2d(torch.rand(1, 3, 8, 8, device=cuda), torch.rand(1, 4, 4, device=cuda))


def test_clamp_round_and_pair_pool(cuda):
    from trainner_redux_amd.data import otf

    x = torch.randn(2, 3, 9, 11, generator=torch.Generator().manual_seed(1)) * 0.7 + 0.5
    assert torch.equal(otf.clamp_round_u8(x.to(cuda)).cpu(), torch.clamp((x * 255.0).round(), 0, 255) / 255.0)
    pool = otf.PairPool(queue_size=6)
    seen = []
    for i in range(3):                                                  # filling: pairs pass through unchanged
        lq, gt = torch.full((2, 1, 2, 2), float(i), device=cuda), torch.full((2, 1, 4, 4), float(i), device=cuda)
        a, b = pool(lq, gt)
        assert torch.equal(a, lq) and torch.equal(b, gt)
    torch.manual_seed(0)
    lq, gt = torch.full((2, 1, 2, 2), 9.0, device=cuda), torch.full((2, 1, 4, 4), 9.0, device=cuda)
    a, b = pool(lq, gt)                                                 # full: shuffled, first b handed out, replaced by the newcomers
    assert a.shape == lq.shape and set(a.flatten().tolist()) <= {0.0, 1.0, 2.0}
    assert torch.equal(a[:, 0, 0, 0], b[:, 0, 0, 0])                    # lq / gt stay paired through the shuffle
    vals = sorted(pool.queue_lr[:, 0, 0, 0].tolist())
    assert vals.count(9.0) == 2 and len(vals) == 6
