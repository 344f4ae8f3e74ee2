"""The adversarial operands of tests/split_cases.py keep their promises (CPU): conv1 exact by construction, the cancelling pairs
really cancel, the float64 reference runs and is finite.  The GPU side is tests/test_gpu_parity.py::test_split_precision_error_vs_fp32_mfma."""
import pytest
import torch
import torch.nn.functional as F

from tests import split_cases as SC


@pytest.mark.parametrize("case", SC.CASES)
def test_case_construction(case):
    n = 2
    x, w1, b1, w2, b2, dp2 = SC.make(case, n)
    a32 = F.conv2d(x, w1, b1, stride=2, padding=1)
    a64 = F.conv2d(x.double(), w1.double(), b1.double(), stride=2, padding=1)
    if case != "end_to_end":
        assert torch.equal(a32.double(), a64)                      # one tap per channel, 12-bit pixel x 12-bit weight: exact in fp32
        assert int((a32 > 0).sum()) > 0.9 * a32.numel() or case.startswith("range")
    a1 = torch.relu(a32)
    y2 = F.conv2d(a1, w2, b2, stride=2, padding=1)
    assert torch.isfinite(y2).all() and torch.isfinite(dp2).all()
    if case.startswith("cancel"):
        # input channels 0..23 in pairs with identical a1, output channels 0..39 in pairs with identical rows
        assert torch.equal(a1[:, 0:24:2], a1[:, 1:24:2])
        assert torch.equal(w2[0:40:2], w2[1:40:2]) and torch.equal(y2[:, 0:40:2], y2[:, 1:40:2])
        if case == "cancel_exact":
            assert torch.equal(w2[:, 0:24:2], -w2[:, 1:24:2]) and torch.equal(dp2[:, 0:40:2], -dp2[:, 1:40:2])
            assert float(a1[:, :24].mean()) > 1e4 * float(a1[:, 24:].mean())      # the pairs sit 2^20 above the free channels
    # the float64 reference under the routing of a plain fp32 forward
    win = y2.view(n, 48, 16, 2, 16, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, 48, 16, 16, 4)
    am = win.argmax(-1)
    p2 = win.gather(4, am.unsqueeze(-1)).squeeze(-1)
    ref_p2, grads = SC.ref64(x, w1, b1, w2, b2, dp2, ((a32 > 0).float(), am, (p2 > 0).float()))
    assert torch.isfinite(ref_p2).all() and all(torch.isfinite(g).all() for g in grads)
    err = (torch.relu(p2).double() - ref_p2).abs().max() / ref_p2.abs().max().clamp_min(1e-300)
    assert float(err) < (0.5 if case.startswith("cancel") else 1e-5)          # (the cancelling cases lose the small result in fp32 too)
