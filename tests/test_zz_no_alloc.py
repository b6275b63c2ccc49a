"""The pose-grid scoring entry point that allocates nothing (nemo_grid_score_static), called through the C ABI
and held to exact equality with nemo_grid_score (same kernels, the partial sums in the library's fixed device
array instead of a caller workspace): two calls in a row, the second with a smaller input, plus the refusal of a
size beyond the array.  Collected last so that a failure here does not hide other tests."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(N, B, C, H, W, seed, dev):
    g = torch.Generator().manual_seed(seed)
    pre = torch.nn.functional.normalize(torch.randn(N, C, H, W, generator=g), dim=1).to(dev).contiguous()
    pre[:, :, : H // 3] = 0.0                               # background rows, as in rendered maps
    fm = torch.nn.functional.normalize(torch.randn(B, C, H, W, generator=g), dim=1).to(dev).contiguous()
    clu = (0.1 * torch.randn(B, H, W, generator=g)).to(dev).contiguous()
    return pre, fm, clu


def _both(pre, fm, clu):
    from nemo_amd import ops
    N, B = pre.shape[0], fm.shape[0]
    loss = torch.full((B, N), float("nan"), device=fm.device)
    amin = torch.full((B,), -7, dtype=torch.int32, device=fm.device)
    mn = torch.full((B,), float("nan"), device=fm.device)
    ops.grid_score_static(pre, fm, clu, loss, amin, mn)
    ref_loss, ref_amin, ref_mn = ops.grid_score(pre, fm, clu)
    torch.cuda.synchronize()
    return (loss, amin, mn), (ref_loss, ref_amin, ref_mn)


def test_static_scratch_grid_score_matches_workspace_form(gpu):
    from nemo_amd import _lib
    # first call: 144 poses x 9 images on a 40 x 50 map (8 pixel tiles); second, smaller: 37 poses x 3 images on 17 x 23
    for shape, seed in (((144, 9, 32, 40, 50), 0), ((37, 3, 24, 17, 23), 1)):
        pre, fm, clu = _inputs(*shape, seed, gpu)
        got, ref = _both(pre, fm, clu)
        assert torch.equal(got[0], ref[0]), "losses differ"
        assert torch.equal(got[1], ref[1]), "arg-min differs"
        assert torch.equal(got[2], ref[2]), "minimum differs"
        assert torch.equal(got[1].long(), got[0].argmin(dim=1))
    # without a clutter map
    pre, fm, _ = _inputs(20, 2, 16, 10, 12, 2, gpu)
    got, ref = _both(pre, fm, None)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    # partial sums beyond the fixed array are refused before any launch
    lib = _lib.load()
    limit = 16 << 20
    N, B, HW = 1024, 1024, 8000
    assert lib.nemo_grid_score_workspace(N, B, HW) > limit
    one = torch.zeros(1, device=gpu)
    amin = torch.zeros(1, dtype=torch.int32, device=gpu)
    rc = lib.nemo_grid_score_static(_lib.ptr(one), _lib.ptr(one), None, N, B, 16, HW, None, _lib.ptr(amin, torch.int32), None, None)
    assert rc == -2 and b"static scratch" in lib.nemo_last_error()
