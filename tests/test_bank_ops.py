"""The feature-bank bookkeeping ops (HIP replacement of the reference's CuNeMo extension) against the CPU
restatement oracle/cunemo_ref.py.  Index outputs bit-exact; float sums to float32 summation tolerance (the
reference itself accumulates with atomics in no defined order)."""
import numpy as np
import pytest
import torch

from oracle import cunemo_ref as CR


def _case(seed, N=5, M=3, C=24, mesh_n=(7, 5, 0, 9), pad=3):
    rng = np.random.default_rng(seed)
    L = len(mesh_n)
    idx = rng.integers(-1, L, size=(N, M)).astype(np.int32)
    idx[0] = -1
    idx[0, :2] = [1, 0]                                    # the reference's own example order for image 0
    idx[1, :] = -1                                         # an image without objects
    idx[2, 0] = -1                                         # an empty entry in front of real ones
    verts_num, shifts = CR.vert_shifts(idx, mesh_n)
    valid = (shifts[:, -1] + verts_num[:, -1]).astype(np.int32)
    K = int(valid.max()) + pad
    feat = rng.standard_normal((N, K, C)).astype(np.float32)
    w = (rng.random((N, K)) > 0.3).astype(np.float32) * rng.random((N, K)).astype(np.float32)
    bank_shifts = (np.cumsum(mesh_n) - np.asarray(mesh_n)).astype(np.int32)
    return dict(idx=idx, verts_num=verts_num, shifts=shifts, valid=valid, K=K, feat=feat, w=w, bank_shifts=bank_shifts,
                mesh_n=list(mesh_n), KK=int(sum(mesh_n)))


def test_oracle_on_the_reference_example():
    """The input example of cu_layers/CuNeMo/gather_features.py:80-96 (meshes of 7 and 5 vertices, one image
    holding mesh 1 then mesh 0): slots 0-4 are mesh 1 -> bank rows 7-11, slots 5-11 are mesh 0 -> rows 0-6."""
    rng = np.random.default_rng(0)
    feat = rng.random((1, 12, 3)).astype(np.float32)
    w = np.ones((1, 12), np.float32)
    f, ws, fmap = CR.gather_features(feat, w, np.array([12]), np.array([0, 7]), np.array([[0, 5]]), np.array([[1, 0]]), 12)
    assert fmap[0].tolist() == [7, 8, 9, 10, 11, 0, 1, 2, 3, 4, 5, 6]
    assert np.array_equal(f[7:12], feat[0, 0:5]) and np.array_equal(f[0:7], feat[0, 5:12])
    assert np.array_equal(ws, np.ones(12, np.float32))
    verts_num, shifts = CR.vert_shifts(np.array([[1, 0]], np.int32), [7, 5])
    assert verts_num.tolist() == [[5, 7]] and shifts.tolist() == [[0, 5]]
    oi, sv, vs, nv = CR.gather_idx(np.array([12]), shifts, np.array([[1, 0]]), verts_num, np.array([0, 7]), 14)
    assert oi[0].tolist() == [1] * 5 + [0] * 7 + [0, 0] and sv[0].tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5, 6, 0, 0]
    assert vs[0].tolist() == [7] * 5 + [0] * 9 and nv[0].tolist() == [5] * 5 + [7] * 7 + [0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,C", [(0, 24), (1, 128), (2, 200)])
def test_gather_features_forward_and_backward(gpu, seed, C):
    from nemo_amd import cu_nemo
    c = _case(seed, C=C)
    dev = gpu
    feat = torch.from_numpy(c["feat"]).to(dev).requires_grad_(True)
    w = torch.from_numpy(c["w"]).to(dev).requires_grad_(True)
    idx = torch.from_numpy(c["idx"]).to(dev)
    gf, gw = cu_nemo.gather_features(feat, w, idx, c["mesh_n"])
    rf, rw, rmap = CR.gather_features(c["feat"], c["w"], c["valid"], c["bank_shifts"], c["shifts"], c["idx"], c["KK"])
    assert gf.shape == (c["KK"], C) and gw.shape == (c["KK"],)
    assert np.abs(gf.detach().cpu().numpy() - rf).max() < 1e-5 * max(1.0, np.abs(rf).max())
    assert np.abs(gw.detach().cpu().numpy() - rw).max() < 1e-5
    # the map itself, through the autograd function
    _, _, fmap = cu_nemo._GatherFeatures.apply(feat, w, torch.from_numpy(c["valid"]).to(dev), torch.from_numpy(c["bank_shifts"]).to(dev),
                                               torch.from_numpy(c["shifts"]).to(dev), idx, c["KK"])
    assert np.array_equal(fmap.cpu().numpy(), rmap)
    # backward against the restated kernel and against torch autograd of the same sum written with index_add
    g = torch.Generator().manual_seed(seed)
    up_f, up_w = torch.randn(c["KK"], C, generator=g), torch.randn(c["KK"], generator=g)
    (gf * up_f.to(dev)).sum().backward(retain_graph=True)
    (gw * up_w.to(dev)).sum().backward()
    bf, bw = CR.gather_features_backward(up_f.numpy(), up_w.numpy(), c["feat"], c["w"], rmap)
    assert np.abs(feat.grad.cpu().numpy() - bf).max() < 1e-5 * max(1.0, np.abs(bf).max())
    assert np.abs(w.grad.cpu().numpy() - bw).max() < 1e-4 * max(1.0, np.abs(bw).max())
    ft, wt = torch.from_numpy(c["feat"]).requires_grad_(True), torch.from_numpy(c["w"]).requires_grad_(True)
    sel = torch.from_numpy(rmap.reshape(-1) >= 0)
    rows = torch.from_numpy(rmap.reshape(-1)).long()[sel]
    tf = torch.zeros(c["KK"], C).index_add(0, rows, (ft * wt[..., None]).reshape(-1, C)[sel])
    tw = torch.zeros(c["KK"]).index_add(0, rows, wt.reshape(-1)[sel])
    ((tf * up_f).sum() + (tw * up_w).sum()).backward()
    assert float((feat.grad.cpu() - ft.grad).abs().max()) < 1e-5 * max(1.0, float(ft.grad.abs().max()))
    assert float((w.grad.cpu() - wt.grad).abs().max()) < 1e-4 * max(1.0, float(wt.grad.abs().max()))


@pytest.mark.gpu
def test_gather_is_reproducible_and_writes_every_row(gpu):
    """Two calls give the same bits (fixed summation order); bank rows no slot maps to come back as zeros."""
    from nemo_amd import cu_nemo
    c = _case(3, N=9, M=4, C=64)
    feat, w = torch.from_numpy(c["feat"]).to(gpu), torch.from_numpy(c["w"]).to(gpu)
    idx = torch.from_numpy(c["idx"]).to(gpu)
    a, aw = cu_nemo.gather_features(feat, w, idx, c["mesh_n"])
    b, bw = cu_nemo.gather_features(feat, w, idx, c["mesh_n"])
    assert torch.equal(a, b) and torch.equal(aw, bw)
    _, _, rmap = CR.gather_features(c["feat"], c["w"], c["valid"], c["bank_shifts"], c["shifts"], c["idx"], c["KK"])
    unused = np.setdiff1d(np.arange(c["KK"]), rmap[rmap >= 0])
    if unused.size:
        assert float(a[torch.from_numpy(unused).to(gpu)].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("with_sel", [False, True])
def test_get_mask_matches_restatement(gpu, with_sel):
    from nemo_amd import cu_nemo
    c = _case(5, N=4, M=3, mesh_n=(6, 4, 8))
    n_noise, L = 3, 3
    vmax = max(c["mesh_n"])
    rng = np.random.default_rng(9)
    vdw = rng.random((L, vmax, vmax)).astype(np.float32)
    total = c["KK"] + n_noise
    idx = torch.from_numpy(c["idx"]).to(gpu)
    sel = None
    if with_sel:
        sel_np = rng.random((4, c["K"])) > 0.5
        sel = torch.from_numpy(sel_np).to(gpu)
    out, vert_index = cu_nemo.get_mask(torch.from_numpy(vdw).to(gpu), idx, c["mesh_n"], total, c["K"], 7.5, 2.25, mask_sel=sel, n_noise=n_noise)
    oi, sv, vs, nv = CR.gather_idx(c["valid"], c["shifts"], c["idx"], c["verts_num"], c["bank_shifts"], c["K"])
    g = cu_nemo.gather_idx(torch.from_numpy(c["valid"]).to(gpu), torch.from_numpy(c["shifts"]).to(gpu), idx,
                           torch.from_numpy(c["verts_num"]).to(gpu), torch.from_numpy(c["bank_shifts"]).to(gpu), c["K"])
    for got, ref in zip(g, (oi, sv, vs, nv)):
        assert np.array_equal(got.cpu().numpy(), ref)
    if with_sel:
        oi, sv, vs, nv = oi[sel_np], sv[sel_np], vs[sel_np], nv[sel_np]
    else:
        oi, sv, vs, nv = oi.reshape(-1), sv.reshape(-1), vs.reshape(-1), nv.reshape(-1)
    ref = CR.mask_weight(oi, sv, vs, vs + nv, vdw, 7.5, 2.25, total - n_noise, total)
    assert out.shape == ref.shape and np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(vert_index.cpu().numpy(), c["bank_shifts"][oi].astype(np.int64) + sv)


@pytest.mark.gpu
def test_feature_bank_multi_class_update(gpu):
    """FeatureBankNeMo.forward with object_labels: the momentum update uses the gathered visible features."""
    from nemo_amd.feature_banks import FeatureBankNeMo, MaskCreater
    torch.manual_seed(0)
    mesh_n, C, noise, B = [5, 6], 16, 2, 3
    bank = FeatureBankNeMo(input_size=C, num_pos=sum(mesh_n), num_noise=noise, max_groups=4, momentum=0.9, mesh_n_list=mesh_n).cuda()
    labels = torch.tensor([[0], [1], [0]], dtype=torch.int32, device=gpu)
    K = max(mesh_n)
    x = torch.nn.functional.normalize(torch.randn(B, K + noise, C), dim=2).to(gpu)
    visible = (torch.rand(B, K) > 0.3).to(gpu)
    for b, l in enumerate([0, 1, 0]):
        visible[b, mesh_n[l]:] = False
    before = bank.memory_pos.clone()
    creater = MaskCreater(dist_thr=0.5, kappas={"pos": 0, "near": 3, "clutter": 4, "class": 2}, n_noise=noise * 4,
                          verts_ori=[torch.rand(n, 3) for n in mesh_n], device=gpu)
    weight, vert_index = creater(sample_indexs=labels, K_padded=K, vis_mask=visible)
    sim, noise_sim = bank(x, visible, object_labels=labels, vis_mask=visible)
    assert sim.shape == (int(visible.sum()), sum(mesh_n) + noise * 4) and weight.shape == sim.shape
    assert vert_index.shape == (int(visible.sum()),)
    # expected update in plain torch
    acc = torch.zeros(sum(mesh_n), C, device=gpu)
    starts = [0, mesh_n[0]]
    for b, l in enumerate([0, 1, 0]):
        n = mesh_n[l]
        acc[starts[l]:starts[l] + n] += x[b, :n] * visible[b, :n, None].float()
    want = torch.nn.functional.normalize(before * 0.9 + acc / B * 0.1, dim=1)
    assert float((bank.memory_pos - want).abs().max()) < 1e-6
