"""Pins the torch-CPU primitive semantics the oracle is built on (SURVEY.md
Appx B.2): the oracle has no reference to be checked against (parity unpinned),
so at least its building blocks are nailed down."""
import warnings

import torch
import torch.nn.functional as F


def test_affine_grid_grid_sample_defaults_are_align_corners_false_zeros_bilinear():
    torch.manual_seed(0)
    x = torch.randn(1, 3, 8, 8)
    theta = torch.tensor([[[0.9, -0.2, 0.1], [0.2, 0.9, -0.3]]])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g_def = F.affine_grid(theta, x.shape)
        y_def = F.grid_sample(x, g_def)
    g_exp = F.affine_grid(theta, x.shape, align_corners=False)
    y_exp = F.grid_sample(x, g_exp, mode="bilinear", padding_mode="zeros", align_corners=False)
    assert torch.equal(g_def, g_exp)
    assert torch.equal(y_def, y_exp)


def test_affine_grid_base_is_pixel_centres():
    theta = torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    g = F.affine_grid(theta, (1, 1, 4, 4), align_corners=False)
    expect = torch.tensor([(2 * j + 1) / 4 - 1 for j in range(4)])
    assert torch.allclose(g[0, 0, :, 0], expect, atol=1e-7)
    assert torch.allclose(g[0, :, 0, 1], expect, atol=1e-7)


def test_identity_warp_is_exact():
    torch.manual_seed(1)
    x = torch.randn(1, 4, 16, 16)
    theta = torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    y = F.grid_sample(x, F.affine_grid(theta, x.shape, align_corners=False), align_corners=False)
    assert torch.allclose(x, y, atol=1e-6)


def test_two_pass_warp_differs_from_composed_affine():
    # rotate -> zero-pad -> translate is NOT one resample (SURVEY.md Appx A.4)
    torch.manual_seed(2)
    x = torch.randn(1, 2, 32, 32)
    c, s = 0.9553365, 0.2955202
    rot = torch.tensor([[[c, -s, 0.0], [s, c, 0.0]]])
    tr = torch.tensor([[[1.0, 0.0, 0.3], [0.0, 1.0, -0.2]]])
    two = F.grid_sample(F.grid_sample(x, F.affine_grid(rot, x.shape, align_corners=False),
                                      align_corners=False),
                        F.affine_grid(tr, x.shape, align_corners=False), align_corners=False)
    comp = torch.tensor([[[c, -s, 0.3], [s, c, -0.2]]])
    one = F.grid_sample(x, F.affine_grid(comp, x.shape, align_corners=False), align_corners=False)
    assert (two - one).abs().max() > 0.5


def test_interpolate_default_is_nearest_floor_half():
    x = torch.arange(4.0).view(1, 1, 2, 2)
    y = F.interpolate(x, scale_factor=(2, 2))
    expect = torch.tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [2., 2., 3., 3.], [2., 2., 3., 3.]])
    assert torch.equal(y[0, 0], expect)


def test_exp_sum_softmax_matches_torch_softmax():
    torch.manual_seed(3)
    s = torch.rand(5, 32, 32) * 3
    e = torch.exp(s)
    w = e / e.sum(0, keepdim=True)
    assert torch.allclose(w, torch.softmax(s, 0), atol=1e-6)
