"""Host-side logic (no kernels): masks, weight-image packing, sparsity placement, harness plumbing."""
import numpy as np
import pytest
import torch

import mentflow_amd as mf
from mentflow_amd.generate import packing
from mentflow_amd.generate.masks import conditioner_masks
from mentflow_amd import distributions, harness, ops
from oracle import flow as of
from conftest import load_golden


@pytest.mark.parametrize("d", [2, 3, 4, 6, 7])
@pytest.mark.parametrize("reverse", [False, True])
@pytest.mark.parametrize("total", [2, 59])
def test_closed_form_masks_equal_zuko_procedure(d, reverse, total):
    order = torch.arange(d)
    if reverse:
        order = torch.flipud(order)
    ours = conditioner_masks(order, (64, 64, 64), total)
    ref = of.masked_mlp_masks(of.ar_adjacency(order, total), (64, 64, 64))
    assert len(ours) == len(ref) == 4
    for a, b in zip(ours, ref):
        assert torch.equal(a, b)


def _unpack_image(image, d, L, K):
    """Invert the packing with numpy: dense effective matrices in natural unit order."""
    g = packing.image_layout(d, L, d)
    phys = packing.hidden_placement(d)
    W0 = image[g["offW0"]:g["offW0"] + 64 * g["S0"]].reshape(64, g["S0"])[phys][:, :d]
    b0 = image[g["offB0"]:g["offB0"] + 64][phys]
    Ws, bs = [W0], [b0]
    for l in range(1, L):
        base = g["offWh"] + (l - 1) * (64 * 65 + 64)
        W = image[base:base + 64 * 65].reshape(64, 65)[phys][:, phys]
        Ws.append(W)
        bs.append(image[base + 64 * 65:base + 64 * 65 + 64][phys])
    q = 3 * K - 1
    W3 = np.zeros((d * q, 64), dtype=np.float32)
    b3 = np.zeros(d * q, dtype=np.float32)
    for i in range(d):
        blk = image[g["offW3"] + i * 64 * 65:g["offW3"] + (i + 1) * 64 * 65].reshape(64, 65)
        bb = image[g["offB3"] + i * 64:g["offB3"] + (i + 1) * 64]
        for r in range(64):
            hh, m = packing.slot_of_row(r)
            t = packing.rqs_logical_param(hh, m, K)
            if t >= 0:
                W3[i * q + t] = blk[r][phys]
                b3[i * q + t] = bb[r]
            else:
                assert not blk[r].any() and bb[r] == 0
    Ws.append(W3)
    bs.append(b3)
    return Ws, bs


@pytest.mark.parametrize("d,K", [(6, 20), (2, 20), (3, 8)])
def test_weight_image_packing_roundtrip(d, K):
    torch.manual_seed(0)
    gen = mf.generate.build_generator("nsf", input_features=d, output_features=d, hidden_layers=3, hidden_units=64,
                                      transforms=2, bins=K)
    image_index, grad_index, floats = gen.build_index_maps()
    assert floats == packing.image_layout(d, 3, d)["total"] and image_index.size == 2 * floats
    flat = gen.flat_parameters().detach().numpy()
    images = np.where(image_index >= 0, flat[np.maximum(image_index, 0)], 0.0).astype(np.float32).reshape(2, floats)
    x = torch.randn(13, d)
    for t, layer in enumerate(gen.layers):
        Ws, bs = _unpack_image(images[t], d, 3, K)
        h = x.numpy()
        for i, (W, b) in enumerate(zip(Ws, bs)):
            h = h @ W.T + b
            if i < 3:
                h = np.maximum(h, 0)
        lins = layer.linears()
        ref = of.conditioner(x, of.ARLayer(layer.order, [l.weight.detach() for l in lins], [l.bias.detach() for l in lins],
                                           [l.mask for l in lins]), 3 * K - 1)
        np.testing.assert_allclose(h.reshape(13, d, -1), ref.numpy(), rtol=1e-5, atol=1e-5)
    # gradient index is the inverse map; masked-out weights have no image position
    numel = flat.size
    pos = grad_index[grad_index >= 0]
    assert len(np.unique(pos)) == len(pos)
    assert (image_index[pos] == np.nonzero(grad_index >= 0)[0]).all()
    masks = np.concatenate([np.concatenate([l.mask.numpy().reshape(-1), np.ones(l.bias.numel(), bool)])
                            for layer in gen.layers for l in layer.linears()])
    assert ((grad_index >= 0) == masks).all() and numel == masks.size


@pytest.mark.parametrize("d", [2, 3, 6, 7])
def test_hidden_placement_is_class_sorted_per_kstep(d):
    phys = packing.hidden_placement(d)
    assert sorted(phys.tolist()) == list(range(64))
    cls = 1 + (np.arange(64) % (d - 1))
    # k-step s covers image rows (row0(s), row0(s) + 4); classes must be non-decreasing with s
    by_row = np.empty(64, dtype=int)
    by_row[phys] = cls
    seq = []
    for s_ in range(32):
        r0 = 32 * (s_ >> 4) + packing.rho(0, s_ & 15)
        seq += [by_row[r0], by_row[r0 + 4]]
    assert seq == sorted(seq)
    cum = packing.class_counts(d)
    assert cum[0] == 0 and cum[-1] == 64


def test_slot_row_maps_are_inverse():
    seen = set()
    for hh in (0, 1):
        for m in range(32):
            r = packing.rho(hh, m)
            assert packing.slot_of_row(r) == (hh, m)
            seen.add(r)
    assert seen == set(range(64))
    used = [packing.rqs_logical_param(hh, m, 20) for hh in (0, 1) for m in range(32)]
    assert sorted(t for t in used if t >= 0) == list(range(59))


def test_kde_radius():
    assert ops.kde_radius(0.5) == 4 and ops.kde_radius(1.0) == 9 and ops.kde_radius(0.1) == 1


@pytest.mark.parametrize("name,kws", [("rings", dict(ndim=6, seed=2, decay=0.2)), ("gaussian_mixture", dict(ndim=6, seed=0)),
                                      ("swissroll", dict(ndim=2, seed=21))])
def test_distributions_equal_reference_samples(name, kws):
    g = load_golden(f"ref_dist_{name}")
    x = distributions.get_distribution(name, **kws).sample(int(g["n"]))
    assert torch.equal(x[:2048], g["x"])
    torch.testing.assert_close(x.mean(0), g["mean"], rtol=0, atol=0)


@pytest.mark.parametrize("seed,P", [(2, 25), (0, 100)])
def test_harness_directions_equal_reference(seed, P):
    g = load_golden(f"ref_directions_seed{seed}_P{P}_d6")
    assert torch.equal(harness.make_directions(P, 6, seed), g["V"])
    tfs = harness.make_transforms_nd_1d(P, 6, seed)
    assert torch.equal(tfs[3].matrix[0], g["V"][3]) and torch.equal(tfs[3].matrix[1:], torch.eye(6)[1:])


def test_corner_and_rotation_optics_equal_reference():
    g = load_golden("ref_mentflow_loss_nd2d_corner15")
    assert torch.equal(torch.stack([t.matrix for t in harness.make_transforms_nd_2d_corner(6)]), g["matrices"])
    g = load_golden("ref_mentflow_loss_2d_P7")
    assert torch.equal(torch.stack([t.matrix for t in harness.make_transforms_2d_linear(7)]), g["matrices"])
    for t in harness.make_transforms_nd_2d_random(5, 6, 0):
        a, b = t.matrix[0], t.matrix[2]
        assert abs(float(a @ a) - 1) < 1e-6 and abs(float(b @ b) - 1) < 1e-6 and abs(float(a @ b)) < 1e-6


def test_group_measurements_sends_what_is_not_fused_to_the_generic_loop():
    diag = mf.diagnostics.Histogram1D(edges=torch.linspace(-1, 1, 9))
    lin = mf.simulate.LinearTransform(torch.eye(2))
    groups, generic = mf.simulate.group_measurements([torch.nn.Identity()], [[diag]])
    assert not groups and generic == [(0, 0)]                      # an arbitrary nn.Module transport
    groups, generic = mf.simulate.group_measurements([lin], [[mf.diagnostics.Projection(0)]])
    assert not groups and generic == [(0, 0)]                      # a non-histogram diagnostic
    kick_last = mf.simulate.CompositeTransform(lin, mf.simulate.MultipoleTransform(order=3, strength=0.5))
    groups, generic = mf.simulate.group_measurements([lin, kick_last], [[diag, mf.diagnostics.Projection(1)], [diag]])
    assert len(groups) == 1 and list(groups.values())[0][2] == [(0, 0)] and generic == [(0, 1), (1, 0)]
    # direction projections fold the transport matrix into the projection row
    d2 = mf.diagnostics.Histogram1D(edges=torch.linspace(-1, 1, 9), direction=torch.tensor([3.0, 4.0]))
    M = torch.tensor([[0.0, 1.0], [1.0, 0.0]])
    rows = d2.projection_rows(M)
    torch.testing.assert_close(rows[0], torch.tensor([0.8, 0.6]))
