"""mtgs_amd.densify.update_statistics against the reference's masked-tensor formulation
(/root/reference/mtgs/scene_model/mtgs_scene_graph.py:1157-1183 and
 /root/reference/mtgs/scene_model/gaussian_model/vanilla_gaussian_splatting.py:448-474), restated line by line."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(stats, radii, xys_absgrad, submodel_mask, W, H):
    """update_submodel_statistics + after_train for one node (tensors on CPU)."""
    xys_grad_norm, vis_counts, max_2Dsize = stats
    grads = xys_absgrad[0, submodel_mask].detach()
    image_size = grads.new_tensor([W, H]).unsqueeze(0)
    grads = (grads * image_size * 0.5).norm(dim=-1)
    node_radii = radii[0, submodel_mask]
    visible_mask = (node_radii > 0).flatten()
    vis_counts[visible_mask] += +1
    xys_grad_norm[visible_mask] += grads[visible_mask]
    newradii = node_radii.detach()[visible_mask]
    max_2Dsize[visible_mask] = torch.maximum(max_2Dsize[visible_mask], newradii)


@pytest.mark.parametrize("sizes", [(5000,), (1200, 1, 3333, 64)])
def test_update_statistics_matches_reference(hip_lib, sizes):
    from mtgs_amd.densify import update_statistics
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(4)
    N, W, H = sum(sizes), 960, 540
    model_id = torch.cat([torch.full((n,), i) for i, n in enumerate(sizes)])
    stats_ref = [[torch.rand(n, generator=g), torch.ones(n) + torch.randint(0, 5, (n,), generator=g).float(),
                  torch.rand(n, generator=g) * 30] for n in sizes]
    stats_dev = [[t.clone().to(dev) for t in s] for s in stats_ref]
    for step in range(3):
        radii = (torch.randint(0, 40, (1, N), generator=g) * (torch.rand(1, N, generator=g) < 0.3)).int()
        absgrad = torch.rand(1, N, 2, generator=g) * 1e-3
        start = 0
        for i, n in enumerate(sizes):
            _reference(stats_ref[i], radii, absgrad, model_id == i, W, H)
            update_statistics(*stats_dev[i], radii.to(dev), absgrad.to(dev), W, H, start=start)
            start += n
    for sr, sd in zip(stats_ref, stats_dev):
        for a, b, name in zip(sr, sd, ("xys_grad_norm", "vis_counts", "max_2Dsize")):
            assert torch.allclose(a, b.cpu(), rtol=1e-6, atol=1e-7), name


def test_update_statistics_all_nodes_one_launch(hip_lib):
    """update_statistics_all (one launch for the scene graph) == update_statistics per node, bit for bit, with 120 nodes
    of awkward sizes (0, 1, 255, 256, 257 ...)."""
    from mtgs_amd.densify import update_statistics, update_statistics_all
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    sizes = [4000, 0, 1, 255, 256, 257, 63, 64] + [int(x) for x in torch.randint(1, 700, (112,), generator=g)]
    N, W, H = sum(sizes), 960, 540
    mk = lambda: [[torch.rand(n, generator=g).to(dev), (torch.ones(n) + torch.randint(0, 5, (n,), generator=g).float()).to(dev),
                   (torch.rand(n, generator=g) * 30).to(dev)] for n in sizes]
    a = mk()
    b = [[t.clone() for t in s] for s in a]
    for step in range(2):
        radii = (torch.randint(0, 40, (1, N), generator=g) * (torch.rand(1, N, generator=g) < 0.3)).int().to(dev)
        absgrad = (torch.rand(1, N, 2, generator=g) * 1e-3).to(dev)
        start = 0
        for i, n in enumerate(sizes):
            update_statistics(*a[i], radii, absgrad, W, H, start=start)
            start += n
        update_statistics_all([tuple(s) for s in b], radii, absgrad, W, H)
    for sa, sb in zip(a, b):
        for x, y in zip(sa, sb):
            assert torch.equal(x, y)
