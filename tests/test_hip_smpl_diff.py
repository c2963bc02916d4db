"""-m gpu: the hand-written differentiable skinning (csrc/smpl.hip nm_smpl_vertex_forward / nm_smpl_vertex_backward behind
SMPLDiff.vertex_forward = HumanNeRF.vertex_forward, models/human_nerf.py:92-122 over models/smpl.py:266-360) against the same chain
as torch tensor algebra under torch's autograd (SMPLDiff.vertex_forward_torch, itself held to the reference's autograd by
tests/golden/smpl_grad.npz): outputs and the gradients to pose, shape and alignment, for gradients arriving through the vertices,
through the transforms, and through both."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def body():
    from neuman_hip import smpl, synthetic
    return smpl.SMPLDiff(synthetic.smpl_like_model(0), 'cuda'), synthetic


def leaf(x):
    return torch.tensor(np.ascontiguousarray(x), dtype=torch.float32, device='cuda', requires_grad=True)


@pytest.mark.parametrize("through", ["world", "T", "both"])
def test_forward_and_gradients_equal_torch_autograd(body, through):
    b, syn = body
    pose, betas, align = syn.smpl_like_frames(3, 0)
    g = torch.Generator(device='cuda').manual_seed(5)
    V = b.v_template.shape[0]
    gw = torch.randn((1, V, 3), device='cuda', generator=g)
    gT = torch.randn((1, V, 4, 4), device='cuda', generator=g)
    for f in range(3):
        al = np.concatenate([align[f'{f:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1).astype(np.float32)
        res = {}
        for name in ("hip", "torch"):
            p, be, a = leaf(pose[f][None] * 0.7), leaf(betas[f][None] * 0.5), leaf(al)
            fn = b.vertex_forward if name == "hip" else b.vertex_forward_torch
            world, T = fn(p, be, a, 1.3)
            loss = (world * gw).sum() * (through != "T") + (T * gT).sum() * (through != "world")
            loss.backward()
            res[name] = [x.detach().double().cpu().numpy() for x in (world, T, p.grad, be.grad, a.grad)]
        for what, x, y in zip(("world", "T", "g_pose", "g_beta", "g_align"), res["hip"], res["torch"]):
            scale = np.abs(y).max() + 1e-30
            e = np.abs(x - y).max() / scale
            print(f"[smpl-diff] frame {f}, gradient through {through}: {what} max |hip - torch| / max |torch| = {e:.2e}")
            assert e < (2e-5 if what in ("world", "T") else 2e-4), (f, what, e)


def test_run_to_run_bit_identical_and_no_leftover_state(body):
    b, syn = body
    pose, betas, align = syn.smpl_like_frames(2, 1)
    al = np.concatenate([align['00001.png'], np.array([[0.], [0.], [0.], [1.]])], 1).astype(np.float32)
    outs = []
    for _ in range(2):
        p, be, a = leaf(pose[1][None]), leaf(betas[1][None]), leaf(al)
        world, T = b.vertex_forward(p, be, a, 1.0)
        (world.square().sum() + T.square().sum()).backward()
        outs.append([x.detach().clone() for x in (world, T, p.grad, be.grad, a.grad)])
        b.vertex_forward(leaf(pose[0][None]), leaf(betas[0][None]), leaf(al), 2.0)               # another frame in between
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("runs", [False, True])
def test_fused_warp_apply_equals_the_reference_shaped_lines(body, runs):
    """(runs: consecutive points near one vertex, as neighbouring samples of a ray are -- the backward kernels add up a wave's runs of equal triangles
    before their atomics -- and a ragged count)
    ray_utils.warp_points_to_canonical_diff (one kernel each way for blend + inverse + product) against
    warp_samples_to_canonical_diff followed by the batched product, both under autograd: canonical points and the gradients to the
    vertex transforms and the posed vertices"""
    from neuman_hip import ray_utils
    b, syn = body
    pose, betas, align = syn.smpl_like_frames(1, 0)
    al = np.concatenate([align['00000.png'], np.array([[0.], [0.], [0.], [1.]])], 1).astype(np.float32)
    faces = syn.smpl_like_model(0)['f'].astype(np.int32)
    g = torch.Generator(device='cuda').manual_seed(9)
    res = {}
    for name in ("fused", "lines"):
        p, be, a = leaf(pose[0][None] * 0.5), leaf(betas[0][None] * 0.5), leaf(al)
        world, T = b.vertex_forward_torch(p, be, a, 1.0)
        verts, T = world[0], T[0]
        if name == "fused":
            g.manual_seed(9)
        else:
            g.manual_seed(9)
        n = 20001 if runs else 20000
        at = torch.randint(0, verts.shape[0], (n,), device='cuda', generator=g)
        if runs:                                                                     # runs of 1..12 points around one vertex
            at = torch.repeat_interleave(at[:n // 4], torch.randint(1, 13, (n // 4,), device='cuda', generator=g))[:n]
            assert at.shape[0] == n
        pts = (verts.detach()[at] + (0.004 if runs else 0.03) * torch.randn((n, 3), device='cuda', generator=g)).contiguous()
        gc = torch.randn((n, 3), device='cuda', generator=g)
        if name == "fused":
            can, _, _ = ray_utils.warp_points_to_canonical_diff(pts, verts, faces, T)
        else:
            Ts, _, _ = ray_utils.warp_samples_to_canonical_diff(pts, verts, faces, T)
            can = (Ts @ ray_utils.to_homogeneous(pts)[..., None])[:, :3, 0]
        (can * gc).sum().backward()
        res[name] = [x.detach().double().cpu().numpy() for x in (can, p.grad, be.grad, a.grad)]
    for what, x, y in zip(("can_pts", "g_pose", "g_beta", "g_align"), res["fused"], res["lines"]):
        e = np.abs(x - y).max() / (np.abs(y).max() + 1e-30)
        print(f"[warp-apply] {what}: max |fused - lines| / max |lines| = {e:.2e}")
        assert e < (2e-5 if what == "can_pts" else 5e-4), (what, e)


@pytest.mark.parametrize("runs", [False, True])
def test_barycentric_kernels_equal_the_reference_lines(runs):
    """nm_bary_forward / nm_bary_backward against the reference's cross / dot / divide lines (utils/ray_utils.py:72-84) in float64 under
    torch autograd: coordinates and the gradient that reaches the vertices"""
    from neuman_hip import ray_utils
    g = torch.Generator(device='cuda').manual_seed(4)
    V, N = 500, (30001 if runs else 30000)
    verts = torch.randn((V, 3), device='cuda', generator=g)
    pick = torch.randint(0, 64, (N,), device='cuda', generator=g)
    if runs:                                                                         # runs of 1..100 samples of one triangle (longer than a wave too), ragged count
        pick = torch.repeat_interleave(pick[:N // 8], torch.randint(1, 101, (N // 8,), device='cuda', generator=g))[:N]
        assert pick.shape[0] == N
    tri = torch.stack([torch.randperm(V, device='cuda', generator=g)[:3] for _ in range(64)])[pick].to(torch.int32).contiguous()
    wts = torch.rand((N, 3), device='cuda', generator=g)
    wts = wts / wts.sum(1, keepdim=True)
    closest = (verts[tri.long()] * wts[..., None]).sum(1).contiguous()            # points inside their triangles
    gb = torch.randn((N, 3), device='cuda', generator=g)
    v32 = verts.clone().requires_grad_(True)
    bary = ray_utils._BaryFn.apply(v32, tri, closest)
    (bary * gb).sum().backward()
    v64 = verts.double().requires_grad_(True)
    t = v64[tri.long()]
    c = closest.double()
    Nn = torch.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0], dim=1)
    den = (Nn * Nn).sum(1)
    u = (Nn * torch.cross(t[:, 2] - t[:, 1], c - t[:, 1], dim=1)).sum(1) / den
    v = (Nn * torch.cross(t[:, 0] - t[:, 2], c - t[:, 2], dim=1)).sum(1) / den
    ref = torch.stack([u, v, 1 - u - v], 1)
    (ref * gb.double()).sum().backward()
    eb = float((bary.detach().double() - ref.detach()).abs().max())
    eg = float((v32.grad.double() - v64.grad).abs().max() / v64.grad.abs().max())
    print(f"[bary] coordinates Linf {eb:.2e} (they reproduce the blend weights: {float((bary.detach() - wts).abs().max()):.2e}); vertex gradient {eg:.2e} of its largest entry")
    assert eb < 2e-4 and eg < 1e-4


def test_device_copy_follows_the_model_buffers():
    """the hand-written kernels work on a device copy of the body model made on first use: editing a buffer afterwards (a load_state_dict of
    HumanNeRF's body_model.* entries, an in-place change) rebuilds it, so they keep agreeing with the torch path that reads the buffers"""
    from neuman_hip import smpl, synthetic
    b = smpl.SMPLDiff(synthetic.smpl_like_model(0), 'cuda')
    pose, betas, align = synthetic.smpl_like_frames(1, 0)
    al = np.concatenate([align['00000.png'], np.array([[0.], [0.], [0.], [1.]])], 1).astype(np.float32)
    args = lambda: (leaf(pose[0][None] * 0.5), leaf(betas[0][None] * 0.5), leaf(al), 1.0)      # noqa: E731
    w0, _ = b.vertex_forward(*args())
    with torch.no_grad():
        b.v_template.mul_(1.1)
        b.shapedirs.add_(0.01)
    w1, T1 = b.vertex_forward(*args())
    w2, T2 = b.vertex_forward_torch(*args())
    assert float((w1 - w0).abs().max()) > 1e-3
    assert float((w1 - w2).abs().max()) < 2e-5 * float(w2.abs().max()) and float((T1 - T2).abs().max()) < 2e-5 * float(T2.abs().max())


def test_mesh_update_equals_a_fresh_build():
    """nm_mesh_update (the same faces, moved vertices: what every training iteration has) against a mesh built from scratch on the moved vertices:
    closest points, face ids and signed distances bit for bit, through two updates, and through the cached handle of the differentiable warp"""
    from neuman_hip import ray_utils, synthetic
    model = synthetic.smpl_like_model(0)
    faces = model['f'].astype(np.int32)
    g = torch.Generator(device='cuda').manual_seed(17)
    v0 = torch.as_tensor(np.asarray(model['v_template'], np.float32)).cuda()
    pts = (v0[torch.randint(0, v0.shape[0], (30000,), device='cuda', generator=g)] + 0.05 * torch.randn((30000, 3), device='cuda', generator=g)).contiguous()
    mesh = ray_utils.Mesh(v0, faces, None, 'cuda')
    ray_utils.signed_distance_dev(pts, mesh)                             # (builds the normals: the update must refresh them)
    for step in (1, 2):
        v = (v0 * (1.0 + 0.03 * step) + 0.01 * step * torch.sin(7.0 * v0.flip(1))).contiguous()       # a smooth, non-rigid move
        got = ray_utils.signed_distance_dev(pts, mesh.update(v))
        want = ray_utils.signed_distance_dev(pts, ray_utils.Mesh(v, faces, None, 'cuda'))
        for a, b, what in zip(got, want, ("signed distance", "face", "closest point")):
            assert torch.equal(a, b), (step, what)
        assert float((got[0] < 0).float().mean()) > 0.2 and float((got[0] > 0).float().mean()) > 0.2
    with pytest.raises(Exception):
        mesh.update(v[:-1])
    # the differentiable warp keeps ONE handle per face array and updates it: the second call (moved vertices) equals a cold call
    T = torch.eye(4, device='cuda').repeat(v0.shape[0], 1, 1).contiguous()
    ray_utils._DIFF_CACHE.clear()
    ray_utils.warp_points_to_canonical_diff(pts, v0, faces, T)
    warm = ray_utils.warp_points_to_canonical_diff(pts, v, faces, T)
    assert len(ray_utils._DIFF_CACHE) == 1
    ray_utils._DIFF_CACHE.clear()
    cold = ray_utils.warp_points_to_canonical_diff(pts, v, faces, T)
    for a, b in zip(warm, cold):
        assert torch.equal(a, b)


def test_mesh_update_with_non_finite_vertices_is_defined_and_recovers():
    """nm_mesh_update takes no read-back, so vertices that went NaN (diverged SMPL parameters) are not refused the way nm_mesh_create
    refuses them; the search must then not descend a tree built from NaN boxes: it finishes (the all-triangles loop on the device-side flag), and
    the next update with finite vertices gives a fresh build's answers again, bit for bit"""
    from neuman_hip import ray_utils, synthetic
    model = synthetic.smpl_like_model(0)
    faces = model['f'].astype(np.int32)
    v0 = torch.as_tensor(np.asarray(model['v_template'], np.float32)).cuda()
    g = torch.Generator(device='cuda').manual_seed(3)
    pts = (v0[torch.randint(0, v0.shape[0], (2048,), device='cuda', generator=g)] + 0.05 * torch.randn((2048, 3), device='cuda', generator=g)).contiguous()
    with pytest.raises(Exception):
        ray_utils.Mesh(torch.full_like(v0, float('nan')), faces, None, 'cuda')          # creation still refuses
    mesh = ray_utils.Mesh(v0, faces, None, 'cuda')
    want = ray_utils.signed_distance_dev(pts, mesh)
    bad = v0.clone()
    bad[100] = float('nan')
    s, f, c = ray_utils.signed_distance_dev(pts, mesh.update(bad))
    torch.cuda.synchronize()                                                            # (finishes: no descent of NaN boxes)
    assert int(f.min()) >= 0 and int(f.max()) < faces.shape[0]
    got = ray_utils.signed_distance_dev(pts, mesh.update(v0))
    for a, b in zip(got, want):
        assert torch.equal(a, b)
