"""Pin the CPU oracle against the reference's own outputs (tests/golden/*.npz, made by make_golden.py)."""
import types

import numpy as np

from oracle import compositing, nerf_mlp, ray_ops, render


def test_shot_rays(golden):
    g = golden['ray_ops']
    coords = ray_ops.all_pixel_coords((12, 16))
    o, d = ray_ops.shot_rays(g['cam_K'], g['cam_c2w'], coords)
    np.testing.assert_allclose(o, g['shot_rays_o'], atol=0)
    np.testing.assert_allclose(d, g['shot_rays_d'], atol=1e-12)
    o, d = ray_ops.shot_all_rays(g['cam_K'], g['cam_c2w'], (12, 16))
    np.testing.assert_allclose(d, g['shot_all_d'], atol=1e-12)
    np.testing.assert_allclose(o, g['shot_all_o'], atol=0)


def test_linspace(golden):
    np.testing.assert_allclose(ray_ops.linspace_f32(0, 1, 32), golden['ray_ops']['linspace32'], atol=6e-8)


def test_ray_to_samples(golden):
    g = golden['ray_ops']
    a = (g['rs_o'], g['rs_d'], g['rs_near'], g['rs_far'], 32)
    for tag, kw in [('lin', {}), ('disp', {'lindisp': True})]:
        p, d, z = ray_ops.ray_to_samples(*a, **kw)
        np.testing.assert_allclose(z, g[f'rs_{tag}_z'], atol=5e-7)
        np.testing.assert_allclose(p, g[f'rs_{tag}_pts'], atol=2e-6)
        np.testing.assert_array_equal(d, g[f'rs_{tag}_dirs'])
    p, d, z = ray_ops.ray_to_samples(*a, t_rand=g['rs_perturb_trand'])
    np.testing.assert_allclose(z, g['rs_perturb_z'], atol=1e-6)
    np.testing.assert_allclose(p, g['rs_perturb_pts'], atol=2e-6)
    # with the reference's own t the lerp is bit-exact
    _, _, z = ray_ops.ray_to_samples(*a, t_vals=g['linspace32'])
    np.testing.assert_array_equal(z, g['rs_lin_z'])


def close_except_cdf_ties(a, ref, atol, max_frac=0.015, max_jump=1.0):
    """The inverse-CDF lookup is a step function of (u - cdf[i]): where u lands within an ulp of a cdf entry
    (always possible for u = 1 against cdf[-1] ~ 1) a different summation order picks the neighbouring bin
    (reference ray_utils.py:180-192; SURVEY H2-class discontinuity).  Those few samples may differ by up to a bin."""
    bad = np.abs(a - ref) > atol
    assert bad.mean() <= max_frac, f"{bad.sum()} of {bad.size} samples differ"
    assert np.abs(a - ref).max() <= max_jump
    return bad


def test_sample_pdf_and_importance(golden):
    g = golden['ray_ops']
    s = ray_ops.sample_pdf(g['pdf_bins'], g['pdf_w'], 16)
    close_except_cdf_ties(s, g['pdf_samples'], 2e-6, max_jump=np.diff(g['pdf_bins'], axis=1).max())
    p, d, z = ray_ops.ray_to_importance_samples(g['rs_o'], g['rs_d'], g['rs_lin_z'], g['imp_w'], 24)
    gap = np.diff(g['rs_lin_z'], axis=1).max()
    bad = close_except_cdf_ties(z, g['imp_z'], 2e-6, max_jump=gap)
    assert np.abs(p - g['imp_pts'])[~bad].max() < 5e-6
    assert (np.diff(z, axis=1) >= 0).all()
    _, _, z = ray_ops.ray_to_importance_samples(g['rs_o'], g['rs_d'], g['rs_lin_z'], g['imp_w'], 24, including_old=False)
    close_except_cdf_ties(z, g['imp_z_new_only'], 2e-6, max_jump=gap)


def test_near_far(golden):
    g = golden['ray_ops']
    n, f = ray_ops.geometry_guided_near_far(g['nf_o'], g['nf_d'], g['nf_verts'], 0.2)
    for ref_n, ref_f in [(g['nf_near_torch'], g['nf_far_torch']), (g['nf_near_np'], g['nf_far_np'])]:
        hit = ref_n < ref_f
        assert 5 < hit.sum() < 60
        np.testing.assert_array_equal(n < f, hit)
        np.testing.assert_allclose(n[hit], ref_n[hit], atol=2e-5)
        np.testing.assert_allclose(f[hit], ref_f[hit], atol=2e-5)
        assert np.isposinf(n[~hit]).all() and np.isneginf(f[~hit]).all()


def test_raw2outputs(golden):
    g = golden['ray_ops']
    for tag, wb in [('white', True), ('black', False)]:
        rgb, disp, acc, w, depth = compositing.raw2outputs(g['c_raw'], g['c_z'], g['c_d'], white_bkg=wb)
        np.testing.assert_allclose(w, g[f'c_{tag}_w'], atol=2e-7)
        np.testing.assert_allclose(rgb, g[f'c_{tag}_rgb'], atol=1e-6)
        np.testing.assert_allclose(acc, g[f'c_{tag}_acc'], atol=1e-6)
        np.testing.assert_allclose(depth, g[f'c_{tag}_depth'], atol=2e-6)
        np.testing.assert_allclose(disp, g[f'c_{tag}_disp'], rtol=1e-5)


def test_pe_and_mlp(golden, nets):
    g = golden['mlp']
    for seed, mapping in [(0, 'posenc'), (2, 'rotate')]:
        _, sd, spec = nets[seed]
        np.testing.assert_allclose(nerf_mlp.embed(g['pts'], mapping, *spec.pos), g[f'{mapping}_pos_pe'], atol=2e-4 if mapping == 'rotate' else 1e-6)
        np.testing.assert_allclose(nerf_mlp.embed(g['dirs'], mapping, *spec.dir), g[f'{mapping}_dir_pe'], atol=1e-6)
        out = nerf_mlp.joiner_forward(sd, spec, g['pts'], g['dirs'])
        ref = g[f'{mapping}_out']
        np.testing.assert_allclose(out[:, :3], ref[:, :3], atol=2e-5 if mapping == 'posenc' else 2e-4)
        np.testing.assert_allclose(out[:, 3], ref[:, 3], atol=1e-4 if mapping == 'posenc' else 1e-3)


def _cap(shape, c2w, fx):
    return types.SimpleNamespace(shape=shape, intrinsic_matrix=np.array([[fx, 0, shape[1] / 2], [0, fx, shape[0] / 2], [0, 0, 1.]]),
                                 cam_pose=types.SimpleNamespace(camera_to_world=c2w), near={'bkg': 0.0}, far={'bkg': 3.14})


def test_render_vanilla_c1(golden, nets):
    g = golden['render']
    cap = _cap((64, 64), g['c1_c2w'], 1.25 * 64)
    coarse, fine = (nets[0][1], nets[0][2]), (nets[1][1], nets[1][2])
    rgb, depth = render.render_vanilla(coarse, cap, fine, rays_per_batch=2048, samples_per_ray=32, importance_samples_per_ray=32,
                                       return_depth=True)
    # Two-pass frames inherit the inverse-CDF step function: the u = 1 importance sample sits on cdf[-1] ~ 1 and moves by
    # ~0.5 % of a bin per ulp of the running sum, which a semi-transparent ray (terminal 1e10 interval) turns into ~1e-3 of
    # colour.  The reference disagrees with ITSELF at this level between summation orders (DESIGN.md "conditioning"), so
    # the frame is pinned statistically here and exactly, stage by stage, in the other tests.
    err = np.abs(rgb - g['c1_rgb']).max(-1)
    assert (err > 1e-4).mean() < 0.01 and err.max() < 5e-3
    derr = np.abs(depth - g['c1_depth'])
    assert (derr > 1e-4).mean() < 0.02 and derr.max() < 5e-2
    # the single-pass frame has no such step: tight
    rgb = render.render_vanilla(coarse, cap, None, rays_per_batch=4096, samples_per_ray=32)
    assert np.abs(rgb - g['c1_coarse_only_rgb']).max() < 2e-6


def test_render_smpl_canonical_c3(golden, nets):
    from neuman_hip import synthetic
    g = golden['render']
    cap = _cap((48, 48), g['c3_c2w'], float(g['c3_fx']))
    human = (nets[2][1], nets[2][2])
    rgb, depth, acc = render.render_smpl_nerf(human, cap, synthetic.human_vertex_cloud(0), None, None, rays_per_batch=1024,
                                              samples_per_ray=32, render_can=True, geo_threshold=0.2, return_depth=True,
                                              return_mask=True, interval_comp=0.7)
    hit = g['c3_acc'] > 0
    assert 0.1 < hit.mean() < 0.9
    # rays grazing the vertex-sphere union can flip hit/miss on a 1-ulp difference; they must be rare
    flips = (acc > 0) != hit
    assert flips.mean() < 2e-3
    ok = ~flips
    assert np.abs(rgb - g['c3_rgb'])[ok].max() < 2e-3
    assert np.abs(acc - g['c3_acc'])[ok].max() < 2e-3
    assert np.abs(depth - g['c3_depth'])[ok].max() < 5e-3
