"""Synthetic workloads for the hot path: no dataset, checkpoint or SMPL asset exists offline.

Everything here is deterministic in its seed and is shared by tests/, bench.py and __graft_entry__.smoke()
so that the HIP path and the CPU oracle consume identical rays, weights and meshes.
"""
import types

import numpy as np
import torch

from . import vanilla


def default_opt(**over):
    """Render-time options with the reference defaults (options/options.py:47-81)."""
    o = dict(use_cuda=False, nerf_depth=8, nerf_width=256, use_viewdirs=True, specular_can=True, raw_pos_dim=3,
             pos_min_freq=0, pos_max_freq=9, pos_N_freqs=10, raw_dir_dim=3, dir_max_freq=3, dir_N_freqs=4,
             log_sampling=True, include_input=True, can_posenc='rotate', rays_per_batch=2048, samples_per_ray=128,
             white_bkg=True)
    o.update(over)
    return types.SimpleNamespace(**o)


class _Pose:
    def __init__(self, c2w):
        # the matrix keeps a float32 dtype when it is given one: the reference's CameraPose builds its matrix from float32 quaternions, and shot_rays'
        # subtraction, norm and division then run in float32 (numpy's result type; utils/ray_utils.py:23-29) -- a capture rebuilt from a recorded float32
        # matrix must shoot the reference's rays bit for bit (csrc/frame.hip mode 2), not the float64 chain's, which differ in the last place
        c2w = np.asarray(c2w)
        self.camera_to_world = c2w if c2w.dtype == np.float32 else c2w.astype(np.float64)

    @property
    def camera_center_in_world(self):
        return self.camera_to_world[:3, 3]


class SimpleCapture:
    """The slice of the reference's capture duck-type the hot path consumes (cameras/captures.py:21-63)."""

    def __init__(self, width, height, fx=None, fy=None, cx=None, cy=None, c2w=None, near=0.0, far=3.14):
        self.width, self.height = int(width), int(height)
        self.fx = 1.25 * width if fx is None else fx
        self.fy = self.fx if fy is None else fy
        self.cx = width / 2 if cx is None else cx
        self.cy = height / 2 if cy is None else cy
        self.cam_pose = _Pose(np.eye(4) if c2w is None else c2w)
        self.near = {'bkg': near}
        self.far = {'bkg': far}

    @property
    def shape(self):
        return (self.height, self.width)

    size = shape

    @property
    def intrinsic_matrix(self):
        return np.array([[self.fx, 0., self.cx], [0., self.fy, self.cy], [0., 0., 1.]])


def spherical_c2w(theta_deg, phi_deg, radius):
    """Camera on a sphere looking at the origin (same convention as render_utils.pose_spherical, :42-56)."""
    th, ph = np.deg2rad(theta_deg), np.deg2rad(phi_deg)
    t = np.eye(4)
    t[2, 3] = radius
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1.]])
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1.]])
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.]]) @ rt @ rp @ t
    return c2w @ np.diag([1., -1., -1., 1.])


def make_joiner(seed, mapping='posenc', dense=True, pos_min_freq=0, preset=None):
    """One Joiner with torch.manual_seed(seed) default nn.Linear init; `dense` applies the synthetic-dense
    preset of SURVEY 8d (alpha_linear.weight *= 40, alpha_linear.bias = 0.5, rgb_linear.weight *= 8) that keeps
    sigma away from the 1e10-interval step at 0 and gives non-trivial transmittance.

    preset='opaque': the same random field read as SURFACES -- alpha_linear.weight *= 40000, bias = 400: sigma is zero in two
    thirds of the volume and in the hundreds elsewhere, so a ray crosses empty space and is absorbed within a couple of samples
    of the first blob it meets (what trained scenes look like to a renderer: the workload early ray termination is for,
    which the dense preset -- every sample of every ray still visible -- cannot show).  Use the SAME seed for the coarse and the
    fine net, as a trained pair agrees about where the surfaces are.

    preset='fog': the same random field as a PARTICIPATING MEDIUM -- alpha_linear.weight *= 10, bias = 1.5: sigma is positive everywhere (about 1.5
    +- 0.5 over the volume, optical depth ~ 4.7 along a ray), so every coarse bin carries compositing weight far above sample_pdf's 1e-5 floor and
    the inverse CDF that places the importance samples is WELL CONDITIONED on every ray: the reference's own float32 frame sits 3e-7 from its
    float64 evaluation (tests/golden/make_golden_f64.py), where the dense and opaque presets -- with their empty bins at the floor -- leave 0.7 % / 0.005 %
    of the rays beyond 1e-4.  The workload on which the 1e-4 contract can be held on EVERY ray of a frame; same seed for both nets."""
    opt = default_opt(posenc=mapping, pos_min_freq=pos_min_freq)
    torch.manual_seed(seed)
    net, _ = vanilla.build_nerf(opt)
    if preset == 'fog':
        with torch.no_grad():
            net.nerf.alpha_linear.weight *= 10.
            net.nerf.alpha_linear.bias.fill_(1.5)
            net.nerf.rgb_linear.weight *= 8.
    elif preset == 'opaque':
        with torch.no_grad():
            net.nerf.alpha_linear.weight *= 40000.
            net.nerf.alpha_linear.bias.fill_(400.)
            net.nerf.rgb_linear.weight *= 8.
    elif preset is not None:
        raise ValueError(preset)
    elif dense:
        with torch.no_grad():
            net.nerf.alpha_linear.weight *= 40.
            net.nerf.alpha_linear.bias.fill_(0.5)
            net.nerf.rgb_linear.weight *= 8.
    return net.eval()                    # rendering workloads; train() + grad enabled selects the differentiable forward


def densify(net):
    """The synthetic-dense preset on whichever head a Joiner has (ours or the reference's): sigma away from the 1e10-interval
    step at 0 and non-trivial transmittance."""
    with torch.no_grad():
        if net.nerf.use_viewdirs:
            net.nerf.alpha_linear.weight *= 40.
            net.nerf.alpha_linear.bias.fill_(0.5)
            net.nerf.rgb_linear.weight *= 8.
        else:                                            # plain head: output_linear rows (r, g, b, sigma)
            net.nerf.output_linear.weight[:3] *= 8.
            net.nerf.output_linear.weight[3] *= 40.
            net.nerf.output_linear.bias[3] = 0.5
    return net


def make_variant_joiner(seed, **opt_over):
    """A dense Joiner of one of the variants beside the default net: use_viewdirs=False (the plain head of `--specular_can no`),
    raw_pos_dim=4 (the time-conditioned net of `--ablate_nerft`), posenc='rotate' ...  Same seed and construction order as the
    reference's build_nerf, so the weights equal the ones tests/golden/make_golden_heads.py gave the reference."""
    opt = default_opt(**opt_over)
    torch.manual_seed(seed)
    net, _ = vanilla.build_nerf(opt)
    return densify(net).eval()


def state_numpy(joiner):
    """Joiner weights as {reference state_dict name: f32 numpy} (what oracle/nerf_mlp.py consumes)."""
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in joiner.state_dict().items()}


def human_vertex_cloud(seed=0, n=6890):
    """Stand-in for SMPL vertices in canonical renders."""
    return (np.random.default_rng(seed).normal(size=(n, 3)) * np.array([0.25, 0.6, 0.15])).astype(np.float32)


def capsule_mesh(n_rings=84, n_seg=82, radius=(0.25, 0.6, 0.15)):
    """Closed genus-0 triangle mesh with SMPL's counts for the defaults: V = 84*82 + 2 = 6890, F = 2*82*84 = 13776."""
    th = np.linspace(0, np.pi, n_rings + 2)[1:-1]
    ph = np.linspace(0, 2 * np.pi, n_seg, endpoint=False)
    ring = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.cos(th), np.ones_like(ph)),
                     np.outer(np.sin(th), np.sin(ph))], -1).reshape(-1, 3)
    verts = np.concatenate([[[0, 1, 0]], ring, [[0, -1, 0]]]) * np.asarray(radius)
    idx = lambda r, s: 1 + r * n_seg + (s % n_seg)
    faces = []
    last = 1 + n_rings * n_seg
    for s in range(n_seg):
        faces.append([0, idx(0, s + 1), idx(0, s)])
        faces.append([last, idx(n_rings - 1, s), idx(n_rings - 1, s + 1)])
    for r in range(n_rings - 1):
        for s in range(n_seg):
            a, b, c, d = idx(r, s), idx(r, s + 1), idx(r + 1, s), idx(r + 1, s + 1)
            faces.append([a, b, c])
            faces.append([b, d, c])
    return verts.astype(np.float32), np.asarray(faces, dtype=np.int64)


def twist_transforms(can_verts, twist=0.8, shift=(0.05, -0.02, 0.03)):
    """Per-vertex rigid canonical->observation transforms (f64 [V,4,4]): rotation about y by an angle
    proportional to height, plus a translation -- a smooth stand-in for SMPL LBS.  Returns (posed_verts f32, T)."""
    v = can_verts.astype(np.float64)
    ang = twist * v[:, 1]
    T = np.tile(np.eye(4), (v.shape[0], 1, 1))
    T[:, 0, 0], T[:, 0, 2] = np.cos(ang), np.sin(ang)
    T[:, 2, 0], T[:, 2, 2] = -np.sin(ang), np.cos(ang)
    T[:, :3, 3] = np.asarray(shift)
    posed = np.einsum('vij,vj->vi', T[:, :3, :3], v) + T[:, :3, 3]
    return posed.astype(np.float32), T


SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)   # SMPL's kinematic tree


def smpl_like_model(seed=0, n_betas=10):
    """A body model with SMPL's shapes and file layout (models/smpl.py:74-107 reads these keys) but synthetic content: the
    SMPL assets are licensed and absent.  V = 6890 capsule vertices, 24 joints on SMPL's kinematic tree, dense joint
    regressor and skinning weights (rows sum to one), small random shape directions; pose blend shapes are zero (the
    reference computes and then ignores them, smpl.py:320-330)."""
    rng = np.random.default_rng(seed)
    verts, faces = capsule_mesh()
    V, J = verts.shape[0], len(SMPL_PARENTS)
    t = np.linspace(0.0, 1.0, J)
    anchors = np.stack([0.18 * np.sin(2 * np.pi * t * 4.6), -0.55 + 1.1 * t, 0.1 * np.cos(2 * np.pi * t * 3.2)], 1)
    d2 = ((verts[:, None, :].astype(np.float64) - anchors[None]) ** 2).sum(-1)                     # [V,J]
    w = np.exp(-d2 / 0.02)
    w /= w.sum(1, keepdims=True)
    jr = np.exp(-d2.T / 0.005)                                                                     # [J,V]
    jr /= jr.sum(1, keepdims=True)
    kintree = np.stack([np.asarray(SMPL_PARENTS, np.int64), np.arange(J)], 0).astype(np.int64)
    kintree[0, 0] = 2 ** 32 - 1                                                                    # as in the SMPL files
    return {
        'f': faces[:, :3].astype(np.uint32),
        'v_template': verts.astype(np.float64),
        'shapedirs': (rng.normal(size=(V, 3, n_betas)) * 0.004).astype(np.float64),
        'J_regressor': jr.astype(np.float64),
        'posedirs': np.zeros((V, 3, (J - 1) * 9), np.float64),
        'kintree_table': kintree,
        'weights': w.astype(np.float64),
    }


def smpl_like_frames(n_frames=3, seed=0, n_betas=10):
    """Per-frame SMPL parameters and scene alignments as `smpl_output_*.pkl` / `alignments.npy` hold them
    (neuman_helper.py:281-293): pose [n,72] f32, betas [n,10] f32, alignments {name: [4,3] f64}."""
    rng = np.random.default_rng(seed + 99)
    pose = (rng.normal(size=(n_frames, 72)) * 0.35).astype(np.float32)
    pose[0, 6:9] = 0.0                                             # a joint at rest: Rodrigues at (near) zero angle
    betas = (rng.normal(size=(n_frames, n_betas)) * 0.8).astype(np.float32)
    align = {}
    for i in range(n_frames):
        a = rng.normal(size=3)
        a /= np.linalg.norm(a)
        ang = rng.uniform(0.2, 1.2)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        M = np.eye(4)
        M[:3, :3] = Rm * rng.uniform(0.8, 1.3)
        M[:3, 3] = rng.normal(size=3) * 0.5
        align[f"{i:05d}.png"] = np.ascontiguousarray(M.T[:, :3])  # stored transposed, [4,3] (neuman_helper.py:292, 316)
    return pose, betas, align
