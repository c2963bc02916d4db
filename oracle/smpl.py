"""Oracle: SMPL linear blend skinning -> per-frame posed vertices and per-vertex canonical->scene transforms.

Restates, in numpy float32 (the reference computes this part in torch float32 on the CPU):
  * models/smpl.py:407-438  batch_rodrigues
  * models/smpl.py:454-505  batch_rigid_transform (kinematic chain, sequential over the joints)
  * models/smpl.py:266-360  lbs  (shape blend, joint regression, skinning; the pose blend shapes are computed and then
                                  NOT applied: `v_posed = v_shaped`, :332)
  * models/smpl.py:109-162, 164-216  SMPL.verts_transformations / SMPL.forward
and the two callers that turn it into what the renderers consume:
  * data_io/neuman_helper.py:288-330  NeuManReader.read_smpls (numpy: f32 4x4 inverse, then float64 from the alignment on)
  * models/human_nerf.py:92-122       HumanNeRF.vertex_forward (torch float32 throughout)

PINNED: tests/golden/smpl.npz holds outputs of the reference itself (tests/golden/make_golden_smpl.py runs read_smpls,
verts_transformations, batch_rodrigues and vertex_forward unmodified on a synthetic SMPL-layout model).  Matrix products
and reductions run through BLAS in both, in unspecified summation order, so the comparison is to float32 tolerance
(1e-5 relative to the values' magnitude), not bit for bit.  Test infrastructure only.
"""
import numpy as np

F32 = np.float32


class Model:
    """The arrays models/smpl.py:74-107 registers, as float32 (`to_tensor(to_np(x), dtype=float32)`)."""

    def __init__(self, data):
        self.v_template = np.asarray(data['v_template'], F32)                 # [V,3]
        self.shapedirs = np.asarray(data['shapedirs'], F32)                   # [V,3,NB]
        self.J_regressor = np.asarray(data['J_regressor'], F32)               # [J,V]
        parents = np.asarray(data['kintree_table'])[0].astype(F32).astype(np.int64)   # to_np(float32) then .long() (:100)
        parents[0] = -1
        self.parents = parents
        self.lbs_weights = np.asarray(data['weights'], F32)                   # [V,J]


def batch_rodrigues(rot_vecs):
    """[N,3] f32 -> [N,3,3] f32 (smpl.py:407-438)."""
    rot_vecs = np.asarray(rot_vecs, F32)
    angle = np.linalg.norm(rot_vecs + F32(1e-8), axis=1, keepdims=True).astype(F32)
    rot_dir = rot_vecs / angle
    cos, sin = np.cos(angle)[:, None, :].astype(F32), np.sin(angle)[:, None, :].astype(F32)
    rx, ry, rz = rot_dir[:, 0:1], rot_dir[:, 1:2], rot_dir[:, 2:3]
    zeros = np.zeros_like(rx)
    K = np.concatenate([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], 1).reshape(-1, 3, 3)
    ident = np.eye(3, dtype=F32)[None]
    return (ident + sin * K + (F32(1) - cos) * np.matmul(K, K)).astype(F32)


def batch_rigid_transform(rot_mats, joints, parents):
    """rot_mats [J,3,3], joints [J,3] f32 -> posed joints [J,3], relative transforms A [J,4,4] (smpl.py:454-505)."""
    J = joints.shape[0]
    rel = joints.copy()
    rel[1:] -= joints[parents[1:]]
    mats = np.zeros((J, 4, 4), F32)
    mats[:, :3, :3] = rot_mats
    mats[:, :3, 3] = rel
    mats[:, 3, 3] = 1
    chain = [mats[0]]
    for i in range(1, J):
        chain.append(np.matmul(chain[parents[i]], mats[i]).astype(F32))
    transforms = np.stack(chain)
    posed = transforms[:, :3, 3].copy()
    jh = np.concatenate([joints, np.zeros((J, 1), F32)], 1)[..., None]        # F.pad(joints, [0,0,0,1])
    tj = np.matmul(transforms, jh)[..., 0]                                    # [J,4]
    rel_transforms = transforms.copy()
    rel_transforms[:, :, 3] -= tj                                             # only the last column is non-zero after the pad
    return posed, rel_transforms.astype(F32)


def lbs(model, betas, pose, concat_joints=False):
    """betas [NB], pose [J*3] f32 -> (T [V(+J),4,4] f32, v_posed [V(+J),3] f32, verts [V,3] f32, J_transformed [J,3] f32)
    i.e. both return forms of smpl.py:266-360 at once."""
    betas, pose = np.asarray(betas, F32), np.asarray(pose, F32)
    v_shaped = (model.v_template + np.einsum('l,mkl->mk', betas, model.shapedirs)).astype(F32)
    J = np.einsum('ik,ji->jk', v_shaped, model.J_regressor).astype(F32)
    rot = batch_rodrigues(pose.reshape(-1, 3))
    J_transformed, A = batch_rigid_transform(rot, J, model.parents)
    T = np.matmul(model.lbs_weights, A.reshape(-1, 16)).reshape(-1, 4, 4).astype(F32)
    vh = np.concatenate([v_shaped, np.ones((v_shaped.shape[0], 1), F32)], 1)
    verts = np.matmul(T, vh[..., None])[:, :3, 0].astype(F32)
    if concat_joints:
        return np.concatenate([T, A], 0), np.concatenate([v_shaped, J], 0), verts, J_transformed
    return T, v_shaped, verts, J_transformed


def da_pose(n_joints=24):
    """The canonical "da" pose: legs apart (neuman_helper.py:294-299, human_nerf.py:24-29)."""
    da = np.zeros((n_joints, 3), F32)
    da[1] = (0, 0, 1.0)
    da[2] = (0, 0, -1.0)
    return da.reshape(-1)


def read_smpl_frame(model, pose, betas, alignment_4x3, scale=1.0):
    """One iteration of the loop at neuman_helper.py:288-329.
    -> world_verts [V,3] f32, world_joints [J,3] f32, static_verts [V,3] f32, static_joints [J,3] f32, Ts [V+J,4,4] f64"""
    V = model.v_template.shape[0]
    align = np.eye(4)
    align[:, :3] = alignment_4x3
    T_t2pose, _, _, _ = lbs(model, betas, pose, concat_joints=True)
    T_t2da, _, da_verts, da_joints = lbs(model, betas, da_pose(model.parents.shape[0]), concat_joints=True)
    T_da2pose = np.matmul(T_t2pose, np.linalg.inv(T_t2da))                    # float32 @ float32-inverse
    T_da2scene = align.T @ T_da2pose                                          # float64 from here
    s = np.eye(4)
    s[:3, :3] *= scale
    T_da2scene = s @ T_da2scene
    pts = np.concatenate([da_verts, da_joints], 0)
    ph = np.concatenate([pts, np.ones((pts.shape[0], 1), pts.dtype)], 1)
    world = np.einsum('BNi,Bi->BN', T_da2scene, ph)[:, :3].astype(F32)
    return world[:V], world[V:], da_verts, da_joints, T_da2scene


def vertex_forward(model, pose, betas, alignment_4x4, scale=1.0):
    """human_nerf.py:92-122, float32 throughout -> world_verts [V,3] f32, T_da2scene [V,4,4] f32."""
    T_t2pose, _, _, _ = lbs(model, betas, pose)
    T_t2da, _, da_verts, _ = lbs(model, betas, da_pose(model.parents.shape[0]))
    T_da2pose = np.matmul(T_t2pose, np.linalg.inv(T_t2da)).astype(F32)
    T = np.matmul(np.asarray(alignment_4x4, F32).T, T_da2pose).astype(F32)
    s = np.eye(4, dtype=F32)
    s[:3, :3] *= F32(scale)
    T = np.matmul(s, T).astype(F32)
    ph = np.concatenate([da_verts, np.ones((da_verts.shape[0], 1), F32)], 1)
    world = np.einsum('bni,bi->bn', T, ph)[:, :3].astype(F32)
    return world, T
