"""a12 SMPL linear blend skinning on the device -- host-side mirror of the reference interface.

    SMPL                      models/smpl.py:54-216      (constructor reads the same SMPL_{GENDER}.pkl; verts_transformations, __call__)
    read_smpls                data_io/neuman_helper.py:258-331 (same files: smpl_output_{type}.pkl via joblib, alignments.npy)
    vertex_forward            models/human_nerf.py:92-122

All arithmetic runs in libneuman_hip (csrc/smpl.hip, `nm_smpl_frames`), batched over frames; there is no CPU path.
"""
import ctypes
import os
import pickle

import numpy as np
import torch

from . import _lib


def da_pose(n_joints=24):
    """The canonical "da" pose, legs apart (neuman_helper.py:294-299, human_nerf.py:24-29)."""
    da = np.zeros((n_joints, 3), np.float32)
    da[1] = (0, 0, 1.0)
    da[2] = (0, 0, -1.0)
    return da.reshape(-1)


def _dense_f32(x):
    if 'scipy.sparse' in str(type(x)):
        x = x.todense()
    return np.ascontiguousarray(np.asarray(x), dtype=np.float32)


class SMPL:
    """`SMPL(model_path, gender, device)` as the reference; `model_path` may also be the unpickled dict."""

    def __init__(self, model_path, gender='neutral', device='cuda'):
        _lib.require_gpu()
        if isinstance(model_path, dict):
            data = model_path
        else:
            path = os.path.join(model_path, f'SMPL_{gender.upper()}.pkl') if os.path.isdir(model_path) else model_path
            if not os.path.exists(path):
                raise FileNotFoundError(f'Path {path} does not exist!')
            with open(path, 'rb') as f:
                data = pickle.load(f, encoding='latin1')
        self.device = torch.device(device)
        self.faces = np.asarray(data['f'])
        vt, sd = _dense_f32(data['v_template']), _dense_f32(data['shapedirs'])
        jr, w = _dense_f32(data['J_regressor']), _dense_f32(data['weights'])
        parents = np.asarray(data['kintree_table'])[0].astype(np.float32).astype(np.int64)      # to_np(float32).long(), smpl.py:100
        parents[0] = -1
        self.parents = parents
        self.V, self.J, self.NB = vt.shape[0], jr.shape[0], sd.shape[-1]
        self.da_smpl = da_pose(self.J)
        self.handle = ctypes.c_void_p()
        p32 = np.ascontiguousarray(parents.astype(np.int32))
        _lib.check(_lib.lib().nm_smpl_create(vt.ctypes.data_as(ctypes.c_void_p), sd.ctypes.data_as(ctypes.c_void_p),
                                             jr.ctypes.data_as(ctypes.c_void_p), p32.ctypes.data_as(ctypes.c_void_p),
                                             w.ctypes.data_as(ctypes.c_void_p), self.da_smpl.ctypes.data_as(ctypes.c_void_p),
                                             self.V, self.J, self.NB, ctypes.byref(self.handle)), "nm_smpl_create")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().nm_smpl_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def frames(self, poses, betas, alignments, scale=1.0, precise=True):
        """poses [B,J*3], betas [B,NB], alignments [B,4,4] (the matrix whose transpose is applied) ->
        T [B,V+J,4,4] f64, world [B,V+J,3] f32, static [B,V+J,3] f32, CUDA tensors (rows V.. are the joints)."""
        dev = self.device
        po = torch.as_tensor(np.asarray(poses, np.float32) if not isinstance(poses, torch.Tensor) else poses).to(dev, torch.float32).reshape(-1, self.J * 3).contiguous()
        be = torch.as_tensor(np.asarray(betas, np.float32) if not isinstance(betas, torch.Tensor) else betas).to(dev, torch.float32).reshape(-1, self.NB).contiguous()
        al = torch.as_tensor(np.asarray(alignments, np.float64) if not isinstance(alignments, torch.Tensor) else alignments).to(dev, torch.float64).reshape(-1, 16).contiguous()
        B = po.shape[0]
        if be.shape[0] != B or al.shape[0] != B:
            raise _lib.NeumanHipError(f"frames: {B} poses, {be.shape[0]} betas, {al.shape[0]} alignments")
        rows = self.V + self.J
        T = torch.empty((B, rows, 4, 4), device=dev, dtype=torch.float64)
        world = torch.empty((B, rows, 3), device=dev, dtype=torch.float32)
        static = torch.empty((B, rows, 3), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().nm_smpl_frames(self.handle, _lib.dev_ptr(po), _lib.dev_ptr(be), _lib.dev_ptr(al, torch.float64), B, float(scale),
                                             1 if precise else 0, _lib.dev_ptr(T, torch.float64), _lib.dev_ptr(world), _lib.dev_ptr(static),
                                             _lib.stream_ptr()), "nm_smpl_frames")
        return T, world, static


def read_smpls(scene_dir, caps, scale=1, smpl_type='romp', body_model=None, model_dir=None, device='cuda'):
    """NeuManReader.read_smpls (neuman_helper.py:258-331): same inputs on disk, same four returns -- smpls (list of per-frame
    dicts with joints_3d / static_joints_3d added), world_verts, static_verts (lists of [6890,3] f32) and Ts (list of
    [6914,4,4] f64) -- as numpy, all frames skinned in one batch on the device."""
    import joblib
    if body_model is None:
        body_model = SMPL(model_dir if model_dir is not None else
                          os.path.join(os.path.abspath(os.path.join(os.path.dirname(__file__), '..')), 'data/smplx/smpl'), 'neutral', device)
    smpl_path = os.path.join(scene_dir, f'smpl_output_{smpl_type}.pkl')
    assert os.path.isfile(smpl_path), f'{smpl_path} is missing'
    raw_smpl = joblib.load(smpl_path)
    assert len(raw_smpl) == 1
    raw_smpl = raw_smpl[list(raw_smpl.keys())[0]]
    raw_alignments = np.load(os.path.join(scene_dir, 'alignments.npy'), allow_pickle=True).item()
    smpls, aligns = [], []
    for cap in caps:
        frame_id = int(os.path.basename(cap.image_path)[:-4])
        out = {}
        for k, v in raw_smpl.items():
            try:
                out[k] = v[frame_id]
            except Exception:
                out[k] = None
        smpls.append(out)
        a = np.eye(4)
        a[:, :3] = raw_alignments[os.path.basename(cap.image_path)]
        aligns.append(a)
    if not smpls:
        return [], [], [], []
    T, world, static = body_model.frames(np.stack([s['pose'] for s in smpls]), np.stack([s['betas'] for s in smpls]), np.stack(aligns), scale, True)
    T, world, static = T.cpu().numpy(), world.cpu().numpy(), static.cpu().numpy()
    V = body_model.V
    for i, s in enumerate(smpls):
        s['joints_3d'] = world[i, V:]
        s['static_joints_3d'] = static[i, V:]
    return smpls, [world[i, :V] for i in range(len(smpls))], [static[i, :V] for i in range(len(smpls))], [T[i] for i in range(len(smpls))]


def vertex_forward(body_model, pose, beta, alignment, scale):
    """HumanNeRF.vertex_forward (human_nerf.py:92-122): pose [1,72], beta [1,10], alignment [4,4] -> world_verts [1,V,3] f32,
    T_da2scene [1,V,4,4] f32 (CUDA tensors)."""
    T, world, _ = body_model.frames(pose, beta, torch.as_tensor(alignment).reshape(1, 4, 4), scale, False)
    V = body_model.V
    return world[:, :V], T[:, :V].to(torch.float32)


# ------------------------------------------------------------------------------------------------
# differentiable skinning for training: gradients to poses / betas / alignments
# ------------------------------------------------------------------------------------------------
class _VertexForwardFn(torch.autograd.Function):
    """HumanNeRF.vertex_forward of one frame on the hand-written kernels (csrc/smpl.hip: nm_smpl_vertex_forward / _backward): two launches
    forward, six backward, instead of the ~800 launches torch's autograd makes of the same chain."""

    @staticmethod
    def forward(ctx, owner, pose, beta, alignment, scale, da_pose):
        h = owner._hip_handle()
        dev = pose.device
        V = owner.v_template.shape[0]
        po, be = pose.detach().reshape(-1).contiguous().float(), beta.detach().reshape(-1).contiguous().float()
        al = alignment.detach().to(torch.float64).contiguous()
        da = da_pose.detach().reshape(-1).contiguous().float()
        ws = torch.empty(int(_lib.lib().nm_smpl_vertex_workspace_floats(h)), device=dev, dtype=torch.float32)
        world = torch.empty((V, 3), device=dev, dtype=torch.float32)
        T = torch.empty((V, 4, 4), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().nm_smpl_vertex_forward(h, _lib.dev_ptr(po), _lib.dev_ptr(be), _lib.dev_ptr(al, torch.float64), float(scale), _lib.dev_ptr(da),
                                                     _lib.dev_ptr(ws), _lib.dev_ptr(world), _lib.dev_ptr(T), _lib.stream_ptr()), "nm_smpl_vertex_forward")
        ctx.owner, ctx.scale, ctx.shapes = owner, float(scale), (pose.shape, beta.shape)
        ctx.save_for_backward(po, be, al, da)
        return world, T

    @staticmethod
    def backward(ctx, g_world, g_T):
        po, be, al, da = ctx.saved_tensors
        owner = ctx.owner
        h = owner._hip_handle()
        dev = po.device
        ws = torch.empty(int(_lib.lib().nm_smpl_vertex_workspace_floats(h)), device=dev, dtype=torch.float32)
        g_pose, g_beta = torch.empty_like(po), torch.empty_like(be)
        g_al = torch.empty(16, device=dev, dtype=torch.float32)
        gw = g_world.contiguous().float() if g_world is not None else None
        gt = g_T.contiguous().float() if g_T is not None else None
        _lib.check(_lib.lib().nm_smpl_vertex_backward(h, _lib.dev_ptr(po), _lib.dev_ptr(be), _lib.dev_ptr(al, torch.float64), ctx.scale, _lib.dev_ptr(da),
                                                      _lib.dev_ptr(gw), _lib.dev_ptr(gt), _lib.dev_ptr(ws), _lib.dev_ptr(g_pose), _lib.dev_ptr(g_beta),
                                                      _lib.dev_ptr(g_al), _lib.stream_ptr()), "nm_smpl_vertex_backward")
        return None, g_pose.reshape(ctx.shapes[0]), g_beta.reshape(ctx.shapes[1]), g_al.reshape(4, 4), None, None


class SMPLDiff(torch.nn.Module):
    """HumanNeRF.vertex_forward (models/human_nerf.py:92-122) with autograd, for the trainer's pose refinement: the skinning
    chain of models/smpl.py:266-360 (shape blend, joint regression, Rodrigues, 24-joint kinematic chain, blend of the joint
    transforms; pose blend shapes are computed and ignored by the reference, :320-334) written as batched tensor algebra so that
    torch differentiates it.  A few small matrix products per training iteration on whatever device the parameters live on
    (the forward-only, all-frames-at-once form used by the renderers is the HIP kernel behind `SMPL.frames`)."""

    def __init__(self, model, device='cuda'):
        super().__init__()
        if not isinstance(model, dict):
            with open(model, 'rb') as f:
                model = pickle.load(f, encoding='latin1')
        dev = torch.device(device)
        t = lambda x: torch.from_numpy(_dense_f32(x)).to(dev)                      # noqa: E731
        # buffers named, shaped and ordered as models/smpl.py:66-98 registers them, so that a HumanNeRF state_dict ('body_model.*')
        # matches the reference's key for key
        self.register_buffer('faces_tensor', torch.from_numpy(np.asarray(model['f']).astype(np.int64)).to(dev))
        self.register_buffer('v_template', t(model['v_template']))                # [V,3]
        self.register_buffer('shapedirs', t(model['shapedirs']))                  # [V,3,NB]
        self.register_buffer('J_regressor', t(model['J_regressor']))              # [J,V]
        posedirs = _dense_f32(model['posedirs'])                                  # computed and ignored by the reference's lbs (:320-334)
        self.register_buffer('posedirs', torch.from_numpy(np.ascontiguousarray(posedirs.reshape(-1, posedirs.shape[-1]).T)).to(dev))
        parents = np.asarray(model['kintree_table'])[0].astype(np.float32).astype(np.int64)
        parents[0] = -1
        self.register_buffer('parents', torch.from_numpy(parents).to(dev))
        self.register_buffer('lbs_weights', t(model['weights']))                  # [V,J]
        self.parents_list = [int(p) for p in parents]
        self.register_buffer('da_smpl', torch.from_numpy(da_pose(len(self.parents_list))).to(dev)[None], persistent=False)

    @staticmethod
    def rodrigues(rot_vecs):
        """axis-angle [N,3] -> rotation matrices [N,3,3]  (models/smpl.py:407-438: angle = |v + 1e-8|, R = I + sin K + (1 - cos) K^2)"""
        angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
        axis = rot_vecs / angle
        x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
        zero = torch.zeros_like(x)
        K = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], dim=1).reshape(-1, 3, 3)
        s, c = torch.sin(angle)[:, :, None], torch.cos(angle)[:, :, None]
        eye = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
        return eye + s * K + (1 - c) * torch.bmm(K, K)

    def transformations(self, pose, beta):
        """pose [1,J*3], beta [1,NB] -> (vertex transforms T [V,4,4] from the shaped template to the posed body, v_shaped [V,3])"""
        J = len(self.parents_list)
        v_shaped = self.v_template + torch.einsum('vkl,l->vk', self.shapedirs, beta[0])            # smpl.py:312
        joints = self.J_regressor @ v_shaped                                                       # :315
        R = self.rodrigues(pose.reshape(J, 3))                                                      # :319-320
        rel = joints.clone()
        rel[1:] = joints[1:] - joints[[p for p in self.parents_list[1:]]]                               # :474-476
        local = torch.cat([torch.cat([R, rel[:, :, None]], 2),
                           torch.tensor([0., 0., 0., 1.], dtype=R.dtype, device=R.device).expand(J, 1, 4)], 1)    # [J,4,4]
        chain = [local[0]]
        for j in range(1, J):                                                                       # :487-493
            chain.append(chain[self.parents_list[j]] @ local[j])
        G = torch.stack(chain)
        # remove the rest pose: A_j = G_j - [0 | G_j[:3,:3] J_j]                                    # :499-503
        shift = torch.einsum('jab,jb->ja', G[:, :3, :3], joints)
        A = G.clone()
        A[:, :3, 3] = G[:, :3, 3] - shift
        T = torch.einsum('vj,jab->vab', self.lbs_weights, A)                                        # :338-341
        return T, v_shaped

    def _model_key(self):
        """identity of the model data behind the device copy: a load_state_dict / .to() / in-place edit of a buffer changes it"""
        return tuple((t.data_ptr(), t._version, str(t.device)) for t in (self.v_template, self.shapedirs, self.J_regressor, self.lbs_weights, self.da_smpl))

    def _hip_handle(self):
        """the device copy of the model behind the hand-written kernels (created on first use, rebuilt when the buffers it was made from change)"""
        if getattr(self, '_handle', None) is not None and getattr(self, '_handle_key', None) != self._model_key():
            _lib.lib().nm_smpl_destroy(self._handle)
            self._handle = None
        if getattr(self, '_handle', None) is None:
            import ctypes
            self._handle_key = self._model_key()
            f = lambda t: np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))      # noqa: E731
            vt, sd, jr, w = f(self.v_template), f(self.shapedirs), f(self.J_regressor), f(self.lbs_weights)
            p32 = np.ascontiguousarray(np.asarray(self.parents_list, np.int32))
            da = f(self.da_smpl).reshape(-1)
            self._handle = ctypes.c_void_p()
            _lib.check(_lib.lib().nm_smpl_create(vt.ctypes.data_as(ctypes.c_void_p), sd.ctypes.data_as(ctypes.c_void_p), jr.ctypes.data_as(ctypes.c_void_p),
                                                 p32.ctypes.data_as(ctypes.c_void_p), w.ctypes.data_as(ctypes.c_void_p), da.ctypes.data_as(ctypes.c_void_p),
                                                 vt.shape[0], jr.shape[0], sd.shape[-1], ctypes.byref(self._handle)), "nm_smpl_create")
        return self._handle

    def __del__(self):
        try:
            if getattr(self, '_handle', None):
                _lib.lib().nm_smpl_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def __getstate__(self):                                          # (copies / pickles rebuild the device copy on first use)
        st = self.__dict__.copy()
        st['_handle'] = None
        return st

    def vertex_forward(self, pose, beta, alignment, scale, da_pose=None):
        """pose [1,J*3], beta [1,NB], alignment [4,4] (human_nerf.py's self.alignments[idx]: its TRANSPOSE is applied), scale ->
        world_verts [1,V,3], T_da2scene [1,V,4,4]; differentiable in pose, beta and alignment.  `da_pose` [1,J*3]: the canonical
        pose to use instead of the built-in one (HumanNeRF keeps it as its `da_smpl` parameter).  On the GPU: csrc/smpl.hip's
        kernels, forward and backward (`vertex_forward_torch` is the same chain as torch tensor algebra, kept as the checker)."""
        if pose.is_cuda and pose.dtype == torch.float32 and isinstance(scale, (int, float)):
            da = self.da_smpl if da_pose is None else da_pose
            world, T = _VertexForwardFn.apply(self, pose, beta, alignment, float(scale), da)
            return world[None], T[None]
        return self.vertex_forward_torch(pose, beta, alignment, scale, da_pose)

    def vertex_forward_torch(self, pose, beta, alignment, scale, da_pose=None):
        """vertex_forward as batched tensor algebra under torch's autograd (any device, any dtype)"""
        T_pose, v_shaped = self.transformations(pose, beta)
        T_da, _ = self.transformations((self.da_smpl if da_pose is None else da_pose).to(pose.dtype), beta)
        T_da2pose = T_pose @ torch.inverse(T_da)                                                    # human_nerf.py:109
        T = alignment.T @ T_da2pose                                                                 # :110
        s = torch.eye(4, dtype=T.dtype, device=T.device)
        s[:3, :3] *= scale                                                                          # :111-113
        T = s @ T
        hom = torch.cat([v_shaped, torch.ones_like(v_shaped[:, :1])], 1)
        da_verts = torch.einsum('vab,vb->va', T_da, hom)                                            # the body in the da pose, :114-120
        world = torch.einsum('vab,vb->va', T, da_verts)[:, :3]                                      # :121
        return world[None], T[None]
