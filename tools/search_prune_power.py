import sys, numpy as np
ROOT = __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..')); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/ml-neuman_amd')
from oracle import warp as OW, ray_ops as O
from neuman_hip import synthetic
verts_c, faces = synthetic.capsule_mesh()
posed, T = synthetic.twist_transforms(verts_c)
posed = np.asarray(posed, np.float64); faces = np.asarray(faces)[:, :3]
cap = synthetic.SimpleCapture(512, 512, fx=1.6 * 512, c2w=synthetic.spherical_c2w(40., 0., 3.0))
o, d = O.shot_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, O.all_pixel_coords(cap.shape))
near, far = O.geometry_guided_near_far(o, d, posed.astype(np.float32), 0.2)
hit = np.nonzero(near < far)[0]
rng = np.random.default_rng(0)
rays = rng.choice(hit, 40, replace=False)
z = near[rays, None] + (far[rays] - near[rays])[:, None] * np.linspace(0, 1, 128)[None]
pts = (o[rays, None] + d[rays, None] * z[..., None]).reshape(-1, 3)[::8]          # 640 samples
a, b, c = posed[faces[:, 0]], posed[faces[:, 1]], posed[faces[:, 2]]
# exact distances to all triangles
N = pts.shape[0]
dex = np.empty((N, len(faces)))
for i in range(N):
    q = OW.closest_point_on_triangles(np.broadcast_to(pts[i], a.shape), a, b, c)
    q = q[0] if isinstance(q, tuple) else q
    dex[i] = np.linalg.norm(q - pts[i], axis=-1)
dmin = dex.min(1)
lo, hi = np.minimum(np.minimum(a, b), c), np.maximum(np.maximum(a, b), c)
dbox = np.sqrt((np.maximum(np.maximum(lo[None] - pts[:, None], pts[:, None] - hi[None]), 0) ** 2).sum(-1))
# disc: min enclosing circle
ab, ac, bc = b - a, c - a, c - b
n = np.cross(ab, ac); nn = (n * n).sum(-1)
lab, lac, lbc = (ab * ab).sum(-1), (ac * ac).sum(-1), (bc * bc).sum(-1)
cc = a + (lac[:, None] * np.cross(n, ab) + lab[:, None] * np.cross(ac, n)) / (2 * nn[:, None])
# obtuse -> longest edge midpoint
E = np.stack([lab, lac, lbc], 1); k = E.argmax(1)
P = np.where((k == 0)[:, None], a, np.where((k == 1)[:, None], a, b)); Q = np.where((k == 0)[:, None], b, c); R = np.where((k == 0)[:, None], c, np.where((k == 1)[:, None], b, a))
obt = ((P - R) * (Q - R)).sum(-1) <= 0
cc = np.where(obt[:, None], 0.5 * (P + Q), cc)
r = np.max(np.stack([np.linalg.norm(v - cc, axis=-1) for v in (a, b, c)], 1), 1)
nu = n / np.sqrt(nn)[:, None]
v = pts[:, None] - cc[None]
h = (v * nu[None]).sum(-1); vv = (v * v).sum(-1)
rho = np.sqrt(np.maximum(vv - h * h, 0))
ddisc = np.sqrt(np.maximum(rho - r[None], 0) ** 2 + h ** 2)
dsph = np.maximum(np.sqrt(vv) - r[None], 0)
thr = dmin * 1.0001 + 2e-5
print("samples", N, "mean dmin", dmin.mean())
print("triangles with box  <= best:", (dbox <= thr[:, None]).sum(1).mean())
print("triangles with disc <= best:", (ddisc <= thr[:, None]).sum(1).mean())
print("triangles with sphere <= best:", (dsph <= thr[:, None]).sum(1).mean())
print("triangles with exact <= best*1.0001:", (dex <= thr[:, None]).sum(1).mean())
print("r mean", r.mean(), "edge mean", np.sqrt(lab).mean())
assert (ddisc <= dex + 1e-9).all()
