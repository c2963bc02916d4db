"""Scene ingestion for the hot path's callers: the on-disk formats the reference's render / train scripts read
before the first ray is shot, restated without the reference's dependency stack (imageio, open3d, tqdm ...).

    COLMAP ASCII model       cameras.txt / images.txt / points3D.txt     data_io/colmap_helper.py:22-149
    captures with poses      world->camera quaternion + translation      cameras/camera_pose.py:15-108, geometry/basics.py:10-120
    near / far per capture   95th percentile of the projected cloud      data_io/neuman_helper.py:199-226, geometry/pcd_projector.py:156-186
    scene normalisation      p95(far) -> 3.14                            data_io/neuman_helper.py:229-242
    train / val / test       *_split.txt                                 data_io/neuman_helper.py:149-196
    checkpoints              checkpoint.pth.tar, safe_load_weights       train.py:97-103, utils/utils.py:225-254, models/human_nerf.py:53-74
    options                  params.json                                 options/options.py:36-45
    SMPL parameter files     see neuman_hip/smpl.py (read_smpls)

Host-side Python like the reference's own readers (they run once per scene, never per ray).  The objects returned carry
exactly the attributes the renderers consume (`cap.shape / .size / .intrinsic_matrix / .cam_pose.camera_to_world /
.cam_pose.camera_center_in_world / .near / .far / .frame_id`), with the reference's own float32 / float64 conventions so that
the rays shot from them are the reference's rays.
"""
import json
import os
import re
from collections import namedtuple

import numpy as np

_EPS = np.finfo(float).eps * 4.0


# ------------------------------------------------------------------------------------------------
# poses (geometry/basics.py Rotation / Translation, cameras/camera_pose.py CameraPose)
# ------------------------------------------------------------------------------------------------
def quaternion_matrix(quat):
    """Homogeneous rotation matrix of a (w, x, y, z) quaternion, float64 (the reference uses transformations.quaternion_matrix)."""
    q = np.array(quat, dtype=np.float64)
    n = float(q @ q)
    if n < _EPS:
        return np.eye(4)
    w, x, y, z = q * np.sqrt(2.0 / n)
    return np.array([[1.0 - y * y - z * z, x * y - z * w, x * z + y * w, 0.0],
                     [x * y + z * w, 1.0 - x * x - z * z, y * z - x * w, 0.0],
                     [x * z - y * w, y * z + x * w, 1.0 - x * x - y * y, 0.0],
                     [0.0, 0.0, 0.0, 1.0]])


class CameraPose:
    """World->camera translation (float32 [3]) and unit quaternion (float32 (w, x, y, z)), as COLMAP stores them."""

    def __init__(self, translation, quaternion):
        t = np.asarray(translation)
        q = np.asarray(quaternion)
        assert t.shape == (3,) and t.dtype == np.float32 and q.shape == (4,)
        if not np.isclose(np.linalg.norm(q), 1.0):                                    # geometry/basics.py:54-56
            q = q / np.linalg.norm(q)
        self.translation_vector, self.quaternion = t, q
        self._rotation_override = None                                                 # set by the camera-centre setter's sibling paths

    @property
    def rotation_matrix(self):
        return quaternion_matrix(self.quaternion).astype(np.float32)                   # geometry/basics.py:33-34

    @property
    def translation_matrix(self):
        m = np.eye(4)
        m[:3, 3] = self.translation_vector
        return m.astype(np.float32)                                                    # geometry/basics.py:98-99

    @property
    def world_to_camera(self):
        m = np.matmul(self.translation_matrix, self.rotation_matrix)                   # camera_pose.py:71-74
        m /= m[3, 3]
        return m

    @property
    def extrinsic_matrix(self):
        return self.world_to_camera[0:3, 0:4]

    @property
    def camera_to_world(self):
        m = np.linalg.inv(self.world_to_camera)                                        # camera_pose.py:85-88
        m /= m[3, 3]
        return m

    @property
    def camera_center_in_world(self):
        return self.camera_to_world[:3, 3]

    @camera_center_in_world.setter
    def camera_center_in_world(self, value):
        """camera_pose.py:98-103: move the centre, keep the orientation (the translation of the inverted matrix, float32)."""
        c2w = self.camera_to_world
        c2w[:3, 3] = value
        w2c = np.linalg.inv(c2w)
        w2c /= w2c[3, 3]
        self.translation_vector = w2c[:3, 3].astype(np.float32)


class PinholeCamera:
    """cameras/pinhole_camera.py:13-46"""

    def __init__(self, width, height, fx, fy, cx, cy):
        self.width, self.height = int(width), int(height)
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy

    @property
    def shape(self):
        return (self.height, self.width)

    size = shape

    @property
    def intrinsic_matrix(self):
        return np.array([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]])


def resize_pinhole_camera(cam, tgt_size):
    """cameras/pinhole_camera.py:49-56"""
    h, w = tgt_size
    sh, sw = h / cam.shape[0], w / cam.shape[1]
    return PinholeCamera(w, h, cam.fx * sw, cam.fy * sh, cam.cx * sw, cam.cy * sh)


class Capture:
    """What a renderer needs of the reference's capture classes (cameras/captures.py:21-63): camera, pose, near / far, frame id."""

    def __init__(self, image_path, pinhole_cam, cam_pose, frame_id=None):
        self.image_path, self.pinhole_cam, self.cam_pose = image_path, pinhole_cam, cam_pose
        self.near, self.far = {}, {}
        if frame_id is not None:
            self.frame_id = frame_id

    @property
    def shape(self):
        return self.pinhole_cam.shape

    size = shape

    @property
    def intrinsic_matrix(self):
        return self.pinhole_cam.intrinsic_matrix

    @property
    def extrinsic_matrix(self):
        return self.cam_pose.extrinsic_matrix


# ------------------------------------------------------------------------------------------------
# COLMAP ASCII (data_io/colmap_helper.py)
# ------------------------------------------------------------------------------------------------
ImageMeta = namedtuple('ImageMeta', ['image_id', 'camera_pose', 'camera_id', 'image_path'])
_NUM = r"[-+]?\d*\.\d+|\d+"


class ColmapAsciiReader:
    @classmethod
    def read_scene(cls, scene_dir, images_dir, tgt_size=None, order='default', check_files=True):
        """-> (captures, point_cloud [N,6] float32 xyz rgb)  (colmap_helper.py:26-34)"""
        captures = cls.read_captures(os.path.join(scene_dir, 'images.txt'), os.path.join(scene_dir, 'cameras.txt'), images_dir, tgt_size, order,
                                     check_files)
        return captures, cls.read_point_cloud(os.path.join(scene_dir, 'points3D.txt'))

    @staticmethod
    def read_point_cloud(points_txt_path):
        """colmap_helper.py:36-57"""
        with open(points_txt_path, "r") as fid:
            assert fid.readline() == '# 3D point list with one line of data per point:\n'
            assert fid.readline() == '#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n'
            line = fid.readline()
            assert re.search(r'^# Number of points: \d+, mean track length: [-+]?\d*\.\d+|\d+\n$', line)
            num_points = int(re.findall(_NUM, line)[0])
            pcd = np.zeros((num_points, 6), dtype=np.float32)
            for i in range(num_points):
                elems = fid.readline().split()
                pcd[i] = list(map(float, elems[1:7]))
        return pcd

    @classmethod
    def read_cameras(cls, cameras_txt_path):
        """colmap_helper.py:89-117: SIMPLE_RADIAL / PINHOLE / OPENCV, distortion ignored (the scenes are undistorted)"""
        cameras = {}
        with open(cameras_txt_path, "r") as fid:
            assert fid.readline() == '# Camera list with one line of data per camera:\n'
            assert fid.readline() == '#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n'
            line = fid.readline()
            assert re.search(r'^# Number of cameras: \d+\n$', line)
            for _ in range(int(re.findall(_NUM, line)[0])):
                elems = fid.readline().split()
                camera_id, model, vals = int(elems[0]), elems[1], list(map(float, elems[2:]))
                if model == 'SIMPLE_RADIAL':
                    width, height, f, cx, cy, _radial = vals
                    cam = PinholeCamera(width, height, f, f, cx, cy)
                elif model == 'PINHOLE':
                    cam = PinholeCamera(*vals)
                elif model == 'OPENCV':
                    cam = PinholeCamera(*vals[:6])
                else:
                    raise ValueError(f'unsupported camera: {model}')
                assert camera_id not in cameras
                cameras[camera_id] = cam
        return cameras

    @classmethod
    def read_images_meta(cls, images_txt_path, images_dir, check_files=True):
        """colmap_helper.py:119-149: pose = CameraPose(Translation(t f32), Rotation(q f32)), world -> camera"""
        images_meta = {}
        with open(images_txt_path, "r") as fid:
            assert fid.readline() == '# Image list with two lines of data per image:\n'
            assert fid.readline() == '#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n'
            assert fid.readline() == '#   POINTS2D[] as (X, Y, POINT3D_ID)\n'
            line = fid.readline()
            assert re.search(r'^# Number of images: \d+, mean observations per image: [-+]?\d*\.\d+|\d+\n$', line)
            for _ in range(int(re.findall(_NUM, line)[0])):
                elems = fid.readline().split()
                assert len(elems) == 10
                fid.readline()                                                          # the POINTS2D line
                image_path = os.path.join(images_dir, elems[9])
                if check_files:
                    assert os.path.isfile(image_path), f'missing file: {image_path}'
                image_id = int(elems[0])
                qw, qx, qy, qz, tx, ty, tz = list(map(float, elems[1:8]))
                pose = CameraPose(np.array([tx, ty, tz], dtype=np.float32), np.array([qw, qx, qy, qz], dtype=np.float32))
                assert image_id not in images_meta, f'duplicated image, id: {image_id}, path: {image_path}'
                images_meta[image_id] = ImageMeta(image_id, pose, int(elems[8]), image_path)
        return images_meta

    @classmethod
    def read_captures(cls, images_txt_path, cameras_txt_path, images_dir, tgt_size=None, order='default', check_files=True):
        """colmap_helper.py:59-87: order='video' sorts by file name and stamps frame_id / total_frames (the time of `--ablate_nerft`)"""
        cameras = cls.read_cameras(cameras_txt_path)
        meta = cls.read_images_meta(images_txt_path, images_dir, check_files)
        if order == 'default':
            keys = list(meta.keys())
        elif order == 'video':
            keys = [k for _, k in sorted(zip([os.path.basename(v.image_path) for v in meta.values()], meta.keys()))]
        else:
            raise ValueError(f'unknown order: {order}')
        caps = []
        for i, key in enumerate(keys):
            cam = cameras[meta[key].camera_id]
            if tgt_size is not None:
                cam = resize_pinhole_camera(cam, tgt_size)
            cap = Capture(meta[key].image_path, cam, meta[key].camera_pose)
            if order == 'video':
                cap.frame_id = {'frame_id': i, 'total_frames': len(meta)}
            caps.append(cap)
        return caps


# ------------------------------------------------------------------------------------------------
# near / far and scene normalisation (data_io/neuman_helper.py:199-242)
# ------------------------------------------------------------------------------------------------
def projected_depths(points, cap):
    """z of the points that project inside the image with positive depth (pcd_projector.py:156-186 with crop, filter_neg;
    what project_point_cloud_at_capture(..., render_type='pcd')[:, 2] holds)."""
    pts = np.asarray(points)[:, :3]
    xyzw = np.concatenate([pts, np.ones_like(pts[:, 0:1])], axis=1)
    cam = np.matmul(np.matmul(cap.intrinsic_matrix, cap.extrinsic_matrix), xyzw.T).T
    cam = cam[cam[:, 2] > 0.0]
    img = cam / cam[:, 2:3]
    h, w = cap.size
    keep = (img[:, 0] >= 0) * (img[:, 0] < w - 1) * (img[:, 1] >= 0) * (img[:, 1] < h - 1)
    return cam[keep][:, 2]


def update_near_far(captures, key, points_per_capture, range_scale):
    """neuman_helper.py:200-226.  key 'bkg': points = the COLMAP cloud, near 0, far = p95 of the projected depths; key 'human':
    points = that frame's posed vertices, near / far = min / max; both then widened about their centre by range_scale."""
    for i, cap in enumerate(captures):
        z = projected_depths(points_per_capture(i), cap)
        if key == 'bkg':
            near, far = 0, np.percentile(z, 95)
        elif key == 'human':
            near, far = z.min(), z.max()
        else:
            raise ValueError(key)
        center, length = (near + far) / 2, (far - near) * range_scale
        cap.near[key] = max(0.0, float(center - length / 2))
        cap.far[key] = float(center + length / 2)


def normalize_scene(captures, point_cloud):
    """neuman_helper.py:229-242: scale the scene so that the 95th percentile of the background far bounds is 3.14; camera centres,
    near / far and the cloud are scaled in place.  -> scale"""
    fars = np.array([cap.far['bkg'] for cap in captures])
    scale = 3.14 / np.percentile(fars, 95)
    for cap in captures:
        cap.cam_pose.camera_center_in_world = cap.cam_pose.camera_center_in_world * scale
        cap.near['bkg'], cap.far['bkg'] = cap.near['bkg'] * scale, cap.far['bkg'] * scale
    point_cloud[:, :3] *= scale
    return scale


def read_scene(scene_dir, tgt_size=None, normalize=False, bkg_range_scale=1.1, check_files=True):
    """The camera / bounds part of NeuManReader.read_scene (neuman_helper.py:198-247; `sparse/` + `images/`, video order):
    -> (captures, point_cloud, scale).  The SMPL part is neuman_hip.smpl.read_smpls (human bounds: update_near_far(..., 'human', ...))."""
    caps, pcd = ColmapAsciiReader.read_scene(os.path.join(scene_dir, 'sparse'), os.path.join(scene_dir, 'images'), tgt_size, 'video', check_files)
    update_near_far(caps, 'bkg', lambda i: pcd, bkg_range_scale)
    scale = normalize_scene(caps, pcd) if normalize else 1
    assert len(caps) > 0
    return caps, pcd, scale


# ------------------------------------------------------------------------------------------------
# splits (data_io/neuman_helper.py:149-196)
# ------------------------------------------------------------------------------------------------
def split_indices(scene_length):
    """neuman_helper.py:153-165: every fifth frame held out, the first half of those for testing -> (train, val, test) index lists"""
    num_val = scene_length // 5
    length = int(1 / num_val * scene_length)
    offset = length // 2
    val_list = list(range(scene_length))[offset::length]
    train_list = list(set(range(scene_length)) - set(val_list))
    test_list = val_list[:len(val_list) // 2]
    val_list = val_list[len(val_list) // 2:]
    assert len(train_list) > 0 and len(test_list) > 0 and len(val_list) > 0
    return train_list, val_list, test_list


def create_split_files(scene_dir, captures):
    """neuman_helper.py:149-181: {train,val,test}_split.txt with one image file name per line"""
    paths = []
    for idx, split in zip(split_indices(len(captures)), ['train', 'val', 'test']):
        path = os.path.join(scene_dir, f'{split}_split.txt')
        keep = set(idx)
        with open(path, 'w') as f:
            for i, cap in enumerate(captures):
                if i in keep:
                    f.write("%s\n" % os.path.basename(cap.image_path))
        paths.append(path)
    return paths


def read_text(txt_file):
    """neuman_helper.py:184-196"""
    assert os.path.isfile(txt_file)
    with open(txt_file, "r") as fid:
        return [line.strip() for line in fid if line]


def captures_of_split(captures, split_file):
    names = set(read_text(split_file))
    return [cap for cap in captures if os.path.basename(cap.image_path) in names]


# ------------------------------------------------------------------------------------------------
# checkpoints and options
# ------------------------------------------------------------------------------------------------
def safe_load_weights(model, saved_weights):
    """utils/utils.py:225-254: exact keys, then without / with a DataParallel 'module.' prefix, then whatever matches by name and
    shape (returns the set of keys left untouched; raises if nothing matches instead of the reference's exit())."""
    for remap in (lambda k: k, lambda k: k.replace('module.', ''), lambda k: 'module.' + k):
        try:
            model.load_state_dict({remap(k): v for k, v in saved_weights.items()})
            return set()
        except RuntimeError:
            continue
    model_dict = model.state_dict()
    match = {k: v for k, v in saved_weights.items() if k in model_dict and model_dict[k].shape == v.shape}
    if not match:
        raise RuntimeError("pretrained weights loading failed: no tensor matches by name and shape")
    model_dict.update(match)
    model.load_state_dict(model_dict)
    return set(model.state_dict().keys()) - set(match.keys())


def load_background_checkpoint(path, coarse_net, fine_net):
    """models/human_nerf.py:53-61, render_vanilla callers: a NeRFTrainer checkpoint holds 'coarse_model_state_dict' / 'fine_model_state_dict'"""
    import torch
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    safe_load_weights(coarse_net, ckpt['coarse_model_state_dict'])
    safe_load_weights(fine_net, ckpt['fine_model_state_dict'])
    return ckpt


def load_hybrid_checkpoint(path, net):
    """train.py:97-103 / render_*.py: a HumanNeRFTrainer checkpoint holds 'hybrid_model_state_dict' for the whole HumanNeRF module"""
    import torch
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    safe_load_weights(net, ckpt['hybrid_model_state_dict'])
    return ckpt


def load_canonical_human(path, human_net):
    """models/human_nerf.py:63-74: only the `coarse_human_net.` tensors of a hybrid checkpoint"""
    import torch
    sd = torch.load(path, map_location='cpu', weights_only=False)['hybrid_model_state_dict']
    safe_load_weights(human_net, {k.split('coarse_human_net.', 1)[1]: v for k, v in sd.items() if 'coarse_human_net.' in k})


def save_opt(opt, out_dir=None):
    """options/options.py:36-45"""
    out_dir = out_dir or opt.out
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, 'params.json')
    with open(path, 'w') as fp:
        json.dump(vars(opt), fp, indent=0, sort_keys=True)
    return path


def read_params(path):
    """-> argparse-like namespace of a run's params.json"""
    import types
    with open(path) as fp:
        return types.SimpleNamespace(**json.load(fp))
