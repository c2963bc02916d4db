"""-m gpu: device ray generation (a1) and frame egress through the C ABI vs the CPU oracle and the reference goldens."""
import numpy as np
import pytest
import torch

from oracle import frame as OF, ray_ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    import types
    from neuman_hip import ray_utils, render_utils, synthetic
    return types.SimpleNamespace(ray=ray_utils, render=render_utils, syn=synthetic)


def ulps32(a, b):
    """distance in f32 units in the last place between two f32 arrays"""
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


CAMS = [dict(width=64, height=48), dict(width=800, height=800), dict(width=1920, height=1080, fx=1777.7, fy=1801.3, cx=955.2, cy=544.9),
        dict(width=33, height=17, fx=40.0, cx=10.0, cy=3.5)]


@pytest.mark.parametrize("cam", CAMS)
@pytest.mark.parametrize("pose", ["eye", "spherical"])
def test_shot_all_rays_dev_vs_reference_chain(H, cam, pose):
    """mode 0: the f64 chain of ray_utils.py:32-38 per ray.  The oracle is the numpy restatement (pinned against the
    reference's own output in tests/test_oracle_golden.py); BLAS may contract its 3- and 4-term dot products differently,
    so a direction component may land on the other side of an f32 rounding boundary: <= 1 ulp, on <= 1e-4 of the values."""
    c2w = None if pose == "eye" else H.syn.spherical_c2w(37.0, -21.0, 2.7)
    cap = H.syn.SimpleCapture(c2w=c2w, **cam)
    o, d = H.ray.shot_all_rays_dev(cap, torch.device('cuda'))
    ro, rd = O.shot_all_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, cap.shape)
    assert o.shape == d.shape == (cap.shape[0] * cap.shape[1], 3) and o.dtype == d.dtype == torch.float32
    np.testing.assert_array_equal(o.cpu().numpy(), np.asarray(ro, dtype=np.float32))
    u = ulps32(d.cpu().numpy(), np.asarray(rd, dtype=np.float32))
    assert u.max() <= 1 and (u > 0).mean() <= 1e-4, (u.max(), (u > 0).mean())
    n = torch.linalg.norm(d.double(), dim=1)
    assert (n - 1).abs().max() < 1e-7


@pytest.mark.parametrize("cam", CAMS[:3])
def test_shot_rays_dev_vs_reference_chain(H, cam):
    """mode 1 (ray_utils.py:23-29: world points cast to f32 before the centre is subtracted), arbitrary pixel lists."""
    cap = H.syn.SimpleCapture(c2w=H.syn.spherical_c2w(-80.0, 12.0, 3.0), **cam)
    h, w = cap.shape
    rng = np.random.default_rng(3)
    xy = np.stack([rng.integers(0, w, 5000), rng.integers(0, h, 5000)], axis=1)
    xy[:4] = [[0, 0], [w - 1, 0], [0, h - 1], [w - 1, h - 1]]
    o, d = H.ray.shot_rays_dev(cap, torch.as_tensor(xy, device='cuda'))
    ro, rd = O.shot_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, xy)
    np.testing.assert_array_equal(o.cpu().numpy(), np.asarray(ro, dtype=np.float32))
    u = ulps32(d.cpu().numpy(), np.asarray(rd, dtype=np.float32))
    assert u.max() <= 1 and (u > 0).mean() <= 1e-3, (u.max(), (u > 0).mean())


def test_shot_rays_golden_and_empty(H, golden):
    """the reference's own shot_rays / shot_all_rays outputs (tests/golden/ray_ops.npz)"""
    g = golden['ray_ops']
    K, c2w = g['cam_K'], g['cam_c2w']

    class Cap:
        intrinsic_matrix = K
        size = shape = (12, 16)

        class cam_pose:
            camera_to_world = c2w
    o, d = H.ray.shot_all_rays_dev(Cap, torch.device('cuda'))
    assert ulps32(d.cpu().numpy(), g['shot_all_d'].astype(np.float32)).max() <= 1
    np.testing.assert_array_equal(o.cpu().numpy(), g['shot_all_o'].astype(np.float32))
    coords = np.argwhere(np.ones((12, 16)))[:, ::-1].copy()
    o, d = H.ray.shot_rays_dev(Cap, torch.as_tensor(coords, device='cuda'))
    assert g['cam_c2w'].dtype == np.float32                    # the reference's pose matrix is f32 -> mode 2 (all-f32 tail)
    u = ulps32(d.cpu().numpy(), g['shot_rays_d'])
    assert u.max() <= 1 and (u > 0).mean() < 0.01, (u.max(), (u > 0).mean())
    o, d = H.ray.shot_rays_dev(Cap, torch.zeros((0, 2), dtype=torch.int32, device='cuda'))
    assert o.shape == (0, 3) and d.shape == (0, 3)


def test_renderers_use_identical_rays_either_way(H, monkeypatch, nets):
    """device-generated rays vs the host (numpy) mirror through a whole render: same frame except where a 1-ulp ray
    difference meets a tie (none expected at this size)."""
    j = nets[0][0].cuda()
    cap = H.syn.SimpleCapture(48, 40)
    a = H.render.render_vanilla(j, cap, None, samples_per_ray=24, importance_samples_per_ray=0)
    monkeypatch.setattr(H.render, "HOST_RAYS", True)
    b = H.render.render_vanilla(j, cap, None, samples_per_ray=24, importance_samples_per_ray=0)
    assert np.abs(a - b).max() < 1e-6


@pytest.mark.parametrize("n", [0, 1, 255, 257, 800 * 800 * 3])
def test_frame_to_uint8_and_psnr_vs_oracle(H, n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-0.05, 1.05, n).astype(np.float32)
    if n > 8:
        x[:8] = [0.0, 1.0, 0.5, 127.5 / 255, np.nextafter(np.float32(1), np.float32(2)), -1e-9, 0.4999999 / 255, 254.5 / 255]
    got = H.render.frame_to_uint8(torch.as_tensor(x, device='cuda'))
    ref = OF.to_uint8(x)
    assert got.dtype == torch.uint8 and got.shape == (n,)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)                              # bit-exact: byte work
    if n:
        y = np.clip(x + rng.normal(0, 0.02, n).astype(np.float32), 0, 1)
        gy = H.render.frame_to_uint8(torch.as_tensor(y, device='cuda'))
        p = H.render.psnr_uint8(got, gy)
        assert p == pytest.approx(OF.psnr_uint8(ref, OF.to_uint8(y)), rel=1e-12)
        assert H.render.psnr_uint8(got, got) == float('inf')


def test_frame_roundtrip_properties_full_size(H):
    """size-independent properties at 1080p: monotone, idempotent through the float round trip, PSNR symmetric"""
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.rand((1080, 1920, 3), device='cuda', generator=g)
    u = H.render.frame_to_uint8(x)
    assert u.shape == x.shape
    again = H.render.frame_to_uint8(u.float() / 255)
    assert torch.equal(u, again)
    order = torch.argsort(x.reshape(-1)[:100000])
    assert (torch.diff(u.reshape(-1)[:100000][order].int()) >= 0).all()
    v = H.render.frame_to_uint8((x + 0.01).clamp(0, 1))
    assert H.render.psnr_uint8(u, v) == H.render.psnr_uint8(v, u)
    with pytest.raises(Exception):
        H.render.psnr_uint8(u.cpu(), v.cpu())


@pytest.mark.parametrize("shape", [(7, 7, 1), (31, 45, 3), (200, 320, 3), (800, 800, 3)])
def test_ssim_vs_oracle(H, shape):
    """nm_ssim_u8 vs oracle/frame.py:ssim_uint8 (scipy uniform_filter, as scikit-image computes it)"""
    rng = np.random.default_rng(shape[0])
    yy, xx = np.mgrid[:shape[0], :shape[1]]
    base = (127 + 100 * np.sin(xx / 9.0)[..., None] * np.cos(yy / 13.0)[..., None] + rng.normal(0, 6, shape)).clip(0, 255)
    gt = base.astype(np.uint8)
    pred = (base + rng.normal(0, 9, shape)).clip(0, 255).astype(np.uint8)
    got = H.render.ssim_uint8(torch.as_tensor(pred, device='cuda'), torch.as_tensor(gt, device='cuda'))
    ref = OF.ssim_uint8(pred, gt)
    assert got == pytest.approx(ref, abs=1e-12), (got, ref)
    assert 0.0 < got < 1.0
    assert H.render.ssim_uint8(torch.as_tensor(gt, device='cuda'), torch.as_tensor(gt, device='cuda')) == pytest.approx(1.0, abs=1e-15)
    assert H.render.ssim_uint8(torch.as_tensor(gt, device='cuda'), torch.as_tensor(pred, device='cuda')) == pytest.approx(got, abs=1e-15)   # symmetric


def test_save_png_roundtrip(H, tmp_path):
    """the PNG written from a float frame decodes (independent mini-decoder: zlib + filter 0) to imageio's uint8 pixels"""
    import struct
    import zlib
    rng = np.random.default_rng(2)
    x = rng.uniform(-0.02, 1.02, (37, 53, 3)).astype(np.float32)
    p = tmp_path / "frame.png"
    H.render.save_png(str(p), x)
    b = p.read_bytes()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(b):
        n, tag = struct.unpack(">I", b[pos:pos + 4])[0], b[pos + 4:pos + 8]
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + data) & 0xffffffff)
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        if tag == b"IDAT":
            idat += data
        pos += 12 + n
    assert hdr == (53, 37, 8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(37, 1 + 53 * 3)
    assert (rows[:, 0] == 0).all()
    np.testing.assert_array_equal(rows[:, 1:].reshape(37, 53, 3), OF.to_uint8(x.reshape(-1)).reshape(37, 53, 3))
