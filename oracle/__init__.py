"""CPU oracle for the NeuMan ray-march hot path (TEST INFRASTRUCTURE ONLY).

This package is a numpy restatement of the reference algorithm
(apple/ml-neuman: utils/ray_utils.py, utils/render_utils.py:69-461,
models/vanilla.py, geometry/pcd_projector.py:85-153).  It exists to CHECK
the HIP path; it is never the thing shipped or measured:

* only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
  ``bench.py`` may import it;
* nothing under ``ml-neuman_amd/`` imports it (``tests/test_no_oracle_leak.py``
  enforces that).

Pinning status
--------------
* ray_ops / compositing / nerf_mlp / render (vanilla, smpl canonical):
  PINNED against outputs of the reference itself, generated in the build
  container by ``tests/golden/make_golden.py`` (imports /root/reference with
  stubs for the absent third-party wheels) and committed as ``tests/golden/*.npz``.
* warp (closest point on mesh + barycentric blend): **parity unpinned**.  The
  reference delegates the closest-point query to libigl 2.2.1
  (``igl.point_mesh_squared_distance`` / ``igl.barycentric_coordinates_tri``,
  environment.yml:13), which is not vendored and not installable here, and the
  reference has no test that pins it.  ``oracle/warp.py`` restates the
  published algorithm (exact Euclidean closest point on each triangle, global
  arg-min, barycentrics of the closest point) and is anchored on the
  reference's call site utils/ray_utils.py:48-66 and on the in-repo formula of
  utils/ray_utils.py:73-88 for the barycentric ordering.
* smpl (linear blend skinning -> per-frame verts / Ts): PINNED against the reference's own
  ``read_smpls`` / ``verts_transformations`` / ``batch_rodrigues`` / ``vertex_forward`` run on a synthetic SMPL-layout
  model (``tests/golden/make_golden_smpl.py`` -> ``tests/golden/smpl.npz``), to float32 tolerance.
* train (one training step of the background NeRF: losses and parameter gradients): PINNED against the
  reference's own ``NeRFTrainer.loss_func`` + ``backward()`` (``tests/golden/make_golden_train.py`` ->
  ``tests/golden/train.npz``); ``oracle/train.py`` is torch float64 + autograd.
* frame (float -> uint8, uint8 PSNR, SSIM): **parity unpinned**.  imageio and
  scikit-image (environment.yml:22, :30, no versions) are absent;
  ``oracle/frame.py`` restates their published rules.
"""
