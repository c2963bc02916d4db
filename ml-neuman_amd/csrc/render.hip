// Fused per-ray entry points (SURVEY 8b's proposed ABI: nm_render_rays_bkg / _human, nm_merge_composite): ONE C call per pass of the
// reference's renderers, the kernels of the pass enqueued back to back on the caller's stream -- no host synchronisation inside, no
// hidden allocation (every intermediate lives in the caller's workspace or output arrays), same kernels and therefore the same bits
// as the step-by-step entry points.  Reference: utils/render_utils.py:131-151 / 287-297 (two-pass background), :213-229 / 320-329
// (human pass of already compacted hit rays), :330-345 / 441-456 (merge + composite).
#include "common.h"

namespace {
inline int64_t align4(int64_t n) { return (n + 3) & ~int64_t(3); }        // keep every sub-array 16-byte aligned
}  // namespace

extern "C" {

int64_t nm_render_rays_bkg_workspace_floats(int64_t R, int S, int N) {
    // coarse z [R,S] | coarse raw [R,S,4] | coarse weights [R,S] | per-ray scratch of the coarse composite [R,6]
    return N > 0 ? align4(R * S) + align4(R * S * 4) + align4(R * S) + align4(R * 6) : align4(R * 6);
}

int nm_render_rays_bkg(nm_mlp_t coarse, nm_mlp_t fine, const float* origin, const float* direction, const float* near, const float* far,
                       int64_t R, int S, int N, const float* t_vals, const float* u, int white_bkg, int precision_coarse, int precision_fine,
                       float* workspace, float* raw_out, float* z_out, float* rgb, float* depth, float* acc, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (coarse && origin && direction && near && far && t_vals && workspace && raw_out && z_out), "nm_render_rays_bkg: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1 && N >= 0 && (N == 0) == (fine == nullptr), "nm_render_rays_bkg: a fine net and N > 0 importance samples go together (S=%d N=%d)", S, N);
    NM_REQUIRE(N == 0 || u, "nm_render_rays_bkg: u [N] is missing");
    if (R == 0) return NM_OK;
    int rc;
    float* scratch = workspace + (N > 0 ? align4(R * S) + align4(R * S * 4) + align4(R * S) : 0);      // [R,6]
    if (!fine) {                                                  // one pass: its output is what is composited
        if ((rc = nm_ray_to_samples(origin, direction, near, far, R, S, t_vals, 0, nullptr, nullptr, nullptr, z_out, stream))) return rc;
        if ((rc = nm_mlp_forward_rays(coarse, origin, direction, z_out, R, S, precision_coarse, 1.f, raw_out, stream))) return rc;
    } else {
        float* zc = workspace;
        float* rawc = zc + align4(R * S);
        float* wc = rawc + align4(R * S * 4);
        // (scratch: rgb [R,3] | disp | acc | depth of the coarse composite, discarded as the reference discards them, :139-141)
        if ((rc = nm_ray_to_samples(origin, direction, near, far, R, S, t_vals, 0, nullptr, nullptr, nullptr, zc, stream))) return rc;
        if ((rc = nm_mlp_sigma_rays(coarse, origin, direction, zc, R, S, precision_coarse, 1.f, rawc, stream))) return rc;
        if ((rc = nm_composite(rawc, zc, direction, R, S, white_bkg, nullptr, scratch, scratch + 3 * R, scratch + 4 * R, wc, scratch + 5 * R, stream))) return rc;
        if ((rc = nm_importance_z(zc, wc, R, S, u, N, 1, z_out, stream))) return rc;
        if ((rc = nm_mlp_forward_rays(fine, origin, direction, z_out, R, S + N, precision_fine, 1.f, raw_out, stream))) return rc;
    }
    if (rgb) {
        NM_REQUIRE(depth && acc, "nm_render_rays_bkg: rgb, depth and acc go together");
        if ((rc = nm_composite(raw_out, z_out, direction, R, S + N, white_bkg, nullptr, rgb, scratch + 3 * R, acc, nullptr, depth, stream))) return rc;
    }
    return NM_OK;
}

int64_t nm_render_rays_human_workspace_floats(int64_t R, int S, int posed) {
    // posed: observation-space points [R,S,3] | canonical points | canonical directions;  + disp [R]
    return (posed ? 3 * align4(R * S * 3) : 0) + align4(R);
}

int nm_render_rays_human(nm_mlp_t human, nm_mesh_t mesh, const double* T, const float* origin, const float* direction, const float* near,
                         const float* far, int64_t R, int S, const float* t_vals, int white_bkg, float sigma_scale, int precision,
                         float* workspace, float* raw_out, float* z_out, float* rgb, float* depth, float* acc, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (human && origin && direction && near && far && t_vals && workspace && raw_out && z_out), "nm_render_rays_human: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1 && (mesh == nullptr) == (T == nullptr), "nm_render_rays_human: a posed mesh and its transforms go together");
    if (R == 0) return NM_OK;
    int rc;
    float* disp = workspace + (mesh ? 3 * align4(R * S * 3) : 0);
    if (!mesh) {                                                  // canonical render (render_can=True, :213-216): the camera ray is the view direction
        if ((rc = nm_ray_to_samples(origin, direction, near, far, R, S, t_vals, 0, nullptr, nullptr, nullptr, z_out, stream))) return rc;
        if ((rc = nm_mlp_forward_rays(human, origin, direction, z_out, R, S, precision, sigma_scale, raw_out, stream))) return rc;
    } else {                                                      // posed: warp the samples, directions = differences of warped points (:217-227)
        float* pts = workspace;
        float* can_pts = pts + align4(R * S * 3);
        float* can_dirs = can_pts + align4(R * S * 3);
        if ((rc = nm_ray_to_samples(origin, direction, near, far, R, S, t_vals, 0, nullptr, pts, nullptr, z_out, stream))) return rc;
        if ((rc = nm_warp_to_canonical(mesh, pts, R, S, T, can_pts, can_dirs, nullptr, stream))) return rc;
        if ((rc = nm_mlp_forward(human, can_pts, can_dirs, R * S, precision, sigma_scale, raw_out, stream))) return rc;
    }
    if (rgb) {
        NM_REQUIRE(depth && acc, "nm_render_rays_human: rgb, depth and acc go together");
        if ((rc = nm_composite(raw_out, z_out, direction, R, S, white_bkg, nullptr, rgb, disp, acc, nullptr, depth, stream))) return rc;
    }
    return NM_OK;
}

int64_t nm_merge_composite_workspace_floats(int64_t R, int Sa, int Sb) { return align4(R * (Sa + Sb)) + align4(R * (Sa + Sb) * 4) + align4(R); }

int nm_merge_composite(const float* za, const float* rawa, int Sa, const float* zb, const float* rawb, int Sb, int64_t R, const float* rays_d,
                       int white_bkg, float* workspace, float* rgb, float* depth, float* acc, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (za && rawa && zb && rawb && rays_d && workspace && rgb && depth && acc), "nm_merge_composite: null pointer");
    if (R == 0) return NM_OK;
    float* z = workspace;
    float* raw = z + align4(R * (Sa + Sb));
    float* disp = raw + align4(R * (Sa + Sb) * 4);
    int rc;
    if ((rc = nm_merge_sorted(za, rawa, Sa, zb, rawb, Sb, R, z, raw, stream))) return rc;
    return nm_composite(raw, z, rays_d, R, Sa + Sb, white_bkg, nullptr, rgb, disp, acc, nullptr, depth, stream);
}

}  // extern "C"
