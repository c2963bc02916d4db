// Sequenced per-ray entry points (SURVEY 8b's proposed ABI: nm_render_rays_bkg / _human, nm_merge_composite): ONE C call per pass of the
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
        // (the coarse composite's colours are discarded, as the reference discards them, :139-141: ONE kernel from sigma to the samples)
        (void)wc;
        if ((rc = nm_importance_from_raw(rawc, zc, direction, R, S, u, N, z_out, nullptr, stream))) return rc;
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

// (the merged list lives in LDS now: nothing is needed; kept for callers that size a workspace)
int64_t nm_merge_composite_workspace_floats(int64_t R, int Sa, int Sb) { (void)R; (void)Sa; (void)Sb; return 4; }

int nm_merge_composite(const float* za, const float* rawa, int Sa, const float* zb, const float* rawb, int Sb, int64_t R, const float* rays_d,
                       int white_bkg, float* workspace, float* rgb, float* depth, float* acc, nm_stream_t stream) {
    (void)workspace;
    NM_REQUIRE(R == 0 || (za && rawa && zb && rawb && rays_d && rgb && depth && acc), "nm_merge_composite: null pointer");
    if (R == 0) return NM_OK;
    const float* zs[2] = {za, zb};
    const float* raws[2] = {rawa, rawb};
    const int Ss[2] = {Sa, Sb};
    return nm_merge_composite_lists(2, zs, raws, nullptr, Ss, R, rays_d, white_bkg, rgb, depth, acc, stream);
}

// ---- render_hybrid_nerf's per-batch body (utils/render_utils.py:287-353) as one call: two-pass background for every ray and its
// composite (what a ray that misses the body keeps, :303-311), near / far against the posed body, compaction of the hit rays (the one
// host read of the call: their count -- the reference's boolean-mask indexing implies the same), human pass of the hit rays, merged
// composite (:330-345) and the human-only accumulation (:345-350) scattered back.  Built from the entry points above and below: same
// kernels, same bits as calling them one by one (tests/test_hip_fused.py).
namespace {
struct HybridWs {
    float *near_b, *far_b, *bkg_ws, *raw_b, *z_b, *near_h, *far_h, *ho, *hd, *hn, *hf, *bz, *braw, *h_raw, *h_z, *human_ws, *merge_ws, *rgb_h, *depth_h, *acc_h,
        *scratch;
    int32_t *hit, *counts, *cws;
    int64_t total;
};
inline HybridWs hybrid_layout(float* base, int64_t R, int S, int N, int Sh) {
    HybridWs w;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* p = base ? base + o : nullptr; o += align4(n); return p; };
    const int Sb = S + N;
    w.near_b = take(R); w.far_b = take(R);
    w.bkg_ws = take(nm_render_rays_bkg_workspace_floats(R, S, N));
    w.raw_b = take(R * Sb * 4); w.z_b = take(R * Sb);
    w.near_h = take(R); w.far_h = take(R);
    w.hit = reinterpret_cast<int32_t*>(take(R)); w.counts = reinterpret_cast<int32_t*>(take(4));
    w.cws = reinterpret_cast<int32_t*>(take(nm_compact_workspace_ints(R)));
    w.ho = take(R * 3); w.hd = take(R * 3); w.hn = take(R); w.hf = take(R);
    w.bz = nullptr; w.braw = nullptr;                             // (the merge reads the background rows in place)
    w.h_raw = take(R * Sh * 4); w.h_z = take(R * Sh);
    w.human_ws = take(nm_render_rays_human_workspace_floats(R, Sh, 1));
    w.merge_ws = nullptr;
    w.rgb_h = take(R * 3); w.depth_h = take(R); w.acc_h = take(R); w.scratch = take(R * 6);
    w.total = o;
    return w;
}
}  // namespace

int64_t nm_render_rays_hybrid_workspace_floats(int64_t R, int S, int N, int S_human) { return hybrid_layout(nullptr, R, S, N, S_human).total; }

int nm_render_rays_hybrid(nm_mlp_t coarse, nm_mlp_t fine, nm_mlp_t human, nm_mesh_t mesh, const double* T, const float* verts, int V,
                          double geo_threshold, const float* origin, const float* direction, int64_t R, float bkg_near, float bkg_far, int S, int N,
                          int S_human, const float* t_vals, const float* u, const float* t_vals_human, int white_bkg, int precision_coarse,
                          int precision_fine, int precision_human, float* workspace, float* rgb, float* depth, float* acc, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (coarse && human && mesh && T && verts && origin && direction && t_vals && t_vals_human && workspace && rgb && depth && acc),
               "nm_render_rays_hybrid: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1 && N >= 0 && S_human >= 2 && V >= 1, "nm_render_rays_hybrid: bad sizes");
    if (R == 0) return NM_OK;
    hipStream_t st = nm::as_stream(stream);
    const HybridWs w = hybrid_layout(workspace, R, S, N, S_human);
    const int Sb = S + N;
    int rc;
    union { float f; uint32_t u; } nb{bkg_near}, fb{bkg_far};
    if ((rc = nm::check_hip(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.near_b), (int)nb.u, (size_t)R, st), "nm_render_rays_hybrid: near"))) return rc;
    if ((rc = nm::check_hip(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(w.far_b), (int)fb.u, (size_t)R, st), "nm_render_rays_hybrid: far"))) return rc;
    // every ray: the background-only composite; acc is forced to 0 where the body is missed (:311)
    if ((rc = nm_render_rays_bkg(coarse, fine, origin, direction, w.near_b, w.far_b, R, S, N, t_vals, u, white_bkg, precision_coarse, precision_fine, w.bkg_ws,
                                 w.raw_b, w.z_b, rgb, depth, acc, stream))) return rc;
    if ((rc = nm::check_hip(hipMemsetAsync(acc, 0, (size_t)R * 4, st), "nm_render_rays_hybrid: acc"))) return rc;
    if ((rc = nm_near_far(origin, direction, R, verts, V, geo_threshold, w.near_h, w.far_h, stream))) return rc;
    if ((rc = nm_compact_hits(w.near_h, w.far_h, R, w.hit, nullptr, w.counts, w.cws, stream))) return rc;
    int32_t n_hit = 0;
    if ((rc = nm::check_hip(hipMemcpyAsync(&n_hit, w.counts, 4, hipMemcpyDeviceToHost, st), "nm_render_rays_hybrid: hit count"))) return rc;
    if ((rc = nm::check_hip(hipStreamSynchronize(st), "nm_render_rays_hybrid: hit count"))) return rc;
    if (n_hit == 0) return NM_OK;
    // the hit rays: overwritten by the merged human + background composite (:313-353)
    if ((rc = nm_gather_rows(origin, w.hit, nullptr, n_hit, 3, w.ho, stream))) return rc;
    if ((rc = nm_gather_rows(direction, w.hit, nullptr, n_hit, 3, w.hd, stream))) return rc;
    if ((rc = nm_gather_rows(w.near_h, w.hit, nullptr, n_hit, 1, w.hn, stream))) return rc;
    if ((rc = nm_gather_rows(w.far_h, w.hit, nullptr, n_hit, 1, w.hf, stream))) return rc;
    if ((rc = nm_render_rays_human(human, mesh, T, w.ho, w.hd, w.hn, w.hf, n_hit, S_human, t_vals_human, white_bkg, 1.f, precision_human, w.human_ws, w.h_raw,
                                   w.h_z, nullptr, nullptr, nullptr, stream))) return rc;
    {                                                                 // merged composite: the background lists of the hit rays read in place
        const float* zs[2] = {w.z_b, w.h_z};
        const float* raws[2] = {w.raw_b, w.h_raw};
        const int32_t* rows[2] = {w.hit, nullptr};
        const int Ss[2] = {Sb, S_human};
        if ((rc = nm_merge_composite_lists(2, zs, raws, rows, Ss, n_hit, w.hd, white_bkg, w.rgb_h, w.depth_h, w.acc_h, stream))) return rc;
    }
    if ((rc = nm_composite(w.h_raw, w.h_z, w.hd, n_hit, S_human, white_bkg, nullptr, w.scratch, w.scratch + 3 * (int64_t)n_hit, w.acc_h, nullptr,
                           w.scratch + 5 * (int64_t)n_hit, stream))) return rc;                     // the human-only accumulation (:345-350)
    if ((rc = nm_scatter_rows(w.rgb_h, w.hit, nullptr, n_hit, 3, rgb, stream))) return rc;
    if ((rc = nm_scatter_rows(w.depth_h, w.hit, nullptr, n_hit, 1, depth, stream))) return rc;
    return nm_scatter_rows(w.acc_h, w.hit, nullptr, n_hit, 1, acc, stream);
}

}  // extern "C"
