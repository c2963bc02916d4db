// a11 observation -> canonical warp of ray samples for gfx950.
// Replaces reference utils/ray_utils.py:48-66 (warp_samples_to_canonical), whose closest-point query is
// libigl's CPU AABB tree behind a device->host->device round trip per ray batch
// (utils/render_utils.py:218-227).  Here the whole warp stays on the GPU:
//
//   prep kernel : per triangle {a, b, c, bounding sphere} -> 64 B record (workspace), so the search
//                 loop reads one wave-uniform record per triangle through the scalar cache;
//   warp kernel : one workgroup per ray, one lane per sample.  Nearest-vertex pass gives an upper
//                 bound, then every triangle whose bounding sphere can beat the bound gets the exact
//                 Voronoi-region closest-point test (f32, like the reference's f32 query).  The
//                 winning triangle's barycentrics, the blended 4x4 (f64, reference T is f64), its
//                 inverse and the canonical point are f64; canonical points of the ray are staged in
//                 LDS so the finite-difference directions (:62-64) need no second pass over HBM.
#include "common.h"

namespace {

struct TriRec {          // 16 floats
    float ax, ay, az, bx, by, bz, cx, cy, cz;
    float sx, sy, sz, sr;   // bounding sphere
    float pad[3];
};

__global__ __launch_bounds__(256) void tri_prep_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int F,
                                                       TriRec* __restrict__ rec) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    TriRec t;
    t.ax = verts[i0 * 3]; t.ay = verts[i0 * 3 + 1]; t.az = verts[i0 * 3 + 2];
    t.bx = verts[i1 * 3]; t.by = verts[i1 * 3 + 1]; t.bz = verts[i1 * 3 + 2];
    t.cx = verts[i2 * 3]; t.cy = verts[i2 * 3 + 1]; t.cz = verts[i2 * 3 + 2];
    t.sx = (t.ax + t.bx + t.cx) * (1.f / 3.f);
    t.sy = (t.ay + t.by + t.cy) * (1.f / 3.f);
    t.sz = (t.az + t.bz + t.cz) * (1.f / 3.f);
    const float da = (t.ax - t.sx) * (t.ax - t.sx) + (t.ay - t.sy) * (t.ay - t.sy) + (t.az - t.sz) * (t.az - t.sz);
    const float db = (t.bx - t.sx) * (t.bx - t.sx) + (t.by - t.sy) * (t.by - t.sy) + (t.bz - t.sz) * (t.bz - t.sz);
    const float dc = (t.cx - t.sx) * (t.cx - t.sx) + (t.cy - t.sy) * (t.cy - t.sy) + (t.cz - t.sz) * (t.cz - t.sz);
    t.sr = sqrtf(fmaxf(da, fmaxf(db, dc))) * 1.0001f + 1e-7f;
    t.pad[0] = t.pad[1] = t.pad[2] = 0.f;
    rec[f] = t;
}

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// exact closest point on triangle (a,b,c) to p: Voronoi-region test (Ericson, RTCD 5.1.5)
__device__ __forceinline__ V3 closest_on_tri(V3 p, V3 a, V3 b, V3 c) {
    const V3 ab = sub(b, a), ac = sub(c, a), ap = sub(p, a);
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    const V3 bp = sub(p, b);
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    const V3 cp = sub(p, c);
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    float v, w;   // q = a + v*ab + w*ac
    if (d1 <= 0.f && d2 <= 0.f) { v = 0.f; w = 0.f; }                              // vertex A
    else if (d3 >= 0.f && d4 <= d3) { v = 1.f; w = 0.f; }                          // vertex B
    else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { v = d1 / (d1 - d3); w = 0.f; } // edge AB
    else if (d6 >= 0.f && d5 <= d6) { v = 0.f; w = 1.f; }                          // vertex C
    else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { v = 0.f; w = d2 / (d2 - d6); } // edge AC
    else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {                  // edge BC
        w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        v = 1.f - w;
    } else {                                                                       // interior
        const float denom = 1.f / (va + vb + vc);
        v = vb * denom;
        w = vc * denom;
    }
    return {a.x + ab.x * v + ac.x * w, a.y + ab.y * v + ac.y * w, a.z + ab.z * v + ac.z * w};
}

// inverse of a general 4x4 (row-major) by cofactors, f64
__device__ __forceinline__ void inv4x4(const double* m, double* o) {
    const double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const double inv = 1.0 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    o[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * inv;
    o[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * inv;
    o[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * inv;
    o[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * inv;
    o[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * inv;
    o[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * inv;
    o[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * inv;
    o[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * inv;
    o[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * inv;
    o[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * inv;
    o[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * inv;
    o[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * inv;
    // last row not needed: only the first three components of T^-1 [p;1] are used (ray_utils.py:58)
}

__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ pts, int S, const float* __restrict__ verts, int V,
                                                   const int32_t* __restrict__ faces, int F, const TriRec* __restrict__ rec,
                                                   const double* __restrict__ T, float* __restrict__ can_pts,
                                                   float* __restrict__ can_dirs, float* __restrict__ closest) {
    extern __shared__ double can_lds[];                     // [S][3]
    const int64_t r = blockIdx.x;
    for (int s0 = 0; s0 < S; s0 += blockDim.x) {
        const int s = s0 + threadIdx.x;
        const bool live = s < S;
        const int64_t i = r * S + (live ? s : S - 1);
        const V3 p = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
        // ---- upper bound: nearest vertex
        float best = INFINITY;
#pragma unroll 4
        for (int v = 0; v < V; ++v) {
            const float dx = verts[v * 3] - p.x, dy = verts[v * 3 + 1] - p.y, dz = verts[v * 3 + 2] - p.z;
            best = fminf(best, dx * dx + dy * dy + dz * dz);
        }
        best = best * 1.0001f + 1e-12f;                      // keep the bound an upper bound under rounding
        float sb = sqrtf(best);
        int bf = 0;
        V3 q = p;
        // ---- exact search with bounding-sphere culling
        for (int f = 0; f < F; ++f) {
            const TriRec t = rec[f];                         // wave-uniform -> scalar loads
            const float cx = t.sx - p.x, cy = t.sy - p.y, cz = t.sz - p.z;
            const float lim = sb + t.sr;
            if (cx * cx + cy * cy + cz * cz <= lim * lim) {
                const V3 c = closest_on_tri(p, {t.ax, t.ay, t.az}, {t.bx, t.by, t.bz}, {t.cx, t.cy, t.cz});
                const V3 d = sub(c, p);
                const float d2 = dot(d, d);
                if (d2 < best) {                              // strict: ties keep the lowest face id
                    best = d2; sb = sqrtf(d2); bf = f; q = c;
                }
            }
        }
        // ---- barycentrics of q in the winning triangle, igl.barycentric_coordinates_tri (ray_utils.py:55), f64
        const int i0 = faces[bf * 3], i1 = faces[bf * 3 + 1], i2 = faces[bf * 3 + 2];
        const double ax = verts[i0 * 3], ay = verts[i0 * 3 + 1], az = verts[i0 * 3 + 2];
        const double v0x = (double)verts[i1 * 3] - ax, v0y = (double)verts[i1 * 3 + 1] - ay, v0z = (double)verts[i1 * 3 + 2] - az;
        const double v1x = (double)verts[i2 * 3] - ax, v1y = (double)verts[i2 * 3 + 1] - ay, v1z = (double)verts[i2 * 3 + 2] - az;
        const double v2x = (double)q.x - ax, v2y = (double)q.y - ay, v2z = (double)q.z - az;
        const double d00 = v0x * v0x + v0y * v0y + v0z * v0z, d01 = v0x * v1x + v0y * v1y + v0z * v1z;
        const double d11 = v1x * v1x + v1y * v1y + v1z * v1z;
        const double d20 = v2x * v0x + v2y * v0y + v2z * v0z, d21 = v2x * v1x + v2y * v1y + v2z * v1z;
        const double den = d00 * d11 - d01 * d01;
        const double bv = (d11 * d20 - d01 * d21) / den, bw = (d00 * d21 - d01 * d20) / den, bu = 1.0 - bv - bw;
        // ---- blended transform (ray_utils.py:56), inverse (:57), canonical point (:58)
        double M[16], Mi[12];
        const double* T0 = T + (int64_t)i0 * 16;
        const double* T1 = T + (int64_t)i1 * 16;
        const double* T2 = T + (int64_t)i2 * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) M[k] = T0[k] * bu + T1[k] * bv + T2[k] * bw;
        inv4x4(M, Mi);
        const double px = p.x, py = p.y, pz = p.z;
        const double cxp = Mi[0] * px + Mi[1] * py + Mi[2] * pz + Mi[3];
        const double cyp = Mi[4] * px + Mi[5] * py + Mi[6] * pz + Mi[7];
        const double czp = Mi[8] * px + Mi[9] * py + Mi[10] * pz + Mi[11];
        if (live) {
            can_lds[s * 3] = cxp; can_lds[s * 3 + 1] = cyp; can_lds[s * 3 + 2] = czp;
            can_pts[i * 3] = (float)cxp; can_pts[i * 3 + 1] = (float)cyp; can_pts[i * 3 + 2] = (float)czp;
            if (closest) { closest[i * 3] = q.x; closest[i * 3 + 1] = q.y; closest[i * 3 + 2] = q.z; }
        }
    }
    __syncthreads();
    // ---- canonical ray directions: forward differences, last one repeated, normalised (ray_utils.py:62-64)
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const int a = s < S - 1 ? s : S - 2;
        const double dx = can_lds[(a + 1) * 3] - can_lds[a * 3];
        const double dy = can_lds[(a + 1) * 3 + 1] - can_lds[a * 3 + 1];
        const double dz = can_lds[(a + 1) * 3 + 2] - can_lds[a * 3 + 2];
        const double nrm = sqrt(dx * dx + dy * dy + dz * dz);
        const int64_t i = r * S + s;
        can_dirs[i * 3] = (float)(dx / nrm); can_dirs[i * 3 + 1] = (float)(dy / nrm); can_dirs[i * 3 + 2] = (float)(dz / nrm);
    }
}

}  // namespace

extern "C" {

int64_t nm_warp_workspace_floats(int F) { return (int64_t)F * 16; }

int nm_warp_to_canonical(const float* pts, int64_t R, int S, const float* verts, int V, const int32_t* faces, int F,
                         const double* T, float* can_pts, float* can_dirs, float* closest, float* workspace,
                         nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (pts && verts && faces && T && can_pts && can_dirs && workspace), "nm_warp_to_canonical: null pointer");
    NM_REQUIRE(R >= 0 && S >= 2 && V >= 3 && F >= 1, "nm_warp_to_canonical: bad sizes R=%lld S=%d V=%d F=%d", (long long)R, S, V, F);
    NM_REQUIRE((size_t)S * 24 <= 64 * 1024, "nm_warp_to_canonical: S=%d exceeds the LDS staging budget", S);
    NM_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 63) == 0, "nm_warp_to_canonical: workspace must be 64-byte aligned");
    NM_REQUIRE(R < (1ll << 31), "nm_warp_to_canonical: too many rays for one launch");
    if (R == 0) return NM_OK;
    hipStream_t st = nm::as_stream(stream);
    TriRec* rec = reinterpret_cast<TriRec*>(workspace);
    hipLaunchKernelGGL(tri_prep_kernel, dim3((F + 255) / 256), dim3(256), 0, st, verts, faces, F, rec);
    if (int e = nm::check_launch("tri_prep_kernel")) return e;
    const int threads = S <= 64 ? 64 : (S <= 128 ? 128 : 256);
    hipLaunchKernelGGL(warp_kernel, dim3((unsigned)R), dim3(threads), (size_t)S * 24, st, pts, S, verts, V, faces, F, rec, T,
                       can_pts, can_dirs, closest);
    return nm::check_launch("warp_kernel");
}

}  // extern "C"
