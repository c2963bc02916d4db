// a11 observation -> canonical warp of ray samples for gfx950.
// Replaces reference utils/ray_utils.py:48-66 (warp_samples_to_canonical), whose closest-point query is
// libigl's CPU AABB tree behind a device->host->device round trip per ray batch
// (utils/render_utils.py:218-227).  Here the whole warp stays on the GPU.
//
// nm_mesh_create (once per posed mesh, i.e. per frame and actor) builds an exact acceleration structure:
//   * per triangle a 64 B record {a, b, c, bounding sphere};
//   * a uniform grid over the vertex AABB inflated by `reach` (the largest distance a query point can have from
//     the vertices: geo_threshold for the render paths).  For each cell C with centre c and half diagonal hd,
//     ub(C) = |c - nearest vertex| + hd bounds the mesh distance of every point of C, so only triangles with
//     key(t) = |sphere centre - c| - sphere radius <= ub(C) + hd can be the closest triangle of a point in C.
//     Those candidates are stored per cell, bucketed by key into kRings distance rings (counting sort).
// warp_kernel (one workgroup per ray, one lane per sample) walks the rings of its cell in order and stops as soon as a
// ring's lower bound exceeds the best distance found; each candidate first gets a bounding-sphere test, then the exact
// Voronoi-region closest-point test (f32, like the reference's f32 query).  Cells farther than `reach` from every
// vertex hold no list; points there (never produced by the render paths) take the brute-force loop over all triangles,
// so the result is exact everywhere.  The winning triangle's barycentrics, the blended 4x4 (f64, the reference's T is
// f64), its inverse and the canonical point are f64; the ray's canonical points are staged in LDS so the
// finite-difference directions (:62-64) need no second pass over HBM.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace {

constexpr int kRings = 8;
constexpr int kMaxCells = 48 * 1024;

struct TriRec {          // 16 floats
    float sx, sy, sz, sr;   // bounding sphere (first 16 B: the cull test reads only this)
    float ax, ay, az, bx, by, bz, cx, cy, cz;
    float pad[3];
};

struct Grid {
    float lox, loy, loz, h, inv_h, half_diag, ring_w, reach;
    int nx, ny, nz, ncells;
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

__device__ __forceinline__ void cell_centre(const Grid& g, int cell, float& cx, float& cy, float& cz) {
    const int ix = cell % g.nx, iy = (cell / g.nx) % g.ny, iz = cell / (g.nx * g.ny);
    cx = g.lox + (ix + .5f) * g.h;
    cy = g.loy + (iy + .5f) * g.h;
    cz = g.loz + (iz + .5f) * g.h;
}

// ub(C) per cell; < 0 marks a cell farther than `reach` from every vertex (no list, brute force)
__global__ __launch_bounds__(256) void cell_bound_kernel(Grid g, const float* __restrict__ verts, int V, float* __restrict__ ub) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= g.ncells) return;
    float cx, cy, cz;
    cell_centre(g, cell, cx, cy, cz);
    float best = INFINITY;
#pragma unroll 4
    for (int v = 0; v < V; ++v) {
        const float dx = verts[v * 3] - cx, dy = verts[v * 3 + 1] - cy, dz = verts[v * 3 + 2] - cz;
        best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
    const float dv = sqrtf(best);
    ub[cell] = (dv - g.half_diag <= g.reach) ? (dv + g.half_diag) * 1.0001f + 1e-6f : -1.f;
}

// FILL = false: count candidates per (cell, ring); FILL = true: write triangle ids at the scanned offsets
template <bool FILL>
__global__ __launch_bounds__(256) void cell_lists_kernel(Grid g, const TriRec* __restrict__ rec, int F, const float* __restrict__ ub,
                                                         int32_t* __restrict__ counts_or_offsets, int32_t* __restrict__ list) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= g.ncells) return;
    const float u = ub[cell];
    int n[kRings];
#pragma unroll
    for (int r = 0; r < kRings; ++r) n[r] = FILL ? counts_or_offsets[cell * kRings + r] : 0;
    if (u >= 0.f) {
        float cx, cy, cz;
        cell_centre(g, cell, cx, cy, cz);
        const float lim = u + g.half_diag;
        for (int f = 0; f < F; ++f) {
            const float4 s = *reinterpret_cast<const float4*>(&rec[f]);      // wave-uniform -> scalar load
            const float dx = s.x - cx, dy = s.y - cy, dz = s.z - cz;
            const float key = sqrtf(dx * dx + dy * dy + dz * dz) - s.w;
            if (key <= lim) {
                int r = (int)(fmaxf(key, 0.f) / g.ring_w);
                r = r < kRings - 1 ? r : kRings - 1;
#pragma unroll
                for (int q = 0; q < kRings; ++q)
                    if (q == r) {
                        if (FILL) list[n[q]] = f;
                        n[q]++;
                    }
            }
        }
    }
    if (!FILL) {
#pragma unroll
        for (int r = 0; r < kRings; ++r) counts_or_offsets[cell * kRings + r] = n[r];
    }
}

// exclusive scan of `n` int32 in place (single block), total into total[0]
__global__ __launch_bounds__(1024) void scan_kernel(int32_t* __restrict__ a, int n, int32_t* __restrict__ total) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? a[i] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wave_tot[wid] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wave_tot[w];
        const int carry = carry_s;
        if (i < n) a[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[0] = carry_s;
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

// inverse of a general 4x4 (row-major) by cofactors, f64; only the first three rows are produced
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
}

struct Best {
    float d2, sd;   // best squared distance, its square root
    int f;
    V3 q;
};

__device__ __forceinline__ void try_tri(const TriRec* __restrict__ rec, int f, V3 p, Best& b) {
    const float4 s = *reinterpret_cast<const float4*>(&rec[f]);
    const float cx = s.x - p.x, cy = s.y - p.y, cz = s.z - p.z;
    const float lim = b.sd + s.w;
    if (cx * cx + cy * cy + cz * cz <= lim * lim) {
        const float4 t0 = reinterpret_cast<const float4*>(&rec[f])[1];
        const float4 t1 = reinterpret_cast<const float4*>(&rec[f])[2];
        const float t2 = rec[f].cz;
        const V3 c = closest_on_tri(p, {t0.x, t0.y, t0.z}, {t0.w, t1.x, t1.y}, {t1.z, t1.w, t2});
        const V3 d = sub(c, p);
        const float d2 = dot(d, d);
        if (d2 < b.d2 || (d2 == b.d2 && f < b.f)) {              // ties: lowest face id, independent of visiting order
            b.d2 = d2; b.sd = sqrtf(d2); b.f = f; b.q = c;
        }
    }
}

__global__ __launch_bounds__(256) void warp_kernel(Grid g, const float* __restrict__ pts, int S, const float* __restrict__ verts, int V,
                                                   const int32_t* __restrict__ faces, int F, const TriRec* __restrict__ rec,
                                                   const float* __restrict__ ub, const int32_t* __restrict__ offsets,
                                                   const int32_t* __restrict__ list, const double* __restrict__ T,
                                                   float* __restrict__ can_pts, float* __restrict__ can_dirs,
                                                   float* __restrict__ closest) {
    extern __shared__ double can_lds[];                     // [S][3]
    const int64_t r = blockIdx.x;
    for (int s0 = 0; s0 < S; s0 += blockDim.x) {
        const int s = s0 + threadIdx.x;
        const bool live = s < S;
        const int64_t i = r * S + (live ? s : S - 1);
        const V3 p = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
        Best b;
        b.d2 = INFINITY; b.sd = INFINITY; b.f = 0; b.q = p;
        // ---- grid lookup
        const float fx = (p.x - g.lox) * g.inv_h, fy = (p.y - g.loy) * g.inv_h, fz = (p.z - g.loz) * g.inv_h;
        const int ix = (int)floorf(fx), iy = (int)floorf(fy), iz = (int)floorf(fz);
        bool brute = !(fx >= 0.f && fy >= 0.f && fz >= 0.f && ix < g.nx && iy < g.ny && iz < g.nz);   // also catches NaN
        if (!brute) {
            const int cell = (iz * g.ny + iy) * g.nx + ix;
            const float u = ub[cell];
            if (u < 0.f) brute = true;
            else {
                b.d2 = u * u; b.sd = u; b.f = 0x7fffffff;     // the closest triangle is strictly inside this bound
                for (int ring = 0; ring < kRings; ++ring) {
                    if (ring * g.ring_w - g.half_diag > b.sd) break;     // every triangle of this ring (and beyond) is farther
                    const int e = offsets[cell * kRings + ring + 1];
                    for (int k = offsets[cell * kRings + ring]; k < e; ++k) try_tri(rec, list[k], p, b);
                }
                if (b.f == 0x7fffffff) brute = true;          // cannot happen for a consistent grid; stay exact anyway
            }
        }
        if (brute) {
            b.d2 = INFINITY; b.sd = INFINITY; b.f = 0;
            for (int f = 0; f < F; ++f) try_tri(rec, f, p, b);
        }
        const int bf = b.f;
        const V3 q = b.q;
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

struct nm_mesh_s {
    int V, F;
    Grid g;
    float* d_verts;      // owned copies: the handle outlives the caller's tensors
    int32_t* d_faces;
    TriRec* d_rec;
    float* d_ub;
    int32_t* d_offsets;  // ncells*kRings + 1
    int32_t* d_list;
    int64_t list_len;
};

extern "C" {

int nm_mesh_destroy(nm_mesh_t m) {
    if (!m) return NM_OK;
    if (m->d_verts) (void)hipFree(m->d_verts);
    if (m->d_faces) (void)hipFree(m->d_faces);
    if (m->d_rec) (void)hipFree(m->d_rec);
    if (m->d_ub) (void)hipFree(m->d_ub);
    if (m->d_offsets) (void)hipFree(m->d_offsets);
    if (m->d_list) (void)hipFree(m->d_list);
    delete m;
    return NM_OK;
}

int nm_mesh_create(const float* verts, int V, const int32_t* faces, int F, float reach, nm_mesh_t* out, nm_stream_t stream) {
    NM_REQUIRE(verts && faces && out, "nm_mesh_create: null pointer");
    NM_REQUIRE(V >= 3 && F >= 1 && reach >= 0.f, "nm_mesh_create: bad sizes V=%d F=%d reach=%g", V, F, (double)reach);
    hipStream_t st = nm::as_stream(stream);
    nm_mesh_s* m = new nm_mesh_s();
    memset(m, 0, sizeof(*m));
    m->V = V; m->F = F;
    int rc = NM_OK;
#define NM_TRY(expr, what) if (!rc) rc = nm::check_hip((expr), what)
    NM_TRY(hipMalloc(&m->d_verts, (size_t)V * 12), "nm_mesh_create: hipMalloc(verts)");
    NM_TRY(hipMalloc(&m->d_faces, (size_t)F * 12), "nm_mesh_create: hipMalloc(faces)");
    NM_TRY(hipMalloc(&m->d_rec, (size_t)F * sizeof(TriRec)), "nm_mesh_create: hipMalloc(rec)");
    NM_TRY(hipMemcpyAsync(m->d_verts, verts, (size_t)V * 12, hipMemcpyDeviceToDevice, st), "nm_mesh_create: copy verts");
    NM_TRY(hipMemcpyAsync(m->d_faces, faces, (size_t)F * 12, hipMemcpyDeviceToDevice, st), "nm_mesh_create: copy faces");
    std::vector<float> hv((size_t)V * 3);
    NM_TRY(hipMemcpyAsync(hv.data(), verts, (size_t)V * 12, hipMemcpyDeviceToHost, st), "nm_mesh_create: read verts");
    NM_TRY(hipStreamSynchronize(st), "nm_mesh_create: sync");
    if (rc) { nm_mesh_destroy(m); return rc; }
    // ---- grid geometry on the host: vertex AABB inflated by reach, <= kMaxCells cells
    float lo[3] = {hv[0], hv[1], hv[2]}, hi[3] = {hv[0], hv[1], hv[2]};
    for (int v = 0; v < V; ++v)
        for (int k = 0; k < 3; ++k) {
            const float x = hv[(size_t)v * 3 + k];
            if (!(x == x) || x > 3e38f || x < -3e38f) { nm::set_error("nm_mesh_create: non-finite vertex %d", v); nm_mesh_destroy(m); return NM_ERR_ARG; }
            lo[k] = x < lo[k] ? x : lo[k];
            hi[k] = x > hi[k] ? x : hi[k];
        }
    const float margin = reach * 1.01f + 1e-4f;
    float ext[3];
    for (int k = 0; k < 3; ++k) { lo[k] -= margin; hi[k] += margin; ext[k] = hi[k] - lo[k]; }
    float h = cbrtf(ext[0] * ext[1] * ext[2] / (float)(kMaxCells / 2));
    const float hmin = fmaxf(reach * 0.125f, 1e-4f);
    if (h < hmin) h = hmin;
    Grid& g = m->g;
    for (;;) {
        g.nx = (int)ceilf(ext[0] / h); g.ny = (int)ceilf(ext[1] / h); g.nz = (int)ceilf(ext[2] / h);
        if (g.nx < 1) g.nx = 1;
        if (g.ny < 1) g.ny = 1;
        if (g.nz < 1) g.nz = 1;
        if ((int64_t)g.nx * g.ny * g.nz <= kMaxCells) break;
        h *= 1.1f;
    }
    g.ncells = g.nx * g.ny * g.nz;
    g.lox = lo[0]; g.loy = lo[1]; g.loz = lo[2];
    g.h = h; g.inv_h = 1.f / h;
    g.half_diag = 0.5f * sqrtf(3.f) * h * 1.0001f;
    g.reach = reach;
    g.ring_w = (reach + 3.f * g.half_diag) / (float)kRings + 1e-6f;
    const int nslots = g.ncells * kRings;
    NM_TRY(hipMalloc(&m->d_ub, (size_t)g.ncells * 4), "nm_mesh_create: hipMalloc(ub)");
    NM_TRY(hipMalloc(&m->d_offsets, (size_t)(nslots + 1) * 4), "nm_mesh_create: hipMalloc(offsets)");
    if (rc) { nm_mesh_destroy(m); return rc; }
    const int cb = (g.ncells + 255) / 256;
    hipLaunchKernelGGL(tri_prep_kernel, dim3((F + 255) / 256), dim3(256), 0, st, m->d_verts, m->d_faces, F, m->d_rec);
    hipLaunchKernelGGL(cell_bound_kernel, dim3(cb), dim3(256), 0, st, g, m->d_verts, V, m->d_ub);
    hipLaunchKernelGGL(cell_lists_kernel<false>, dim3(cb), dim3(256), 0, st, g, m->d_rec, F, m->d_ub, m->d_offsets, (int32_t*)nullptr);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, m->d_offsets, nslots, m->d_offsets + nslots);
    rc = nm::check_launch("nm_mesh_create: build kernels");
    int32_t total = 0;
    NM_TRY(hipMemcpyAsync(&total, m->d_offsets + nslots, 4, hipMemcpyDeviceToHost, st), "nm_mesh_create: read total");
    NM_TRY(hipStreamSynchronize(st), "nm_mesh_create: sync");
    m->list_len = total;
    NM_TRY(hipMalloc(&m->d_list, (size_t)(total > 0 ? total : 1) * 4), "nm_mesh_create: hipMalloc(list)");
    if (rc) { nm_mesh_destroy(m); return rc; }
    hipLaunchKernelGGL(cell_lists_kernel<true>, dim3(cb), dim3(256), 0, st, g, m->d_rec, F, m->d_ub, m->d_offsets, m->d_list);
    rc = nm::check_launch("cell_lists_kernel<fill>");
#undef NM_TRY
    if (rc) { nm_mesh_destroy(m); return rc; }
    *out = m;
    return NM_OK;
}

int nm_mesh_info(nm_mesh_t m, int32_t* cells_xyz, int64_t* list_len, float* cell_size) {
    NM_REQUIRE(m, "nm_mesh_info: null handle");
    if (cells_xyz) { cells_xyz[0] = m->g.nx; cells_xyz[1] = m->g.ny; cells_xyz[2] = m->g.nz; }
    if (list_len) *list_len = m->list_len;
    if (cell_size) *cell_size = m->g.h;
    return NM_OK;
}

int nm_warp_to_canonical(nm_mesh_t m, const float* pts, int64_t R, int S, const double* T, float* can_pts, float* can_dirs,
                         float* closest, nm_stream_t stream) {
    NM_REQUIRE(m, "nm_warp_to_canonical: null mesh handle");
    NM_REQUIRE(R == 0 || (pts && T && can_pts && can_dirs), "nm_warp_to_canonical: null pointer");
    NM_REQUIRE(R >= 0 && S >= 2, "nm_warp_to_canonical: bad sizes R=%lld S=%d", (long long)R, S);
    NM_REQUIRE((size_t)S * 24 <= 64 * 1024, "nm_warp_to_canonical: S=%d exceeds the LDS staging budget", S);
    NM_REQUIRE(R < (1ll << 31), "nm_warp_to_canonical: too many rays for one launch");
    if (R == 0) return NM_OK;
    const int threads = S <= 64 ? 64 : (S <= 128 ? 128 : 256);
    hipLaunchKernelGGL(warp_kernel, dim3((unsigned)R), dim3(threads), (size_t)S * 24, nm::as_stream(stream), m->g, pts, S, m->d_verts,
                       m->V, m->d_faces, m->F, m->d_rec, m->d_ub, m->d_offsets, m->d_list, T, can_pts, can_dirs, closest);
    return nm::check_launch("warp_kernel");
}

}  // extern "C"
