// a11 observation -> canonical warp of ray samples for gfx950.
// Replaces reference utils/ray_utils.py:48-66 (warp_samples_to_canonical), whose closest-point query is
// libigl's CPU AABB tree behind a device->host->device round trip per ray batch
// (utils/render_utils.py:218-227).  Here the whole warp stays on the GPU.
//
// nm_mesh_create (once per posed mesh, i.e. per frame and actor) builds an exact search tree on the device (~1 ms for SMPL):
//   * per triangle a 64 B record {a, ab, ac, Gram products, reciprocals of the closest-point test's denominators}, sorted along
//     a 30-bit Morton curve of the centroids (rank sort: F is a body mesh, 13,776 faces for SMPL);
//   * an implicit 4-ary tree over the sorted order: node i of level l covers the sorted triangles
//     [i 4^(L-l), (i+1) 4^(L-l)), level L are the triangles themselves.  A node record holds the AABBs of its four
//     children as six float4 (SoA), so one visit = six 16 B loads and four slab distances.
// nm_warp_to_canonical runs two kernels:
//   search_kernel -- per sample, a depth-first, nearest-child-first search with the stack in LDS.  A subtree is skipped when the
//     distance to its box exceeds the best distance found, inflated by a rounding margin (1e-4 relative + ~80 ulp of the
//     coordinates) so that no triangle whose COMPUTED distance could tie or beat the best one is ever skipped: the result is
//     bit-identical to the all-triangles loop (tested), ties going to the lowest face id whatever the visiting order.
//     What makes it fast is how the wave is scheduled, see the kernel: persistent lanes, and two wave-wide phases (expand a
//     node / test a triangle) of which the one with more ready lanes runs.  The round's first version (a per-cell candidate
//     grid, the exact test inside the divergent candidate loop) took 39.5 ms + 10.6 ms of build for the 7.3 M samples of a
//     512 x 512 frame; this one takes 7.8 ms + 1.0 ms, VALU-bound (86–90 % of the issue slots busy, `profiles/r01_warp_pmc.json`).
//   tail_kernel -- one workgroup per ray: the winning triangle's foot recomputed in f64, its barycentrics, the blended 4x4 (f64, the reference's T is f64),
//     its inverse and the canonical point in f64; the ray's canonical points are staged in LDS so the finite-difference
//     directions (:62-64) need no second pass over HBM.
#include <float.h>
#include <math.h>
#include <string.h>

#include "common.h"

namespace {

constexpr int kMaxLevels = 12;           // 4^12 = 16.7 M triangles

// A triangle as the exact test wants it: a corner, the two edges from it, and every quantity of the Voronoi-region test that
// depends on the triangle alone -- the Gram products and the four reciprocals the test would otherwise compute per query.
struct TriRec {          // 16 floats = 64 B = four 16 B loads
    float ax, ay, az, abx, aby, abz, acx, acy, acz;
    float abab, abac, acac;                    // ab.ab, ab.ac, ac.ac
    float i_abab, i_acac, i_bcbc, i_den;       // 1 / |ab|^2, 1 / |ac|^2, 1 / |bc|^2, 1 / |ab x ac|^2
};

struct Node {            // the four children's boxes, SoA
    float4 lox, loy, loz, hix, hiy, hiz;
};

struct Tree {
    int L;                       // levels of nodes; level L = triangles
    int first_lp;                // first node of level L-1 (the "leaf parents", whose children are triangles)
    int F;
    const float* scale;          // DEVICE: largest |coordinate| of the mesh (bbox_kernel's out[7]): sizes the absolute part of the pruning margin.  On the
                                 // device so that nm_mesh_update can rebuild the tree for moved vertices without a read-back
};
// Nodes are stored in heap order -- level l starts at (4^l - 1) / 3, the children of node g are 4g+1 .. 4g+4 -- so a
// stack entry is one integer and a descent needs no per-level table.  Slots of a level beyond its last node are never
// written or read: the parent's record holds the empty box for them.
__host__ __device__ inline int level_base(int l) { return (int)(((1ll << (2 * l)) - 1) / 3); }

constexpr int kPending = 8;              // a lane keeps walking until it holds this many untested triangles
constexpr int kTriSlots = kPending + 3;  // one more expansion can add four

// ---- build ---------------------------------------------------------------------------------------------------------
// vertex AABB + a non-finite flag: out[0..2] = lo, out[3..5] = hi, out[6] = 1 if any coordinate is NaN/Inf, out[7] = the largest |lo| / |hi|
__global__ __launch_bounds__(1024) void bbox_kernel(const float* __restrict__ verts, int V, float* __restrict__ out) {
    __shared__ float red[16][7];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, bad = 0.f;
    for (int v = threadIdx.x; v < V; v += blockDim.x)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float x = verts[v * 3 + k];
            if (!(fabsf(x) <= FLT_MAX)) bad = 1.f;
            lo[k] = fminf(lo[k], x);
            hi[k] = fmaxf(hi[k], x);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
        }
        bad = fmaxf(bad, __shfl_xor(bad, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 3; ++k) { red[w][k] = lo[k]; red[w][3 + k] = hi[k]; }
        red[w][6] = bad;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        const int k = threadIdx.x;
        float r = red[0][k];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = k < 3 ? fminf(r, red[i][k]) : fmaxf(r, red[i][k]);
        out[k] = r;
        red[0][k] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sc = 0.f;
        for (int k = 0; k < 6; ++k) sc = fmaxf(sc, fabsf(red[0][k]));
        out[7] = sc;
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {      // 10 bits -> every third bit
    x &= 1023u;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

// triangle records in the caller's order + sort keys (Morton code of the centroid << 32 | face id: all distinct)
__global__ __launch_bounds__(256) void tri_prep_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int F,
                                                       const float* __restrict__ bbox, TriRec* __restrict__ rec,
                                                       unsigned long long* __restrict__ keys) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
    const float ax = verts[i0 * 3], ay = verts[i0 * 3 + 1], az = verts[i0 * 3 + 2];
    const float bx = verts[i1 * 3], by = verts[i1 * 3 + 1], bz = verts[i1 * 3 + 2];
    const float cx = verts[i2 * 3], cy = verts[i2 * 3 + 1], cz = verts[i2 * 3 + 2];
    TriRec t;
    t.ax = ax; t.ay = ay; t.az = az;
    t.abx = bx - ax; t.aby = by - ay; t.abz = bz - az;
    t.acx = cx - ax; t.acy = cy - ay; t.acz = cz - az;
    t.abab = fmaf(t.abz, t.abz, fmaf(t.aby, t.aby, t.abx * t.abx));
    t.abac = fmaf(t.abz, t.acz, fmaf(t.aby, t.acy, t.abx * t.acx));
    t.acac = fmaf(t.acz, t.acz, fmaf(t.acy, t.acy, t.acx * t.acx));
    t.i_abab = 1.f / t.abab;
    t.i_acac = 1.f / t.acac;
    t.i_bcbc = 1.f / (t.abab - 2.f * t.abac + t.acac);
    t.i_den = 1.f / (t.abab * t.acac - t.abac * t.abac);
    rec[f] = t;
    uint32_t code = 0;
    const float c[3] = {(ax + bx + cx) * (1.f / 3.f), (ay + by + cy) * (1.f / 3.f), (az + bz + cz) * (1.f / 3.f)};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ext = bbox[3 + k] - bbox[k];
        float u = ext > 0.f ? (c[k] - bbox[k]) / ext * 1024.f : 0.f;
        u = fminf(fmaxf(u, 0.f), 1023.f);
        code |= spread10((uint32_t)u) << k;
    }
    keys[f] = ((unsigned long long)code << 32) | (unsigned)f;
}

// rank sort: position of record f in Morton order = number of smaller keys.  F^2 compares, the keys staged through LDS in chunks and read
// back as broadcasts (a loop of wave-uniform global loads is one exposed scalar-cache round trip per key: 0.8 ms for F = 13,776; this: tens
// of microseconds) -- no multi-pass radix sort for a body mesh.
__global__ __launch_bounds__(256) void rank_scatter_kernel(const unsigned long long* __restrict__ keys, int F, const TriRec* __restrict__ rec,
                                                           TriRec* __restrict__ sorted, int32_t* __restrict__ face_of) {
    constexpr int kChunk = 2048;
    __shared__ unsigned long long sk[kChunk];
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long mine = f < F ? keys[f] : 0ull;
    int rank = 0;
    for (int g0 = 0; g0 < F; g0 += kChunk) {
        const int cnt = F - g0 < kChunk ? F - g0 : kChunk;
        __syncthreads();
        for (int i = threadIdx.x; i < kChunk; i += 256) sk[i] = i < cnt ? keys[g0 + i] : ~0ull;       // (padding: never smaller than a key)
        __syncthreads();
#pragma unroll 16
        for (int g = 0; g < kChunk; ++g) rank += sk[g] < mine ? 1 : 0;
    }
    if (f < F) { sorted[rank] = rec[f]; face_of[rank] = f; }
}

__device__ __forceinline__ void set_lane(float4& v, int c, float x) {
    if (c == 0) v.x = x; else if (c == 1) v.y = x; else if (c == 2) v.z = x; else v.w = x;
}
__device__ __forceinline__ float min4(float4 v) { return fminf(fminf(v.x, v.y), fminf(v.z, v.w)); }
__device__ __forceinline__ float max4(float4 v) { return fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)); }

// level L-1: children are the sorted triangles 4i .. 4i+3 (boxes of the caller's vertices); a missing child gets the empty box
// (lo = +inf, hi = -inf)
__global__ __launch_bounds__(256) void leaf_nodes_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
                                                         const int32_t* __restrict__ face_of, int F, int n, Node* __restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Node nd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int t = 4 * i + c;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (t < F) {
            const int f = face_of[t];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int iv = faces[f * 3 + v];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    lo[k] = fminf(lo[k], verts[iv * 3 + k]);
                    hi[k] = fmaxf(hi[k], verts[iv * 3 + k]);
                }
            }
        }
        set_lane(nd.lox, c, lo[0]); set_lane(nd.loy, c, lo[1]); set_lane(nd.loz, c, lo[2]);
        set_lane(nd.hix, c, hi[0]); set_lane(nd.hiy, c, hi[1]); set_lane(nd.hiz, c, hi[2]);
    }
    nodes[i] = nd;
}

// level l from level l+1: the box of child c is the union of the four boxes child c holds
__global__ __launch_bounds__(256) void upper_nodes_kernel(const Node* __restrict__ below, int n_below, int n, Node* __restrict__ nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Node nd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int j = 4 * i + c;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (j < n_below) {
            const Node b = below[j];
            lo[0] = min4(b.lox); lo[1] = min4(b.loy); lo[2] = min4(b.loz);
            hi[0] = max4(b.hix); hi[1] = max4(b.hiy); hi[2] = max4(b.hiz);
        }
        set_lane(nd.lox, c, lo[0]); set_lane(nd.loy, c, lo[1]); set_lane(nd.loz, c, lo[2]);
        set_lane(nd.hix, c, hi[0]); set_lane(nd.hiy, c, hi[1]); set_lane(nd.hiz, c, hi[2]);
    }
    nodes[i] = nd;
}

// ---- search --------------------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

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
    float d2, sd;   // best squared distance (as computed by the exact test), its square root
    float thr2;     // prune a box when its squared distance exceeds this: (sqrt(d2) 1.0001 + slack)^2, capped at FLT_MAX
    int f;          // caller's face id of the best triangle
    V3 q;
    // RUNNER (the warp's search): the best triangle that did NOT win -- float32 distances decide between two feet only down to ~1e-6 of the
    // distance, and deep inside a body two feet on different faces are that close (every edge's bisector plane is a medial sheet there) while
    // lying centimetres apart: the canonical point jumps with the choice.  tail_kernel settles winner against runner-up in float64.
    float d2b;      // its squared distance (INFINITY: none)
    int fb;         // its face id
};

// Exact closest point of sorted triangle t to p: the Voronoi-region test (Ericson, RTCD 5.1.5) without branches -- 64 lanes
// testing 64 different triangles would otherwise run every region's code one after the other.  With bp = ap - ab and
// cp = ap - ac the six dot products reduce to two (d3 = d1 - ab.ab, ...), and the four denominators -- |ab|^2, |ac|^2,
// |bc|^2 and va + vb + vc = |ab x ac|^2 -- are constants of the triangle, stored as reciprocals.  The seven regions become
// selects applied in reverse priority.  Ties between equidistant triangles go to the lowest face id, so the result does not
// depend on the visiting order; the face id is read only when a triangle ties or wins.
template <bool RUNNER = false>
__device__ __forceinline__ void exact_tri(const TriRec* __restrict__ rec, const int32_t* __restrict__ face_of, int t, V3 p, float slack,
                                          Best& b) {
    const float4 r0 = reinterpret_cast<const float4*>(&rec[t])[0];      // ax ay az abx
    const float4 r1 = reinterpret_cast<const float4*>(&rec[t])[1];      // aby abz acx acy
    const float4 r2 = reinterpret_cast<const float4*>(&rec[t])[2];      // acz abab abac acac
    const float4 r3 = reinterpret_cast<const float4*>(&rec[t])[3];      // i_abab i_acac i_bcbc i_den
    const float apx = p.x - r0.x, apy = p.y - r0.y, apz = p.z - r0.z;
    const float d1 = fmaf(r1.y, apz, fmaf(r1.x, apy, r0.w * apx));      // ab.ap
    const float d2 = fmaf(r2.x, apz, fmaf(r1.w, apy, r1.z * apx));      // ac.ap
    const float d3 = d1 - r2.y, d4 = d2 - r2.z;                         // ab.bp, ac.bp
    const float d5 = d1 - r2.z, d6 = d2 - r2.w;                         // ab.cp, ac.cp
    const float vc = fmaf(d1, d4, -(d3 * d2)), vb = fmaf(d5, d2, -(d1 * d6)), va = fmaf(d3, d6, -(d5 * d4));
    const float e43 = d4 - d3, e56 = d5 - d6;
    const float t_ab = d1 * r3.x, t_ac = d2 * r3.y, t_bc = e43 * r3.z;
    float v = vb * r3.w, w = vc * r3.w;                                              // interior; q = a + v*ab + w*ac
    const bool r_bc = va <= 0.f && e43 >= 0.f && e56 >= 0.f;                         // edge BC
    v = r_bc ? 1.f - t_bc : v;  w = r_bc ? t_bc : w;
    const bool r_ac = vb <= 0.f && d2 >= 0.f && d6 <= 0.f;                           // edge AC
    v = r_ac ? 0.f : v;         w = r_ac ? t_ac : w;
    const bool r_c = d6 >= 0.f && d5 <= d6;                                          // vertex C
    v = r_c ? 0.f : v;          w = r_c ? 1.f : w;
    const bool r_ab = vc <= 0.f && d1 >= 0.f && d3 <= 0.f;                           // edge AB
    v = r_ab ? t_ab : v;        w = r_ab ? 0.f : w;
    const bool r_b = d3 >= 0.f && d4 <= d3;                                          // vertex B
    v = r_b ? 1.f : v;          w = r_b ? 0.f : w;
    const bool r_a = d1 <= 0.f && d2 <= 0.f;                                         // vertex A
    v = r_a ? 0.f : v;          w = r_a ? 0.f : w;
    const V3 c = {fmaf(r1.z, w, fmaf(r0.w, v, r0.x)), fmaf(r1.w, w, fmaf(r1.x, v, r0.y)), fmaf(r2.x, w, fmaf(r1.y, v, r0.z))};
    const float dx = c.x - p.x, dy = c.y - p.y, dz = c.z - p.z;
    const float dd = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    if (dd <= b.d2) {
        const int f = face_of[t];
        if (dd < b.d2 || f < b.f) {
            if (RUNNER) { b.d2b = b.d2; b.fb = b.f; }                      // the dethroned winner is the best loser so far
            b.d2 = dd; b.sd = sqrtf(dd); b.f = f; b.q = c;
            const float thr = b.sd * 1.0001f + slack;
            b.thr2 = fminf(thr * thr, FLT_MAX);
        } else if (RUNNER && (dd < b.d2b || f < b.fb)) {                   // (an exact tie with the winner, higher face id: dd == d2 <= d2b)
            b.d2b = dd; b.fb = f;
        }
    } else if (RUNNER && dd <= b.d2b) {
        const int f = face_of[t];
        if (dd < b.d2b || f < b.fb) { b.d2b = dd; b.fb = f; }              // ties among losers: the lowest face id, whatever the visiting order
    }
}

// the all-triangles loop (search mode NM_SEARCH_ALL, and the fallback for points the tree search cannot order)
template <bool RUNNER = false>
__device__ __forceinline__ void search_all(const TriRec* __restrict__ rec, const int32_t* __restrict__ face_of, int F, V3 p, float slack,
                                           Best& b) {
    for (int t = 0; t < F; ++t) exact_tri<RUNNER>(rec, face_of, t, p, slack, b);
}

__device__ __forceinline__ float slab(float lo, float hi, float x) { return fmaxf(fmaxf(lo - x, x - hi), 0.f); }
// child order: the distance^2 (>= 0, so its bit pattern orders like the value) with the child slot in the two low mantissa
// bits -- cleared bits round the distance down, which keeps it a lower bound -- sorts with one min and one max per exchange
__device__ __forceinline__ uint32_t child_key(float k, uint32_t c) { return (__float_as_uint(k) & ~3u) | c; }
__device__ __forceinline__ void cswap(uint32_t& a, uint32_t& b) {                   // afterwards a >= b
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    a = hi; b = lo;
}

// ---- search_kernel: closest triangle and closest point per sample --------------------------------------------------------------
// One wave per workgroup; a wave owns kChunk consecutive samples and its lanes are persistent: a lane that finishes its
// sample takes the next one of the chunk, so the wave stays full although per-sample work varies 3x (a sample 20 cm off
// the body has ~10x the contenders of one at the surface).
// Per-lane LDS: node stack, entries {box distance^2, node}, 3 (L - 1) deep -- the nearest child of an expanded node stays
// in registers --, and a list of pending triangles (kTriSlots).  Entry d of lane t is at d * 64 + t.
// Every iteration runs ONE of two wave-wide phases, whichever has more lanes ready for it:
//   WALK: each lane with nodes left and fewer than kPending pending triangles expands one node (four child boxes, kept
//         children pushed farthest first);
//   TEST: each lane with a pending triangle runs the exact closest-point test on the nearest one.
// Lanes walk ahead of their tests ("speculative traversal"), so pending triangles are tested against a slightly stale
// bound; that costs a few extra tests and cannot change the result (any superset of the contenders gives the same
// minimum, ties going to the lowest face id).
// Output: q -> q_out[i*3 ..], face -> f_out[i * f_stride] (the warp passes can_pts / can_dirs with stride 3: tail_kernel consumes
// and overwrites them; nm_signed_distance passes its own arrays with stride 1).
// The three constants below were swept on the posed-frame workload (pending 4 / 6 / 8 / 12: 8.2 / 8.0 / 7.8 / 7.8 ms; refill
// threshold 3 / 8: 8.2 / 7.8 ms).
constexpr int kChunk = 512;
// samples per wave of a launch: kChunk when there are enough samples to fill the chip that way (a frame's warp: millions), fewer for the
// small batches of a training iteration (92 k samples = 180 waves of 512 on 256 CUs: 2.5 ms per search, latency-bound; 2048 waves of 64
// take a tenth of that).  A sample's result does not depend on the chunking.  Round 6 (tools/search_small_batch.py): such a launch lasts as
// long as its slowest sample's chain of dependent loads (92 k ray samples: 0.95 ms at 16, 32, 64 or 128 samples per wave alike), so a lane should
// not queue a second sample behind its first while SIMDs idle: one sample per lane up to 262 k samples (184 k points: 0.97 ms against 1.27 at 128).
inline int chunk_for(int64_t N) {
    if (const char* e = getenv("NEUMAN_SEARCH_CHUNK")) { const int v = atoi(e); if (v >= 8 && v <= kChunk) return v; }     // (tools: sweeps)
    int64_t c = (N / 4096 + 63) / 64 * 64;
    return (int)(c < 64 ? 64 : c > kChunk ? kChunk : c);
}
constexpr int kRefill = 8;                            // idle lanes that trigger a refill (or any, when no lane has work)

template <bool SMALL, bool RUNNER = false>      // RUNNER: f_out[i * f_stride + 1] = face id of the best loser within the pruning bound (else -1); f_stride >= 2
__global__ __launch_bounds__(64) void search_kernel(Tree tr, int search_all_mode, const float* __restrict__ pts, int64_t N,
                                                                 const TriRec* __restrict__ rec, const int32_t* __restrict__ face_of,
                                                                 const Node* __restrict__ nodes, float* __restrict__ q_out,
                                                                 int32_t* __restrict__ f_out, int f_stride, int chunk) {
    extern __shared__ double lds_raw[];
    const int depth = 3 * (tr.L - 1);
    // A mesh whose vertices went non-finite AFTER creation (nm_mesh_update never reads back: a training iteration must not stall; the flag is
    // bbox_kernel's out[6], next to the scale): boxes and margins built from NaN order nothing, so the tree is not descended at all -- every sample
    // takes the all-triangles loop, whose answer is defined (NaN distances never win: face 0, q = p) and whose NaN reaches the trainer's loss guard.
    if (tr.scale[-1] != 0.f) search_all_mode = 1;
    // node stack entry: {distance^2, node} as float2, or -- SMALL: at most 65,536 nodes and triangles -- one dword holding the
    // distance^2 truncated to its upper 16 bits (rounded towards zero: still a lower bound) above the node index, and
    // uint16 pending triangles.  LDS per lane decides how many waves share a CU, and the walk is latency-bound.
    float2* nstack = reinterpret_cast<float2*>(lds_raw);
    uint32_t* nstack_s = reinterpret_cast<uint32_t*>(lds_raw);
    int* tlist = reinterpret_cast<int*>(nstack + (size_t)(depth + 1) * 64);                          // + 1: the spare slot
    uint16_t* tlist_s = reinterpret_cast<uint16_t*>(nstack_s + (size_t)(depth + 1) * 64);
    auto push_tri = [&](int slot, int t) {
        if (SMALL) tlist_s[slot * 64 + threadIdx.x] = (uint16_t)t;
        else tlist[slot * 64 + threadIdx.x] = t;
    };
    auto pop_tri = [&](int slot) -> int { return SMALL ? (int)tlist_s[slot * 64 + threadIdx.x] : tlist[slot * 64 + threadIdx.x]; };
    auto push_node = [&](int slot, float k, int id) {
        if (SMALL) nstack_s[slot * 64 + threadIdx.x] = (__float_as_uint(k) & 0xffff0000u) | (uint32_t)id;
        else nstack[slot * 64 + threadIdx.x] = make_float2(k, __int_as_float(id));
    };
    auto pop_node = [&](int slot, float& k, int& id) {
        if (SMALL) {
            const uint32_t e = nstack_s[slot * 64 + threadIdx.x];
            k = __uint_as_float(e & 0xffff0000u); id = (int)(e & 0xffffu);
        } else {
            const float2 e = nstack[slot * 64 + threadIdx.x];
            k = e.x; id = __float_as_int(e.y);
        }
    };
    int64_t next = (int64_t)blockIdx.x * chunk;                                  // wave-uniform: first sample not handed out
    const int64_t end = next + chunk < N ? next + chunk : N;
    bool active = false;
    int64_t i = 0;
    V3 p = {0.f, 0.f, 0.f};
    float slack = 0.f;
    const float mesh_scale = *tr.scale;
    Best b;
    b.d2 = INFINITY; b.sd = INFINITY; b.thr2 = FLT_MAX; b.f = 0x7fffffff; b.q = p; b.d2b = INFINITY; b.fb = 0x7fffffff;
    bool has_cur = false;
    float ck = 0.f;
    int cid = 0, nsp = 0, ntri = 0;
    for (;;) {
        bool can_walk = active && (has_cur || nsp > 0) && ntri < kPending;
        unsigned long long bw = __ballot(can_walk), bt = __ballot(active && ntri > 0);
        const unsigned long long bi = __ballot(!active);
        if (next < end && (__popcll(bi) >= kRefill || !(bw | bt))) {               // ---- hand out samples to the idle lanes
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(bi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bi, 0));
            if (!active && next + rank < end) {
                i = next + rank;
                p = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
                const float pmax = fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z)));
                slack = 1e-5f * (1.f + fmaxf(pmax, mesh_scale));
                b.d2 = INFINITY; b.sd = INFINITY; b.thr2 = FLT_MAX; b.f = 0x7fffffff; b.q = p; b.d2b = INFINITY; b.fb = 0x7fffffff;
                has_cur = pmax <= FLT_MAX && !search_all_mode;                   // NaN / Inf, all-triangles mode: straight to the loop
                ck = 0.f; cid = 0; nsp = 0; ntri = 0;                            // the root
                active = true;
            }
            next += __popcll(bi);
            can_walk = active && (has_cur || nsp > 0) && ntri < kPending;
            bw = __ballot(can_walk); bt = __ballot(active && ntri > 0);
        }
        if (__popcll(bt) < __popcll(bw)) {                                       // ---- WALK: expand one node per lane
            if (can_walk) {
                while (!has_cur && nsp > 0) {                                    // next entry the bound has not overtaken
                    float ek; int eid;
                    pop_node(--nsp, ek, eid);
                    if (ek <= b.thr2) { ck = ek; cid = eid; has_cur = true; }
                }
                if (has_cur && ck > b.thr2) has_cur = false;
                if (has_cur) {
                    const Node* nd = nodes + cid;
                    const float4 lox = nd->lox, loy = nd->loy, loz = nd->loz, hix = nd->hix, hiy = nd->hiy, hiz = nd->hiz;
                    const float x0 = slab(lox.x, hix.x, p.x), y0 = slab(loy.x, hiy.x, p.y), z0 = slab(loz.x, hiz.x, p.z);
                    const float x1 = slab(lox.y, hix.y, p.x), y1 = slab(loy.y, hiy.y, p.y), z1 = slab(loz.y, hiz.y, p.z);
                    const float x2 = slab(lox.z, hix.z, p.x), y2 = slab(loy.z, hiy.z, p.y), z2 = slab(loz.z, hiz.z, p.z);
                    const float x3 = slab(lox.w, hix.w, p.x), y3 = slab(loy.w, hiy.w, p.y), z3 = slab(loz.w, hiz.w, p.z);
                    uint32_t k0 = child_key(fmaf(z0, z0, fmaf(y0, y0, x0 * x0)), 0), k1 = child_key(fmaf(z1, z1, fmaf(y1, y1, x1 * x1)), 1);
                    uint32_t k2 = child_key(fmaf(z2, z2, fmaf(y2, y2, x2 * x2)), 2), k3 = child_key(fmaf(z3, z3, fmaf(y3, y3, x3 * x3)), 3);
                    cswap(k0, k1); cswap(k2, k3);                                // sort descending: k0 >= k1 >= k2 >= k3
                    cswap(k0, k2); cswap(k1, k3);
                    cswap(k1, k2);
                    // keep a child when its key is <= the bound's (4 ulp of slop on the keeping side); an empty child's
                    // distance is +inf, above any bound (thr2 <= FLT_MAX): never kept
                    const uint32_t tb = __float_as_uint(b.thr2) | 3u;
                    const bool lp = cid >= tr.first_lp;                          // children are triangles
                    const int child = lp ? 4 * (cid - tr.first_lp) : 4 * cid + 1;
                    // The keys are sorted, so the kept children are the last `kept` of the four.  They are stored farthest first
                    // (the stack and the list are popped from their ends) without branches: child j goes to slot
                    // top + j - (4 - kept), and a child that is not kept goes to the lane's spare slot instead.
                    const int kept = (k0 <= tb) + (k1 <= tb) + (k2 <= tb) + (k3 <= tb);
                    if (lp) {
                        const int base = ntri - (4 - kept);
                        push_tri(k0 <= tb ? base : kTriSlots, child + (int)(k0 & 3u));
                        push_tri(k1 <= tb ? base + 1 : kTriSlots, child + (int)(k1 & 3u));
                        push_tri(k2 <= tb ? base + 2 : kTriSlots, child + (int)(k2 & 3u));
                        push_tri(k3 <= tb ? base + 3 : kTriSlots, child + (int)(k3 & 3u));
                        ntri += kept;
                        has_cur = false;
                    } else {
                        const int base = nsp - (4 - kept);
                        push_node(k0 <= tb ? base : depth, __uint_as_float(k0 & ~3u), child + (int)(k0 & 3u));
                        push_node(k1 <= tb ? base + 1 : depth, __uint_as_float(k1 & ~3u), child + (int)(k1 & 3u));
                        push_node(k2 <= tb ? base + 2 : depth, __uint_as_float(k2 & ~3u), child + (int)(k2 & 3u));
                        nsp += kept > 0 ? kept - 1 : 0;
                        has_cur = kept > 0;                                      // the nearest child is expanded next, from registers
                        ck = __uint_as_float(k3 & ~3u); cid = child + (int)(k3 & 3u);
                    }
                }
            }
        } else if (bt) {                                                         // ---- TEST: one pending triangle per lane
            if (active && ntri > 0) exact_tri<RUNNER>(rec, face_of, pop_tri(--ntri), p, slack, b);
        }
        if (active && !has_cur && nsp == 0 && ntri == 0) {                       // ---- this lane's sample is done
            if (b.f == 0x7fffffff) {
                // all-triangles mode, or nothing found (non-finite point, overflowing distances): the plain loop, whose
                // answer for such points is face 0 and q = p
                b.d2 = INFINITY; b.sd = INFINITY; b.thr2 = FLT_MAX; b.f = 0; b.q = p; b.d2b = INFINITY; b.fb = 0x7fffffff;
                search_all<RUNNER>(rec, face_of, tr.F, p, slack, b);
            }
            q_out[i * 3] = b.q.x; q_out[i * 3 + 1] = b.q.y; q_out[i * 3 + 2] = b.q.z;
            f_out[i * f_stride] = b.f;
            // every triangle within the final pruning bound has been tested (a box is only dropped beyond the bound of its time, and bounds only shrink):
            // the runner-up reported is the same triangle whatever the visiting order, tree or all-triangles mode
            if (RUNNER) f_out[i * f_stride + 1] = (b.d2b <= b.thr2 && b.fb != 0x7fffffff) ? b.fb : -1;
            active = false;
        }
        if (next >= end && !__any(active)) break;
    }
}

// closest point of triangle `face` to p in float64 (Ericson 5.1.5, with branches: one lane, one or two triangles) -> out = {x, y, z, squared distance};
// false when the result is not finite
__device__ __forceinline__ bool foot64(const float* __restrict__ verts, const int32_t* __restrict__ faces, int face, V3 p, double* out) {
    const int i0 = faces[face * 3], i1 = faces[face * 3 + 1], i2 = faces[face * 3 + 2];
    const double ax = verts[i0 * 3], ay = verts[i0 * 3 + 1], az = verts[i0 * 3 + 2];
    const double v0x = (double)verts[i1 * 3] - ax, v0y = (double)verts[i1 * 3 + 1] - ay, v0z = (double)verts[i1 * 3 + 2] - az;
    const double v1x = (double)verts[i2 * 3] - ax, v1y = (double)verts[i2 * 3 + 1] - ay, v1z = (double)verts[i2 * 3 + 2] - az;
    const double px_ = p.x, py_ = p.y, pz_ = p.z;
    const double apx = px_ - ax, apy = py_ - ay, apz = pz_ - az;
    const double e1 = v0x * apx + v0y * apy + v0z * apz, e2 = v1x * apx + v1y * apy + v1z * apz;      // d1, d2
    const double bpx = apx - v0x, bpy = apy - v0y, bpz = apz - v0z;
    const double e3 = v0x * bpx + v0y * bpy + v0z * bpz, e4 = v1x * bpx + v1y * bpy + v1z * bpz;      // d3, d4
    const double cpx = apx - v1x, cpy = apy - v1y, cpz = apz - v1z;
    const double e5 = v0x * cpx + v0y * cpy + v0z * cpz, e6 = v1x * cpx + v1y * cpy + v1z * cpz;      // d5, d6
    const double vc = e1 * e4 - e3 * e2, vb = e5 * e2 - e1 * e6, va = e3 * e6 - e5 * e4;
    double fv, fw;
    if (e1 <= 0 && e2 <= 0) { fv = 0; fw = 0; }                                       // vertex A
    else if (e3 >= 0 && e4 <= e3) { fv = 1; fw = 0; }                                 // vertex B
    else if (vc <= 0 && e1 >= 0 && e3 <= 0) { fv = e1 / (e1 - e3); fw = 0; }          // edge AB
    else if (e6 >= 0 && e5 <= e6) { fv = 0; fw = 1; }                                 // vertex C
    else if (vb <= 0 && e2 >= 0 && e6 <= 0) { fv = 0; fw = e2 / (e2 - e6); }          // edge AC
    else if (va <= 0 && (e4 - e3) >= 0 && (e5 - e6) >= 0) { fw = (e4 - e3) / ((e4 - e3) + (e5 - e6)); fv = 1 - fw; }   // edge BC
    else { const double dn = 1.0 / (va + vb + vc); fv = vb * dn; fw = vc * dn; }      // interior
    const double tx = ax + v0x * fv + v1x * fw, ty = ay + v0y * fv + v1y * fw, tz = az + v0z * fv + v1z * fw;
    out[0] = tx; out[1] = ty; out[2] = tz;
    out[3] = (tx - px_) * (tx - px_) + (ty - py_) * (ty - py_) + (tz - pz_) * (tz - pz_);
    return tx == tx && ty == ty && tz == tz && fabs(tx) <= DBL_MAX && fabs(ty) <= DBL_MAX && fabs(tz) <= DBL_MAX;
}

// ---- tail_kernel: one workgroup per ray, one lane per sample ---------------------------------------------------------------------
// reads the search result from can_pts (q) and can_dirs (face id) and overwrites both with the outputs
__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ pts, int S, const float* __restrict__ verts,
                                                   const int32_t* __restrict__ faces, const double* __restrict__ T,
                                                   float* __restrict__ can_pts, float* __restrict__ can_dirs, float* __restrict__ closest) {
    extern __shared__ double can_lds[];                     // [S][3]
    const int64_t r = blockIdx.x;
    for (int s0 = 0; s0 < S; s0 += blockDim.x) {
        const int s = s0 + threadIdx.x;
        const bool live = s < S;
        if (!live) continue;
        const int64_t i = r * S + s;
        const V3 p = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
        int bf = reinterpret_cast<const int32_t*>(can_dirs)[i * 3];
        const int rf = reinterpret_cast<const int32_t*>(can_dirs)[i * 3 + 1];       // the search's runner-up (-1: none within its bound)
        const V3 q = {can_pts[i * 3], can_pts[i * 3 + 1], can_pts[i * 3 + 2]};
        // The search's distances are float32 arithmetic (like libigl's on float32 input): two feet whose distances differ by less than ~1e-6 of the
        // distance are decided by rounding, and inside a body such feet lie on different faces centimetres apart.  Winner and runner-up are both
        // re-evaluated here in float64 (the same Voronoi-region test, Ericson 5.1.5) and the nearer one -- ties: the lower face id -- is the foot, as in
        // a float64 evaluation of utils/ray_utils.py:53; its closest point is the float64 one either way: canonical points to ~1e-7, finite-difference
        // directions to ~1e-5 of that evaluation.  A non-finite query keeps the search's q.
        double qx = q.x, qy = q.y, qz = q.z;
        {
            double f0[4], f1[4];
            const bool ok0 = foot64(verts, faces, bf, p, f0);
            if (rf >= 0 && ok0 && foot64(verts, faces, rf, p, f1) && (f1[3] < f0[3] || (f1[3] == f0[3] && rf < bf))) {
                bf = rf;
                qx = f1[0]; qy = f1[1]; qz = f1[2];
            } else if (ok0) {
                qx = f0[0]; qy = f0[1]; qz = f0[2];
            }
        }
        // ---- barycentrics of q in the winning triangle, igl.barycentric_coordinates_tri (ray_utils.py:55), f64
        const int i0 = faces[bf * 3], i1 = faces[bf * 3 + 1], i2 = faces[bf * 3 + 2];
        const double ax = verts[i0 * 3], ay = verts[i0 * 3 + 1], az = verts[i0 * 3 + 2];
        const double v0x = (double)verts[i1 * 3] - ax, v0y = (double)verts[i1 * 3 + 1] - ay, v0z = (double)verts[i1 * 3 + 2] - az;
        const double v1x = (double)verts[i2 * 3] - ax, v1y = (double)verts[i2 * 3 + 1] - ay, v1z = (double)verts[i2 * 3 + 2] - az;
        const double d00 = v0x * v0x + v0y * v0y + v0z * v0z, d01 = v0x * v1x + v0y * v1y + v0z * v1z;
        const double d11 = v1x * v1x + v1y * v1y + v1z * v1z;
        const double v2x = qx - ax, v2y = qy - ay, v2z = qz - az;
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
            if (closest) { closest[i * 3] = (float)qx; closest[i * 3 + 1] = (float)qy; closest[i * 3 + 2] = (float)qz; }
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

// ---- signed distance: the sign of the closest-point query by angle-weighted pseudonormals (Baerentzen & Aanaes 2005), what
// igl.signed_distance does for a triangle mesh (reference utils/ray_utils.py:70, trainers/human_nerf_trainer.py:310, 326).
// Per face (caller's id) 21 floats: unit face normal | pseudonormal of edge 0 (v0-v1), 1 (v1-v2), 2 (v2-v0) = sum of the unit
// normals of the faces sharing it | angle-weighted pseudonormal of v0, v1, v2.  Built on the first query (seven small kernels,
// sums in face order: deterministic), not by nm_mesh_create: the renderers never ask for a sign.
constexpr int kPnFloats = 21;

__global__ __launch_bounds__(256) void face_normal_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int F, float* __restrict__ fn) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float* a = verts + faces[f * 3] * 3;
    const float* b = verts + faces[f * 3 + 1] * 3;
    const float* c = verts + faces[f * 3 + 2] * 3;
    const float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2], vx = c[0] - a[0], vy = c[1] - a[1], vz = c[2] - a[2];
    const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const float inv = 1.f / fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-30f);
    fn[f * 3] = nx * inv; fn[f * 3 + 1] = ny * inv; fn[f * 3 + 2] = nz * inv;
}

// ---- angle-weighted vertex pseudonormals and edge pseudonormals (the sum of the normals of the faces on the edge) through a
// vertex -> incident-faces table (CSR, each list in ascending face order, so every sum runs in face order: deterministic).
// O(F); loops over all faces per vertex / per edge took 3 ms per SMPL mesh -- paid once per training iteration, because the
// posed body changes with the pose parameters.
__global__ __launch_bounds__(256) void corner_count_kernel(const int32_t* __restrict__ faces, int F, int V, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * F) return;
    const int v = faces[i];
    if (v >= 0 && v < V) atomicAdd(cnt + v, 1);
}
__global__ __launch_bounds__(1024) void adj_offsets_kernel(const int* __restrict__ cnt, int V, int* __restrict__ off) {   // one workgroup
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (V + 1023) / 1024;
    const int lo = t * per < V ? t * per : V, hi = lo + per < V ? lo + per : V;
    int s = 0;
    for (int v = lo; v < hi; ++v) s += cnt[v];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int k = 0; k < 1024; ++k) { const int c = part[k]; part[k] = run; run += c; }
        off[V] = run;
    }
    __syncthreads();
    int run = part[t];
    for (int v = lo; v < hi; ++v) { off[v] = run; run += cnt[v]; }
}
__global__ __launch_bounds__(256) void corner_fill_kernel(const int32_t* __restrict__ faces, int F, int V, const int* __restrict__ off,
                                                          int* __restrict__ cur, int* __restrict__ adj) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * F) return;
    const int v = faces[i];
    if (v >= 0 && v < V) adj[off[v] + atomicAdd(cur + v, 1)] = i / 3;
}
__global__ __launch_bounds__(256) void adj_sort_kernel(const int* __restrict__ off, int V, int* __restrict__ adj) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int lo = off[v], hi = off[v + 1];
    for (int i = lo + 1; i < hi; ++i) {                              // insertion sort: a vertex has a handful of faces
        const int x = adj[i];
        int j = i - 1;
        while (j >= lo && adj[j] > x) { adj[j + 1] = adj[j]; --j; }
        adj[j + 1] = x;
    }
}
__global__ __launch_bounds__(256) void vertex_normal_adj_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, int V,
                                                                const int* __restrict__ off, const int* __restrict__ adj,
                                                                const float* __restrict__ fn, float* __restrict__ vn) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float s[3] = {0.f, 0.f, 0.f};
    int last = -1;
    for (int q = off[v]; q < off[v + 1]; ++q) {
        const int f = adj[q];
        if (f == last) continue;                                     // a degenerate face naming the vertex twice is one term, as above
        last = f;
        const int i0 = faces[f * 3], i1 = faces[f * 3 + 1], i2 = faces[f * 3 + 2];
        const int p1 = i0 == v ? i1 : (i1 == v ? i2 : i0), p2 = i0 == v ? i2 : (i1 == v ? i0 : i1);
        float e1[3], e2[3], l1 = 0.f, l2 = 0.f, d = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            e1[k] = verts[p1 * 3 + k] - verts[v * 3 + k];
            e2[k] = verts[p2 * 3 + k] - verts[v * 3 + k];
            l1 += e1[k] * e1[k]; l2 += e2[k] * e2[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) d += e1[k] * e2[k];
        const float ang = acosf(fminf(fmaxf(d / fmaxf(sqrtf(l1 * l2), 1e-30f), -1.f), 1.f));
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += ang * fn[f * 3 + k];
    }
    vn[v * 3] = s[0]; vn[v * 3 + 1] = s[1]; vn[v * 3 + 2] = s[2];
}
__global__ __launch_bounds__(256) void pseudonormal_pack_adj_kernel(const int32_t* __restrict__ faces, int F, int V, const int* __restrict__ off,
                                                                    const int* __restrict__ adj, const float* __restrict__ fn,
                                                                    const float* __restrict__ vn, float* __restrict__ pn) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;              // (face, edge)
    if (i >= 3 * F) return;
    const int f = i / 3, e = i - 3 * f;
    const int a = faces[f * 3 + e], b = faces[f * 3 + (e + 1) % 3];
    float s[3] = {0.f, 0.f, 0.f};
    int last = -1;
    if (a >= 0 && a < V)
        for (int q = off[a]; q < off[a + 1]; ++q) {                  // faces on corner a, ascending; those that also hold b are on the edge
            const int g = adj[q];
            if (g == last) continue;
            last = g;
            const int j0 = faces[g * 3], j1 = faces[g * 3 + 1], j2 = faces[g * 3 + 2];
            if (j0 == b || j1 == b || j2 == b) {
#pragma unroll
                for (int k = 0; k < 3; ++k) s[k] += fn[g * 3 + k];
            }
        }
    float* o = pn + (int64_t)f * kPnFloats;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[3 + 3 * e + k] = s[k];
        o[12 + 3 * e + k] = vn[a * 3 + k];
        if (e == 0) o[k] = fn[f * 3 + k];
    }
}

// sign for N solved queries: the Voronoi region of the query in its winning triangle picks the pseudonormal
__global__ __launch_bounds__(256) void signed_distance_kernel(const float* __restrict__ pts, int64_t N, const float* __restrict__ verts,
                                                              const int32_t* __restrict__ faces, const float* __restrict__ pn,
                                                              const int32_t* __restrict__ face, const float* __restrict__ closest,
                                                              float* __restrict__ sdist) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int f = face[i];
    const V3 p = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]}, q = {closest[i * 3], closest[i * 3 + 1], closest[i * 3 + 2]};
    const float* va = verts + faces[f * 3] * 3;
    const float* vb = verts + faces[f * 3 + 1] * 3;
    const float* vc = verts + faces[f * 3 + 2] * 3;
    const V3 ab = {vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]}, ac = {vc[0] - va[0], vc[1] - va[1], vc[2] - va[2]};
    const V3 ap = {p.x - va[0], p.y - va[1], p.z - va[2]};
    const float abab = fmaf(ab.z, ab.z, fmaf(ab.y, ab.y, ab.x * ab.x)), abac = fmaf(ab.z, ac.z, fmaf(ab.y, ac.y, ab.x * ac.x));
    const float acac = fmaf(ac.z, ac.z, fmaf(ac.y, ac.y, ac.x * ac.x));
    const float d1 = fmaf(ab.z, ap.z, fmaf(ab.y, ap.y, ab.x * ap.x)), d2 = fmaf(ac.z, ap.z, fmaf(ac.y, ap.y, ac.x * ap.x));
    const float d3 = d1 - abab, d4 = d2 - abac, d5 = d1 - abac, d6 = d2 - acac;
    const float vcc = fmaf(d1, d4, -(d3 * d2)), vbb = fmaf(d5, d2, -(d1 * d6)), vaa = fmaf(d3, d6, -(d5 * d4));
    // same regions, same priority as exact_tri: 0 face, 1 / 2 / 3 edge v0v1 / v1v2 / v2v0, 4 / 5 / 6 vertex v0 / v1 / v2
    int region = 0;
    if (vaa <= 0.f && d4 - d3 >= 0.f && d5 - d6 >= 0.f) region = 2;
    if (vbb <= 0.f && d2 >= 0.f && d6 <= 0.f) region = 3;
    if (d6 >= 0.f && d5 <= d6) region = 6;
    if (vcc <= 0.f && d1 >= 0.f && d3 <= 0.f) region = 1;
    if (d3 >= 0.f && d4 <= d3) region = 5;
    if (d1 <= 0.f && d2 <= 0.f) region = 4;
    const float* n = pn + (int64_t)f * kPnFloats + (region == 0 ? 0 : (region <= 3 ? 3 + 3 * (region - 1) : 12 + 3 * (region - 4)));
    const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
    const float dist = sqrtf(dx * dx + dy * dy + dz * dz);
    sdist[i] = (dx * n[0] + dy * n[1] + dz * n[2]) < 0.f ? -dist : dist;
}

}  // namespace

struct nm_mesh_s {
    int V, F, search;
    bool force_wide;     // tests: use the two-dword stack entries even for a small tree
    Tree tr;
    int n_nodes;
    float* d_verts;      // owned copies: the handle outlives the caller's tensors
    int32_t* d_faces;
    TriRec* d_rec;       // Morton order
    int32_t* d_face;     // caller's face id of sorted triangle t
    Node* d_nodes;
    float* d_pn;         // pseudonormals [F][21], built by the first nm_signed_distance
    void* d_pn_scratch;  // face / vertex normals and the vertex -> faces table of that build (freed with the handle: no sync on the way)
    bool pn_stale;       // nm_mesh_update moved the vertices: the normals (not the vertex -> faces table) are rebuilt by the next nm_signed_distance
    TriRec* d_tmp;       // the build's scratch, kept for nm_mesh_update: unsorted records, sort keys,
    unsigned long long* d_keys;
    float* d_bbox;       //   and the vertex box [8] (Tree::scale points at its last entry)
    int n_level[kMaxLevels + 1];
};

// =====================================================================================================================================
// The differentiable warp of the human trainer (reference utils/ray_utils.py:85-93 + trainers/human_nerf_trainer.py:262-266): per sample
// T_interp = sum_k bary_k T[tri_k] (blend of the closest triangle's vertex transforms), canonical point = inv(T_interp) [p; 1].  Forward
// and backward as ONE kernel each instead of a gather, a product, a sum, a batched LU inverse and a batched 4x4 product under autograd
// (and their adjoints, among them the sort-based index_put of the gather).  Gradients: to the vertex transforms T (float atomics: the
// order in which the samples of a vertex arrive is not fixed) and to the barycentric coordinates (which torch differentiates on to the
// posed vertices).  The points carry no gradient (the trainer detaches them, :262).
namespace {

__device__ __forceinline__ void inv4x4_all(const double* m, double* o) {
    const double s0 = m[0] * m[5] - m[4] * m[1], s1 = m[0] * m[6] - m[4] * m[2], s2 = m[0] * m[7] - m[4] * m[3];
    const double s3 = m[1] * m[6] - m[5] * m[2], s4 = m[1] * m[7] - m[5] * m[3], s5 = m[2] * m[7] - m[6] * m[3];
    const double c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    const double c2 = m[8] * m[15] - m[12] * m[11], c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    const double inv = 1.0 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    o[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * inv;    o[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * inv;
    o[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * inv; o[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * inv;
    o[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * inv;   o[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * inv;
    o[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * inv; o[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * inv;
    o[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * inv;    o[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * inv;
    o[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * inv; o[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * inv;
    o[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * inv;  o[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * inv;
    o[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * inv; o[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * inv;
}

// blended transform of sample i and its inverse (float32 values; the inverse through float64 cofactors)
__device__ __forceinline__ void blend_inverse(const float* __restrict__ T, const int32_t* __restrict__ tri, const float* __restrict__ bary, int64_t i,
                                              float* inv) {
    const float b0 = bary[i * 3], b1 = bary[i * 3 + 1], b2 = bary[i * 3 + 2];
    const float* T0 = T + (int64_t)tri[i * 3] * 16;
    const float* T1 = T + (int64_t)tri[i * 3 + 1] * 16;
    const float* T2 = T + (int64_t)tri[i * 3 + 2] * 16;
    double M[16], Mi[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) M[q] = (double)(T0[q] * b0 + T1[q] * b1 + T2[q] * b2);
    inv4x4_all(M, Mi);
#pragma unroll
    for (int q = 0; q < 16; ++q) inv[q] = (float)Mi[q];
}

__global__ __launch_bounds__(256) void warp_apply_forward_kernel(const float* __restrict__ T, const int32_t* __restrict__ tri,
                                                                 const float* __restrict__ bary, const float* __restrict__ pts, int64_t N,
                                                                 float* __restrict__ can) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float inv[16];
    blend_inverse(T, tri, bary, i, inv);
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
#pragma unroll
    for (int k = 0; k < 3; ++k) can[i * 3 + k] = inv[k * 4] * x + inv[k * 4 + 1] * y + inv[k * 4 + 2] * z + inv[k * 4 + 3];
}

// Consecutive samples -- neighbours on a ray -- mostly share their closest triangle, and a float atomic to one address is served one lane at a time: the
// scatter kernels below first add up, inside the wave, the contributions of every RUN of lanes with the same triangle and let the run's first lane issue
// the atomics (human trainer batch: 5x fewer of them).  Runs(): which lanes d = 1, 2, 4 .. 32 above this one still belong to its run (bit log2(d)), and
// whether this lane is its run's first; run_sum(): the run's total, valid in the first lane (a suffix sum by doubling that stops at run ends).
struct Runs {
    unsigned ok;       // bit j: lane + 2^j is in this lane's run
    bool head;
};
__device__ __forceinline__ Runs find_runs(int k0, int k1, int k2, bool valid) {
    const int lane = threadIdx.x & 63;
    const int p0 = __shfl_up(k0, 1, 64), p1 = __shfl_up(k1, 1, 64), p2 = __shfl_up(k2, 1, 64);
    const bool pvalid = __shfl_up((int)valid, 1, 64) != 0;
    Runs r;
    r.head = lane == 0 || !valid || !pvalid || p0 != k0 || p1 != k1 || p2 != k2;          // (a lane past the end is a run of its own, adding zeros)
    const unsigned long long heads = __ballot(r.head);
    const unsigned long long above = lane == 63 ? ~0ull : ((heads >> (lane + 1)) | (~0ull << (63 - lane)));   // bit j: lane + 1 + j starts a run (or is past the wave)
    r.ok = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j)
        if ((above & ((1ull << (1 << j)) - 1ull)) == 0) r.ok |= 1u << j;
    return r;
}
__device__ __forceinline__ float run_sum(float v, const Runs& r) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float t = __shfl_down(v, 1 << j, 64);
        v += ((r.ok >> j) & 1u) ? t : 0.f;
    }
    return v;
}

__global__ __launch_bounds__(256) void warp_apply_backward_kernel(const float* __restrict__ T, const int32_t* __restrict__ tri,
                                                                  const float* __restrict__ bary, const float* __restrict__ pts,
                                                                  const float* __restrict__ g_can, int64_t N, float* __restrict__ g_T,
                                                                  float* __restrict__ g_bary) {
    const int64_t i_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i_raw < N;
    const int64_t i = valid ? i_raw : N - 1;                                             // (past the end: the last sample's arithmetic, nothing stored)
    const Runs runs = find_runs(tri[i * 3], tri[i * 3 + 1], tri[i * 3 + 2], valid);
    float inv[16];
    blend_inverse(T, tri, bary, i, inv);
    const float h[4] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], 1.f};
    const float g[3] = {g_can[i * 3], g_can[i * 3 + 1], g_can[i * 3 + 2]};
    // can = inv[:3] h: g_inv[k][l] = g[k] h[l] (k < 3);  M = inv^-1: g_M = -inv^T g_inv inv^T
    //   (inv^T g_inv)[a][l] = (sum_k inv[k][a] g[k]) h[l] = u[a] h[l];  g_M[a][b] = -u[a] (sum_l h[l] inv[b][l]) = -u[a] w[b]
    float u[4], w[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        u[a] = inv[a] * g[0] + inv[4 + a] * g[1] + inv[8 + a] * g[2];
        w[a] = inv[a * 4] * h[0] + inv[a * 4 + 1] * h[1] + inv[a * 4 + 2] * h[2] + inv[a * 4 + 3] * h[3];
    }
    float gb[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t v = tri[i * 3 + k];
        const float b = valid ? bary[i * 3 + k] : 0.f;
        const float* Tv = T + v * 16;
        float* gTv = g_T + v * 16;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gm = -u[a] * w[c];
                gb[k] += gm * Tv[a * 4 + c];
                const float total = run_sum(b * gm, runs);
                if (runs.head && valid) atomicAdd(gTv + a * 4 + c, total);
            }
    }
    if (valid)
#pragma unroll
        for (int k = 0; k < 3; ++k) g_bary[i * 3 + k] = gb[k];
}

// ---- barycentric coordinates of the closest point in its triangle, the reference's lines (utils/ray_utils.py:72-84) in float32, forward and adjoint.
//   N = e01 x e02, u = N . (e12 x (P - B)) / N.N, v = N . (e20 x (P - C)) / N.N, w = 1 - u - v      (A, B, C = the triangle's vertices; P constant)
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 crs3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__global__ __launch_bounds__(256) void bary_forward_kernel(const float* __restrict__ verts, const int32_t* __restrict__ tri, const float* __restrict__ closest,
                                                           int64_t N, float* __restrict__ bary) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const V3 A = ld3(verts + (int64_t)tri[i * 3] * 3), B = ld3(verts + (int64_t)tri[i * 3 + 1] * 3), C = ld3(verts + (int64_t)tri[i * 3 + 2] * 3);
    const V3 P = ld3(closest + i * 3);
    const V3 e01 = sub3(B, A), e02 = sub3(C, A), e12 = sub3(C, B), e20 = sub3(A, C);
    const V3 Nn = crs3(e01, e02);
    const float den = dot3(Nn, Nn);
    const float u = dot3(Nn, crs3(e12, sub3(P, B))) / den, v = dot3(Nn, crs3(e20, sub3(P, C))) / den;
    bary[i * 3] = u; bary[i * 3 + 1] = v; bary[i * 3 + 2] = 1.f - u - v;
}
__global__ __launch_bounds__(256) void bary_backward_kernel(const float* __restrict__ verts, const int32_t* __restrict__ tri, const float* __restrict__ closest,
                                                            const float* __restrict__ g_bary, int64_t N, float* __restrict__ g_verts) {
    const int64_t i_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i_raw < N;
    const int64_t i = valid ? i_raw : N - 1;
    const int64_t ia = tri[i * 3], ib = tri[i * 3 + 1], ic = tri[i * 3 + 2];
    const Runs runs = find_runs((int)ia, (int)ib, (int)ic, valid);
    const V3 A = ld3(verts + ia * 3), B = ld3(verts + ib * 3), C = ld3(verts + ic * 3), P = ld3(closest + i * 3);
    const V3 e01 = sub3(B, A), e02 = sub3(C, A), e12 = sub3(C, B), e20 = sub3(A, C), p1 = sub3(P, B), p2 = sub3(P, C);
    const V3 Nn = crs3(e01, e02), a = crs3(e12, p1), b = crs3(e20, p2);
    const float den = dot3(Nn, Nn), u = dot3(Nn, a) / den, v = dot3(Nn, b) / den;
    const float gw = valid ? g_bary[i * 3 + 2] : 0.f, gu = (valid ? g_bary[i * 3] : 0.f) - gw, gv = (valid ? g_bary[i * 3 + 1] : 0.f) - gw;
    const float k = 2.f * (gu * u + gv * v) / den;
    const V3 gN = {(gu * a.x + gv * b.x) / den - k * Nn.x, (gu * a.y + gv * b.y) / den - k * Nn.y, (gu * a.z + gv * b.z) / den - k * Nn.z};
    const V3 ga = {gu * Nn.x / den, gu * Nn.y / den, gu * Nn.z / den}, gb = {gv * Nn.x / den, gv * Nn.y / den, gv * Nn.z / den};
    // c = x cross y: g_x = y cross g_c, g_y = g_c cross x
    const V3 g01 = crs3(e02, gN), g02 = crs3(gN, e01), g12 = crs3(p1, ga), gp1 = crs3(ga, e12), g20 = crs3(p2, gb), gp2 = crs3(gb, e20);
    const V3 gA = {-g01.x - g02.x + g20.x, -g01.y - g02.y + g20.y, -g01.z - g02.z + g20.z};
    const V3 gB = {g01.x - g12.x - gp1.x, g01.y - g12.y - gp1.y, g01.z - g12.z - gp1.z};
    const V3 gC = {g02.x + g12.x - g20.x - gp2.x, g02.y + g12.y - g20.y - gp2.y, g02.z + g12.z - g20.z - gp2.z};
    const float vals[9] = {gA.x, gA.y, gA.z, gB.x, gB.y, gB.z, gC.x, gC.y, gC.z};
    const int64_t at[3] = {ia, ib, ic};
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const float total = run_sum(vals[q], runs);                                   // (the triangle's samples of this wave: one atomic per run)
        if (runs.head && valid) atomicAdd(g_verts + at[q / 3] * 3 + q % 3, total);
    }
}

}  // namespace

extern "C" {

int nm_bary_forward(const float* verts, const int32_t* tri, const float* closest, int64_t N, float* bary, nm_stream_t stream) {
    NM_REQUIRE(N == 0 || (verts && tri && closest && bary), "nm_bary_forward: null pointer");
    if (N == 0) return NM_OK;
    hipLaunchKernelGGL(bary_forward_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nm::as_stream(stream), verts, tri, closest, N, bary);
    return nm::check_launch("bary_forward_kernel");
}

int nm_bary_backward(const float* verts, const int32_t* tri, const float* closest, const float* g_bary, int64_t N, int64_t V, float* g_verts,
                     nm_stream_t stream) {
    NM_REQUIRE(verts && g_verts && V >= 1, "nm_bary_backward: null pointer");
    hipStream_t st = nm::as_stream(stream);
    if (int rc = nm::check_hip(hipMemsetAsync(g_verts, 0, (size_t)V * 12, st), "nm_bary_backward: clear g_verts")) return rc;
    if (N == 0) return NM_OK;
    NM_REQUIRE(tri && closest && g_bary, "nm_bary_backward: null pointer");
    hipLaunchKernelGGL(bary_backward_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, verts, tri, closest, g_bary, N, g_verts);
    return nm::check_launch("bary_backward_kernel");
}

// the tree of m->d_verts / m->d_faces into the handle's buffers: five kernels, nothing read back
static int build_tree(nm_mesh_s* m, hipStream_t st) {
    const Tree& tr = m->tr;
    const int F = m->F;
    hipLaunchKernelGGL(bbox_kernel, dim3(1), dim3(1024), 0, st, m->d_verts, m->V, m->d_bbox);
    hipLaunchKernelGGL(tri_prep_kernel, dim3((F + 255) / 256), dim3(256), 0, st, m->d_verts, m->d_faces, F, m->d_bbox, m->d_tmp, m->d_keys);
    hipLaunchKernelGGL(rank_scatter_kernel, dim3((F + 255) / 256), dim3(256), 0, st, m->d_keys, F, m->d_tmp, m->d_rec, m->d_face);
    hipLaunchKernelGGL(leaf_nodes_kernel, dim3((m->n_level[tr.L - 1] + 255) / 256), dim3(256), 0, st, m->d_verts, m->d_faces, m->d_face, F,
                       m->n_level[tr.L - 1], m->d_nodes + tr.first_lp);
    for (int l = tr.L - 2; l >= 0; --l)
        hipLaunchKernelGGL(upper_nodes_kernel, dim3((m->n_level[l] + 255) / 256), dim3(256), 0, st, m->d_nodes + level_base(l + 1),
                           m->n_level[l + 1], m->n_level[l], m->d_nodes + level_base(l));
    return nm::check_launch("mesh build kernels");
}

int nm_mesh_destroy(nm_mesh_t m) {
    if (!m) return NM_OK;
    if (m->d_tmp) (void)hipFree(m->d_tmp);
    if (m->d_keys) (void)hipFree(m->d_keys);
    if (m->d_bbox) (void)hipFree(m->d_bbox);
    if (m->d_verts) (void)hipFree(m->d_verts);
    if (m->d_faces) (void)hipFree(m->d_faces);
    if (m->d_rec) (void)hipFree(m->d_rec);
    if (m->d_face) (void)hipFree(m->d_face);
    if (m->d_nodes) (void)hipFree(m->d_nodes);
    if (m->d_pn) (void)hipFree(m->d_pn);
    if (m->d_pn_scratch) (void)hipFree(m->d_pn_scratch);
    delete m;
    return NM_OK;
}

int nm_mesh_create(const float* verts, int V, const int32_t* faces, int F, int search, nm_mesh_t* out, nm_stream_t stream) {
    NM_REQUIRE(verts && faces && out, "nm_mesh_create: null pointer");
    NM_REQUIRE(V >= 3 && F >= 1 && F <= (1 << 24), "nm_mesh_create: bad sizes V=%d F=%d", V, F);
    NM_REQUIRE(search == NM_SEARCH_TREE || search == NM_SEARCH_ALL || search == NM_SEARCH_TREE_WIDE, "nm_mesh_create: search mode %d", search);
    hipStream_t st = nm::as_stream(stream);
    nm_mesh_s* m = new nm_mesh_s();
    memset(m, 0, sizeof(*m));
    m->V = V; m->F = F;
    m->search = search == NM_SEARCH_ALL ? NM_SEARCH_ALL : NM_SEARCH_TREE;
    m->force_wide = search == NM_SEARCH_TREE_WIDE;
    Tree& tr = m->tr;
    tr.F = F;
    tr.L = 1;
    while ((1ll << (2 * tr.L)) < F) ++tr.L;                          // 4^L >= F
    int* n_level = m->n_level;
    n_level[tr.L] = F;
    for (int l = tr.L - 1; l >= 0; --l) n_level[l] = (n_level[l + 1] + 3) / 4;
    tr.first_lp = level_base(tr.L - 1);
    const int total = tr.first_lp + n_level[tr.L - 1];               // heap order: the last level is stored up to its last node
    m->n_nodes = total;
    int rc = NM_OK;
#define NM_TRY(expr, what) if (!rc) rc = nm::check_hip((expr), what)
    NM_TRY(hipMalloc(&m->d_verts, (size_t)V * 12), "nm_mesh_create: hipMalloc(verts)");
    NM_TRY(hipMalloc(&m->d_faces, (size_t)F * 12), "nm_mesh_create: hipMalloc(faces)");
    NM_TRY(hipMalloc(&m->d_rec, (size_t)F * sizeof(TriRec)), "nm_mesh_create: hipMalloc(records)");
    NM_TRY(hipMalloc(&m->d_face, (size_t)F * 4), "nm_mesh_create: hipMalloc(face ids)");
    NM_TRY(hipMalloc(&m->d_nodes, (size_t)total * sizeof(Node)), "nm_mesh_create: hipMalloc(nodes)");
    NM_TRY(hipMalloc(&m->d_tmp, (size_t)F * sizeof(TriRec)), "nm_mesh_create: hipMalloc(unsorted records)");
    NM_TRY(hipMalloc(&m->d_keys, (size_t)F * 8), "nm_mesh_create: hipMalloc(keys)");
    NM_TRY(hipMalloc(&m->d_bbox, 8 * 4), "nm_mesh_create: hipMalloc(bbox)");
    tr.scale = m->d_bbox + 7;
    NM_TRY(hipMemcpyAsync(m->d_verts, verts, (size_t)V * 12, hipMemcpyDeviceToDevice, st), "nm_mesh_create: copy verts");
    NM_TRY(hipMemcpyAsync(m->d_faces, faces, (size_t)F * 12, hipMemcpyDeviceToDevice, st), "nm_mesh_create: copy faces");
    float hb[7] = {0, 0, 0, 0, 0, 0, 0};
    if (!rc) rc = build_tree(m, st);
    // one read-back, at creation only: the non-finite flag of the vertex box (a mesh with a NaN / Inf vertex is refused; after an nm_mesh_update the
    // same flag, left on the device, switches search_kernel to its all-triangles loop)
    NM_TRY(hipMemcpyAsync(hb, m->d_bbox, 7 * 4, hipMemcpyDeviceToHost, st), "nm_mesh_create: read bbox");
    NM_TRY(hipStreamSynchronize(st), "nm_mesh_create: sync");
#undef NM_TRY
    if (!rc && hb[6] != 0.f) { nm::set_error("nm_mesh_create: non-finite vertex coordinate"); rc = NM_ERR_ARG; }
    if (rc) { nm_mesh_destroy(m); return rc; }
    *out = m;
    return NM_OK;
}

int nm_mesh_update(nm_mesh_t m, const float* verts, nm_stream_t stream) {
    NM_REQUIRE(m && verts, "nm_mesh_update: null pointer");
    hipStream_t st = nm::as_stream(stream);
    if (int rc = nm::check_hip(hipMemcpyAsync(m->d_verts, verts, (size_t)m->V * 12, hipMemcpyDeviceToDevice, st), "nm_mesh_update: copy verts")) return rc;
    m->pn_stale = true;
    return build_tree(m, st);
}

int nm_mesh_info(nm_mesh_t m, int32_t* levels, int64_t* nodes, int64_t* bytes) {
    NM_REQUIRE(m, "nm_mesh_info: null handle");
    if (levels) *levels = m->tr.L;
    if (nodes) *nodes = m->n_nodes;
    if (bytes) *bytes = (int64_t)m->n_nodes * (int64_t)sizeof(Node) + (int64_t)m->F * (int64_t)(sizeof(TriRec) + 4);
    return NM_OK;
}

int nm_warp_to_canonical(nm_mesh_t m, const float* pts, int64_t R, int S, const double* T, float* can_pts, float* can_dirs,
                         float* closest, nm_stream_t stream) {
    NM_REQUIRE(m, "nm_warp_to_canonical: null mesh handle");
    NM_REQUIRE(R == 0 || (pts && T && can_pts && can_dirs), "nm_warp_to_canonical: null pointer");
    NM_REQUIRE(R >= 0 && S >= 2, "nm_warp_to_canonical: bad sizes R=%lld S=%d", (long long)R, S);
    NM_REQUIRE(R < (1ll << 31), "nm_warp_to_canonical: too many rays for one launch");
    NM_REQUIRE((size_t)S * 24 <= 64 * 1024, "nm_warp_to_canonical: S=%d exceeds the LDS staging budget", S);
    if (R == 0) return NM_OK;
    const bool small = m->n_nodes <= 65536 && m->F <= 65536 && !m->force_wide;
    const int64_t N = R * (int64_t)S;
    NM_REQUIRE(N < (1ll << 31) * (int64_t)kChunk, "nm_warp_to_canonical: too many samples for one launch");
    const int chunk = chunk_for(N);
    const unsigned waves = (unsigned)((N + chunk - 1) / chunk);
    const size_t lds = (size_t)(3 * (m->tr.L - 1) + 1) * 64 * (small ? 4 : 8) + (size_t)(kTriSlots + 1) * 64 * (small ? 2 : 4);
    hipStream_t st = nm::as_stream(stream);
    int32_t* f_out = reinterpret_cast<int32_t*>(can_dirs);
    const int all = m->search == NM_SEARCH_ALL ? 1 : 0;
    if (small) hipLaunchKernelGGL((search_kernel<true, true>), dim3(waves), dim3(64), lds, st, m->tr, all, pts, N, m->d_rec, m->d_face, m->d_nodes, can_pts, f_out, 3, chunk);
    else hipLaunchKernelGGL((search_kernel<false, true>), dim3(waves), dim3(64), lds, st, m->tr, all, pts, N, m->d_rec, m->d_face, m->d_nodes, can_pts, f_out, 3, chunk);
    if (int rc = nm::check_launch("search_kernel")) return rc;
    const int threads = S <= 64 ? 64 : (S <= 128 ? 128 : 256);
    hipLaunchKernelGGL(tail_kernel, dim3((unsigned)R), dim3(threads), (size_t)S * 24, st, pts, S, m->d_verts, m->d_faces, T, can_pts, can_dirs, closest);
    return nm::check_launch("tail_kernel");
}

int nm_signed_distance(nm_mesh_t m, const float* pts, int64_t N, float* sdist, int32_t* face, float* closest, nm_stream_t stream) {
    NM_REQUIRE(m, "nm_signed_distance: null mesh handle");
    NM_REQUIRE(N >= 0 && N < (1ll << 31) * (int64_t)kChunk, "nm_signed_distance: bad N=%lld", (long long)N);
    if (N == 0) return NM_OK;
    NM_REQUIRE(pts && sdist && face && closest, "nm_signed_distance: null pointer");
    hipStream_t st = nm::as_stream(stream);
    if (!m->d_pn) {
        const int F = m->F, V = m->V;
        // scratch: fn [F,3] | vn [V,3] | cnt [V] | cur [V] | off [V+1] | adj [3F]
        const size_t n_f = (size_t)F * 3 + (size_t)V * 3, n_i = (size_t)2 * V + (size_t)V + 1 + (size_t)3 * F;
        int rc = nm::check_hip(hipMalloc(&m->d_pn, (size_t)F * kPnFloats * 4), "nm_signed_distance: hipMalloc(pseudonormals)");
        if (!rc) rc = nm::check_hip(hipMalloc(&m->d_pn_scratch, (n_f + n_i) * 4), "nm_signed_distance: hipMalloc(pseudonormal scratch)");
        if (!rc) {
            float* fn = reinterpret_cast<float*>(m->d_pn_scratch);
            float* vn = fn + (size_t)F * 3;
            int* cnt = reinterpret_cast<int*>(vn + (size_t)V * 3);
            int *cur = cnt + V, *off = cur + V, *adj = off + V + 1;
            rc = nm::check_hip(hipMemsetAsync(cnt, 0, (size_t)2 * V * 4, st), "nm_signed_distance: memset");
            if (!rc) {
                const dim3 gc((3 * F + 255) / 256), gv((V + 255) / 256);
                hipLaunchKernelGGL(face_normal_kernel, dim3((F + 255) / 256), dim3(256), 0, st, m->d_verts, m->d_faces, F, fn);
                hipLaunchKernelGGL(corner_count_kernel, gc, dim3(256), 0, st, m->d_faces, F, V, cnt);
                hipLaunchKernelGGL(adj_offsets_kernel, dim3(1), dim3(1024), 0, st, cnt, V, off);
                hipLaunchKernelGGL(corner_fill_kernel, gc, dim3(256), 0, st, m->d_faces, F, V, off, cur, adj);
                hipLaunchKernelGGL(adj_sort_kernel, gv, dim3(256), 0, st, off, V, adj);
                hipLaunchKernelGGL(vertex_normal_adj_kernel, gv, dim3(256), 0, st, m->d_verts, m->d_faces, V, off, adj, fn, vn);
                hipLaunchKernelGGL(pseudonormal_pack_adj_kernel, gc, dim3(256), 0, st, m->d_faces, F, V, off, adj, fn, vn, m->d_pn);
                rc = nm::check_launch("pseudonormal kernels");
            }
        }
        if (rc) {
            if (m->d_pn) (void)hipFree(m->d_pn);
            if (m->d_pn_scratch) (void)hipFree(m->d_pn_scratch);
            m->d_pn = nullptr; m->d_pn_scratch = nullptr;
            return rc;
        }
        m->pn_stale = false;
    } else if (m->pn_stale) {                                            // nm_mesh_update moved the vertices: the normals again, on the table of the first build
        const int F = m->F, V = m->V;
        float* fn = reinterpret_cast<float*>(m->d_pn_scratch);
        float* vn = fn + (size_t)F * 3;
        int* cnt = reinterpret_cast<int*>(vn + (size_t)V * 3);
        int *off = cnt + 2 * V, *adj = off + V + 1;
        hipLaunchKernelGGL(face_normal_kernel, dim3((F + 255) / 256), dim3(256), 0, st, m->d_verts, m->d_faces, F, fn);
        hipLaunchKernelGGL(vertex_normal_adj_kernel, dim3((V + 255) / 256), dim3(256), 0, st, m->d_verts, m->d_faces, V, off, adj, fn, vn);
        hipLaunchKernelGGL(pseudonormal_pack_adj_kernel, dim3((3 * F + 255) / 256), dim3(256), 0, st, m->d_faces, F, V, off, adj, fn, vn, m->d_pn);
        if (int rc = nm::check_launch("pseudonormal kernels (update)")) return rc;
        m->pn_stale = false;
    }
    const bool small = m->n_nodes <= 65536 && m->F <= 65536 && !m->force_wide;
    const int chunk = chunk_for(N);
    const unsigned waves = (unsigned)((N + chunk - 1) / chunk);
    const size_t lds = (size_t)(3 * (m->tr.L - 1) + 1) * 64 * (small ? 4 : 8) + (size_t)(kTriSlots + 1) * 64 * (small ? 2 : 4);
    const int all = m->search == NM_SEARCH_ALL ? 1 : 0;
    if (small) hipLaunchKernelGGL(search_kernel<true>, dim3(waves), dim3(64), lds, st, m->tr, all, pts, N, m->d_rec, m->d_face, m->d_nodes, closest, face, 1, chunk);
    else hipLaunchKernelGGL(search_kernel<false>, dim3(waves), dim3(64), lds, st, m->tr, all, pts, N, m->d_rec, m->d_face, m->d_nodes, closest, face, 1, chunk);
    if (int rc = nm::check_launch("search_kernel")) return rc;
    hipLaunchKernelGGL(signed_distance_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, pts, N, m->d_verts, m->d_faces, m->d_pn, face, closest,
                       sdist);
    return nm::check_launch("signed_distance_kernel");
}

int nm_warp_apply_forward(const float* T, const int32_t* tri, const float* bary, const float* pts, int64_t N, float* can, nm_stream_t stream) {
    NM_REQUIRE(N == 0 || (T && tri && bary && pts && can), "nm_warp_apply_forward: null pointer");
    if (N == 0) return NM_OK;
    hipLaunchKernelGGL(warp_apply_forward_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nm::as_stream(stream), T, tri, bary, pts, N, can);
    return nm::check_launch("warp_apply_forward_kernel");
}

int nm_warp_apply_backward(const float* T, const int32_t* tri, const float* bary, const float* pts, const float* g_can, int64_t N, int64_t V,
                           float* g_T, float* g_bary, nm_stream_t stream) {
    NM_REQUIRE(T && g_T && V >= 1, "nm_warp_apply_backward: null pointer");
    hipStream_t st = nm::as_stream(stream);
    if (int rc = nm::check_hip(hipMemsetAsync(g_T, 0, (size_t)V * 64, st), "nm_warp_apply_backward: clear g_T")) return rc;
    if (N == 0) return NM_OK;
    NM_REQUIRE(tri && bary && pts && g_can && g_bary, "nm_warp_apply_backward: null pointer");
    hipLaunchKernelGGL(warp_apply_backward_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, T, tri, bary, pts, g_can, N, g_T, g_bary);
    return nm::check_launch("warp_apply_backward_kernel");
}

}  // extern "C"
