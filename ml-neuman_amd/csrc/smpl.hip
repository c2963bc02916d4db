// a12 SMPL linear blend skinning for gfx950: per-frame posed vertices and per-vertex canonical ("da" pose) -> scene transforms,
// batched over frames.  Replaces, on the device,
//   * reference models/smpl.py:266-360 (lbs), :407-438 (batch_rodrigues), :454-505 (batch_rigid_transform), float32;
//   * data_io/neuman_helper.py:288-330 (read_smpls: the numpy chain of the render scripts -- a float32 4x4 inverse and product,
//     then float64 from the alignment on) when `precise`, models/human_nerf.py:92-122 (vertex_forward: float32 throughout) else.
// It is ~3 MFLOP per frame: the point is not speed but that a whole sequence's meshes and transforms are produced where the
// renderers consume them ("device LBS batched over frames"), with no host round trip per frame.
//   smpl_joints_kernel  one workgroup per frame: shape blend + joint regression (the 6890-long reductions), Rodrigues for the
//                       frame's pose and for the da pose, the two kinematic chains (sequential over the 24 joints).
//   smpl_rows_kernel    one lane per (frame, vertex or joint row): skinning blend of the 24 joint transforms for both poses,
//                       da-pose vertex, T_da2pose = T_t2pose inv(T_t2da), alignment and scale, world vertex.
// Sums run in index order in float32 where the reference uses a float32 BLAS call (unspecified order): agreement with the
// reference is to float32 rounding (~1e-6), tested at 2e-5 of the values' magnitude.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace {

constexpr int kMaxJoints = 64;
constexpr int kMaxBetas = 32;

struct SmplDims { int V, J, NB; };

__device__ __forceinline__ void rodrigues(const float* __restrict__ rv, float* __restrict__ R) {   // smpl.py:407-438
    const float x = rv[0] + 1e-8f, y = rv[1] + 1e-8f, z = rv[2] + 1e-8f;
    const float angle = sqrtf(x * x + y * y + z * z);
    const float rx = rv[0] / angle, ry = rv[1] / angle, rz = rv[2] / angle;
    const float c = cosf(angle), s = sinf(angle), omc = 1.f - c;
    const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float kk = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
            R[i * 3 + j] = (i == j ? 1.f : 0.f) + s * K[i * 3 + j] + omc * kk;
        }
}

__device__ __forceinline__ void mat4_mul(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i * 4 + j] = a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j] + a[i * 4 + 2] * b[8 + j] + a[i * 4 + 3] * b[12 + j];
}

// workspace per frame: J [Jn,3] rest joints | Jda [Jn,3] posed joints of the da pose | A_pose [Jn,16] | A_da [Jn,16]
__device__ __host__ inline int ws_floats(int J) { return J * (3 + 3 + 16 + 16); }

__global__ __launch_bounds__(256) void smpl_joints_kernel(SmplDims d, const float* __restrict__ v_template, const float* __restrict__ shapedirs,
                                                          const float* __restrict__ j_reg, const int32_t* __restrict__ parents,
                                                          const float* __restrict__ poses, const float* __restrict__ betas,
                                                          const float* __restrict__ da_pose, float* __restrict__ ws) {
    // (poses / betas / ws of frame b; da_pose: the canonical pose -- the handle's, or the caller's for the differentiable form)
    __shared__ float beta[kMaxBetas];
    __shared__ float red[4][3];
    __shared__ float Js[kMaxJoints][3];
    __shared__ float Rm[2][kMaxJoints][9];
    __shared__ float chain[2][kMaxJoints][16];
    const int b = blockIdx.x, tid = threadIdx.x;
    float* w = ws + (size_t)b * ws_floats(d.J);
    if (tid < d.NB) beta[tid] = betas[(size_t)b * d.NB + tid];
    __syncthreads();
    // rest joints J = J_regressor v_shaped (smpl.py:312-316): one strided pass over the vertices per joint
    for (int j = 0; j < d.J; ++j) {
        float acc[3] = {0.f, 0.f, 0.f};
        for (int v = tid; v < d.V; v += blockDim.x) {
            const float r = j_reg[(size_t)j * d.V + v];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float vs = 0.f;
                for (int l = 0; l < d.NB; ++l) vs += beta[l] * shapedirs[((size_t)v * 3 + k) * d.NB + l];
                vs = v_template[v * 3 + k] + vs;
                acc[k] += r * vs;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] = wave_sum(acc[k]);
        if ((tid & 63) == 0)
#pragma unroll
            for (int k = 0; k < 3; ++k) red[tid >> 6][k] = acc[k];
        __syncthreads();
        if (tid < 3) {
            float s = 0.f;
            for (int q = 0; q < (int)(blockDim.x >> 6); ++q) s += red[q][tid];
            Js[j][tid] = s;
        }
        __syncthreads();
    }
    // Rodrigues: the frame's pose (set 0) and the da pose (set 1)
    if (tid < 2 * d.J) {
        const int set = tid / d.J, j = tid % d.J;
        rodrigues(set == 0 ? poses + (size_t)b * d.J * 3 + j * 3 : da_pose + j * 3, Rm[set][j]);
    }
    __syncthreads();
    // kinematic chains (smpl.py:474-493): sequential over the joints, one lane per pose
    if (tid < 2) {
        const int set = tid;
        for (int j = 0; j < d.J; ++j) {
            const int p = j == 0 ? -1 : parents[j];
            float m[16];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) m[i * 4 + k] = Rm[set][j][i * 3 + k];
                m[i * 4 + 3] = p < 0 ? Js[j][i] : Js[j][i] - Js[p][i];
            }
            m[12] = m[13] = m[14] = 0.f; m[15] = 1.f;
            if (p < 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) chain[set][j][q] = m[q];
            } else {
                mat4_mul(chain[set][p], m, chain[set][j]);
            }
        }
    }
    __syncthreads();
    // rel_transforms = transforms - pad(transforms [J;0]) (smpl.py:499-503): only the last column changes
    if (tid < 2 * d.J) {
        const int set = tid / d.J, j = tid % d.J;
        const float* t = chain[set][j];
        float* A = w + d.J * 6 + ((size_t)set * d.J + j) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float tj = t[i * 4] * Js[j][0] + t[i * 4 + 1] * Js[j][1] + t[i * 4 + 2] * Js[j][2] + t[i * 4 + 3] * 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) A[i * 4 + k] = t[i * 4 + k];
            A[i * 4 + 3] = t[i * 4 + 3] - tj;
        }
        if (set == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) w[j * 3 + k] = Js[j][k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) w[d.J * 3 + j * 3 + k] = t[k * 4 + 3];                      // posed joints of the da pose
        }
    }
}

// general 4x4 inverse by cofactors in f64 (the blended matrices are affine only up to the rounding of sum(w) = 1)
__device__ __forceinline__ void inv4x4_full(const double* m, double* o) {
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
    o[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * inv;
    o[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * inv;
    o[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * inv;
    o[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * inv;
}

__global__ __launch_bounds__(256) void smpl_rows_kernel(SmplDims d, int B, const float* __restrict__ v_template, const float* __restrict__ shapedirs,
                                                        const float* __restrict__ lbs_weights, const float* __restrict__ betas,
                                                        const double* __restrict__ alignments, double scale, int precise,
                                                        const float* __restrict__ ws, double* __restrict__ T_out, float* __restrict__ world,
                                                        float* __restrict__ stat, float* __restrict__ T32) {
    // T32 != null (nm_smpl_vertex_forward): vertex rows only, T as float32 [B,V,16] into T32, world [B,V,3]; T_out / stat unused
    const int rows = d.V + d.J;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)B * rows) return;
    const int b = (int)(gid / rows), r = (int)(gid % rows);
    const float* w = ws + (size_t)b * ws_floats(d.J);
    const float* A_pose = w + d.J * 6;
    const float* A_da = A_pose + d.J * 16;
    float Tp[16], Td[16], pt[3], dap[3];
    if (r < d.V) {
        // T = W A (smpl.py:343-344), both poses; v_shaped (:309); the da-pose vertex T_da [v_shaped; 1] (:355-358)
#pragma unroll
        for (int q = 0; q < 16; ++q) Tp[q] = Td[q] = 0.f;
        for (int j = 0; j < d.J; ++j) {
            const float wj = lbs_weights[(size_t)r * d.J + j];
#pragma unroll
            for (int q = 0; q < 16; ++q) { Tp[q] += wj * A_pose[j * 16 + q]; Td[q] += wj * A_da[j * 16 + q]; }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float vs = 0.f;
            for (int l = 0; l < d.NB; ++l) vs += betas[(size_t)b * d.NB + l] * shapedirs[((size_t)r * 3 + k) * d.NB + l];
            pt[k] = v_template[r * 3 + k] + vs;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dap[k] = Td[k * 4] * pt[0] + Td[k * 4 + 1] * pt[1] + Td[k * 4 + 2] * pt[2] + Td[k * 4 + 3] * 1.f;
    } else {
        const int j = r - d.V;                                       // joint rows (concat_joints=True, smpl.py:347-349)
#pragma unroll
        for (int q = 0; q < 16; ++q) { Tp[q] = A_pose[j * 16 + q]; Td[q] = A_da[j * 16 + q]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) dap[k] = w[d.J * 3 + j * 3 + k];
    }
    // T_da2pose = T_t2pose inv(T_t2da): a float32 inverse and product in the reference (neuman_helper.py:314, human_nerf.py:109)
    double Tdd[16], inv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Tdd[q] = (double)Td[q];
    inv4x4_full(Tdd, inv);
    float invf[16], Tdp[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) invf[q] = (float)inv[q];
    mat4_mul(Tp, invf, Tdp);
    const double* al = alignments + (size_t)b * 16;
    double T[16];
    if (precise) {
        // float64 from here: T = S (align^T T_da2pose) (neuman_helper.py:315-318)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double t = al[0 * 4 + i] * (double)Tdp[j] + al[1 * 4 + i] * (double)Tdp[4 + j] + al[2 * 4 + i] * (double)Tdp[8 + j] +
                                 al[3 * 4 + i] * (double)Tdp[12 + j];
                T[i * 4 + j] = i < 3 ? scale * t : t;
            }
    } else {
        // float32 throughout (human_nerf.py:110-113)
        const float sc = (float)scale;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = (float)al[0 * 4 + i] * Tdp[j] + (float)al[1 * 4 + i] * Tdp[4 + j] + (float)al[2 * 4 + i] * Tdp[8 + j] +
                                (float)al[3 * 4 + i] * Tdp[12 + j];
                T[i * 4 + j] = (double)(i < 3 ? sc * t : t);
            }
    }
    if (T32) {
        if (r >= d.V) return;
        float* To = T32 + ((size_t)b * d.V + r) * 16;
        float* wo = world + ((size_t)b * d.V + r) * 3;
#pragma unroll
        for (int q = 0; q < 16; ++q) To[q] = (float)T[q];
#pragma unroll
        for (int k = 0; k < 3; ++k) wo[k] = (float)T[k * 4] * dap[0] + (float)T[k * 4 + 1] * dap[1] + (float)T[k * 4 + 2] * dap[2] + (float)T[k * 4 + 3] * 1.f;
        return;
    }
    double* To = T_out + ((size_t)b * rows + r) * 16;
#pragma unroll
    for (int q = 0; q < 16; ++q) To[q] = T[q];
    float* wo = world + ((size_t)b * rows + r) * 3;
    float* so = stat + ((size_t)b * rows + r) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (precise)
            wo[k] = (float)(T[k * 4] * (double)dap[0] + T[k * 4 + 1] * (double)dap[1] + T[k * 4 + 2] * (double)dap[2] + T[k * 4 + 3]);
        else
            wo[k] = (float)T[k * 4] * dap[0] + (float)T[k * 4 + 1] * dap[1] + (float)T[k * 4 + 2] * dap[2] + (float)T[k * 4 + 3] * 1.f;
        so[k] = dap[k];
    }
}


// ---- the joints of ONE frame, parallel over the joints (smpl_joints_kernel runs one workgroup per frame: right for a sequence, 2.6 ms for
// a single frame): J = J_regressor v_shaped with v_shaped formed only where the regressor is non-zero; then rotations, chains and A in
// one small workgroup.  Same workspace layout as above.
__global__ __launch_bounds__(256) void smpl_jreg_kernel(SmplDims d, const float* __restrict__ v_template, const float* __restrict__ shapedirs,
                                                        const float* __restrict__ j_reg, const float* __restrict__ betas, float* __restrict__ ws) {
    __shared__ float beta[kMaxBetas];
    __shared__ float red[4][3];
    const int j = blockIdx.x, tid = threadIdx.x;
    if (tid < d.NB) beta[tid] = betas[tid];
    __syncthreads();
    float acc[3] = {0.f, 0.f, 0.f};
    for (int v = tid; v < d.V; v += blockDim.x) {
        const float r = j_reg[(size_t)j * d.V + v];
        if (r != 0.f)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float vs = 0.f;
                for (int l = 0; l < d.NB; ++l) vs += beta[l] * shapedirs[((size_t)v * 3 + k) * d.NB + l];
                acc[k] += r * (v_template[v * 3 + k] + vs);
            }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = wave_sum(acc[k]);
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) red[tid >> 6][k] = acc[k];
    __syncthreads();
    if (tid < 3) ws[j * 3 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__global__ __launch_bounds__(128) void smpl_chain_kernel(SmplDims d, const int32_t* __restrict__ parents, const float* __restrict__ pose,
                                                         const float* __restrict__ da_pose, float* __restrict__ ws) {
    __shared__ float Js[kMaxJoints][3];
    __shared__ float Rm[2][kMaxJoints][9];
    __shared__ float chain[2][kMaxJoints][16];
    const int tid = threadIdx.x;
    for (int i = tid; i < d.J * 3; i += blockDim.x) Js[i / 3][i % 3] = ws[i];
    if (tid < 2 * d.J) {
        const int set = tid / d.J, j = tid % d.J;
        rodrigues(set == 0 ? pose + j * 3 : da_pose + j * 3, Rm[set][j]);
    }
    __syncthreads();
    if (tid < 2) {
        const int set = tid;
        for (int j = 0; j < d.J; ++j) {
            const int p = j == 0 ? -1 : parents[j];
            float m[16];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) m[i * 4 + k] = Rm[set][j][i * 3 + k];
                m[i * 4 + 3] = p < 0 ? Js[j][i] : Js[j][i] - Js[p][i];
            }
            m[12] = m[13] = m[14] = 0.f; m[15] = 1.f;
            if (p < 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) chain[set][j][q] = m[q];
            } else {
                mat4_mul(chain[set][p], m, chain[set][j]);
            }
        }
    }
    __syncthreads();
    if (tid < 2 * d.J) {
        const int set = tid / d.J, j = tid % d.J;
        const float* t = chain[set][j];
        float* A = ws + d.J * 6 + ((size_t)set * d.J + j) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float tj = t[i * 4] * Js[j][0] + t[i * 4 + 1] * Js[j][1] + t[i * 4 + 2] * Js[j][2] + t[i * 4 + 3] * 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) A[i * 4 + k] = t[i * 4 + k];
            A[i * 4 + 3] = t[i * 4 + 3] - tj;
        }
        if (set == 1)
#pragma unroll
            for (int k = 0; k < 3; ++k) ws[d.J * 3 + j * 3 + k] = t[k * 4 + 3];                      // posed joints of the da pose
    }
}

// =====================================================================================================================================
// The differentiable form (HumanNeRF.vertex_forward, models/human_nerf.py:92-122, under autograd in the human trainer:
// gradients of the loss with respect to the frame's pose, shape and alignment): forward = the two kernels above on one frame in
// vertex_forward's float32 arithmetic; backward = four launches, every reduction in a fixed order (no float atomics).
//   world = T [da; 1],  T = D(scale) align^T Tp inv(Td),  da = Td [v_shaped; 1],  Tp = sum_j w_j A_pose_j,  Td = sum_j w_j A_da_j,
//   A_j = [G_j.R | G_j.t - G_j.R J_j],  G_j = G_parent [R_j | J_j - J_parent],  R_j = Rodrigues(pose_j),  J = J_regressor v_shaped,
//   v_shaped = v_template + shapedirs beta.
// ---- per vertex: gradients of Tp and Td (3 x 4 each: the last row of an affine blend is constant), of v_shaped through the da
// vertex, and this vertex's share of the alignment gradient
__global__ __launch_bounds__(256) void smpl_bw_rows_kernel(SmplDims d, const float* __restrict__ v_template, const float* __restrict__ shapedirs,
                                                           const float* __restrict__ lbs_weights, const float* __restrict__ beta,
                                                           const double* __restrict__ alignment, float sc, const float* __restrict__ ws,
                                                           const float* __restrict__ g_world, const float* __restrict__ g_T,
                                                           float* __restrict__ g_rows, float* __restrict__ g_al_part) {
    // g_rows [V][27] = gTp (12) | gTd (12) | g v_shaped through da (3);  g_al_part [blocks][16]
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    float gal[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gal[q] = 0.f;
    if (v < d.V) {
        const float* A_pose = ws + d.J * 6;
        const float* A_da = A_pose + d.J * 16;
        float Tp[16], Td[16], pt[3], dap[3];
#pragma unroll
        for (int q = 0; q < 16; ++q) Tp[q] = Td[q] = 0.f;
        for (int j = 0; j < d.J; ++j) {
            const float wj = lbs_weights[(size_t)v * d.J + j];
#pragma unroll
            for (int q = 0; q < 16; ++q) { Tp[q] += wj * A_pose[j * 16 + q]; Td[q] += wj * A_da[j * 16 + q]; }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float vs = 0.f;
            for (int l = 0; l < d.NB; ++l) vs += beta[l] * shapedirs[((size_t)v * 3 + k) * d.NB + l];
            pt[k] = v_template[v * 3 + k] + vs;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) dap[k] = Td[k * 4] * pt[0] + Td[k * 4 + 1] * pt[1] + Td[k * 4 + 2] * pt[2] + Td[k * 4 + 3];
        double Tdd[16], invd[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) Tdd[q] = (double)Td[q];
        inv4x4_full(Tdd, invd);
        float inv[16], Tdp[16], al[16], T[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { inv[q] = (float)invd[q]; al[q] = (float)alignment[q]; }
        mat4_mul(Tp, inv, Tdp);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = al[0 * 4 + i] * Tdp[j] + al[1 * 4 + i] * Tdp[4 + j] + al[2 * 4 + i] * Tdp[8 + j] + al[3 * 4 + i] * Tdp[12 + j];
                T[i * 4 + j] = i < 3 ? sc * t : t;
            }
        // upstream gradients
        float gT[16], gw[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) gT[q] = g_T ? g_T[(size_t)v * 16 + q] : 0.f;
        if (g_world)
#pragma unroll
            for (int k = 0; k < 3; ++k) gw[k] = g_world[(size_t)v * 3 + k];
        // world = T[:3] [da; 1]
        float gdap[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gdap[c] = T[c] * gw[0] + T[4 + c] * gw[1] + T[8 + c] * gw[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) gT[k * 4 + c] += gw[k] * dap[c];
            gT[k * 4 + 3] += gw[k];
        }
        // T = D (al^T Tdp): gU = D gT;  gTdp[k][j] = sum_i al[k][i] gU[i][j];  gal[k][i] = sum_j gU[i][j] Tdp[k][j]
        float gU[16], gTdp[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) gU[i * 4 + j] = (i < 3 ? sc : 1.f) * gT[i * 4 + j];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) gTdp[k * 4 + j] = al[k * 4] * gU[j] + al[k * 4 + 1] * gU[4 + j] + al[k * 4 + 2] * gU[8 + j] + al[k * 4 + 3] * gU[12 + j];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                gal[k * 4 + i] = gU[i * 4] * Tdp[k * 4] + gU[i * 4 + 1] * Tdp[k * 4 + 1] + gU[i * 4 + 2] * Tdp[k * 4 + 2] + gU[i * 4 + 3] * Tdp[k * 4 + 3];
        // Tdp = Tp inv: gTp = gTdp inv^T;  ginv = Tp^T gTdp;  inv = Td^-1: gTd = -inv^T ginv inv^T
        float gTp[16], ginv[16], tmp[16], gTd[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gTp[i * 4 + j] = gTdp[i * 4] * inv[j * 4] + gTdp[i * 4 + 1] * inv[j * 4 + 1] + gTdp[i * 4 + 2] * inv[j * 4 + 2] + gTdp[i * 4 + 3] * inv[j * 4 + 3];
                ginv[i * 4 + j] = Tp[i] * gTdp[j] + Tp[4 + i] * gTdp[4 + j] + Tp[8 + i] * gTdp[8 + j] + Tp[12 + i] * gTdp[12 + j];
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tmp[i * 4 + j] = inv[i] * ginv[j] + inv[4 + i] * ginv[4 + j] + inv[8 + i] * ginv[8 + j] + inv[12 + i] * ginv[12 + j];          // inv^T ginv
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                gTd[i * 4 + j] = -(tmp[i * 4] * inv[j * 4] + tmp[i * 4 + 1] * inv[j * 4 + 1] + tmp[i * 4 + 2] * inv[j * 4 + 2] + tmp[i * 4 + 3] * inv[j * 4 + 3]);
        // da = Td[:3] [pt; 1]
        float gpt[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gpt[c] = Td[c] * gdap[0] + Td[4 + c] * gdap[1] + Td[8 + c] * gdap[2];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int c = 0; c < 3; ++c) gTd[k * 4 + c] += gdap[k] * pt[c];
            gTd[k * 4 + 3] += gdap[k];
        }
        float* o = g_rows + (size_t)v * 27;
#pragma unroll
        for (int q = 0; q < 12; ++q) { o[q] = gTp[q]; o[12 + q] = gTd[q]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) o[24 + c] = gpt[c];
    }
    // the block's share of the alignment gradient, summed in a fixed order
    __shared__ float red[4][16];
#pragma unroll
    for (int q = 0; q < 16; ++q) gal[q] = wave_sum(gal[q]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[threadIdx.x >> 6][q] = gal[q];
    __syncthreads();
    if (threadIdx.x < 16) g_al_part[(size_t)blockIdx.x * 16 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---- per (chain, joint): gA_j = sum_v w_vj gT_v (12 values), one workgroup each
__global__ __launch_bounds__(256) void smpl_bw_joints_kernel(SmplDims d, const float* __restrict__ lbs_weights, const float* __restrict__ g_rows,
                                                             float* __restrict__ g_A) {
    const int c = blockIdx.x / d.J, j = blockIdx.x % d.J;
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.f;
    for (int v = threadIdx.x; v < d.V; v += blockDim.x) {
        const float w = lbs_weights[(size_t)v * d.J + j];
        if (w != 0.f) {
            const float* g = g_rows + (size_t)v * 27 + 12 * c;
#pragma unroll
            for (int q = 0; q < 12; ++q) acc[q] += w * g[q];
        }
    }
    __shared__ float red[4][12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = wave_sum(acc[q]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 12; ++q) red[threadIdx.x >> 6][q] = acc[q];
    __syncthreads();
    if (threadIdx.x < 12) g_A[((size_t)c * d.J + j) * 12 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// d Rodrigues: R = I + sin(t) K + (1 - cos t) K^2, K = skew(r / t), t = |r + 1e-8|  (smpl.py:407-438)
__device__ __forceinline__ void rodrigues_backward(const float* __restrict__ rv, const float* __restrict__ gR, float* __restrict__ g_rv) {
    const float ex = rv[0] + 1e-8f, ey = rv[1] + 1e-8f, ez = rv[2] + 1e-8f;
    const float t = sqrtf(ex * ex + ey * ey + ez * ez);
    const float u[3] = {rv[0] / t, rv[1] / t, rv[2] / t};
    const float c = cosf(t), s = sinf(t), omc = 1.f - c;
    const float K[9] = {0.f, -u[2], u[1], u[2], 0.f, -u[0], -u[1], u[0], 0.f};
    float KK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) KK[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
    float gt = 0.f, gK[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) gt += gR[q] * (c * K[q] + s * KK[q]);
    // d(K K) = dK K + K dK: gK = s gR + (1 - c) (gR K^T + K^T gR)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float a = gR[i * 3] * K[j * 3] + gR[i * 3 + 1] * K[j * 3 + 1] + gR[i * 3 + 2] * K[j * 3 + 2];       // (gR K^T)[i][j]
            const float b = K[i] * gR[j] + K[3 + i] * gR[3 + j] + K[6 + i] * gR[6 + j];                              // (K^T gR)[i][j]
            gK[i * 3 + j] = s * gR[i * 3 + j] + omc * (a + b);
        }
    const float gu[3] = {gK[7] - gK[5], gK[2] - gK[6], gK[3] - gK[1]};
    // u = r / t: g_r = gu / t + (gt - (gu . r) / t^2) * (r + eps) / t
    const float dot = gu[0] * rv[0] + gu[1] * rv[1] + gu[2] * rv[2];
    const float gtt = (gt - dot / (t * t)) / t;
    g_rv[0] = gu[0] / t + gtt * ex;
    g_rv[1] = gu[1] / t + gtt * ey;
    g_rv[2] = gu[2] / t + gtt * ez;
}

// ---- one workgroup: the two kinematic chains backwards (children before parents) -> g pose [J*3] (pose chain only), g J [J,3]
__global__ __launch_bounds__(64) void smpl_bw_chain_kernel(SmplDims d, const int32_t* __restrict__ parents, const float* __restrict__ pose,
                                                           const float* __restrict__ da_pose, const float* __restrict__ ws,
                                                           const float* __restrict__ g_A, float* __restrict__ g_pose, float* __restrict__ g_J) {
    __shared__ float gJs[2][kMaxJoints][3];
    __shared__ float gGR[2][kMaxJoints][9], gGt[2][kMaxJoints][3];
    const int c = threadIdx.x;                                           // chain 0: the frame's pose, chain 1: the da pose
    if (c < 2) {
        const float* Js = ws;                                            // rest joints [J,3]
        const float* A = ws + d.J * 6 + (size_t)c * d.J * 16;            // A_j: .R = G_j.R, .t = G_j.t - G_j.R J_j
        const float* rv = c == 0 ? pose : da_pose;
        for (int j = 0; j < d.J; ++j) {
#pragma unroll
            for (int q = 0; q < 9; ++q) gGR[c][j][q] = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) { gGt[c][j][q] = 0.f; gJs[c][j][q] = 0.f; }
        }
        for (int j = d.J - 1; j >= 0; --j) {
            const float* gA = g_A + ((size_t)c * d.J + j) * 12;
            const float* Aj = A + j * 16;
            float gR[9], gt[3];
            // A_j = [G.R | G.t - G.R J_j]
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int k = 0; k < 3; ++k) gR[i * 3 + k] = gGR[c][j][i * 3 + k] + gA[i * 4 + k] - gA[i * 4 + 3] * Js[j * 3 + k];
                gt[i] = gGt[c][j][i] + gA[i * 4 + 3];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) gJs[c][j][k] -= Aj[k] * gA[3] + Aj[4 + k] * gA[7] + Aj[8 + k] * gA[11];          // -G.R^T gA.t
            float Rj[9], gRj[9], gtj[3];
            rodrigues(rv + j * 3, Rj);
            const int p = j == 0 ? -1 : parents[j];
            if (p >= 0) {
                const float* Ap = A + p * 16;                                // G_p.R
                float tj[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) tj[k] = Js[j * 3 + k] - Js[p * 3 + k];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        gGR[c][p][i * 3 + k] += gR[i * 3] * Rj[k * 3] + gR[i * 3 + 1] * Rj[k * 3 + 1] + gR[i * 3 + 2] * Rj[k * 3 + 2] + gt[i] * tj[k];
                        gRj[i * 3 + k] = Ap[i] * gR[k] + Ap[4 + i] * gR[3 + k] + Ap[8 + i] * gR[6 + k];               // G_p.R^T gR
                    }
                    gGt[c][p][i] += gt[i];
                    gtj[i] = Ap[i] * gt[0] + Ap[4 + i] * gt[1] + Ap[8 + i] * gt[2];
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) { gJs[c][j][k] += gtj[k]; gJs[c][p][k] -= gtj[k]; }
            } else {
#pragma unroll
                for (int q = 0; q < 9; ++q) gRj[q] = gR[q];
#pragma unroll
                for (int k = 0; k < 3; ++k) gJs[c][j][k] += gt[k];
            }
            if (c == 0) rodrigues_backward(rv + j * 3, gRj, g_pose + j * 3);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d.J * 3; i += blockDim.x) g_J[i] = gJs[0][i / 3][i % 3] + gJs[1][i / 3][i % 3];
}

// ---- per vertex: g v_shaped = (through the da vertex) + J_regressor^T g J;  g beta = shapedirs^T g v_shaped, block partials
__global__ __launch_bounds__(256) void smpl_bw_shape_kernel(SmplDims d, const float* __restrict__ shapedirs, const float* __restrict__ j_reg,
                                                            const float* __restrict__ g_rows, const float* __restrict__ g_J,
                                                            float* __restrict__ g_beta_part) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    float gb[kMaxBetas];
#pragma unroll
    for (int l = 0; l < kMaxBetas; ++l) gb[l] = 0.f;
    if (v < d.V) {
        float g[3] = {g_rows[(size_t)v * 27 + 24], g_rows[(size_t)v * 27 + 25], g_rows[(size_t)v * 27 + 26]};
        for (int j = 0; j < d.J; ++j) {
            const float r = j_reg[(size_t)j * d.V + v];
            if (r != 0.f)
#pragma unroll
                for (int k = 0; k < 3; ++k) g[k] += r * g_J[j * 3 + k];
        }
        for (int l = 0; l < d.NB; ++l)
            gb[l] = shapedirs[((size_t)v * 3 + 0) * d.NB + l] * g[0] + shapedirs[((size_t)v * 3 + 1) * d.NB + l] * g[1] + shapedirs[((size_t)v * 3 + 2) * d.NB + l] * g[2];
    }
    __shared__ float red[4][kMaxBetas];
    for (int l = 0; l < d.NB; ++l) {
        const float t = wave_sum(gb[l]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][l] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < d.NB) g_beta_part[(size_t)blockIdx.x * d.NB + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ---- the block partials of the alignment and shape gradients, added in block order
__global__ void smpl_bw_finish_kernel(int nblocks, int NB, const float* __restrict__ g_al_part, const float* __restrict__ g_beta_part,
                                      float* __restrict__ g_align, float* __restrict__ g_beta) {
    const int t = threadIdx.x;
    if (t < 16) {
        float s = 0.f;
        for (int b = 0; b < nblocks; ++b) s += g_al_part[(size_t)b * 16 + t];
        g_align[t] = s;
    } else if (t - 16 < NB) {
        float s = 0.f;
        for (int b = 0; b < nblocks; ++b) s += g_beta_part[(size_t)b * NB + (t - 16)];
        g_beta[t - 16] = s;
    }
}

}  // namespace

struct nm_smpl_s {
    SmplDims d;
    float *v_template, *shapedirs, *j_reg, *weights, *da_pose, *ws;
    int32_t* parents;
    int ws_frames;
};

extern "C" {

int nm_smpl_destroy(nm_smpl_t m) {
    if (!m) return NM_OK;
    void* p[] = {m->v_template, m->shapedirs, m->j_reg, m->weights, m->da_pose, m->ws, m->parents};
    for (void* q : p)
        if (q) (void)hipFree(q);
    delete m;
    return NM_OK;
}

int nm_smpl_create(const float* v_template, const float* shapedirs, const float* j_regressor, const int32_t* parents, const float* lbs_weights,
                   const float* da_pose, int V, int J, int NB, nm_smpl_t* out) {
    NM_REQUIRE(v_template && shapedirs && j_regressor && parents && lbs_weights && da_pose && out, "nm_smpl_create: null pointer");
    NM_REQUIRE(V >= 1 && J >= 1 && J <= kMaxJoints && NB >= 0 && NB <= kMaxBetas, "nm_smpl_create: bad sizes V=%d J=%d NB=%d (J <= %d, NB <= %d)", V,
               J, NB, kMaxJoints, kMaxBetas);
    for (int j = 1; j < J; ++j)
        NM_REQUIRE(parents[j] >= 0 && parents[j] < j, "nm_smpl_create: parents[%d] = %d is not an earlier joint", j, parents[j]);
    nm_smpl_s* m = new nm_smpl_s();
    memset(m, 0, sizeof(*m));
    m->d = {V, J, NB};
    int rc = NM_OK;
    auto up = [&](auto** dst, const void* src, size_t bytes, const char* what) {
        if (rc) return;
        rc = nm::check_hip(hipMalloc(reinterpret_cast<void**>(dst), bytes ? bytes : 4), what);
        if (!rc && bytes) rc = nm::check_hip(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice), what);
    };
    up(&m->v_template, v_template, (size_t)V * 3 * 4, "nm_smpl_create: v_template");
    up(&m->shapedirs, shapedirs, (size_t)V * 3 * NB * 4, "nm_smpl_create: shapedirs");
    up(&m->j_reg, j_regressor, (size_t)J * V * 4, "nm_smpl_create: J_regressor");
    up(&m->weights, lbs_weights, (size_t)V * J * 4, "nm_smpl_create: lbs_weights");
    up(&m->da_pose, da_pose, (size_t)J * 3 * 4, "nm_smpl_create: da pose");
    up(&m->parents, parents, (size_t)J * 4, "nm_smpl_create: parents");
    if (rc) { nm_smpl_destroy(m); return rc; }
    *out = m;
    return NM_OK;
}

int nm_smpl_frames(nm_smpl_t m, const float* poses, const float* betas, const double* alignments, int B, double scale, int precise,
                   double* T_out, float* world_out, float* static_out, nm_stream_t stream) {
    NM_REQUIRE(m, "nm_smpl_frames: null handle");
    NM_REQUIRE(B >= 0, "nm_smpl_frames: negative batch");
    if (B == 0) return NM_OK;
    NM_REQUIRE(poses && betas && alignments && T_out && world_out && static_out, "nm_smpl_frames: null pointer");
    hipStream_t st = nm::as_stream(stream);
    if (B > m->ws_frames) {                                           // grows with the largest batch seen; frees synchronise
        if (m->ws) (void)hipFree(m->ws);
        m->ws = nullptr; m->ws_frames = 0;
        if (int rc = nm::check_hip(hipMalloc(&m->ws, (size_t)B * ws_floats(m->d.J) * 4), "nm_smpl_frames: workspace")) return rc;
        m->ws_frames = B;
    }
    hipLaunchKernelGGL(smpl_joints_kernel, dim3(B), dim3(256), 0, st, m->d, m->v_template, m->shapedirs, m->j_reg, m->parents, poses, betas,
                       m->da_pose, m->ws);
    if (int rc = nm::check_launch("smpl_joints_kernel")) return rc;
    const int64_t n = (int64_t)B * (m->d.V + m->d.J);
    hipLaunchKernelGGL(smpl_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->d, B, m->v_template, m->shapedirs, m->weights, betas,
                       alignments, scale, precise, m->ws, T_out, world_out, static_out, nullptr);
    return nm::check_launch("smpl_rows_kernel");
}

int64_t nm_smpl_vertex_workspace_floats(nm_smpl_t m) {
    if (!m) return 0;
    const int64_t blocks = (m->d.V + 255) / 256;
    // joints workspace | g_rows [V,27] | g_A [2,J,12] | g_J [J,3] | alignment partials | shape partials
    return ws_floats(m->d.J) + (int64_t)m->d.V * 27 + 2 * m->d.J * 12 + m->d.J * 3 + blocks * 16 + blocks * kMaxBetas + 64;
}

int nm_smpl_vertex_forward(nm_smpl_t m, const float* pose, const float* beta, const double* alignment, double scale, const float* da_pose,
                           float* workspace, float* world_out, float* T_out, nm_stream_t stream) {
    NM_REQUIRE(m && pose && beta && alignment && workspace && world_out && T_out, "nm_smpl_vertex_forward: null pointer");
    hipStream_t st = nm::as_stream(stream);
    hipLaunchKernelGGL(smpl_jreg_kernel, dim3(m->d.J), dim3(256), 0, st, m->d, m->v_template, m->shapedirs, m->j_reg, beta, workspace);
    if (int rc = nm::check_launch("smpl_jreg_kernel")) return rc;
    hipLaunchKernelGGL(smpl_chain_kernel, dim3(1), dim3(128), 0, st, m->d, m->parents, pose, da_pose ? da_pose : m->da_pose, workspace);
    if (int rc = nm::check_launch("smpl_chain_kernel")) return rc;
    const int64_t n = m->d.V + m->d.J;
    hipLaunchKernelGGL(smpl_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->d, 1, m->v_template, m->shapedirs, m->weights, beta,
                       alignment, scale, 0, workspace, nullptr, world_out, nullptr, T_out);
    return nm::check_launch("smpl_rows_kernel");
}

int nm_smpl_vertex_backward(nm_smpl_t m, const float* pose, const float* beta, const double* alignment, double scale, const float* da_pose,
                            const float* g_world, const float* g_T, float* workspace, float* g_pose, float* g_beta, float* g_align,
                            nm_stream_t stream) {
    NM_REQUIRE(m && pose && beta && alignment && workspace && g_pose && g_beta && g_align, "nm_smpl_vertex_backward: null pointer");
    NM_REQUIRE(g_world || g_T, "nm_smpl_vertex_backward: no upstream gradient");
    hipStream_t st = nm::as_stream(stream);
    const SmplDims d = m->d;
    const int blocks = (d.V + 255) / 256;
    float* ws = workspace;                                            // recomputed: rest joints, A of both chains
    float* g_rows = ws + ws_floats(d.J);
    float* g_A = g_rows + (size_t)d.V * 27;
    float* g_J = g_A + 2 * d.J * 12;
    float* g_al_part = g_J + d.J * 3;
    float* g_beta_part = g_al_part + (size_t)blocks * 16;
    const float* dap = da_pose ? da_pose : m->da_pose;
    hipLaunchKernelGGL(smpl_jreg_kernel, dim3(d.J), dim3(256), 0, st, d, m->v_template, m->shapedirs, m->j_reg, beta, ws);
    if (int rc = nm::check_launch("smpl_jreg_kernel")) return rc;
    hipLaunchKernelGGL(smpl_chain_kernel, dim3(1), dim3(128), 0, st, d, m->parents, pose, dap, ws);
    if (int rc = nm::check_launch("smpl_chain_kernel")) return rc;
    hipLaunchKernelGGL(smpl_bw_rows_kernel, dim3(blocks), dim3(256), 0, st, d, m->v_template, m->shapedirs, m->weights, beta, alignment, (float)scale, ws,
                       g_world, g_T, g_rows, g_al_part);
    if (int rc = nm::check_launch("smpl_bw_rows_kernel")) return rc;
    hipLaunchKernelGGL(smpl_bw_joints_kernel, dim3(2 * d.J), dim3(256), 0, st, d, m->weights, g_rows, g_A);
    if (int rc = nm::check_launch("smpl_bw_joints_kernel")) return rc;
    hipLaunchKernelGGL(smpl_bw_chain_kernel, dim3(1), dim3(64), 0, st, d, m->parents, pose, dap, ws, g_A, g_pose, g_J);
    if (int rc = nm::check_launch("smpl_bw_chain_kernel")) return rc;
    hipLaunchKernelGGL(smpl_bw_shape_kernel, dim3(blocks), dim3(256), 0, st, d, m->shapedirs, m->j_reg, g_rows, g_J, g_beta_part);
    if (int rc = nm::check_launch("smpl_bw_shape_kernel")) return rc;
    hipLaunchKernelGGL(smpl_bw_finish_kernel, dim3(1), dim3(64), 0, st, blocks, d.NB, g_al_part, g_beta_part, g_align, g_beta);
    return nm::check_launch("nm_smpl_vertex_backward");
}

}  // extern "C"
