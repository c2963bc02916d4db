// Feasibility probe for an "activation-stationary" i8 MLP schedule: every wave keeps its 32 samples' K = 256 input activations in
// registers (B operands: 8 k-steps x {hi, lo} x 16 B) and streams the WEIGHT fragments (A operands) from LDS, one 1 KB fragment per limb and
// k-step, each used for ONE triple of v_mfma_i32_32x32x32_i8 (hi.lo, lo.hi -> cross accumulator; hi.hi -> hh accumulator).  Question: what
// MFMA rate do 8 waves per CU sustain when every 3 MFMAs (96 cycles of one SIMD's pipe) need 2 ds_read_b128 per lane = 85 B/clk/CU of LDS
// reads -- with nothing else in the loop?   hipcc --offload-arch=gfx950 -O3 as_probe.hip -o as_probe && ./as_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

template <int MODE>   // 0: fragments from LDS; 1: no LDS reads (fragment registers reused): the bare MFMA ceiling of this loop shape
__global__ __launch_bounds__(512, 2) void probe(int iters, int* out) {
    __shared__ uint4 lds[8192];                                    // 128 KB of "weights": 64 k-step fragments of 2 KB (hi | lo)
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 512) lds[i] = make_uint4(i * 2654435761u, i ^ 0x5bd1e995u, i * 40503u, i + 7u);
    __syncthreads();
    i32x4 xh[8], xl[8];                                            // resident activations of this wave's 32 samples, K = 256
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        xh[t] = (i32x4){(int)(tid * 31 + t), (int)(tid * 17 + t), (int)(lane + t), (int)(t * 5 + 1)};
        xl[t] = (i32x4){(int)(tid * 13 + t), (int)(tid * 7 + t), (int)(lane * 3 + t), (int)(t * 9 + 2)};
    }
    i32x16 total = {};
    const uint4* base = lds + lane;
    for (int it = 0; it < iters; ++it) {                           // one "output block": 8 k-steps, accumulators combined at the end
        i32x16 ah = {}, ac = {};
        const uint4* p = base + (it & 7) * 1024;                   // 8 blocks x 8 steps x 128 uint4 (hi: +0, lo: +64)
        uint4 wh = p[0], wl = p[64];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            uint4 nh = wh, nl = wl;
            if (MODE == 0 && t < 7) { nh = p[(t + 1) * 128]; nl = p[(t + 1) * 128 + 64]; }
            ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), xl[t], ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wl), xh[t], ac, 0, 0, 0);
            ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), xh[t], ah, 0, 0, 0);
            wh = nh; wl = nl;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) total[r] += (ah[r] << 8) + ac[r];
    }
    int s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += total[r];
    out[blockIdx.x * 512 + tid] = s;
}

int main() {
    int* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, iters, out);
            else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = 256.0 * 8 * iters * 24 * 2.0 * 32 * 32 * 32;
        printf("mode %d (%s): %.2f ms, %.0f Tops/s of i8 MFMA (peak 5000; 32x32x32 measured ceiling ~4400)\n", mode, mode ? "no LDS reads" : "A fragments from LDS", ms, ops / ms / 1e9);
    }
    return 0;
}
