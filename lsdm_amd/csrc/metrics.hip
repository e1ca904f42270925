// Evaluation-side kernel of the CLI (run/test_sdm.py:185-196): nearest-neighbour distances between two point sets.
//
//   nn_dist_kernel   for every point of A the squared Euclidean distance to (and the index of) its nearest point of B,
//                    d^2 = (ax-bx)^2 + (ay-by)^2 + (az-bz)^2 in fp32 -- the K = 1 nearest-neighbour search underneath
//                    pytorch3d.loss.chamfer_distance (run/test_sdm.py:186) and open3d's compute_point_cloud_distance
//                    (util/evaluation.py:37-38, F-score).  Ties keep the lowest index.
//
// HBM-bound by construction (24 B per point pair set, each read once); B is staged through LDS in 1024-point tiles so
// that a cloud pair is read from HBM once per 256 queries (L2 serves the rest).
#include "common.hpp"
#include "kernels.hpp"

namespace lsdm {

constexpr int kNNTile = 1024;

__global__ __launch_bounds__(256) void nn_dist_kernel(const float* __restrict__ A, int na, const float* __restrict__ Bp, int nb,
                                                      float* __restrict__ out_d2, int* __restrict__ out_idx) {
    __shared__ float sb[kNNTile * 3];
    const int batch = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
    const float* a = A + (size_t)batch * na * 3;
    const float* b = Bp + (size_t)batch * nb * 3;
    const bool live = q < na;
    const float ax = live ? a[q * 3] : 0.0f, ay = live ? a[q * 3 + 1] : 0.0f, az = live ? a[q * 3 + 2] : 0.0f;
    float best = __builtin_inff();
    int besti = 0;
    for (int j0 = 0; j0 < nb; j0 += kNNTile) {
        const int cnt = nb - j0 < kNNTile ? nb - j0 : kNNTile;
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 3; e += 256) sb[e] = b[(size_t)j0 * 3 + e];   // coalesced
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < cnt; ++j) {                                                   // broadcast LDS reads
            const float dx = ax - sb[j * 3], dy = ay - sb[j * 3 + 1], dz = az - sb[j * 3 + 2];
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (d < best) { best = d; besti = j0 + j; }
        }
    }
    if (live) {
        out_d2[(size_t)batch * na + q] = best;
        if (out_idx) out_idx[(size_t)batch * na + q] = besti;
    }
}

void launch_nn_dist(const float* A, int na, const float* B, int nb, int batch, float* out_d2, int* out_idx, hipStream_t st) {
    hipLaunchKernelGGL(nn_dist_kernel, dim3((na + 255) / 256, batch), dim3(256), 0, st, A, na, B, nb, out_d2, out_idx);
}

}  // namespace lsdm
