// C[M,N] = alpha * A[M,K] * op(B) in fp32 on the f32-input matrix instruction (v_mfma_f32_32x32x2_f32): bit-for-bit a
// k-ordered fmaf chain per output element, no operand rounding.  Used only by the mean-pooling variant of the reward head
// (rw_model_general_preference.py:398-406 with the SkipCA block of :376-386 evaluated for EVERY token), which is a small,
// off-by-default part of the path (0.8 % of the FLOPs): correctness first, 128x128x16 tiles, one LDS stage.
//   B_NT: B is [N, K] row-major (C = A B^T; torch Linear weights, or the image-token rows as keys)
//   !B_NT: B is [K, N] row-major (C = A B; the image-token rows as values)
//   BT = float or unsigned short (bf16 bits: the SkipCA weights are kept in their checkpoint type).
#include "common.h"
#include "kernels.h"

namespace lr {

template <typename BT> __device__ __forceinline__ float ld_b(const BT* p);
template <> __device__ __forceinline__ float ld_b<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_b<unsigned short>(const unsigned short* p) { return bf16_bits_to_f32(*p); }

template <typename BT, bool B_NT>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, const BT* __restrict__ B, float* __restrict__ C,
                                                       int M, int N, int K, int lda, int ldb, int ldc, float alpha) {
    constexpr int BM = 128, BN = 128, BK = 16, LD = BK + 1;      // +1: the 32 rows a wave reads differ by LD words -> 32 banks
    __shared__ float sA[BM * LD], sB[BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += BK) {
        // ---- stage A [128 x 16] and op(B) [128 x 16] (zero beyond the edges) ----
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = it * 256 + tid;                 // element of the tile: row = e / 16, k = e % 16 (unit stride along k)
            const int r = e >> 4, k = e & 15;
            const int gm = m0 + r, gk = k0 + k;
            sA[r * LD + k] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.f;
        }
        if (B_NT) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int e = it * 256 + tid;
                const int r = e >> 4, k = e & 15;
                const int gn = n0 + r, gk = k0 + k;
                sB[r * LD + k] = (gn < N && gk < K) ? ld_b<BT>(B + (size_t)gn * ldb + gk) : 0.f;
            }
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int e = it * 256 + tid;             // k = e / 128, n = e % 128 (unit stride along n)
                const int k = e >> 7, r = e & 127;
                const int gn = n0 + r, gk = k0 + k;
                sB[r * LD + k] = (gn < N && gk < K) ? ld_b<BT>(B + (size_t)gk * ldb + gn) : 0.f;
            }
        }
        __syncthreads();
        // 32x32x2: lane l holds A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[(wm * 64 + i * 32 + li) * LD + kk + lk];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sB[(wn * 64 + j * 32 + li) * LD + kk + lk];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (row < M) C[(size_t)row * ldc + col] = alpha * acc[i][j][r];
            }
        }
}

void launch_gemm_f32(const float* A, const void* B, int b_is_bf16, int b_nt, float* C, int M, int N, int K, int lda, int ldb,
                     int ldc, float alpha, hipStream_t st) {
    if (M <= 0 || N <= 0) return;
    dim3 grid((N + 127) / 128, (M + 127) / 128), block(256);
    if (b_is_bf16) {
        if (b_nt) hipLaunchKernelGGL((gemm_f32_kernel<unsigned short, true>), grid, block, 0, st, A, (const unsigned short*)B, C, M, N, K, lda, ldb, ldc, alpha);
        else hipLaunchKernelGGL((gemm_f32_kernel<unsigned short, false>), grid, block, 0, st, A, (const unsigned short*)B, C, M, N, K, lda, ldb, ldc, alpha);
    } else {
        if (b_nt) hipLaunchKernelGGL((gemm_f32_kernel<float, true>), grid, block, 0, st, A, (const float*)B, C, M, N, K, lda, ldb, ldc, alpha);
        else hipLaunchKernelGGL((gemm_f32_kernel<float, false>), grid, block, 0, st, A, (const float*)B, C, M, N, K, lda, ldb, ldc, alpha);
    }
}

}  // namespace lr
