// C[M,N] = A[M,K] * W[N,K]^T on the gfx950 matrix cores, fp32 accumulate, fused epilogues.
//
// This one kernel carries every dense contraction of the scoring path: CLIP
// q/k/v/out/fc1/fc2 + patch embedding, the HD projector, and the Phi-3 qkv/o/gate_up/down
// projections (reference call sites: modeling_phi3_v.py:654,715,567,572 and transformers CLIP).
//
// Structure (cdna_hip_programming.md §5, "minimum 2-phase" form of T3):
//   * block tile BM x BN x 64, WM x WN waves, each wave (BM/WM) x (BN/WN) as 32x32x16 MFMA tiles;
//   * both operand tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip),
//     two LDS stages, the load of K-tile t+1 in flight while tile t is multiplied, one barrier per tile;
//   * LDS image: rows of 64 elements (128 B) are paired into 256-B bank rows and the 16-byte chunk
//     index is XORed with (pair & 15).  LDS-DMA writes linearly (base + lane*16), so the swizzle is
//     applied to the per-lane SOURCE address and again on the ds_read_b128 side (rule 21);
//     every 16-lane ds_read_b128 group then touches 16 distinct 16-B slots: conflict-free.
//   * blockIdx -> tile map is XCD-aware: the 8 XCDs (round-robin over blockIdx) each walk a
//     contiguous chunk of tiles, ordered as 8-row bands so co-resident blocks share A rows and
//     W rows in that XCD's L2.
//   * no atomics, fixed K order: a row's result does not depend on M or on the launch geometry
//     (batch-invariant rewards, SURVEY.md §7 "preference ordering must be bit-exact").
#include "common.h"
#include "kernels.h"

namespace lr {

template <typename OT, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bt_kernel(GemmParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int BK = 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 32, NI = TN / 32;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int A_LOADS = BM * 8 / NT, B_LOADS = BN * 8 / NT;
    static_assert(A_LOADS * NT == BM * 8 && B_LOADS * NT == BN * 8, "tile/threads mismatch");
    static_assert(NI % 2 == 0, "SwiGLU epilogue pairs adjacent 32-column tiles");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- XCD-aware tile mapping (bijective for any tile count) ----
    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + BN - 1) / BN;
    const int nwg = Mt * Nt;
    int L;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GM = 8;
    const int band = L / (GM * Nt);
    const int within = L - band * (GM * Nt);
    const int rows_in_band = min(GM, Mt - band * GM);
    const int mi = band * GM + within % rows_in_band;
    const int ni = within / rows_in_band;
    const int m0 = mi * BM, n0 = ni * BN;

    // ---- per-thread LDS-DMA source pointers (swizzle lives on the source side) ----
    const unsigned short* gA[A_LOADS];
    const unsigned short* gB[B_LOADS];
#pragma unroll
    for (int it = 0; it < A_LOADS; ++it) {
        const int q = it * NT + tid;
        const int R = q >> 4, Cp = q & 15;
        const int C = Cp ^ (R & 15);
        const int row = 2 * R + (C >> 3), c = C & 7;
        const int grow = min(m0 + row, p.M - 1);
        gA[it] = (const unsigned short*)p.A + (size_t)grow * p.lda + c * 8;
    }
#pragma unroll
    for (int it = 0; it < B_LOADS; ++it) {
        const int q = it * NT + tid;
        const int R = q >> 4, Cp = q & 15;
        const int C = Cp ^ (R & 15);
        const int row = 2 * R + (C >> 3), c = C & 7;
        const int grow = min(n0 + row, p.N - 1);
        gB[it] = (const unsigned short*)p.W + (size_t)grow * p.ldw + c * 8;
    }

    auto stage = [&](int buf, int kt) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
        int koff = kt * BK, koffw = koff;
        ptrdiff_t wsel = 0;
        if (p.kw > 0) {                          // split-operand mode: A = [hi | lo (| hi)], W repeats along K (, then its residuals)
            if (koff >= 2 * p.kw) { koff -= 2 * p.kw; koffw = koff; wsel = (const unsigned short*)p.Wlo - (const unsigned short*)p.W; }
            else if (koff >= p.kw) koffw = koff - p.kw;
        }
#pragma unroll
        for (int it = 0; it < A_LOADS; ++it) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA[it] + koff),
                                             (__attribute__((address_space(3))) void*)(sA + (it * NT + wave * 64) * 16),
                                             16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < B_LOADS; ++it) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB[it] + koffw + wsel),
                                             (__attribute__((address_space(3))) void*)(sB + (it * NT + wave * 64) * 16),
                                             16, 0, 0);
        }
    };

    // ---- fragment read addresses ----
    const int lr_ = lane & 31, lh = lane >> 5;
    int aR[MI], aP[MI], bR[NI], bP[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = wm * TM + i * 32 + lr_;
        aR[i] = row >> 1;
        aP[i] = (row & 1) * 8;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int row = wn * TN + j * 32 + lr_;
        bR[j] = row >> 1;
        bP[j] = (row & 1) * 8;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sA = smem + cur * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 af[MI], bf[NI];
            const int kc = 2 * ks + lh;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                af[i] = *(const uint4*)(sA + aR[i] * 256 + (((aP[i] + kc) ^ (aR[i] & 15)) << 4));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                bf[j] = *(const uint4*)(sB + bR[j] * 256 + (((bP[j] + kc) ^ (bR[j] & 15)) << 4));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = Op<OT>::mfma32(af[i], bf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 tile is col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    const int rbase = m0 + wm * TM + 4 * lh;
    const int cbase = n0 + wn * TN + lr_;
    if (p.epi == EPI_SWIGLU_OP) {
        unsigned short* C = (unsigned short*)p.C;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int jj = 0; jj < NI / 2; ++jj) {
                const int col = ((n0 + wn * TN) >> 1) + jj * 32 + lr_;
                if (n0 + wn * TN + jj * 64 + 32 + lr_ >= p.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
                    if (row < p.M) {
                        float g = acc[i][2 * jj][r], u = acc[i][2 * jj + 1][r];
                        if (p.bias) { const int pc = n0 + wn * TN + jj * 64 + lr_; g += p.bias[pc]; u += p.bias[pc + 32]; }
                        const float v = u * (g / (1.f + expf(-g)));
                        const unsigned short hv = Op<OT>::from_f32(v);
                        C[(size_t)row * p.ldc + col] = hv;
                        if (p.split > 0) C[(size_t)row * p.ldc + p.split + col] = Op<OT>::from_f32(v - Op<OT>::to_f32(hv));
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = cbase + j * 32;
        if (col >= p.N) continue;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= p.M) continue;
                float v = acc[i][j][r] + bv;
                const size_t o = (size_t)row * p.ldc + col;
                if (p.epi == EPI_OUT_OP) {
                    if (p.act == ACT_QUICK_GELU) v = v / (1.f + expf(-1.702f * v));
                    else if (p.act == ACT_GELU_ERF) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
                    const unsigned short hv = Op<OT>::from_f32(v);
                    ((unsigned short*)p.C)[o] = hv;
                    if (p.split > 0) ((unsigned short*)p.C)[o + p.split] = Op<OT>::from_f32(v - Op<OT>::to_f32(hv));
                } else if (p.epi == EPI_OUT_F32) {
                    ((float*)p.C)[o] = v;
                } else {
                    ((float*)p.C)[o] += v;
                }
            }
        }
    }
}

template <typename OT, int BM, int BN, int WM, int WN>
static void launch_cfg(const GemmParams& p, hipStream_t st) {
    constexpr int smem = 2 * (BM + BN) * 64 * 2;
    static bool attr_set = false;
    auto kfn = gemm_bt_kernel<OT, BM, BN, WM, WN>;
    if (!attr_set) {
        LR_HIP_CHECK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(kfn, dim3(Mt * Nt), dim3(WM * WN * 64), smem, st, p);
}

template <typename OT>
static void launch_dt(const GemmParams& p, int tile, hipStream_t st) {
    switch (tile) {
        case 0: launch_cfg<OT, 128, 128, 2, 2>(p, st); break;
        case 1: launch_cfg<OT, 256, 128, 4, 2>(p, st); break;
        default: launch_cfg<OT, 256, 256, 2, 4>(p, st); break;
    }
}

static int pick_tile(const GemmParams& p, int tile) {
    if (tile >= 0) return tile;
    // The deep-pipelined 256x256 kernel for every weight matrix wide and deep enough to give it a tile.  The choice looks at the
    // WEIGHT's shape (N, K) and at alignment only, never at M: the two kernels sum in different orders, and a row's reward must be
    // bit-identical whatever else is in the batch or however the batch is sharded (every operand mode, not only the default one).
    const bool aligned = p.N % 8 == 0 && p.ldc % 8 == 0 && (((uintptr_t)p.C) & 15) == 0 && (!p.bias || (((uintptr_t)p.bias) & 15) == 0);
    return (aligned && p.N >= 256 && p.K >= 128) ? 6 : 0;
}

bool gemm_bt_is_deep(const GemmParams& p, int tile) { return pick_tile(p, tile) >= 3; }

void launch_gemm_bt(const GemmParams& p, int operand_dtype, int tile, hipStream_t st) {
    if (p.M <= 0) return;
    if (p.aexp) { launch_gemm_bt8_mixed(p, operand_dtype, st); return; }      // split-operand mode with the e4m3 residual pass
    if (p.K % 64 != 0) throw std::runtime_error("gemm_bt: K must be a multiple of 64");
    if (p.kw < 0 || (p.kw > 0 && (p.kw % 64 || p.K != (p.Wlo ? 3 : 2) * p.kw)))
        throw std::runtime_error("gemm_bt: split-operand mode needs K == 2 kw (3 kw with Wlo), kw % 64 == 0");
    if (p.split < 0 || p.split % 8) throw std::runtime_error("gemm_bt: split must be a non-negative multiple of 8");
    if (p.epi == EPI_SWIGLU_OP && (p.N % 64) != 0) throw std::runtime_error("gemm_bt: SwiGLU needs N % 64 == 0");
    tile = p.A2 ? 6 : pick_tile(p, tile);          // a K-extension (un-merged adapter) exists in the deep-pipelined kernel only
    if (p.oexp && (tile != 6 || p.split <= 0 || p.split % 128 || !(p.epi == EPI_OUT_OP || p.epi == EPI_SWIGLU_OP)))
        throw std::runtime_error("gemm_bt: one-byte residual output needs the product kernel, a split operand output and columns % 128 == 0");
    if (tile >= 3) { launch_gemm_bt8(p, operand_dtype, tile, st); return; }
    if (p.epi == EPI_ROPE_OP) throw std::runtime_error("gemm_bt: the fused RoPE epilogue exists only in the deep-pipelined kernel");
    if (p.oexp) throw std::runtime_error("gemm_bt: one-byte residual output exists only in the deep-pipelined kernel");
    if (operand_dtype == DT_F16) launch_dt<F16>(p, tile, st);
    else launch_dt<BF16>(p, tile, st);
}

}  // namespace lr
