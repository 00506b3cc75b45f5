// Deep-pipelined variant of gemm_bt (same contract and epilogues as gemm.hip): 256x256x64 block tile,
// 8 waves, LDS-DMA half-tile ring with counted vmcnt, raw barriers, two wave groups staggered by one
// barrier (cdna_hip_programming.md §5 "256^2 8-phase template", T3+T4+T5; MI355X_MICROARCH.md
// "Two waves per SIMD").
//
// Schedule.  A K-tile (64 deep) is 4 half-tiles of 16 KB: A0, B0, B1, A1 (128 rows x 64 k each, same
// swizzled image as gemm.hip).  Half-tile g = 4t + j lives in ring slot g % NS.  A K-tile is consumed
// in 4 phases, one block quadrant (A-half qa, B-half qb) each: (0,0) (0,1) (1,1) (1,0); in a phase every
// wave multiplies its 64x32 piece of that quadrant over the full K-tile (8 MFMA 32x32x16 or 16 MFMA
// 16x16x32), so the register fragments of one half are reused by the next phase (ds_read_b128 per
// phase: 12, 4, 8, 4).
// Phase P (global index):   LOAD(P): ds_reads for P's MFMAs; issue half-tile P+PF by LDS-DMA;
//                                    s_waitcnt vmcnt(2*(PF-2))  -> everything up to half-tile P+2 landed
//                           barrier; COMPUTE(P): MFMAs; barrier.
// Hazards: half-tile g is first read at phase >= g-2, i.e. one phase after the wait that retires it
// (RAW); its slot is restaged by half-tile g+NS, issued at phase g+NS-PF, while its last read is at
// phase <= g+2 of the lagging wave group (WAR) => NS - PF >= 4.  NS = 10 slots = all 160 KB of LDS.
// Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while
// its partner issues LDS reads / DMA.
//
// Column mapping inside a block tile: wave wc owns columns wc*64 + qb*32 + [0,32) for qb = 0,1, so the
// SwiGLU pair (gate block, up block: weight rows interleaved in 32s) stays in one lane/register.
#include "common.h"
#include "kernels.h"

namespace lr {

#define LR_BARRIER() do { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <typename OT>
__device__ __forceinline__ void epi_store(const GemmParams& p, int row, int col, float v) {
    const size_t o = (size_t)row * p.ldc + col;
    if (p.epi == EPI_OUT_OP) {
        if (p.act == ACT_QUICK_GELU) v = v / (1.f + expf(-1.702f * v));
        else if (p.act == ACT_GELU_ERF) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        ((unsigned short*)p.C)[o] = Op<OT>::from_f32(v);
    } else if (p.epi == EPI_OUT_F32) {
        ((float*)p.C)[o] = v;
    } else {
        ((float*)p.C)[o] += v;
    }
}

// M16: 16x16x32 MFMAs (16 per phase) instead of 32x32x16 (8 per phase).
// DBG: 0 = product; 1 = every K-tile re-reads K-tile 0 (cache-resident operands: LDS+MFMA ceiling; results invalid).
template <typename OT, int PF, int NS, bool M16, int DBG>
__global__ __launch_bounds__(512) void gemm_bt8_kernel(GemmParams p) {
    constexpr int BM = 256, BN = 256, BK = 64;
    constexpr int HT = 16384;                      // bytes per half-tile slot
    static_assert(NS - PF >= 4 && PF >= 3, "ring hazard distances");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- XCD-aware tile mapping (as gemm.hip) ----
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

    // ---- LDS-DMA source pointers: [half][it]; swizzle on the source side ----
    const unsigned short* gA[2][2];
    const unsigned short* gB[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = it * 512 + tid;
        const int R = q >> 4, Cp = q & 15;
        const int C = Cp ^ (R & 15);
        const int row = 2 * R + (C >> 3), c = C & 7;            // row of the 128-row half-tile image
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ga = min(m0 + h * 128 + row, p.M - 1);
            gA[h][it] = (const unsigned short*)p.A + (size_t)ga * p.lda + c * 8;
            const int wrow = (row >> 5) * 64 + h * 32 + (row & 31);   // image row -> tile column
            const int gb = min(n0 + wrow, p.N - 1);
            gB[h][it] = (const unsigned short*)p.W + (size_t)gb * p.ldw + c * 8;
        }
    }
    const int nk = p.K / BK;
    const int Gtot = 4 * nk;

    // half-tile j of a K-tile: 0 = A0, 1 = B0, 2 = B1, 3 = A1.
    // The LDS-DMA is issued from inline asm on purpose: hipcc's waitcnt pass would otherwise put
    // `s_waitcnt vmcnt(0)` in front of every ds_read (it cannot prove the pending DMA does not alias)
    // and drain the ring each phase.  Ordering is ours: counted vmcnt, then a barrier, then the read
    // (cdna_hip_programming.md §5.7 item 1).  M0 is saved/restored inside the statement.
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;
    auto issue = [&](int j, int kt, int slot) {
        const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds_base + slot * HT + wave * 1024);
        const int koff = DBG == 1 ? 0 : kt * BK;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const unsigned short* src = ((j == 0) ? gA[0][it] : (j == 1) ? gB[0][it] : (j == 2) ? gB[1][it] : gA[1][it]) + koff;
            const unsigned dst = dst0 + it * 8192;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };

    // ---- fragment read offsets inside a half-tile image: 8 A reads + 4 B reads per phase ----
    // 32x32x16: lane (r = lane&31, h = lane>>5) reads row r, 16-byte chunk 2*ks + h      (ks = 0..3)
    // 16x16x32: lane (r = lane&15, q = lane>>4) reads row r, 16-byte chunk 4*ks + q      (ks = 0..1)
    int aoff[8], boff[4];
    auto img_off = [](int row, int kc) { const int R = row >> 1; return R * 256 + (((((row & 1) << 3) + kc) ^ (R & 15)) << 4); };
    if constexpr (M16) {
        const int r16 = lane & 15, q4 = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) aoff[i * 2 + ks] = img_off(wr * 64 + i * 16 + r16, 4 * ks + q4);
#pragma unroll
            for (int j = 0; j < 2; ++j) boff[j * 2 + ks] = img_off(wc * 32 + j * 16 + r16, 4 * ks + q4);
        }
    } else {
        const int r32 = lane & 31, h2 = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) aoff[i * 4 + ks] = img_off(wr * 64 + i * 32 + r32, 2 * ks + h2);
            boff[ks] = img_off(wc * 32 + r32, 2 * ks + h2);
        }
    }

    f32x16 acc32[4][2];       // [quadrant (0,0) (0,1) (1,1) (1,0)][row tile of 32]
    f32x4 acc16[4][4][2];     // [quadrant][row tile of 16][col tile of 16]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[q][i][r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc16[q][i][j][r] = 0.f;
    }

    // ---- prologue: PF half-tiles in flight, the first two landed ----
    int islot = 0;         // ring slot of the next half-tile to issue
#pragma unroll
    for (int g = 0; g < PF; ++g) {
        if (g < Gtot) issue(g & 3, g >> 2, islot);
        islot = (islot + 1 == NS) ? 0 : islot + 1;
    }
    if (Gtot > PF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PF - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LR_BARRIER();
    if (wr == 1) LR_BARRIER();                      // stagger the second wave group by one barrier

    uint4 af[8], bf[4];
    int rslot = 0;         // ring slot of half-tile A0 of the current K-tile
    int gi = PF;           // index of the next half-tile to issue

    for (int kt = 0; kt < nk; ++kt) {
        int s1 = rslot + 1; s1 = s1 >= NS ? s1 - NS : s1;
        int s2 = rslot + 2; s2 = s2 >= NS ? s2 - NS : s2;
        int s3 = rslot + 3; s3 = s3 >= NS ? s3 - NS : s3;
        const char* sA0 = smem + rslot * HT;
        const char* sB0 = smem + s1 * HT;
        const char* sB1 = smem + s2 * HT;
        const char* sA1 = smem + s3 * HT;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            // ---------------- LOAD ----------------
            if (ph == 0 || ph == 1 || ph == 3) {
                const char* sb = (ph == 1) ? sB1 : sB0;
#pragma unroll
                for (int f = 0; f < 4; ++f) bf[f] = *(const uint4*)(sb + boff[f]);
            }
            if (ph == 0 || ph == 2) {
                const char* sa = (ph == 0) ? sA0 : sA1;
#pragma unroll
                for (int f = 0; f < 8; ++f) af[f] = *(const uint4*)(sa + aoff[f]);
            }
            if (gi < Gtot) {
                issue((ph + PF) & 3, kt + ((ph + PF) >> 2), islot);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PF - 2)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            ++gi;
            islot = (islot + 1 == NS) ? 0 : islot + 1;
            LR_BARRIER();
            // ---------------- COMPUTE ----------------
            __builtin_amdgcn_s_setprio(1);
            if constexpr (M16) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc16[ph][i][j] = Op<OT>::mfma16(af[i * 2 + ks], bf[j * 2 + ks], acc16[ph][i][j]);
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc32[ph][i] = Op<OT>::mfma32(af[i * 4 + ks], bf[ks], acc32[ph][i]);
            }
            __builtin_amdgcn_s_setprio(0);
            LR_BARRIER();
        }
        rslot += 4;
        rslot = rslot >= NS ? rslot - NS : rslot;
    }
    if (wr == 0) LR_BARRIER();                      // balance the stagger barrier

    // ---- epilogue ----  quadrant q -> (qa, qb): 0:(0,0) 1:(0,1) 2:(1,1) 3:(1,0)
    // 32x32 C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5);  16x16: col = lane&15, row = 4*(lane>>4) + r
    const int rbase = m0 + wr * 64, cbase = n0 + wc * 64;
    if (p.epi == EPI_SWIGLU_OP) {
        unsigned short* C = (unsigned short*)p.C;
        if (cbase + 64 <= p.N) {
#pragma unroll
            for (int qa = 0; qa < 2; ++qa) {
                const int qg = qa == 0 ? 0 : 3, qu = qa == 0 ? 1 : 2;
                if constexpr (M16) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = rbase + qa * 128 + i * 16 + 4 * (lane >> 4) + r;
                                const int col = (cbase >> 1) + j * 16 + (lane & 15);
                                if (row < p.M) {
                                    const float g = acc16[qg][i][j][r], u = acc16[qu][i][j][r];
                                    C[(size_t)row * p.ldc + col] = Op<OT>::from_f32(u * (g / (1.f + expf(-g))));
                                }
                            }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = rbase + qa * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            const int col = (cbase >> 1) + (lane & 31);
                            if (row < p.M) {
                                const float g = acc32[qg][i][r], u = acc32[qu][i][r];
                                C[(size_t)row * p.ldc + col] = Op<OT>::from_f32(u * (g / (1.f + expf(-g))));
                            }
                        }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int qa = (q >= 2) ? 1 : 0, qb = (q == 1 || q == 2) ? 1 : 0;
        if constexpr (M16) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = cbase + qb * 32 + j * 16 + (lane & 15);
                if (col >= p.N) continue;
                const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rbase + qa * 128 + i * 16 + 4 * (lane >> 4) + r;
                        if (row < p.M) epi_store<OT>(p, row, col, acc16[q][i][j][r] + bv);
                    }
            }
        } else {
            const int col = cbase + qb * 32 + (lane & 31);
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + qa * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (row < p.M) epi_store<OT>(p, row, col, acc32[q][i][r] + bv);
                }
        }
    }
}

template <typename OT, int PF, bool M16, int DBG>
static void launch8(const GemmParams& p, hipStream_t st) {
    constexpr int NS = 10;
    constexpr int smem = NS * 16384;
    static bool attr_set = false;
    auto kfn = gemm_bt8_kernel<OT, PF, NS, M16, DBG>;
    if (!attr_set) {
        LR_HIP_CHECK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int Mt = (p.M + 255) / 256, Nt = (p.N + 255) / 256;
    hipLaunchKernelGGL(kfn, dim3(Mt * Nt), dim3(512), smem, st, p);
}

template <typename OT>
static void launch8_variant(const GemmParams& p, int variant, hipStream_t st) {
    switch (variant) {
        case 3: launch8<OT, 5, false, 0>(p, st); break;
        case 4: launch8<OT, 6, false, 0>(p, st); break;
        case 5: launch8<OT, 5, true, 0>(p, st); break;
        case 6: launch8<OT, 6, true, 0>(p, st); break;
        case 7: launch8<OT, 6, true, 1>(p, st); break;      // diagnostic only
        default: throw std::runtime_error("gemm_bt8: unknown variant");
    }
}

void launch_gemm_bt8(const GemmParams& p, int operand_dtype, int variant, hipStream_t st) {
    if (operand_dtype == DT_F16) launch8_variant<F16>(p, variant, st);
    else launch8_variant<BF16>(p, variant, st);
}

}  // namespace lr
