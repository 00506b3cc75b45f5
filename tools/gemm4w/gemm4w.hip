// EXPERIMENT (round 4, not part of the product library): the 256 x 256 x 64 f16 GEMM main loop as ONE wave per SIMD --
// 4 waves x 512 unified registers, each wave a 128 x 128 block of the tile (64 accumulator tiles of 16 x 16 = 256 AGPRs) --
// instead of the product kernel's 8 waves x 256 registers with two wave groups alternating LOAD / COMPUTE behind barriers
// (csrc/gemm8.hip).  Same LDS half-tile images, same LDS-DMA ring discipline (counted vmcnt + raw barrier, source-side swizzle).
// What it is meant to show: how much of the product K loop's 29 % idle MFMA issue time is the barrier / role-flip structure.
//   per K-tile and wave: 32 ds_read_b128 (A half 128 x 64, B half 128 x 64: every fragment feeds 8 MFMAs, the product's feed 4),
//   128 MFMA 16x16x32, 16 LDS-DMA instructions, ONE workgroup barrier.
// Schedule (software-pipelined over K-tiles, two fragment register sets):
//   iteration t:  MFMAs of row tiles 0..6 of K-tile t   (fragments already in registers)    + LDS-DMA issue of half-tiles 4t+6 .. 4t+9
//                 s_waitcnt vmcnt(8); s_barrier          (K-tile t+1 has landed for every wave; K-tile t's slots are free)
//                 ds_reads of K-tile t+1's fragments into the OTHER register set
//                 MFMAs of row tile 7 of K-tile t        (cover the latency of those reads)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/gemm4w/gemm4w.hip -o tools/gemm4w/libgemm4w.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct G4Params {
    const unsigned short* A;   // [M, lda] f16
    const unsigned short* W;   // [N, ldw] f16   (C = A W^T)
    float* C;                  // [M, ldc] fp32
    int M, N, K, lda, ldw, ldc;
    // diagnostic (may be null): per workgroup, summed over its tiles, {shader cycles (s_memtime), 100 MHz ticks (s_memrealtime), K-tiles}
    // spent inside the K loop -- written to a buffer nothing else reads (MI355X_MICROARCH.md 'DVFS give-back' item 6)
    unsigned long long* stamps;
};

typedef int i32x4 __attribute__((ext_vector_type(4)));
// The accumulators are PINNED to the AGPR half of the register file ("+a"): with the builtin, hipcc (ROCm 7.2) allocates the 64
// accumulator tiles across both halves and shuffles them through v_accvgpr_read / write around every MFMA (468 moves per two K-tiles).
__device__ __forceinline__ void mfma16(f32x4& c, const uint4 a, const uint4 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(__builtin_bit_cast(i32x4, a)), "v"(__builtin_bit_cast(i32x4, b)));
}

#define G4_BARRIER() do { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// MODE 0: fp32 result stored straight from the C/D layout (correctness);  1: no epilogue (main loop alone; results not written)
// SCHED 0: one barrier per K-tile, ring of 10 half-tile slots: the B halves of K-tile kt+1 and the A halves of kt+2 are issued during kt
//          (a K-tile's slots are free only behind the NEXT barrier, so LDS holds 2.5 K-tiles and the B halves get < 1 K-tile of lead).
// SCHED 1: a second barrier behind the fragment reads frees a K-tile's slots at once: two K-tile buffers, ALL of K-tile kt+2 is issued
//          during kt and has a full K-tile of lead.
// MODE bit 1 (diagnostic): every K-tile re-reads K-tile 0 (cache-resident operands; results invalid).
template <int MODE, int SCHED>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(G4Params p) {
    constexpr int BM = 256, BN = 256, BK = 64, HT = 16384, NS = SCHED ? 8 : 10;
    constexpr bool RESIDENT = (MODE & 2) != 0, LEAN = SCHED >= 2;      // SCHED 2 = SCHED 1 with the lean LDS-DMA sequence
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;

    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + BN - 1) / BN, nwg = Mt * Nt;
    const int nk = p.K / BK;

    // fragment read offsets inside a half-tile image (the image of csrc/gemm.hip: LDS row R holds image rows 2R, 2R+1, 16-byte chunk
    // position = chunk ^ (R & 15)): row tile i, k-step ks -> i * 2048 + foff[i & 1][ks]
    int foff[2][2];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = par * 16 + l15, kc = 4 * ks + l4, R = row >> 1;
            foff[par][ks] = (R & 7) * 256 + (((((row & 1) << 3) + kc) ^ (R & 15)) << 4);
        }

    // LDS-DMA pieces: piece q = it * 256 + tid of a half-tile -> LDS byte q * 16 (lane-linear), source row 2R + (C >> 3), chunk C & 7
    // with R = q >> 4, C = (q & 15) ^ (R & 15).  R & 15 = tid >> 4 for every it, so row(it) = row0 + 32 it.
    const int Rl = tid >> 4, Cc = (tid & 15) ^ Rl;
    const int row0 = 2 * Rl + (Cc >> 3), c8 = (Cc & 7) * 8;

    const int chq = nwg >> 3, chr = nwg & 7;
    auto chunk0 = [&](int x) { return x < chr ? x * (chq + 1) : chr * (chq + 1) + (x - chr) * chq; };

    unsigned long long cyc = 0, ticks = 0, kts = 0;
    for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
        const int L = chunk0(vb & 7) + (vb >> 3);
        constexpr int GM = 8;
        const int band = L / (GM * Nt), within = L - band * (GM * Nt);
        const int rows_in_band = min(GM, Mt - band * GM);
        const int mi = band * GM + within % rows_in_band, ni = within / rows_in_band;
        const int m0 = mi * BM, n0 = ni * BN;

        // LDS-DMA: global_load_lds with a SCALAR base (half-tile, piece and K-tile) + one 32-bit per-thread offset per operand: no
        // 64-bit per-thread pointers live across the K loop (16 of them would cost 32 VGPRs the fragment sets need).
        // M % 256 == N % 256 == 0 in this experiment (no row clamps).
        const unsigned voffA = (unsigned)(row0 * p.lda + c8) * 2u, voffB = (unsigned)(row0 * p.ldw + c8) * 2u;
        const char* tileA = (const char*)p.A + (size_t)m0 * p.lda * 2;
        const char* tileB = (const char*)p.W + (size_t)n0 * p.ldw * 2;
        int islot = 0;                            // ring slot of the next half-tile to issue
        auto issue1 = [&](int j, int ktc, int it) {       // piece `it` of half-tile j (0: A0, 1: A1, 2: B0, 3: B1) of K-tile ktc -> the next ring slot
            const int h = j & 1;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + islot * HT + it * 4096 + wave * 1024);
            // image row row0 + 32 it: A row h * 128 + that; W row (tile column) it * 64 + h * 32 + row0 (the product's SwiGLU-friendly map)
            if (RESIDENT) ktc = 0;
            const char* base = j < 2 ? tileA + ((size_t)(h * 128 + 32 * it) * p.lda + (size_t)ktc * BK) * 2
                                     : tileB + ((size_t)(it * 64 + h * 32) * p.ldw + (size_t)ktc * BK) * 2;
            if constexpr (LEAN) {          // M0 is not saved / restored (nothing else in this kernel reads it) and no s_nop: 2 instructions per piece
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(j < 2 ? voffA : voffB), "s"(base), "s"(dst) : "memory", "m0");
            } else {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(j < 2 ? voffA : voffB), "s"(base), "s"(dst) : "memory");
            }
            if (it == 3) islot = islot + 1 == NS ? 0 : islot + 1;
        };
        auto issue = [&](int j, int ktc) { issue1(j, ktc, 0); issue1(j, ktc, 1); issue1(j, ktc, 2); issue1(j, ktc, 3); };

        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- prologue ----
        issue(0, 0); issue(1, 0); issue(2, 0); issue(3, 0);
        issue(0, 1); issue(1, 1);                 // (nk >= 2)
        if (SCHED) { issue(2, 1); issue(3, 1); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        G4_BARRIER();
        int rslot = 0;                            // ring slot of half-tile A0 of the K-tile whose fragments are read next

        // fragments: A row tiles 0..6 of the current K-tile (re-loaded in place), row tile 7 and B in two sets used in turn (their
        // MFMAs run after the barrier, while the next K-tile's fragments are on their way)
        uint4 a[7][2], a7x[2], a7y[2], b0[8][2], b1[8][2];
        auto load_a = [&](int kt, uint4 (&a7)[2]) {
            int sl = rslot + wr; sl = sl >= NS ? sl - NS : sl;             // half-tiles of a K-tile: A0, A1, B0, B1
            const char* sa = smem + sl * HT;
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[i][ks] = *(const uint4*)(sa + i * 2048 + foff[i & 1][ks]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a7[ks] = *(const uint4*)(sa + 7 * 2048 + foff[1][ks]);
        };
        auto load_b = [&](int kt, uint4 (&b)[8][2]) {
            int sl = rslot + 2 + wc; sl = sl >= NS ? sl - NS : sl;
            const char* sb = smem + sl * HT;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b[i][ks] = *(const uint4*)(sb + i * 2048 + foff[i & 1][ks]);
        };
        auto ktile = [&](int kt, const uint4 (&b)[8][2], const uint4 (&a7)[2], uint4 (&bn)[8][2], uint4 (&a7n)[2]) {
            // main part: row tiles 0..6 (112 MFMAs); this iteration's LDS-DMA (B halves of K-tile kt+1, A halves of kt+2: 16 instructions)
            // is spread over it, one instruction behind every 6 MFMAs of the first 96 -- the MFMAs are `asm volatile`, so this IS the
            // issue order.  Past the end of K: a harmless re-load of the last K-tile into free slots (no branch in the K loop).
            const int k1 = min(kt + 1, nk - 1), k2 = min(kt + 2, nk - 1);
            int n = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        mfma16(acc[i][j], a[i][ks], b[j][ks]);
                        ++n;
                        if (n % 6 == 0 && n <= 96) {
                            const int d = n / 6 - 1;              // 0..15: half-tile d >> 2 of this iteration, piece d & 3
                            if (SCHED) issue1(d >> 2, k2, d & 3);                                      // all of K-tile kt+2
                            else issue1(d < 4 ? 2 : d < 8 ? 3 : d < 12 ? 0 : 1, d < 8 ? k1 : k2, d & 3);
                        }
                    }
            __builtin_amdgcn_sched_barrier(0);
            if (SCHED) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // everything but this iteration's 16: K-tile kt+1 has landed
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            G4_BARRIER();
            rslot = rslot + 4 >= NS ? rslot + 4 - NS : rslot + 4;
            load_b(kt + 1, bn);            // (the last K-tile reads a stale slot: valid LDS, never used)
            load_a(kt + 1, a7n);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) mfma16(acc[7][j], a7[ks], b[j][ks]);      // row tile 7: covers the latency of those reads
            __builtin_amdgcn_sched_barrier(0);
            if (SCHED) {       // every wave holds K-tile kt+1 in registers: its slots are free for the next iteration's LDS-DMA
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                G4_BARRIER();
            }
        };

        load_b(0, b0);
        load_a(0, a7x);
        if (SCHED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); G4_BARRIER(); }
        unsigned long long t0 = 0, r0 = 0;
        if (p.stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            ktile(kt, b0, a7x, b1, a7y);
            ktile(kt + 1, b1, a7y, b0, a7x);
        }
        if (kt < nk) ktile(kt, b0, a7x, b1, a7y);
        if (p.stamps) { cyc += __builtin_amdgcn_s_memtime() - t0; ticks += __builtin_amdgcn_s_memrealtime() - r0; kts += nk; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        if constexpr ((MODE & 1) == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
        } else {
            // C/D map of 16x16x32: col = lane & 15, row = 4 (lane >> 4) + r
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int jrow = j * 16 + l15;
                    const int col = n0 + (jrow >> 5) * 64 + wc * 32 + (jrow & 31);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + wr * 128 + i * 16 + 4 * l4 + r;
                        if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = acc[i][j][r];
                    }
                }
        }
        __syncthreads();          // every wave is out of the ring before the next tile's prologue refills it
    }
    if (p.stamps && tid == 0) { p.stamps[3 * blockIdx.x] = cyc; p.stamps[3 * blockIdx.x + 1] = ticks; p.stamps[3 * blockIdx.x + 2] = kts; }
}

template <int MODE, int SCHED>
static void g4_go(const G4Params& p, int grid, hipStream_t st) {
    constexpr int smem = 10 * 16384;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gemm4w_kernel<MODE, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); attr = true; }
    hipLaunchKernelGGL((gemm4w_kernel<MODE, SCHED>), dim3(grid), dim3(256), smem, st, p);
}

// mode: bit 0 = no epilogue, bit 1 = cache-resident diagnostic; sched: 0 / 1 (see the kernel)
extern "C" int g4_launch(const void* A, const void* W, void* C, int M, int N, int K, int mode, int sched, void* stream, void* stamps) {
    if (K % 64 || K < 128 || M % 256 || N % 256) return 1;
    G4Params p{(const unsigned short*)A, (const unsigned short*)W, (float*)C, M, N, K, K, K, N, (unsigned long long*)stamps};
    const int tiles = (M / 256) * (N / 256);
    int dev = 0, ncu = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = tiles < ncu ? tiles : ncu;
    hipStream_t st = (hipStream_t)stream;
    switch ((sched == 2 ? 8 : sched ? 4 : 0) | (mode & 3)) {
        case 0: g4_go<0, 0>(p, grid, st); break;
        case 1: g4_go<1, 0>(p, grid, st); break;
        case 3: g4_go<3, 0>(p, grid, st); break;
        case 4: g4_go<0, 1>(p, grid, st); break;
        case 5: g4_go<1, 1>(p, grid, st); break;
        case 7: g4_go<3, 1>(p, grid, st); break;
        case 8: g4_go<0, 2>(p, grid, st); break;
        case 9: g4_go<1, 2>(p, grid, st); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
