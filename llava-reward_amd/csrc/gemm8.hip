// Deep-pipelined, persistent variant of gemm_bt (same contract and epilogues as gemm.hip):
// 256x256x64 block tile, 8 waves, LDS-DMA half-tile ring with counted vmcnt, raw barriers, two wave
// groups staggered by one barrier, 16x16x32 MFMAs, LDS-staged coalesced epilogue, one resident
// workgroup per CU that walks its tiles (cdna_hip_programming.md §5 "256^2 8-phase template",
// T1/T3/T4/T5; MI355X_MICROARCH.md "Two waves per SIMD").
//
// Schedule.  A K-tile (64 deep) is 4 half-tiles of 16 KB: A0, B0, B1, A1 (128 rows x 64 k each, same
// swizzled image as gemm.hip).  Half-tile g = 4t + j lives in ring slot g % NS.  A K-tile is consumed
// in 4 phases, one block quadrant (A-half qa, B-half qb) each: (0,0) (0,1) (1,1) (1,0); in a phase every
// wave multiplies its 64x32 piece of that quadrant over the full K-tile (16 MFMA 16x16x32), so the
// register fragments of one half are reused by the next phase (ds_read_b128 per phase: 12, 4, 8, 4).
// Phase P (global index):   LOAD(P): ds_reads for P's MFMAs; issue half-tile P+PF by LDS-DMA;
//                                    s_waitcnt vmcnt(2*(PF-2))  -> everything up to half-tile P+2 landed
//                           barrier; COMPUTE(P): MFMAs; barrier.
// Hazards: half-tile g is first read at phase >= g-2, i.e. one phase after the wait that retires it
// (RAW); its slot is restaged by half-tile g+NS, issued at phase g+NS-PF, while its last read is at
// phase <= g+2 of the lagging wave group (WAR) => NS - PF >= 4.  NS = 10 slots = all 160 KB of LDS (4-phase A/B forms).
// The product schedule (PB = 2, below) issues 6 half-tiles ahead on 8 slots -- it measures the same as 8 on 10 -- and keeps the
// other 32 KB for the residual K-tiles' scale bytes.
// Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while
// its partner issues LDS reads / DMA.
// PB = 1 (product): the LOAD segments are the critical path (~300-350 cycles against 256 of MFMA), so the B fragments are
// kept in two register sets -- B0 for the whole K-tile (read once instead of twice), B1 -- and requested from inside the
// previous COMPUTE segment; LOAD keeps only the 8 A reads of phases 0 and 2.  +12 VGPRs, 2-5 % on every shape.
// Measured (MI355X, random data): main loop 1.16-1.28 PFLOP/s on every shape of the path; direct
// stores from the C/D layout cost 30 % of all GEMM time and a per-tile relaunch exposes the store
// acknowledgements, hence the staged epilogue and the persistent tile loop.
//
// Column mapping inside a block tile: wave wc owns columns wc*64 + qb*32 + [0,32) for qb = 0,1, so the
// SwiGLU pair (gate block, up block: weight rows interleaved in 32s) stays in one lane/register.
#include <map>
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "common.h"
#include "kernels.h"

#ifndef LR_GEMM_SADDR
#define LR_GEMM_SADDR 1
#endif
#ifndef LR_GEMM_COMP_WAIT
#define LR_GEMM_COMP_WAIT 1
#endif

namespace lr {

#define LR_BARRIER() do { asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// DBG: 0 = product; 1 = every K-tile re-reads K-tile 0 (cache-resident operands; results invalid);
//      2 = no epilogue; 3 = in-kernel stamps; 4 = LDS-DMA stream alone; 5 = ds_reads + MFMAs alone (results invalid).
//      Diagnostics for tools/gemm_bench.py / tools/gemm_stamps.py only.
// EPI is a template parameter so that each instantiation carries ONE epilogue: with all of them
// unrolled in one kernel the code was ~130 KB and every tile's epilogue ran out of the instruction cache.
// F8: the operands are OCP e4m3 bytes (W8A8 mode).  Nothing about the tile images, the DMA stream or the
// fragment reads changes -- p.K / lda / ldw count 2-byte units, a K-tile is still 128 bytes of a row -- only the matrix
// instruction: v_mfma_scale_f32_16x16x128_f8f6f4 takes 32 bytes per lane, and the two 16-byte fragments a lane already holds
// for the two f16 k-steps (chunks q and 4+q of the row) are those 32 bytes; A and B use the same chunk order, so the
// contraction sees a consistent permutation of k.  One instruction per 16x16 tile and K-tile (twice the K at the same MFMA
// time); the per-row / per-channel dequantisation scales multiply the accumulators in front of the epilogue.
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
// MFMA operand fragments of the product K loop as NATIVE vectors (round 6).  As HIP's uint4 -- a struct -- the fragment arrays were
// taken apart by the optimiser into {x}, {y, z}, {w} pieces; the scaled matrix instruction's 8-register operands were then register
// sequences of six pieces each, and the backend left copies of the {y, z} pairs in front of the MFMAs (see join32).
typedef unsigned frag_t __attribute__((ext_vector_type(4)));
typedef unsigned frag2_t __attribute__((ext_vector_type(2)));
// The 32 operand bytes of a lane = the two 16-byte fragments it holds, joined as TWO 128-bit halves (one shufflevector: a register
// sequence the allocator coalesces with the ds_read_b128 destinations).  Built element by element -- {a0.x, a0.y, ...}, as until round 6
// -- the backend left identity copies through a temporary pair in front of the MFMAs (v_pk_mov_b32 + 2 v_mov_b32 per fragment, ~20
// VALU operations per COMPUTE segment) whose operands made it wait for the B1 fragments (lgkmcnt(0)) in front of the SECOND matrix
// instruction of the segment instead of the ninth: the reads' LDS latency was exposed in every residual K-tile (in-kernel stamps,
// profiles/r6_fp6_stamps.log: COMPUTE 900 cycles for 16 e4m3 MFMAs against 606 for the 32 f16 ones of the same nominal MFMA time).
__device__ __forceinline__ v8i_t join32(const frag_t lo, const frag_t hi) {
    return __builtin_shufflevector(__builtin_bit_cast(v4i_t, lo), __builtin_bit_cast(v4i_t, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 mfma_f8(const frag_t a0, const frag_t a1, const frag_t b0, const frag_t b1, const f32x4 c) {
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(join32(a0, a1), join32(b0, b1), c, 0, 0, 0, 127, 0, 127);     // e4m3 x e4m3, scales 2^0
}

// SEL: which byte of `ea` holds the E8M0 scale of the A rows (four row tiles share one register)
template <int SEL>
__device__ __forceinline__ f32x4 mfma_f8s(const frag_t a0, const frag_t a1, const frag_t b0, const frag_t b1, const f32x4 c, const int ea, const int eb) {
#if defined(LR_GEMM_DIAG_NOSCALE)      // diagnostic (results invalid): the un-scaled encoding of the same matrix instruction -- what do the scale operands cost?
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(join32(a0, a1), join32(b0, b1), c, 0, 0, 0, 0, 0, 0);
#endif
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(join32(a0, a1), join32(b0, b1), c, 0, 0, SEL, ea, 0, eb);   // x 2^(ea.byte[SEL] - 127) x 2^(eb - 127)
}
// FP6 (OCP MX e2m3, cbsz = blgp = 2): 24 bytes per lane and operand -- the 16-byte and the 8-byte part of the lane's 32 elements --
// and one E8M0 scale per lane (= per row and 32-element block) and operand, byte SELA of `ea` / SELB of `eb`.  Half the cycles of the
// e4m3 form (MI355X_MICROARCH.md: FP6 at the FP4 rate).  F8 == 3 kernels only (round 6 A/B, tools/fp6).
template <int SELA, int SELB>
__device__ __forceinline__ f32x4 mfma_f6s(const frag_t a0, const frag2_t a1, const frag_t b0, const frag2_t b1, const f32x4 c, const int ea, const int eb) {
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    const v4i_t ah = __builtin_bit_cast(v4i_t, a0), bh = __builtin_bit_cast(v4i_t, b0);
    const v2i_t al = __builtin_bit_cast(v2i_t, a1), bl = __builtin_bit_cast(v2i_t, b1);
    const v4i_t al4 = __builtin_shufflevector(al, al, 0, 1, -1, -1), bl4 = __builtin_shufflevector(bl, bl, 0, 1, -1, -1);
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(ah, al4, 0, 1, 2, 3, 4, 5, 6, 7),
                                                            __builtin_shufflevector(bh, bl4, 0, 1, 2, 3, 4, 5, 6, 7), c, 2, 2, SELA, ea, SELB, eb);
}
template <int I> struct IC { static constexpr int value = I; };
template <typename F> __device__ __forceinline__ void for4(F&& f) { f(IC<0>{}); f(IC<1>{}); f(IC<2>{}); f(IC<3>{}); }

// F8 == 2: split-operand mode with an e4m3 residual pass: A rows are [hi f16 x kw | lo e4m3 x kw bytes], the K loop
// runs kw / 64 f16 K-tiles against W and then kw / 128 e4m3 K-tiles against W8 (the e4m3 twin of W, in the rows of p.Wlo), with
// the power-of-two scales of the residuals (p.aexp, one E8M0 byte per row and K-tile) and of W8 (p.wexp, one per tensor) applied by
// the matrix instruction itself, so both passes meet in the same accumulators and every epilogue is the one of the 16-bit form.
// NW: narrow tiles for problems that do not fill the chip with 256 x 256 ones (round 4).  1 = 128 x 256: the A1 half of every K-tile
// is not multiplied (quadrants (1,*) do not exist), 2 = 256 x 128: the B1 half is not (quadrants (*,1)); the ring, the DMA stream and
// the barriers are the full tile's -- the unused half-tile is fetched from rows the tile reads anyway (cache hits) -- so a narrow tile
// costs ~85 % of a full one for half its flops (the K loop's skeleton, not its MFMAs, sets the pace), and twice as many of them fill
// twice the CUs when the full tiles leave half the chip idle.  An output element's K order, MFMA
// kinds and scales are exactly the full tile's: results are bit-identical whichever tile shape computed them, so the launcher may
// choose by M (it does: narrow_choice).  NW == 2 exists for the operand-out epilogue only (the adapters' t = x A^T, N = rank).
template <typename OT, int PF, int NS, int DBG, int EPI, int PB, int F8 = 0, int NW = 0>
__global__ __launch_bounds__(512) void gemm_bt8_kernel(GemmParams p) {
    static_assert(!F8 || PB == 2, "the fp8 operand path exists in the super-phase schedule only");
    static_assert(F8 != 3 || (NW == 0 && LR_GEMM_SADDR), "the FP6 residual form (A/B): full tiles, saddr DMA");
    // F8 == 3 (round 6 A/B, tools/fp6/): as F8 == 2, but the residual K-tiles are OCP MX FP6 -- e2m3 elements, 96 bytes per row and
    // 128-deep K-tile stored [4 x 16 B | 4 x 8 B] (lane q of a row's four takes 16-byte part q and 8-byte part q), one E8M0 scale per
    // (row, 32 elements) for A_lo AND for the weight twin.  A half-tile is 12 KB in its 16 KB ring slot: plane H = 128 rows x 64 B
    // (ds_read_b128), plane L = 128 rows x 32 B behind it (ds_read_b64), 12 LDS-DMA pieces instead of 16; the two half-tiles of a LOAD
    // segment go out as 3 pieces per wave (waves 0-3 the first, 4-7 the second).  Exact weights, no adapter, no third segment.
    constexpr bool MIX = F8 == 2 || F8 == 3;
    static_assert(NW == 0 || PB == 2, "narrow tiles exist in the super-phase schedule only");
    static_assert(NW != 2 || (EPI & 15) == EPI_OUT_OP, "256 x 128 tiles: operand-out epilogue only");
    constexpr int E_ = EPI & 15;          // epilogue selector; bit 4 = bias present (SwiGLU / RoPE epilogues)
    constexpr bool BIAS_ = (EPI & 16) != 0;

    constexpr int BM = NW == 1 ? 128 : 256, BN = NW == 2 ? 128 : 256, BK = 64;
    constexpr int HT = 16384;                      // bytes per half-tile slot
    static_assert(PB == 2 || (NS - PF >= 4 && PF >= 3), "ring hazard distances");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;

    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + BN - 1) / BN;
    const int nwg = Mt * Nt;
    // PB == 2 (product): the K loop is the segment list the launcher built (GemmParams::seg); the A/B variants keep the older
    // closed-form addressing (16-bit operands only)
    const int nk = PB == 2 ? p.nk : p.K / BK;
    const int nk_hi = (PB == 2 && MIX) ? p.nk_f16 : nk;      // K-tiles of 16-bit operands; the rest are e4m3 (F8 == 3: FP6)
    const int nk_lo = (PB == 2 && MIX) ? p.nk_e1 : nk;       // end of the residual segment; beyond it: A_hi8 x Wlo8 (inexact weights)
    const int Gtot = 4 * nk;
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;
    const ptrdiff_t wlo_delta = p.Wlo ? (const unsigned short*)p.Wlo - (const unsigned short*)p.W : 0;

    // ---- fragment read offsets inside a half-tile image: 8 A reads + 4 B reads per phase ----
    // 16x16x32: lane (r = lane&15, q = lane>>4) reads row r, 16-byte chunk 4*ks + q   (ks = 0..1)
    int aoff[8], boff[4];
    auto img_off = [](int row, int kc) { const int R = row >> 1; return R * 256 + (((((row & 1) << 3) + kc) ^ (R & 15)) << 4); };
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) aoff[i * 2 + ks] = img_off(wr * 64 + i * 16 + l15, 4 * ks + l4);
#pragma unroll
        for (int j = 0; j < 2; ++j) boff[j * 2 + ks] = img_off(wc * 32 + j * 16 + l15, 4 * ks + l4);
    }

    // DBG == 3: in-kernel stamps (cdna_hip_programming.md §7): cycles per segment, summed over all K-tiles, per phase
    unsigned segs[4][4];     // [phase][LOAD, BAR1, COMPUTE, BAR2]
    unsigned lsegs[3] = {0, 0, 0};   // LOAD split: LDS reads (incl. their latency), DMA issue, vmcnt wait
    if constexpr (DBG == 3) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) segs[a][b2] = 0;
    }
    auto stamp = [&]() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };

    // ---- persistent walk over tiles: virtual block id vb keeps the XCD-aware order of gemm.hip ----
    int tile_it = 0;           // DBG == 9: timeline stamps (100 MHz clock) per workgroup and tile -> p.ascale: tile start, K loop start, epilogue start, end
    auto tstamp = [&](int k) {
        if constexpr (DBG == 9) {
            if (tid == 0 && tile_it < 128) ((unsigned long long*)p.ascale)[((size_t)blockIdx.x * 128 + tile_it) * 16 + k] = __builtin_amdgcn_s_memrealtime();
        }
    };
    // Tile order L: XCD x owns the contiguous chunk [chunk0(x), chunk0(x) + chunk_n(x)) and walks it in order, so that the 32
    // tiles its CUs hold at any time form an 8 x 4 patch of the band order below (A / W slices shared in that XCD's L2).
    // Static walk: workgroup slot i of the XCD takes i, i + 32, ...  Dynamic (p.sched, launch8): the first tile is the static one,
    // every further tile is claimed from the XCD's counter while the previous tile's epilogue runs; an XCD that runs ahead of the
    // others -- they differ by several per cent in sustained speed -- then takes tiles from the chunk with the most left.
    // Dynamic walk over many bands (p.band_chunks, launch8): the chunks are cut at band boundaries, so every XCD starts a band at
    // its first column tile and the eight of them move along N together -- the W tiles one XCD fetches are still in the Infinity
    // Cache when the other seven ask for them, whatever the size of W (the chunks differ by up to one band; the XCDs that finish
    // first take tiles from the longest one, as before).
    const int chq = nwg >> 3, chr = nwg & 7;
    const int nbands = (Mt + p.gm - 1) / p.gm, band_tiles = p.gm * Nt;
    const bool bandc = PB == 2 && p.sched != nullptr && p.band_chunks;
    auto chunk0 = [&](int x) { return bandc ? min(nwg, (x * nbands >> 3) * band_tiles) : x < chr ? x * (chq + 1) : chr * (chq + 1) + (x - chr) * chq; };
    auto chunk_n = [&](int x) { return bandc ? min(nwg, ((x + 1) * nbands >> 3) * band_tiles) - min(nwg, (x * nbands >> 3) * band_tiles) : chq + (x < chr ? 1 : 0); };
    const int my_xcd = (int)blockIdx.x & 7, per_xcd = (int)gridDim.x >> 3;
    const bool dyn = PB == 2 && p.sched != nullptr;
    int Ldyn = -1;
    for (int vb = blockIdx.x; vb < nwg;) {
        tstamp(0);
        int L;
        if (Ldyn >= 0) L = Ldyn;
        else L = chunk0(vb & 7) + (vb >> 3);
        const int GM = p.gm;
        const int band = L / (GM * Nt);
        const int within = L - band * (GM * Nt);
        const int rows_in_band = min(GM, Mt - band * GM);
        const int mi = band * GM + within % rows_in_band;
        const int ni = within / rows_in_band;
        const int m0 = mi * BM, n0 = ni * BN;

        // ---- LDS-DMA source pointers: [half][it]; swizzle on the source side ----
        // PB == 2: gA / gB are RUNNING pointers -- they address the K-tile that is issued next and move on by one K-tile (64 units)
        // once its four half-tiles are out; at a segment boundary they are rebuilt for the next segment (load_seg): another
        // operand pair (the adapter's t / B, the e4m3 twins), another row pitch, another column.  The rebuild sits inside the K
        // loop but runs a handful of times per tile; its index arithmetic is kept from being hoisted (it would live across the
        // loop in VGPRs the accumulators need).
        const unsigned short* gA[2][2];
        const unsigned short* gB[2][2];
        // SADDR (round 5 experiment, LR_GEMM_SADDR): the DMA's source as a UNIFORM 64-bit base (SGPR pair, one per operand, advanced per
        // K-tile with scalar adds) + a per-thread 32-bit byte offset that is constant over a segment -- the saddr form of the instruction
        // -- instead of eight running 64-bit per-lane pointers (16 VGPRs, eight 64-bit vector adds per K-tile).
        constexpr bool SADDR = PB == 2 && LR_GEMM_SADDR;
        const unsigned short* bA = nullptr;
        const unsigned short* bB = nullptr;
        unsigned oA[2][2], oB[2][2];
        auto uni64 = [&](const unsigned short* q) {
            const unsigned long long u = (unsigned long long)q;
            return (const unsigned short*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
        };
        int iseg = -1, iseg_end = 0, ikt = 0;          // segment being issued, its last K-tile + 1, K-tile being issued (uniform)
        unsigned o6A[3], o6B[3];      // F8 == 3: byte offsets of this wave's 3 pieces of its A / its B half-tile (load_seg6)
        auto load_seg = [&]() {
            ++iseg;
            const GemmParams::KSeg sg = p.seg[iseg];
            iseg_end = sg.kt_end;
            const bool a2 = (sg.src & GemmParams::SRC_A2) != 0, w2 = (sg.src & GemmParams::SRC_W2) != 0;
            const bool lo = (sg.src & GemmParams::SRC_LO) != 0;
            const unsigned short* Ab = (const unsigned short*)(a2 ? p.A2 : p.A);
            const unsigned short* Wb = (const unsigned short*)(w2 ? (lo ? p.W2lo : p.W2) : (sg.src & GemmParams::SRC_LO16) ? p.Wlo16 : lo ? p.Wlo : p.W);
            const int la = a2 ? p.lda2 : p.lda, lw = w2 ? p.ldw2 : p.ldw;
            int t = tid;
            asm volatile("" : "+v"(t));

#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int q = it * 512 + t;
                const int R = q >> 4, Cp = q & 15;
                const int C = Cp ^ (R & 15);
                const int row = 2 * R + (C >> 3), c = C & 7;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ga = min(m0 + (NW == 1 ? 0 : h * 128) + row, p.M - 1);
                    const int wrow = NW == 2 ? row : (row >> 5) * 64 + h * 32 + (row & 31);     // 256 x 128: image row = tile column
                    const int gb = min(n0 + wrow, p.N - 1);
                    if constexpr (SADDR) {      // byte offsets inside the tile; the tile's (row m0 / n0, column, K-tile) lives in the uniform bases
                        oA[h][it] = (unsigned)((ga - m0) * la + c * 8) * 2u;
                        oB[h][it] = (unsigned)((gb - n0) * lw + c * 8) * 2u;
                    } else {
                        gA[h][it] = Ab + (size_t)ga * la + c * 8 + sg.a_col;
                        gB[h][it] = Wb + (size_t)gb * lw + c * 8 + sg.w_col;
                    }
                }
            }
            if constexpr (SADDR) {
                bA = uni64(Ab + (size_t)m0 * la + sg.a_col);
                bB = uni64(Wb + (size_t)n0 * lw + sg.w_col);
            }
        };
        // F8 == 3: the FP6 residual segment (always the last one), entered from the peeled K-tile that issues its first half-tiles:
        // waves 0-3 carry A half 0 and B half 1, waves 4-7 A half 1 and B half 0
        auto load_seg6 = [&]() {
            ++iseg;
            const GemmParams::KSeg sg = p.seg[iseg];
            iseg_end = sg.kt_end;
            int t = tid;
            asm volatile("" : "+v"(t));
            const int ln = t & 63, hA = wr, hB = 1 - wr;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int pc = 3 * wc + i;                      // piece of the half-tile: 0-7 plane H (16 rows each), 8-11 plane L (32 rows each)
                int row, boff6;
                if (pc < 8) { row = 16 * pc + (ln >> 2); boff6 = (((ln & 3) ^ ((0 - (ln >> 4)) & 3)) << 4); }
                else { row = 32 * (pc - 8) + (ln >> 1); boff6 = 64 + (((ln & 1) ^ ((row >> 3) & 1)) << 4); }
                const int ga = min(m0 + hA * 128 + row, p.M - 1);
                const int gb = min(n0 + (row >> 5) * 64 + hB * 32 + (row & 31), p.N - 1);
                o6A[i] = (unsigned)((ga - m0) * p.lda * 2 + boff6);
                o6B[i] = (unsigned)((gb - n0) * p.ldw * 2 + boff6);
            }
            bA = uni64((const unsigned short*)p.A + (size_t)m0 * p.lda + sg.a_col);
            bB = uni64((const unsigned short*)p.Wlo + (size_t)n0 * p.ldw + sg.w_col);
        };
        if constexpr (PB != 2) {
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
        }

        // half-tile j of a K-tile: 0 = A0, 1 = B0, 2 = B1, 3 = A1.
        // The LDS-DMA is issued from inline asm on purpose: hipcc's waitcnt pass would otherwise put
        // `s_waitcnt vmcnt(0)` in front of every ds_read (it cannot prove the pending DMA does not alias)
        // and drain the ring each phase.  Ordering is ours: counted vmcnt, then a barrier, then the read
        // (cdna_hip_programming.md §5.7 item 1).  M0 is saved/restored inside the statement.
        auto issue1 = [&](int j, int kt, int slot, int it) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * HT + wave * 1024 + it * 8192);
            if constexpr (SADDR) {
                const unsigned short* base = (j == 0 || j == 3) ? bA : bB;
                const unsigned off = (j == 0) ? oA[0][it] : (j == 1) ? oB[0][it] : (j == 2) ? oB[1][it] : oA[1][it];
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
                return;
            }
            const unsigned short* src;
            if constexpr (PB == 2) {             // running pointers: already at this K-tile
                src = (j == 0) ? gA[0][it] : (j == 1) ? gB[0][it] : (j == 2) ? gB[1][it] : gA[1][it];
            } else {
                int koff = DBG == 1 ? 0 : kt * BK, koffw = koff;
                ptrdiff_t wsel = 0;
                if (p.kw > 0) {                  // split-operand mode: A = [hi | lo (| hi)], W repeats along K (, then its residuals)
                    if (koff >= 2 * p.kw) { koff -= 2 * p.kw; koffw = koff; wsel = wlo_delta; }
                    else if (koff >= p.kw) koffw = koff - p.kw;
                }
                src = (j == 0) ? gA[0][it] + koff : (j == 1) ? gB[0][it] + koffw + wsel : (j == 2) ? gB[1][it] + koffw + wsel : gA[1][it] + koff;
            }
            // (round 4 A/B: declaring M0 clobbered instead of saving / restoring it -- 3 instructions per piece instead of 5 -- changes
            //  nothing here, +-0.5 % on five shapes: the partner wave's MFMAs cover this wave's issue slots)
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        };
        auto issue = [&](int j, int kt, int slot) { issue1(j, kt, slot, 0); issue1(j, kt, slot, 1); };
        // PB == 2: in front of the first / behind the last half-tile of the K-tile being issued
        auto ktile_begin = [&]() { if (ikt == iseg_end) load_seg(); };
        auto ktile_end = [&]() {
            if constexpr (F8 == 3) { const int adv = ikt >= nk_hi ? 48 : BK; bA += adv; bB += adv; ++ikt; return; }      // an FP6 K-tile is 96 bytes of a row (scalar arithmetic)
            ++ikt;
            if constexpr (SADDR) { bA += BK; bB += BK; }
            else if constexpr (DBG != 1) {
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int h = 0; h < 2; ++h) { gA[h][it] += BK; gB[h][it] += BK; }
            }
        };

        f32x4 acc[4][4][2];     // [quadrant (0,0) (0,1) (1,1) (1,0)][row tile of 16][col tile of 16]
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[q][i][j][r] = 0.f;

        // F8 == 2: E8M0 scales.  Residual segment: one byte per (row, K-tile) (common.h lo8_scale_at), i.e. per K-tile one dword per
        // lane and A half -- byte i = row tile i of the lane's rows -- see issue_slice below.  Third segment
        // (weights inexact in the operand type: A_hi as e4m3, one exponent per row): ordinary loads, retired here, in front of the DMA
        // stream, so that no compiler-placed vmcnt wait can appear inside the K loop.
        int ea[2] = {0x7F7F7F7F, 0x7F7F7F7F};        // [A half]: byte i = exponent of row tile i, of the K-tile being multiplied
        int ea2[2] = {0x7F7F7F7F, 0x7F7F7F7F};       // the same for the third segment
        const int eb2 = p.wexp2;
        const int eb = p.wexp;
        if constexpr (F8 == 2) {
            if (nk > nk_lo) {                    // weights inexact in the operand type: third segment, A_hi as e4m3 against e4m3(W_lo)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e1 = p.aexp2[min(m0 + (NW == 1 ? 0 : h2 * 128) + wr * 64 + i * 16 + l15, p.M - 1)] & 255;
                        ea2[h2] = i == 0 ? e1 : (ea2[h2] | (e1 << (8 * i)));
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("" : "+v"(ea2[0]), "+v"(ea2[1]));
        }
        // The scale bytes of the residual K-tiles travel through LDS: group g (the 4 x 256 bytes of this row tile for residual K-tiles
        // 4g .. 4g+3, common.h lo8_scale_at) is ONE LDS-DMA instruction of one wave, issued in the tile's prologue -- in front of the
        // first half-tiles, whose latency is exposed there anyway -- into the 32 KB behind the ring (32 groups; beyond that a group
        // follows the one it replaces), so that the residual K-tiles, which are LOAD-bound, only pay one 8-byte LDS read per lane
        // and K-tile.  (Issued inside the K loop, a scale group is a cold line at the head of the in-order queue: ~0.6 us per group.)
        constexpr int SC_GROUPS = 32;
        const unsigned sc_voff = (unsigned)((wr * 16 + l15) * 8);
        // (128-row tiles: the slices are laid out per 256 rows, tile mi is A half (mi & 1) of slice row mi >> 1)
        const unsigned char* sc_tile = p.aexp + (size_t)(NW == 1 ? mi >> 1 : mi) * 1024;
        const size_t sc_plane = (size_t)((p.M + 255) >> 8) * 1024;
        const int nsl = nk_lo - nk_hi, nsg = (nsl + 3) >> 2;
        const bool late_sc = nsg > SC_GROUPS;          // more scale groups than LDS slots: later groups follow the ones they replace
        auto issue_scales = [&](int g) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + NS * HT + (g & (SC_GROUPS - 1)) * 1024);
            int l16 = lane * 16;
            asm volatile("" : "+v"(l16));          // rebuilt at each call: hoisted, the address is one more 64-bit value spilled across the K loop
            const unsigned char* src = sc_tile + (size_t)g * sc_plane + l16;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        };

        // F8 == 3: the FP6 K-tiles' scales, one byte per (row, 32 elements) for both operands: per residual K-tile j one KB of A scales
        // (lane (row, q) of wave group wr: 8 bytes = its 4 row tiles of A half 0, then of A half 1, block q) and one KB of W scales
        // (wave column wc: 4 bytes per lane = column tiles 0, 1 of B half 0, then of B half 1), in LDS slots 2j and 2j + 1 (mod 32)
        // behind the ring: 16 K-tiles resident, the first 16 issued in the tile's prologue, K-tile j + 14 into the slots of K-tile
        // j - 2 while K-tile j is multiplied.  Source layout: [K-tile][256-row tile] x 1 KB (p.aexp) and [K-tile][256-column tile] x 1 KB
        // (p.wscale, bytes).
        auto issue_scales6 = [&](int x) {          // piece x: K-tile x >> 1, A (even) or W (odd)
            const int j6 = x >> 1;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + NS * HT + ((x & 31) << 10));
            int l16 = lane * 16;
            asm volatile("" : "+v"(l16));
            const unsigned char* src = (x & 1) ? (const unsigned char*)p.wscale + ((size_t)j6 * Nt + ni) * 1024 + l16
                                               : p.aexp + ((size_t)j6 * Mt + mi) * 1024 + l16;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        };
        int a6H = 0, a6L = 0, b6H = 0, b6L = 0, ew6 = 0x7F7F7F7F;      // F8 == 3: fragment offsets inside an FP6 half-tile image; W scales of the K-tile

        if constexpr (PB == 2) {
        // ================= super-phase schedule (product) =================
        // Two quadrants per barrier interval: LOAD = 8 A reads + both half-tiles of the interval by LDS-DMA + counted wait;
        // COMPUTE = 32 MFMAs (an A half against B0 and B1) with the 4 B reads of the next use riding inside it.  Half the
        // barriers and priority flips per K-tile, and the LOAD segment (the critical path of the 4-phase form) is shorter than
        // its partner's COMPUTE.  Super-phase Q = 2 kt + s:
        //   LOAD(2kt)    af <- A0(kt)                 issue half-tiles 4kt+6, 4kt+7     wait: all but the newest 2 half-tiles landed
        //   COMPUTE(2kt)   bg <- B1(kt);  (0,0) = af x bf;  (0,1) = af x bg
        //   LOAD(2kt+1)  af <- A1(kt)                 issue half-tiles 4kt+8, 4kt+9
        //   COMPUTE(2kt+1) (1,0) = af x bf;  bf <- B0(kt+1);  (1,1) = af x bg
        // RAW: a read inside COMPUTE(Q) of the leading group needs the LAGGING group's LOAD(Q-1) wait (it is still inside its own
        // LOAD(Q)), a read in LOAD(Q+1) its LOAD(Q) wait; with PF2 half-tiles issued ahead and PF2 - 4 allowed in flight LOAD(Q)
        // retires 2Q+5: B1(kt) = 4kt+2 <= 2(2kt-1)+5, A1 = 4kt+3 and B0(kt+1) = 4kt+5 <= 2(2kt)+5, A0(kt+1) = 4kt+4 <= 2(2kt+1)+5.
        // WAR: half-tile g+NS is issued at LOAD(floor((g+NS-PF2)/2)); the lagging group is then in COMPUTE of the interval before
        // and its reads of g (A0: LOAD(g/2); B1: COMPUTE(g/2-1); A1: LOAD((g-1)/2); B0: COMPUTE((g-5)/2)) are behind it iff
        // NS - PF2 >= 2.  A half-tile has one phase and a half between its issue and the wait that retires it.
        constexpr int PF2 = 6, WAIT2 = 2 * (PF2 - 4);       // 6 half-tiles ahead measure the same as 8 and leave 32 KB of LDS
        static_assert(F8 != 2 || (NS + 2) * HT <= 160 * 1024, "the scale slices live behind the ring");
        static_assert(NS - PF2 >= 2, "ring hazard distances (super-phase schedule)");
        int islot = 0;
        if constexpr (F8 == 2) {         // this tile's scale groups first: the oldest entries of the queue, retired by the prologue's wait
            for (int g = wave; g < min(nsg, SC_GROUPS); g += 8) issue_scales(g);
        }
        if constexpr (F8 == 3) {
            for (int x = wave; x < min(2 * nsl, 32); x += 8) issue_scales6(x);
        }
        // the two half-tiles of an FP6 LOAD segment (first = B1 / A0, second = A1 / B0 for sp = 0 / 1) as 3 pieces per wave
        auto issue6 = [&](int sp, int slot0) {
            const bool isA = (wr == 0) == (sp == 1);
            int slot = slot0 + wr; slot = slot >= NS ? slot - NS : slot;
            const unsigned short* base = isA ? bA : bB;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int pc = 3 * wc + i;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * HT + (pc < 8 ? pc * 1024 : 8192 + (pc - 8) * 1024));
                const unsigned off = isA ? o6A[i] : o6B[i];
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
            }
        };
#pragma unroll
        for (int g = 0; g < PF2; ++g) {
            if (g < Gtot) {
                if ((g & 3) == 0) ktile_begin();
                issue(g & 3, g >> 2, islot);
                if ((g & 3) == 3) ktile_end();
            }
            islot = (islot + 1 == NS) ? 0 : islot + 1;
        }
        if (Gtot > PF2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT2) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LR_BARRIER();
        if (wr == 1) LR_BARRIER();                      // stagger the second wave group by one barrier
        tstamp(1);
        frag_t af[8], bf[4], bg[4];
        frag2_t bfl[2], bgl[2];          // F8 == 3: the 8-byte parts of the FP6 B fragments (their 16-byte parts live in bf[0..1] / bg[0..1])
        auto init6 = [&]() {           // FP6 half-tile image: plane H rows of 64 B (chunk q ^ f(row)), plane L rows of 32 B behind it
            int ln = lane;
            asm volatile("" : "+v"(ln));           // (computed behind the 16-bit K-tiles: not live across them)
            const int r15 = ln & 15, q4 = ln >> 4;
            const int sw4 = (q4 ^ ((0 - (r15 >> 2)) & 3)) << 4, sw8 = (q4 ^ (((r15 >> 3) & 1) << 1)) << 3;
            a6H = (wr * 64 + r15) * 64 + sw4; a6L = 8192 + (wr * 64 + r15) * 32 + sw8;
            b6H = (wc * 32 + r15) * 64 + sw4; b6L = 8192 + (wc * 32 + r15) * 32 + sw8;
        };
        if constexpr (DBG != 4) {
#pragma unroll
            for (int f = 0; f < 4; ++f) bf[f] = *(const frag_t*)(smem + 1 * HT + boff[f]);      // B0 of K-tile 0 (slot 1)
        }
        int rslot = 0, gi = PF2;
        bool scl = false;          // this wave issued a scale slice in this LOAD segment: one more entry in its vmcnt queue
        // One K-tile.  A generic lambda so that the mixed form (F8 == 2) can run two loops, 16-bit tiles then e4m3 tiles, each
        // with its own straight-line body: a run-time branch around the two MFMA kinds merges 64 accumulator registers behind it
        // and spills inside the K loop.
        auto ktile = [&](auto lo_tag, auto iss_tag, const int kt) {
            constexpr int LO = decltype(lo_tag)::value;          // 0: 16-bit K-tile, 1: e4m3 residual K-tile, 2: e4m3 A_hi x W_lo K-tile, 3: FP6 residual K-tile
            constexpr int ISS = decltype(iss_tag)::value;        // F8 == 3: what this K-tile's LOAD segments ISSUE: 0 = 16-bit half-tiles, 1 = FP6 ones, 2 = 16-bit in sp 0, FP6 in sp 1
            int s2 = rslot + 2; s2 = s2 >= NS ? s2 - NS : s2;
            int s3 = rslot + 3; s3 = s3 >= NS ? s3 - NS : s3;
            int s5 = rslot + 5; s5 = s5 >= NS ? s5 - NS : s5;
            const char* sA0 = smem + rslot * HT;
            const char* sB1 = smem + s2 * HT;
            const char* sA1 = smem + s3 * HT;
            const char* sB0n = smem + s5 * HT;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
                if constexpr (DBG == 3) t0 = stamp();
                // ---------------- LOAD ----------------
                const bool more = gi < Gtot;
                frag_t ah6[4];
                frag2_t al6[4];          // FP6 A fragments of this super-phase: 16-byte and 8-byte part of each row tile
                if constexpr (DBG != 4) {
                    if constexpr (LO == 3) {
                        const char* sa = sp == 0 ? sA0 : sA1;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            ah6[i] = *(const frag_t*)(sa + a6H + i * 1024);
                            al6[i] = *(const frag2_t*)(sa + a6L + i * 512);
                        }
                    } else if (NW != 1 || sp == 0) {
                        const char* sa = sp == 0 ? sA0 : sA1;
#pragma unroll
                        for (int f = 0; f < 8; ++f) af[f] = *(const frag_t*)(sa + aoff[f]);
                    }
                }
                if constexpr (F8 == 3 && LO == 3) {
                    const int j = kt - nk_hi;
                    if (sp == 0) {
                        typedef int v2i_t __attribute__((ext_vector_type(2)));
                        typedef __attribute__((address_space(3))) const volatile v2i_t lds_scale_t;
                        typedef __attribute__((address_space(3))) const volatile int lds_scale1_t;
                        const v2i_t e2 = *(lds_scale_t*)(lds_base + NS * HT + (((2 * j) & 31) << 10) + (wr * 64 + lane) * 8);
                        ea[0] = e2.x; ea[1] = e2.y;
                        ew6 = *(lds_scale1_t*)(lds_base + NS * HT + (((2 * j + 1) & 31) << 10) + (wc * 64 + lane) * 4);
                        // K-tile j + 14 into the slots of K-tile j - 2 (read two K-tiles ago by both wave groups)
                        if (j >= 2 && j + 14 < nsl) {
                            if (wave == (j & 7)) { issue_scales6(2 * (j + 14)); scl = true; }
                            else if (wave == ((j + 4) & 7)) { issue_scales6(2 * (j + 14) + 1); scl = true; }
                        }
                    }
                }
                // the scales of this residual K-tile's rows (both A halves) from LDS; a late scale group (K > 16384) into the slot of
                // the one it replaces
                if constexpr (F8 == 2 && LO == 1) {
                    const int j = kt - nk_hi;
                    if (sp == 0) {
                        typedef int v2i_t __attribute__((ext_vector_type(2)));
                        typedef __attribute__((address_space(3))) const volatile v2i_t lds_scale_t;       // an LDS read, never a flat one
                        const v2i_t e2 = *(lds_scale_t*)(lds_base + NS * HT + (((j >> 2) & (SC_GROUPS - 1)) << 10) + ((j & 3) << 8) + sc_voff);
                        ea[0] = (NW == 1 && (mi & 1)) ? e2.y : e2.x; ea[1] = e2.y;
                        if (late_sc) {                                 // (K > 16384 only: one uniform test per K-tile for every other shape)
                            const int g = (j >> 2) - 1 + SC_GROUPS;    // its slot was last read in the K-tile before this one
                            if ((j & 3) == 0 && j >= 4 && g < nsg && wave == (g & 7)) { issue_scales(g); scl = true; }
                        }
                    }
                }
                if (more) {
                    const bool i6 = F8 == 3 && (ISS == 1 || (ISS == 2 && sp == 1));      // the K-tile being ISSUED is an FP6 one (3 pieces per wave instead of 4); folds
                    if constexpr (DBG != 5) {                 // second half of K-tile kt+1, then first half of kt+2
                        if (sp == 1) { if (F8 == 3 && ISS == 2) load_seg6(); else ktile_begin(); }
                        if (F8 == 3 && i6) {
                            issue6(sp, islot);
                            islot += 2; islot = islot >= NS ? islot - NS : islot;
                        } else {
                            issue(sp == 0 ? 2 : 0, kt + 2, islot);
                            islot = (islot + 1 == NS) ? 0 : islot + 1;
                            issue(sp == 0 ? 3 : 1, kt + 2, islot);
                            islot = (islot + 1 == NS) ? 0 : islot + 1;
                        }
                        if (sp == 0) ktile_end();
                    }
                    if (F8 == 3 && i6) {
                        if (scl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3) : "memory");
                    } else {
                        if (scl) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT2 + 1) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT2) : "memory");
                    }
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                scl = false;
                gi += 2;
                if constexpr (DBG == 3) t1 = stamp();
                LR_BARRIER();
                if constexpr (DBG == 3) t2 = stamp();
                // ---------------- COMPUTE ----------------
                __builtin_amdgcn_s_setprio(1);
                // The A fragments were requested in LOAD, a counted wait and a barrier ago: they have landed.  Retiring them HERE (a
                // real s_waitcnt lgkmcnt(0): vmcnt / expcnt untouched) clears the compiler's in-order scoreboard before the B reads
                // of this segment are issued; without it the waits it counts for the A fragments (lgkmcnt(3), (1), (0) in the middle
                // of the first block) also cover whatever B reads it has interleaved in front of them -- freshly issued, so their
                // LDS latency was exposed inside the block (round 6; -DLR_GEMM_COMP_WAIT=0: the old schedule).
                if constexpr (LR_GEMM_COMP_WAIT) __builtin_amdgcn_s_waitcnt(0xC07F);
                if constexpr (DBG != 4) {
                    if constexpr (LO == 3) {
                        if (sp == 0) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                bg[j] = *(const frag_t*)(sB1 + b6H + j * 1024);
                                bgl[j] = *(const frag2_t*)(sB1 + b6L + j * 512);
                            }
                        }
                    } else if (sp == 0 && NW != 2) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) bg[f] = *(const frag_t*)(sB1 + boff[f]);
                    }
                    if constexpr (LR_GEMM_COMP_WAIT) { if (sp == 0) __builtin_amdgcn_sched_barrier(0); }      // (the B1 requests stay in front of the first block)
                    constexpr int q0 = 0, q1 = 1, q3 = 3, q2 = 2;
                    const int qa = sp == 0 ? q0 : q3, qb = sp == 0 ? q1 : q2;      // first block uses bf (B0), second bg (B1)
                    const bool do_a = NW != 1 || sp == 0, do_b = do_a && NW != 2;        // (compile-time constants after unrolling)
                    if constexpr (LO == 3) {
                        for4([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            acc[qa][i][0] = mfma_f6s<i, 0>(ah6[i], al6[i], bf[0], bfl[0], acc[qa][i][0], ea[sp], ew6);
                            acc[qa][i][1] = mfma_f6s<i, 1>(ah6[i], al6[i], bf[1], bfl[1], acc[qa][i][1], ea[sp], ew6);
                        });
                    } else if (!do_a) {
                    } else if constexpr (F8 == 1 || (F8 == 2 && LO)) {
                        for4([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[qa][i][j] = F8 == 1 ? mfma_f8(af[i * 2], af[i * 2 + 1], bf[j * 2], bf[j * 2 + 1], acc[qa][i][j])
                                                        : mfma_f8s<i>(af[i * 2], af[i * 2 + 1], bf[j * 2], bf[j * 2 + 1], acc[qa][i][j], LO == 2 ? ea2[sp] : ea[sp], LO == 2 ? eb2 : eb);
                        });
                    } else {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[qa][i][j] = Op<OT>::mfma16(af[i * 2 + ks], bf[j * 2 + ks], acc[qa][i][j]);
                    }
                    // bf <- B0 of the next K-tile, BEHIND the first block's MFMAs and unconditionally (round 6).  It used to sit under
                    // `kt + 1 < nk`; in the residual K-tiles the compiler hoisted the four reads to the top of the segment, and because
                    // they sat in a branch its counted waits for the A fragments (requested in LOAD, long landed) had to hold on the
                    // path WITHOUT them too: lgkmcnt(3) in front of the first MFMA = one of the reads just issued: their LDS latency
                    // exposed in every residual super-phase 1 (stamps: COMPUTE 788 cycles against 650 in super-phase 0).  After the last
                    // K-tile the reads fetch bytes nobody uses (the slot is inside the ring; LDS reads cannot fault).
                    // (a scheduling barrier between the two blocks in BOTH super-phases: in super-phase 0 the scheduler otherwise pulls
                    //  second-block MFMAs -- which need the B1 fragments requested at the top of this segment -- into the first block,
                    //  with a counted wait in front of them ~100 cycles after the request)
                    if constexpr (LR_GEMM_COMP_WAIT) __builtin_amdgcn_sched_barrier(0);
                    else if (sp == 1) __builtin_amdgcn_sched_barrier(0);
                    if (F8 == 3 && ISS == 1 && sp == 1) {      // the next K-tile is an FP6 one
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            bf[j] = *(const frag_t*)(sB0n + b6H + j * 1024);
                            bfl[j] = *(const frag2_t*)(sB0n + b6L + j * 512);
                        }
                    } else if (sp == 1) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) bf[f] = *(const frag_t*)(sB0n + boff[f]);
                    }
                    if constexpr (LO == 3) {
                        for4([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            acc[qb][i][0] = mfma_f6s<i, 2>(ah6[i], al6[i], bg[0], bgl[0], acc[qb][i][0], ea[sp], ew6);
                            acc[qb][i][1] = mfma_f6s<i, 3>(ah6[i], al6[i], bg[1], bgl[1], acc[qb][i][1], ea[sp], ew6);
                        });
                    } else if (!do_b) {
                    } else if constexpr (F8 == 1 || (F8 == 2 && LO)) {
                        for4([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[qb][i][j] = F8 == 1 ? mfma_f8(af[i * 2], af[i * 2 + 1], bg[j * 2], bg[j * 2 + 1], acc[qb][i][j])
                                                        : mfma_f8s<i>(af[i * 2], af[i * 2 + 1], bg[j * 2], bg[j * 2 + 1], acc[qb][i][j], LO == 2 ? ea2[sp] : ea[sp], LO == 2 ? eb2 : eb);
                        });
                    } else {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[qb][i][j] = Op<OT>::mfma16(af[i * 2 + ks], bg[j * 2 + ks], acc[qb][i][j]);
                    }
                    if constexpr (F8 != 0) {
                        // pin this segment's MFMAs in front of its closing barrier: they touch no memory, so nothing else stops
                        // the optimiser from sinking them into the next segment (it did: all 32 ended up in one)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                if (do_a) asm volatile("" : "+v"(acc[qa][i][j]));
                                if (do_b) asm volatile("" : "+v"(acc[qb][i][j]));
                            }
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                if constexpr (DBG == 3) t3 = stamp();
                LR_BARRIER();
                if constexpr (DBG == 3) {
                    t4 = stamp();
                    if (sp == 0) { const unsigned long long t5 = stamp(); lsegs[0] += (unsigned)(t5 - t4); }   // cost of one stamp
                    const int row = sp + (LO ? 2 : 0);              // rows 2, 3: the residual K-tiles (e4m3 / FP6), summed apart from the 16-bit ones
                    segs[row][0] += (unsigned)(t1 - t0); segs[row][1] += (unsigned)(t2 - t1);
                    segs[row][2] += (unsigned)(t3 - t2); segs[row][3] += (unsigned)(t4 - t3);
                }
            }
            rslot += 4;
            rslot = rslot >= NS ? rslot - NS : rslot;
        };
        {
            int kt = 0;
            if constexpr (F8 == 3) {
                // the last two 16-bit K-tiles issue the first FP6 half-tiles: peeled, so that the main loop's body is the product's
                for (; kt < nk_hi - 2; ++kt) ktile(IC<0>{}, IC<0>{}, kt);
                init6();
                ktile(IC<0>{}, IC<2>{}, kt); ++kt;
                ktile(IC<0>{}, IC<1>{}, kt); ++kt;
                for (; kt < nk; ++kt) ktile(IC<3>{}, IC<1>{}, kt);
            } else {
                for (; kt < nk_hi; ++kt) ktile(IC<0>{}, IC<0>{}, kt);
                if constexpr (F8 == 2) {
                    for (; kt < nk_lo; ++kt) ktile(IC<1>{}, IC<0>{}, kt);
                    for (; kt < nk; ++kt) ktile(IC<2>{}, IC<0>{}, kt);
                }
            }
        }
        if (wr == 0) LR_BARRIER();                      // balance the stagger barrier
        } else {
        // ================= 4-phase schedule (A/B variants 3-5) =================
        // ---- prologue: PF half-tiles in flight, the first two landed.  (Stores of the previous tile's
        //      epilogue are older than these DMAs in the vmcnt queue: the wait covers them too.) ----
        int islot = 0;         // ring slot of the next half-tile to issue
#pragma unroll
        for (int g = 0; g < PF; ++g) {
            if (g < Gtot) issue(g & 3, g >> 2, islot);
            islot = (islot + 1 == NS) ? 0 : islot + 1;
        }
        // PB = 0: at the wait of LOAD(P) everything up to half-tile P+2 has landed (this wave's pieces); a half-tile is first read
        // at phase >= g-1, i.e. after the OTHER wave group's LOAD(P-1) wait as well.  PB = 1 reads half-tile P+2 inside COMPUTE(P),
        // while the lagging group is still in ITS LOAD(P): its last completed wait is LOAD(P-1), which therefore has to retire
        // up to (P-1)+3 -- one half-tile more per wait, paid for with one more half-tile of prefetch (PF 6 instead of 5).
        constexpr int WAITN = PB ? 2 * (PF - 3) : 2 * (PF - 2);
        if (Gtot > PF - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LR_BARRIER();
        if (wr == 1) LR_BARRIER();                      // stagger the second wave group by one barrier

        // PB = 1: two B fragment sets.  bf holds B0 for the whole K-tile (phases 0 and 3, read once), bg holds B1 (phases 1, 2); both
        // are requested from inside the preceding COMPUTE segment (bg during phase 0, the next K-tile's bf after phase 3's last
        // MFMA), so the LOAD segments of phases 1 and 3 carry no LDS reads and those of phases 0 and 2 only the 8 A reads.
        uint4 af[8], bf[4], bg[PB ? 4 : 1];
        if constexpr (PB == 1 && DBG != 4) {
#pragma unroll
            for (int f = 0; f < 4; ++f) bf[f] = *(const uint4*)(smem + 1 * HT + boff[f]);      // B0 of K-tile 0 (slot 1)
        }
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
                unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
                if constexpr (DBG == 3) t0 = stamp();
                // ---------------- LOAD ----------------
                const bool more = gi < Gtot;
                if constexpr (DBG != 4 && PB == 0)
                if (ph == 0 || ph == 1 || ph == 3) {
                    const char* sb = (ph == 1) ? sB1 : sB0;
#pragma unroll
                    for (int f = 0; f < 4; ++f) bf[f] = *(const uint4*)(sb + boff[f]);
                }
                if constexpr (DBG != 4)
                if (ph == 0 || ph == 2) {
                    const char* sa = (ph == 0) ? sA0 : sA1;
#pragma unroll
                    for (int f = 0; f < 8; ++f) af[f] = *(const uint4*)(sa + aoff[f]);
                }
                if (more) {
                    if constexpr (DBG != 5) issue((ph + PF) & 3, kt + ((ph + PF) >> 2), islot);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                ++gi;
                islot = (islot + 1 == NS) ? 0 : islot + 1;
                if constexpr (DBG == 3) t1 = stamp();
                LR_BARRIER();
                if constexpr (DBG == 3) t2 = stamp();
                // ---------------- COMPUTE ----------------
                __builtin_amdgcn_s_setprio(1);
                if constexpr (PB == 1 && DBG != 4) {
                    if (ph == 0) {                   // B1 of this K-tile: landed with the wait of LOAD(ph 0), first used in phase 1
#pragma unroll
                        for (int f = 0; f < 4; ++f) bg[f] = *(const uint4*)(sB1 + boff[f]);
                    }
                }
                if constexpr (DBG != 4)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if constexpr (PB == 1) {
                                const uint4 bb = (ph == 1 || ph == 2) ? bg[j * 2 + ks] : bf[j * 2 + ks];
                                acc[ph][i][j] = Op<OT>::mfma16(af[i * 2 + ks], bb, acc[ph][i][j]);
                            } else {
                                acc[ph][i][j] = Op<OT>::mfma16(af[i * 2 + ks], bf[j * 2 + ks], acc[ph][i][j]);
                            }
                        }
                if constexpr (PB == 1 && DBG != 4) {
                    if (ph == 3 && kt + 1 < nk) {    // B0 of the next K-tile (half-tile P + 2: landed with the wait of LOAD(ph 3))
                        int sn = rslot + 5; sn = sn >= NS ? sn - NS : sn;
#pragma unroll
                        for (int f = 0; f < 4; ++f) bf[f] = *(const uint4*)(smem + sn * HT + boff[f]);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                if constexpr (DBG == 3) t3 = stamp();
                LR_BARRIER();
                if constexpr (DBG == 3) {
                    t4 = stamp();
                    if (ph == 0) { const unsigned long long t5 = stamp(); lsegs[0] += (unsigned)(t5 - t4); }   // cost of one stamp
                    segs[ph][0] += (unsigned)(t1 - t0); segs[ph][1] += (unsigned)(t2 - t1);
                    segs[ph][2] += (unsigned)(t3 - t2); segs[ph][3] += (unsigned)(t4 - t3);
                }
            }
            rslot += 4;
            rslot = rslot >= NS ? rslot - NS : rslot;
        }
        if (wr == 0) LR_BARRIER();                      // balance the stagger barrier

        }

        if constexpr (DBG == 3) {   // diagnostic: write the segment sums of blocks 0..7 to the buffer passed as `bias`
            if (blockIdx.x < 8 && lane == 0 && vb == (int)blockIdx.x) {
                unsigned* dbg = (unsigned*)p.bias + ((size_t)blockIdx.x * 8 + wave) * 16;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b2 = 0; b2 < 4; ++b2) dbg[a * 4 + b2] = segs[a][b2];
                unsigned* dbg2 = (unsigned*)p.bias + 8 * 8 * 16 + ((size_t)blockIdx.x * 8 + wave) * 4;
                dbg2[0] = lsegs[0]; dbg2[1] = lsegs[1]; dbg2[2] = lsegs[2];
            }
        }
        if constexpr (DBG >= 2 && DBG <= 5) {   // diagnostics 2-5: no epilogue (keep the accumulators live); 6-8: epilogue without its C loads / stores / both
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(acc[q][i][0]), "v"(acc[q][i][1])); }
            vb += gridDim.x;
            continue;
        }

        tstamp(2);
        int claim = 0;              // requested here, looked at behind the epilogue: the round trip hides behind it
        if (dyn && tid == 0) claim = __hip_atomic_fetch_add(&p.sched[my_xcd], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The epilogue's per-lane index arithmetic starts HERE: the lane id is laundered so that nothing derived from it can be
        // hoisted above the K loop (a persistent kernel's tile loop makes all of it loop-invariant).  Hoisted, it lived in registers
        // the accumulators need, was spilled across the K loop, and the reloads' vmcnt waits -- placed by the compiler at the first
        // use, inside the residual K-tile loop -- drained the DMA ring every K-tile of the RoPE kernel.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        {
        const int lane = lane_e, l15 = lane_e & 15, l4 = lane_e >> 4;
        if constexpr (F8 == 1) {    // dequantise: C[m][n] *= ascale[m] * wscale[n]
            // (the pointers are laundered so that these ordinary loads cannot be hoisted above the K loop, where their vmcnt
            //  waits would drain the DMA ring in front of every ds_read)
            const float* asc = p.ascale;
            const float* wsc = p.wscale;
            asm volatile("" : "+s"(asc), "+s"(wsc) :: "memory");
            float sw[2][2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = n0 + (NW == 2 ? wc * 32 : wc * 64 + qb * 32) + j * 16 + l15;
                    sw[qb][j] = col < p.N ? wsc[col] : 0.f;
                }
#pragma unroll
            for (int qa = 0; qa < 2; ++qa)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + qa * 128 + wr * 64 + i * 16 + 4 * l4 + r;
                        const float sa = row < p.M ? asc[row] : 0.f;
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                            for (int j = 0; j < 2; ++j) acc[qa == 0 ? qb : 3 - qb][i][j][r] *= sa * sw[qb][j];
                    }
        }

        // operand out with one-byte residuals (GemmParams::oexp): 8 consecutive columns of a row per lane, 16 lanes (one DPP row) = one
        // 128-column block.  hi as 16 bytes, residuals as 8 e4m3 bytes scaled by the block's power of two, one scale byte per block.
        // Lanes past the matrix edge do not get here; their DPP contribution reads as zero.
        auto store_hi_lo8 = [&](const float (&v)[8], unsigned short* crow, int split, int col, unsigned char* scale_byte, bool writes_scale) {
            unsigned hp[4];
            float r[8];
            float m = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hp[e] = round_pair<OT>(v[2 * e], v[2 * e + 1]);
                r[2 * e] = v[2 * e] - Op<OT>::to_f32((unsigned short)(hp[e] & 0xFFFFu));
                r[2 * e + 1] = v[2 * e + 1] - Op<OT>::to_f32((unsigned short)(hp[e] >> 16));
                m = fmaxf(m, fmaxf(fabsf(r[2 * e]), fabsf(r[2 * e + 1])));
            }
            *(uint4*)(crow + col) = make_uint4(hp[0], hp[1], hp[2], hp[3]);
#if LR_EMU_FP6
            {
                const float bm = quad_max(m);
                m = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { r[e] = emu_e2m3(r[e], bm); m = fmaxf(m, fabsf(r[e])); }
            }
#endif
            const int E = e8m0_of_amax(row16_max(m));
            const float sc = e8m0_inv_scale(E);
            int p0 = 0, p1 = 0;
            p0 = __builtin_amdgcn_cvt_pk_fp8_f32(r[0] * sc, r[1] * sc, p0, false);
            p0 = __builtin_amdgcn_cvt_pk_fp8_f32(r[2] * sc, r[3] * sc, p0, true);
            p1 = __builtin_amdgcn_cvt_pk_fp8_f32(r[4] * sc, r[5] * sc, p1, false);
            p1 = __builtin_amdgcn_cvt_pk_fp8_f32(r[6] * sc, r[7] * sc, p1, true);
            *(uint2*)((unsigned char*)(crow + split) + col) = make_uint2((unsigned)p0, (unsigned)p1);
            if (writes_scale) *scale_byte = (unsigned char)E;
        };

        // ---- epilogue ----  quadrant index q -> (qa, qb): 0:(0,0) 1:(0,1) 2:(1,1) 3:(1,0)
        // The ring is idle (every DMA was retired by the vmcnt(0) of the tail phases), so the tile is staged
        // through LDS, 128 rows at a time, and leaves the CU as whole rows with 16 bytes per lane.
        // C/D map of 16x16x32: col = lane&15, row = 4*(lane>>4) + r.  Row stride 260 floats: 4 rows = 1040
        // floats = 16 banks (mod 32), so the row groups of one ds_write_b32 half-wave use disjoint banks.
        constexpr int SLD = 260;
        float* stg = (float*)smem;
#pragma unroll
        for (int qa = 0; qa < (NW == 1 ? 1 : 2); ++qa) {
            const int rowq = m0 + qa * 128;
            // residual rows are fetched around the staging pass so their latency hides behind it: 8 rows
            // before it, 8 right after the staging writes (when 64 accumulator registers have been freed)
            float4 ca[8], cb[8];
            const int fcol = n0 + lane * 4;
            // A tile that lies inside the matrix takes straight-line code (round 6): no bounds test around the residual-row loads and
            // the stores, so the wait counts the compiler places are exact counts -- not the vmcnt(0) a conditional block forces at
            // every join (see the store loop below) -- and the LDS reads of the loop can be issued ahead of the arithmetic.
            const bool interior = m0 + BM <= p.M && n0 + BN <= p.N;
            auto cload = [&](int it) {
                const int row = rowq + it * 8 + wave;
                return (row < p.M && fcol < p.N) ? *(const float4*)((const float*)p.C + (size_t)row * p.ldc + fcol)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            };
            auto cload_in = [&](int it) { return *(const float4*)((const float*)p.C + (size_t)(rowq + it * 8 + wave) * p.ldc + fcol); };
            if constexpr (E_ == EPI_RESADD_F32) {
                if (interior) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) ca[it] = (DBG == 6 || DBG == 8) ? make_float4(0.f, 0.f, 0.f, 0.f) : cload_in(it);
                } else {
#pragma unroll
                    for (int it = 0; it < 8; ++it) ca[it] = (DBG == 6 || DBG == 8) ? make_float4(0.f, 0.f, 0.f, 0.f) : cload(it);
                }
            }
            // RoPE epilogue: the (cos, sin) rows of iterations 0-3 are prefetched the same way (2 float4 each)
            const bool rot = E_ == EPI_ROPE_OP && n0 < p.rope_cols;         // rope_cols is a multiple of the tile width
            auto rload = [&](int it, int k) {
                const int row = min(rowq + it * 16 + wave * 2 + (lane >> 5), p.M - 1);
                const int col = n0 + (lane & 31) * 8;
                return ((const float4*)(p.rope_cs + ((size_t)row * (p.rope_hd >> 1) + ((col % p.rope_hd) >> 1)) * 2))[k];
            };
            // (round 6: fetched for every column tile, rotated or not -- the table is small and the addresses are always valid.  Under
            //  `if (rot)` every fetched register was a merge of two paths, the compiler put the copies right behind the loads, and a
            //  counted wait in front of each copy exposed the loads' latency twice per half)
            if constexpr (E_ == EPI_ROPE_OP) {
#pragma unroll
                for (int it = 0; it < 4; ++it) { ca[2 * it] = rload(it, 0); ca[2 * it + 1] = rload(it, 1); }
            }
            __syncthreads();
            tstamp(4 + 3 * qa);
#pragma unroll
            for (int qb = 0; qb < (NW == 2 ? 1 : 2); ++qb) {
                const int q = qa == 0 ? qb : 3 - qb;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            stg[(wr * 64 + i * 16 + 4 * l4 + r) * SLD + (NW == 2 ? wc * 32 : wc * 64 + qb * 32) + j * 16 + l15] = acc[q][i][j][r];
            }
            if constexpr (E_ == EPI_RESADD_F32) {
                if (interior) {
#pragma unroll
                    for (int it = 0; it < 8; ++it) cb[it] = (DBG == 6 || DBG == 8) ? make_float4(0.f, 0.f, 0.f, 0.f) : cload_in(8 + it);
                } else {
#pragma unroll
                    for (int it = 0; it < 8; ++it) cb[it] = (DBG == 6 || DBG == 8) ? make_float4(0.f, 0.f, 0.f, 0.f) : cload(8 + it);
                }
            }
            __syncthreads();
            tstamp(5 + 3 * qa);
            if constexpr (E_ == EPI_SWIGLU_OP) {
                // 128 output columns per row = 16 chunks of 8: 16 lanes per row, 4 rows per wave-iteration
                const int c8 = lane & 15, wcc = c8 >> 2, cc = c8 & 3;
                const int ocol = (n0 >> 1) + c8 * 8;
                float4 bg0 = make_float4(0.f, 0.f, 0.f, 0.f), bg1 = bg0, bu0 = bg0, bu1 = bg0;
                if constexpr (BIAS_) {
                    const int bc = n0 + wcc * 64 + cc * 8;                   // packed column of the gate chunk; up = +32
                    if (n0 + wcc * 64 + 64 <= p.N) {
                        bg0 = *(const float4*)(p.bias + bc); bg1 = *(const float4*)(p.bias + bc + 4);
                        bu0 = *(const float4*)(p.bias + bc + 32); bu1 = *(const float4*)(p.bias + bc + 36);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);          // the bias is retired in straight-line code (see the operand-out loop below)
                auto glu_rows = [&](auto in_tag) {
                constexpr bool IN = decltype(in_tag)::value != 0;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int rl = it * 32 + wave * 4 + (lane >> 4);
                    const int row = rowq + rl;
                    const float* g = stg + rl * SLD + wcc * 64 + cc * 8;
                    float4 g0 = *(const float4*)g, g1 = *(const float4*)(g + 4);
                    float4 u0 = *(const float4*)(g + 32), u1 = *(const float4*)(g + 36);
                    if constexpr (BIAS_) {
                        g0.x += bg0.x; g0.y += bg0.y; g0.z += bg0.z; g0.w += bg0.w; g1.x += bg1.x; g1.y += bg1.y; g1.z += bg1.z; g1.w += bg1.w;
                        u0.x += bu0.x; u0.y += bu0.y; u0.z += bu0.z; u0.w += bu0.w; u1.x += bu1.x; u1.y += bu1.y; u1.z += bu1.z; u1.w += bu1.w;
                    }
                    if (IN || (row < p.M && n0 + wcc * 64 + 64 <= p.N)) {
                        auto sw = [](float gg, float uu) { return uu * x_sigmoid_fast(gg, 1.f); };
                        const float v[8] = {sw(g0.x, u0.x), sw(g0.y, u0.y), sw(g0.z, u0.z), sw(g0.w, u0.w), sw(g1.x, u1.x), sw(g1.y, u1.y), sw(g1.z, u1.z), sw(g1.w, u1.w)};
                        if (p.oexp) {        // the 16 lanes of this row hold one 128-column block of the output
                            store_hi_lo8(v, (unsigned short*)p.C + (size_t)row * p.ldc, p.split, ocol, p.oexp + lo8_scale_at(row, ocol >> 7, p.M), (lane & 15) == 0);
                        } else {
                            uint4 w, wl;
                            split2p<OT>(v[0], v[1], w.x, wl.x); split2p<OT>(v[2], v[3], w.y, wl.y);
                            split2p<OT>(v[4], v[5], w.z, wl.z); split2p<OT>(v[6], v[7], w.w, wl.w);
                            *(uint4*)((unsigned short*)p.C + (size_t)row * p.ldc + ocol) = w;
                            if (p.split > 0) *(uint4*)((unsigned short*)p.C + (size_t)row * p.ldc + p.split + ocol) = wl;
                        }
                    }
                }
                };
                if (interior) glu_rows(IC<1>{}); else glu_rows(IC<0>{});
            } else if constexpr (E_ == EPI_ROPE_OP) {
                // as OUT_OP, no bias; the 8 columns of a lane are 4 (x[i], x[i+hd/2]) pairs of one head
                const int c8 = lane & 31;
                const int col = n0 + c8 * 8;
                float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f), rb1 = rb0;
                if constexpr (BIAS_) {
                    if (col < p.N) { rb0 = *(const float4*)(p.bias + col); rb1 = *(const float4*)(p.bias + col + 4); }
                }
                auto body = [&](auto in_tag, int it, const float4 a, const float4 bq) {      // a = (c0,s0,c1,s1), bq = (c2,s2,c3,s3)
                    constexpr bool IN = decltype(in_tag)::value != 0;
                    const int rl = it * 16 + wave * 2 + (lane >> 5);
                    const int row = rowq + rl;
                    const float* sp = stg + rl * SLD + c8 * 8;
                    float4 v0 = *(const float4*)sp, v1 = *(const float4*)(sp + 4);
                    if constexpr (BIAS_) {
                        v0.x += rb0.x; v0.y += rb0.y; v0.z += rb0.z; v0.w += rb0.w; v1.x += rb1.x; v1.y += rb1.y; v1.z += rb1.z; v1.w += rb1.w;
                    }
                    {          // (selected, not branched around: the loop stays straight-line code)
                        const float4 x0 = v0, x1 = v1;
                        float4 r0, r1;          // (common.h rope_pair: the reference's own arithmetic, no contraction left to the compiler)
                        rope_pair(x0.x, x0.y, a.x, a.y, r0.x, r0.y); rope_pair(x0.z, x0.w, a.z, a.w, r0.z, r0.w);
                        rope_pair(x1.x, x1.y, bq.x, bq.y, r1.x, r1.y); rope_pair(x1.z, x1.w, bq.z, bq.w, r1.z, r1.w);
                        v0 = rot ? r0 : x0;
                        v1 = rot ? r1 : x1;
                    }
                    if (IN || (row < p.M && col < p.N)) {
                        uint4 w, wl;
                        split2p<OT>(v0.x, v0.y, w.x, wl.x); split2p<OT>(v0.z, v0.w, w.y, wl.y);
                        split2p<OT>(v1.x, v1.y, w.z, wl.z); split2p<OT>(v1.z, v1.w, w.w, wl.w);
                        *(uint4*)((unsigned short*)p.C + (size_t)row * p.ldc + col) = w;
                        if (p.split > 0) *(uint4*)((unsigned short*)p.C + (size_t)row * p.ldc + p.split + col) = wl;
                    }
                };
                // the (cos, sin) rows of iterations 4-7 are requested as iterations 0-3 release their registers (one set of 8 float4
                // instead of two: with both prefetched the epilogue spilled ~100 registers per lane around the staging pass)
                auto rope_rows = [&](auto in_tag) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        body(in_tag, it, ca[2 * it], ca[2 * it + 1]);
                        ca[2 * it] = rload(4 + it, 0); ca[2 * it + 1] = rload(4 + it, 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);          // keep the second half's LDS reads out of the first half's live range
#pragma unroll
                    for (int it = 0; it < 4; ++it) body(in_tag, 4 + it, ca[2 * it], ca[2 * it + 1]);
                };
                if (interior) rope_rows(IC<1>{}); else rope_rows(IC<0>{});
            } else if constexpr (E_ == EPI_OUT_OP) {
                // 256 columns = 32 chunks of 8: 32 lanes per row, 2 rows per wave-iteration (256 x 128 tiles: 16 lanes, 4 rows)
                constexpr int LPR = NW == 2 ? 16 : 32, RPW = 64 / LPR;
                const int c8 = lane & (LPR - 1);
                const int col = n0 + c8 * 8;
                float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
                if (p.bias && col < p.N) { b0 = *(const float4*)(p.bias + col); b1 = *(const float4*)(p.bias + col + 4); }
                // The activation and the residual format are uniform over the launch: the loop is instantiated per (activation, format)
                // and picked ONCE (round 6).  Tested inside the loop -- as until then -- the two `p.act` comparisons were three scalar
                // branches per ELEMENT (64 per wave-iteration: the optimiser did not unswitch the partially unrolled loop) and `p.oexp`
                // one more per row.
                // The bias is retired HERE, in straight-line code (round 6): its first use sits in the loop's conditional block, and the
                // compiler's wait-count pass -- which cannot carry "already waited" across that join and the back edge -- had placed
                // `s_waitcnt vmcnt(0)` at the head of EVERY iteration: each pair of rows waited for the acknowledgement of the stores
                // of the pair before it.  A tile inside the matrix (IN) takes a body without the bounds test.
                __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0); lgkmcnt / expcnt untouched
                auto out_loop = [&](auto act_tag, auto oexp_tag, auto in_tag) {
                    constexpr int ACT = decltype(act_tag)::value;
                    constexpr bool OEXP = decltype(oexp_tag)::value != 0;
                    constexpr bool IN = decltype(in_tag)::value != 0;
#pragma unroll 2        // (4 and 8 measured level on every shape, round 6: the loop is not bound by its own latency chain)
                    for (int it = 0; it < 128 / (8 * RPW); ++it) {
                        const int rl = it * 8 * RPW + wave * RPW + lane / LPR;
                        const int row = rowq + rl;
                        const float* sp = stg + rl * SLD + c8 * 8;
                        float4 v0 = *(const float4*)sp, v1 = *(const float4*)(sp + 4);
                        if (IN || (row < p.M && col < p.N)) {
                            float v[8] = {v0.x + b0.x, v0.y + b0.y, v0.z + b0.z, v0.w + b0.w, v1.x + b1.x, v1.y + b1.y, v1.z + b1.z, v1.w + b1.w};
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                if constexpr (ACT == ACT_QUICK_GELU) v[e] = x_sigmoid_fast(v[e], 1.702f);
                                else if constexpr (ACT == ACT_GELU_ERF) v[e] = 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752440f));
                            }
                            if constexpr (OEXP) {        // each half-wave row holds two 128-column blocks: lanes 0-15 and 16-31 of it
                                store_hi_lo8(v, (unsigned short*)p.C + (size_t)row * p.ldc, p.split, col, p.oexp + lo8_scale_at(row, col >> 7, p.M), (lane & 15) == 0);
                            } else {
                                uint4 w, wl;
                                split2p<OT>(v[0], v[1], w.x, wl.x); split2p<OT>(v[2], v[3], w.y, wl.y);
                                split2p<OT>(v[4], v[5], w.z, wl.z); split2p<OT>(v[6], v[7], w.w, wl.w);
                                *(uint4*)((unsigned short*)p.C + (size_t)row * p.ldc + col) = w;
                                if (p.split > 0) *(uint4*)((unsigned short*)p.C + (size_t)row * p.ldc + p.split + col) = wl;
                            }
                        }
                    }
                };
                const bool oe = p.oexp != nullptr;
                auto pick = [&](auto in_tag) {
                    if (p.act == ACT_QUICK_GELU) { if (oe) out_loop(IC<ACT_QUICK_GELU>{}, IC<1>{}, in_tag); else out_loop(IC<ACT_QUICK_GELU>{}, IC<0>{}, in_tag); }
                    else if (p.act == ACT_GELU_ERF) { if (oe) out_loop(IC<ACT_GELU_ERF>{}, IC<1>{}, in_tag); else out_loop(IC<ACT_GELU_ERF>{}, IC<0>{}, in_tag); }
                    else { if (oe) out_loop(IC<ACT_NONE>{}, IC<1>{}, in_tag); else out_loop(IC<ACT_NONE>{}, IC<0>{}, in_tag); }
                };
                if (interior) pick(IC<1>{}); else pick(IC<0>{});
            } else {
                // fp32 out / residual add: 64 float4 per row, one row per wave-iteration (1 KB contiguous)
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias && fcol < p.N) bv = *(const float4*)(p.bias + fcol);
                // Every load this loop consumes (the residual rows, the bias) is retired HERE, in straight-line code, and the arithmetic
                // stays outside the bounds test (round 6).  With the additions inside `if (row < M ...)` the first use of the loaded
                // registers sat in a conditional block per iteration; the compiler's wait-count pass cannot carry "already waited" across
                // such a join, so it placed `s_waitcnt vmcnt(0)` in EVERY iteration -- behind the previous iteration's store, i.e. each
                // row waited for the acknowledgement of the row before it (4.4 us per 128-row half for 16 stores per wave).
                auto rows_out = [&](auto in_tag) {
                    constexpr bool IN = decltype(in_tag)::value != 0;
                    if constexpr (!IN) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0); lgkmcnt / expcnt untouched
#pragma unroll
                    for (int ib = 0; ib < 16; ib += 8) {          // 8 rows' LDS reads in flight, then their additions and stores
                        float4 r[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) r[i] = *(const float4*)(stg + ((ib + i) * 8 + wave) * SLD + lane * 4);
                        if constexpr (IN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int it = ib + i;
                            const int row = rowq + it * 8 + wave;
                            float4 v = r[i];
                            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                            if constexpr (E_ == EPI_RESADD_F32) { const float4 c = it < 8 ? ca[it & 7] : cb[it & 7]; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
                            if (IN || (row < p.M && fcol < p.N)) {
                                if constexpr (DBG == 7 || DBG == 8) { if (v.x == 1.2345e30f) *(float4*)((float*)p.C + (size_t)row * p.ldc + fcol) = v; }
                                else *(float4*)((float*)p.C + (size_t)row * p.ldc + fcol) = v;
                            }
                        }
                    }
                };
                if (interior) rows_out(IC<1>{}); else rows_out(IC<0>{});
            }
            tstamp(6 + 3 * qa);
        }
        }       // (epilogue scope: laundered lane id)
        tstamp(3);
        ++tile_it;
        if (dyn) {
            typedef __attribute__((address_space(3))) volatile int lds_int_t;                    // LDS accesses, not flat ones
            lds_int_t* next_l = (lds_int_t*)(lds_base + 9 * HT);                                 // behind the staging area
            if (tid == 0) {
                int Ln = -1;
                const int i = per_xcd + claim;
                if (i < chunk_n(my_xcd)) Ln = chunk0(my_xcd) + i;
                else {
                    for (int tries = 0; tries < 16 && Ln < 0; ++tries) {      // own chunk exhausted: help the XCD with the most tiles left
                        int v = -1, best = 0;
                        for (int y = 0; y < 8; ++y) {
                            const int rem = chunk_n(y) - per_xcd - __hip_atomic_load(&p.sched[y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (rem > best) { best = rem; v = y; }
                        }
                        if (v < 0) break;
                        const int j = per_xcd + __hip_atomic_fetch_add(&p.sched[v], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (j < chunk_n(v)) Ln = chunk0(v) + j;
                    }
                }
                if (Ln < 0) {          // this workgroup is done; the last one to get here leaves the words zero for the next launch
                    if (__hip_atomic_fetch_add(&p.sched[8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
                        for (int y = 0; y < 9; ++y) __hip_atomic_store(&p.sched[y], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                *next_l = Ln;
            }
            __syncthreads();      // staging reads done, the claim visible
            Ldyn = __builtin_amdgcn_readfirstlane(*next_l);        // uniform again: the tile coordinates stay in scalar registers
            __syncthreads();      // ... and read by every wave before the next tile's DMA reuses the ring
            if (Ldyn < 0) break;
        } else {
            __syncthreads();      // staging reads done before the next tile's DMA reuses the ring
            vb += gridDim.x;
        }
    }
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        LR_HIP_CHECK(hipGetDevice(&dev));
        LR_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n;
}

// The tile scheduler's words (GemmParams::sched) for callers that bring none (GemmParams::sched_mem: the lr_op_* entry points;
// an engine owns its words): one set per (device, stream) -- launches of one stream run one after the other and each leaves its
// words zero -- allocated and zeroed on that stream's first persistent launch, never freed.  Not for use under stream capture.
static int* sched_words(hipStream_t st) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int*> words;
    int dev = 0;
    LR_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = words.find({dev, st});
    if (it != words.end()) return it->second;
    int* w = nullptr;
    LR_HIP_CHECK(hipMalloc((void**)&w, 64));
    LR_HIP_CHECK(hipMemset(w, 0, 64));
    words[{dev, st}] = w;
    return w;
}

template <typename OT, int PF, int DBG, int EPI, int PB = 0, int F8 = 0, int NW = 0>
static void launch8(const GemmParams& p, bool persistent, hipStream_t st) {
    constexpr int NS = PB == 2 ? 8 : 10;      // ring slots; the product schedule keeps the rest of the 160 KB for scale slices
    constexpr int smem = 10 * 16384;
    static bool attr_set = false;
    auto kfn = gemm_bt8_kernel<OT, PF, NS, DBG, EPI, PB, F8, NW>;
    if (!attr_set) {
        LR_HIP_CHECK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    constexpr int BM = NW == 1 ? 128 : 256, BN = NW == 2 ? 128 : 256;
    const int Mt = (p.M + BM - 1) / BM, Nt = (p.N + BN - 1) / BN;
    // persistent: one resident workgroup per CU (160 KB LDS each) walking its tiles; else one workgroup per tile
    const int grid = persistent ? std::min(Mt * Nt, num_cus()) : Mt * Nt;
    GemmParams q = p;
    // Band height of the tile walk (an XCD's 32 concurrent tiles form a gm x 32/gm patch).  At the L2, 8 x 4 and 4 x 8 are the same
    // optimum; what differs is the Infinity Cache: W plus the A bands of the 8 XCDs in flight (8 gm x 256 rows x K) must fit its
    // 256 MB, or both are re-read from HBM at every patch step.  gate_up: W 151 MB + 151 MB of A at gm = 8 (thrashing) / 75 MB at
    // gm = 4.  Measured (tools/gm_bench.py, profiles/r4_gemm_band_height.log): gm = 4 -2.2 % over the step's GEMMs against gm = 8
    // (qkv -4 %), gm = 16 +6 %; with the XCDs' chunks cut at band boundaries (band_chunks below: -0.8 % more, LLaVA-7B's gate_up /
    // down, whose W alone exceeds the cache, -2.2 / -2.5 %) gm = 4 is best or level on every shape, deep K included.
    { const char* ge = getenv("LR_GEMM_GM"); q.gm = ge ? atoi(ge) : 0; if (q.gm < 1 || q.gm > 32) q.gm = 4; }
    static const bool dynamic = [] { const char* e = getenv("LR_GEMM_DYNAMIC"); return !e || atoi(e) != 0; }();
    q.sched = (persistent && PB == 2 && DBG != 2 && dynamic && grid % 8 == 0 && Mt * Nt >= 4 * grid) ? (p.sched_mem ? p.sched_mem : sched_words(st)) : nullptr;
    { const char* be = getenv("LR_GEMM_BANDCHUNK"); const int nb = (Mt + q.gm - 1) / q.gm;
      q.band_chunks = (q.sched && ((nb >> 3) - 1) * q.gm * Nt >= grid / 8 && (!be || atoi(be) != 0)) ? 1 : 0; }      // (every chunk holds its XCD's first tiles)
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), smem, st, q);
}

#ifdef LR_GEMM8_NARROW_TU
// The narrow-tile instantiations live in a translation unit of their own (gemm8_narrow.hip includes this file) so that the two
// halves compile side by side.  F16 operands, 16-bit (F8 = 0) and mixed (F8 = 2) forms; other modes keep 256 x 256 tiles.
template <int F8>
static void launch8_narrow_f8(const GemmParams& p, int nw, hipStream_t st) {
    if (nw == 2) { launch8<F16, 6, 0, EPI_OUT_OP, 2, F8, 2>(p, true, st); return; }
    switch (p.epi) {
        case EPI_OUT_OP: launch8<F16, 6, 0, EPI_OUT_OP, 2, F8, 1>(p, true, st); break;
        case EPI_OUT_F32: launch8<F16, 6, 0, EPI_OUT_F32, 2, F8, 1>(p, true, st); break;
        case EPI_RESADD_F32: launch8<F16, 6, 0, EPI_RESADD_F32, 2, F8, 1>(p, true, st); break;
        case EPI_SWIGLU_OP:
            if (p.bias) launch8<F16, 6, 0, EPI_SWIGLU_OP | 16, 2, F8, 1>(p, true, st);
            else launch8<F16, 6, 0, EPI_SWIGLU_OP, 2, F8, 1>(p, true, st);
            break;
        case EPI_ROPE_OP:
            if (p.bias) launch8<F16, 6, 0, EPI_ROPE_OP | 16, 2, F8, 1>(p, true, st);
            else launch8<F16, 6, 0, EPI_ROPE_OP, 2, F8, 1>(p, true, st);
            break;
        default: throw std::runtime_error("gemm_bt8: unknown epilogue");
    }
}
void launch8_narrow(const GemmParams& p, int f8, int nw, hipStream_t st) {
    if (f8 == 2) launch8_narrow_f8<2>(p, nw, st);
    else launch8_narrow_f8<0>(p, nw, st);
}
#else
void launch8_narrow(const GemmParams& p, int f8, int nw, hipStream_t st);

// Tile shape for one problem (product schedule): 0 = 256 x 256, 1 = 128 x 256, 2 = 256 x 128 (see the kernel's NW).  Free to look
// at M: every shape gives the same bits.  Measured (tools/narrow_bench.py): a 128-row tile costs 0.85 of a full one -- the K loop's
// DMA / LDS / barrier skeleton is the full tile's, only the MFMAs halve -- so it pays exactly when it does not add a round of tiles
// over the CUs: B = 1 o_proj / down (132 -> 264 tiles: -15 %), the gathered last layer (M = B rows: -16 %); a 256 x 128 tile for
// the adapters' t = x A^T costs 0.94 (-6 %).
int narrow_choice(const GemmParams& p) {
    const char* fe = getenv("LR_GEMM_NARROW");          // A/B and test switch, read per launch: 0 = never, 1 = 128-row tiles always
    const int force = fe ? atoi(fe) : -1;
    if (force == 0) return 0;
    const int E = p.epi;
    if (E == EPI_OUT_OP && p.N <= 128) return 2;
    const int cus = num_cus();
    const long Nt = (p.N + 255) / 256;
    const long full = (p.M + 255) / 256 * Nt, half = (p.M + 127) / 128 * Nt;
    const double t_full = (double)((full + cus - 1) / cus), t_half = 0.85 * (double)((half + cus - 1) / cus);
    return (force == 1 || t_half < 0.9 * t_full) ? 1 : 0;
}


template <typename OT, int PF, int DBG, int PB = 0, int F8 = 0>
static void launch8_epi(const GemmParams& p, bool persistent, hipStream_t st) {
    if constexpr (std::is_same<OT, F16>::value && PB == 2 && DBG == 0 && (F8 == 0 || F8 == 2)) {
        const int nw = narrow_choice(p);
        if (nw) { launch8_narrow(p, F8, nw, st); return; }
    }
    switch (p.epi) {
        case EPI_OUT_OP: launch8<OT, PF, DBG, EPI_OUT_OP, PB, F8>(p, persistent, st); break;
        case EPI_OUT_F32: launch8<OT, PF, DBG, EPI_OUT_F32, PB, F8>(p, persistent, st); break;
        case EPI_RESADD_F32: launch8<OT, PF, DBG, EPI_RESADD_F32, PB, F8>(p, persistent, st); break;
        case EPI_SWIGLU_OP:
            if (p.bias) launch8<OT, PF, DBG, EPI_SWIGLU_OP | 16, PB, F8>(p, persistent, st);
            else launch8<OT, PF, DBG, EPI_SWIGLU_OP, PB, F8>(p, persistent, st);
            break;
        case EPI_ROPE_OP:
            if (p.bias) launch8<OT, PF, DBG, EPI_ROPE_OP | 16, PB, F8>(p, persistent, st);
            else launch8<OT, PF, DBG, EPI_ROPE_OP, PB, F8>(p, persistent, st);
            break;
        default: throw std::runtime_error("gemm_bt8: unknown epilogue");
    }
}

// The K loop of the product kernel as a list of segments (GemmParams::KSeg).  f8: the kernel form (0 = 16-bit, 1 = W8A8, 2 = 16-bit
// pass + e4m3 residual pass).  Reads the caller's description -- kw / Wlo (split operands, inexact weights), aexp2 (e4m3 third
// segment), Wlo16 (16-bit third segment beside an e4m3 residual pass), A2 / W2 / W2lo / k2 (un-merged adapter) -- and fills
// seg / nseg / nk_f16 / nk_e1 / nk.  Columns are in 2-byte units: an e4m3 K-tile is 128 bytes = 64 units of its row.
static void build_segments(GemmParams& p, int f8) {
    int kt = 0, n = 0;
    auto add = [&](int tiles, int src, int a_col, int w_col) {
        if (tiles <= 0) return;
        if (n >= GemmParams::MAX_SEG) throw std::runtime_error("gemm_bt8: too many K segments");
        kt += tiles;
        p.seg[n++] = GemmParams::KSeg{kt, src, a_col, w_col};
    };
    auto adapter = [&](bool split) {         // t_hi x B, t_lo x B (split operands), t_hi x B_lo (B inexact in the operand type)
        if (!p.A2) return;
        if (!p.W2 || p.k2 <= 0 || p.k2 % 64 || p.lda2 % 8 || p.ldw2 % 8 || p.lda2 < (split ? 2 : 1) * p.k2 || p.ldw2 < p.k2 ||
            ((uintptr_t)p.A2 & 15) || ((uintptr_t)p.W2 & 15) || ((uintptr_t)p.W2lo & 15))
            throw std::runtime_error("gemm_bt8: bad K-extension (adapter) operands");
        add(p.k2 / 64, GemmParams::SRC_A2 | GemmParams::SRC_W2, 0, 0);
        if (split) add(p.k2 / 64, GemmParams::SRC_A2 | GemmParams::SRC_W2, p.k2, 0);
        if (p.W2lo) add(p.k2 / 64, GemmParams::SRC_A2 | GemmParams::SRC_W2 | GemmParams::SRC_LO, 0, 0);
    };
    if (f8 == 2) {
        const bool third8 = p.aexp2 != nullptr;
        add(p.kw / 64, 0, 0, 0);                                             // x_hi x W
        if (p.Wlo16) add(p.kw / 64, GemmParams::SRC_LO16, 0, 0);              // x_hi x W_lo, 16-bit
        adapter(true);
        p.nk_f16 = kt;
        add(p.kw / 128, GemmParams::SRC_LO, p.kw, 0);                        // e4m3(x_lo) x e4m3(W)
        p.nk_e1 = kt;
        if (third8) add(p.kw / 128, GemmParams::SRC_LO, p.kw + p.kw / 2, p.kw / 2);      // e4m3(x_hi) x e4m3(W_lo)
    } else if (f8 == 1) {
        if (p.A2) throw std::runtime_error("gemm_bt8_fp8: no K-extension in the W8A8 form (merge the adapter)");
        add(p.K / 64, 0, 0, 0);
        p.nk_f16 = p.nk_e1 = kt;
    } else if (p.kw > 0) {
        add(p.kw / 64, 0, 0, 0);                                             // x_hi x W
        add(p.kw / 64, 0, p.kw, 0);                                          // x_lo x W
        if (p.Wlo) add(p.kw / 64, GemmParams::SRC_LO, 0, 0);                  // x_hi x W_lo
        adapter(true);
        p.nk_f16 = p.nk_e1 = kt;
    } else {
        add(p.K / 64, 0, 0, 0);
        adapter(false);
        p.nk_f16 = p.nk_e1 = kt;
    }
    p.nseg = n;
    p.nk = kt;
}

template <typename OT>
static void launch8_variant(const GemmParams& p, int variant, hipStream_t st) {
    switch (variant) {
        case 3: case 5: launch8_epi<OT, 5, 0>(p, false, st); break;
        case 4: launch8_epi<OT, 5, 0>(p, true, st); break;               // persistent walk, B fragments read in the LOAD segments (A/B)
        case 6: { GemmParams q = p; build_segments(q, 0); launch8_epi<OT, 6, 0, 2>(q, true, st); break; }   // product: persistent walk, super-phase schedule
        case 10: launch8_epi<OT, 6, 0, 1>(p, true, st); break;           // A/B: 4-phase schedule + B fragments prefetched inside COMPUTE
        case 13: { GemmParams q = p; build_segments(q, 0); launch8<OT, 6, 3, EPI_OUT_F32, 2>(q, false, st); break; }  // diagnostic only: stamps of the super-phase schedule
        case 12: { GemmParams q = p; build_segments(q, 0); launch8<OT, 6, 2, EPI_OUT_F32, 2>(q, true, st); break; }   // diagnostic only: the product schedule without its epilogue (results not written)
        case 7: launch8<OT, 5, 1, EPI_OUT_F32>(p, false, st); break;     // diagnostic only: cache-resident operands
        case 8: launch8<OT, 5, 2, EPI_OUT_F32>(p, false, st); break;     // diagnostic only: no epilogue
        case 9: launch8<OT, 5, 3, EPI_OUT_F32>(p, false, st); break;     // diagnostic only: in-kernel stamps -> `bias` buffer
        case 14: launch8<OT, 5, 4, EPI_OUT_F32>(p, false, st); break;     // diagnostic only: LDS-DMA stream alone (no ds_read, no MFMA)
        case 15: launch8<OT, 5, 5, EPI_OUT_F32>(p, false, st); break;     // diagnostic only: ds_reads + MFMA alone (no LDS-DMA in the loop)
        default: throw std::runtime_error("gemm_bt8: unknown variant");
    }
}

// W8A8: A = e4m3 bytes [M, K8] with one fp32 scale per row, W = e4m3 bytes [N, K8] with one scale per output channel; the
// activations C leaves as are operand_dtype (f16 / bf16) or fp32, exactly as in the 16-bit form.  p.K / lda / ldw are given in
// BYTES here and halved for the kernel (2-byte units).
void launch_gemm_bt8_fp8(GemmParams p, int operand_dtype, hipStream_t st) {
    if (p.M <= 0) return;
    if (!p.ascale || !p.wscale) throw std::runtime_error("gemm_bt8_fp8: row / channel scales are required");
    if (p.K % 128 || p.lda % 16 || p.ldw % 16 || ((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15))
        throw std::runtime_error("gemm_bt8_fp8: K must be a multiple of 128 bytes, rows 16-byte aligned");
    if (p.kw || p.split || p.Wlo) throw std::runtime_error("gemm_bt8_fp8: no split-operand form");
    if (p.N % 8 || p.ldc % 8 || ((uintptr_t)p.C & 15) || (p.bias && ((uintptr_t)p.bias & 15)))
        throw std::runtime_error("gemm_bt8_fp8: N and ldc must be multiples of 8 and C/bias 16-byte aligned");
    if (p.epi == EPI_ROPE_OP && (!p.rope_cs || p.rope_hd % 16 || p.rope_cols % 256 || p.rope_cols % p.rope_hd || ((uintptr_t)p.rope_cs & 15)))
        throw std::runtime_error("gemm_bt8_fp8: bad RoPE epilogue parameters");
    p.K /= 2; p.lda /= 2; p.ldw /= 2;
    build_segments(p, 1);
    if (operand_dtype == DT_F16) launch8_epi<F16, 6, 0, 2, 1>(p, true, st);
    else launch8_epi<BF16, 6, 0, 2, 1>(p, true, st);
}

// Split-operand mode with the e4m3 residual pass (kernel form F8 == 2, see there).  p as for the 16-bit split form but
// K = kw + kw / 2 (2-byte units), Wlo = the rows that hold W8, aexp / wexp = the E8M0 scales.
void launch_gemm_bt8_mixed(const GemmParams& p0, int operand_dtype, hipStream_t st, int dbg) {
    if (p0.M <= 0) return;
    GemmParams p = p0;
    if (p.kw <= 0 || p.kw % 128 || !p.Wlo || !p.aexp || (p.aexp2 && p.Wlo16) || (((uintptr_t)p.aexp) & 15))
        throw std::runtime_error("gemm_bt8_mixed: needs kw % 128 == 0, W8 rows and block scales (third segment: e4m3 OR 16-bit, not both)");
    if (p.oexp && (p.split <= 0 || p.split % 128 || !(p.epi == EPI_OUT_OP || p.epi == EPI_SWIGLU_OP)))
        throw std::runtime_error("gemm_bt8_mixed: one-byte residual output needs a split operand output with columns % 128 == 0");
    build_segments(p, 2);
    if (p.N % 8 || p.ldc % 8 || ((uintptr_t)p.C & 15) || (p.bias && ((uintptr_t)p.bias & 15)))
        throw std::runtime_error("gemm_bt8_mixed: N and ldc must be multiples of 8 and C/bias 16-byte aligned");
    if (p.epi == EPI_ROPE_OP && (!p.rope_cs || p.rope_hd % 16 || p.rope_cols % 256 || p.rope_cols % p.rope_hd || ((uintptr_t)p.rope_cs & 15)))
        throw std::runtime_error("gemm_bt8_mixed: bad RoPE epilogue parameters");
    if (dbg == 2) {          // diagnostic (tools/gemm_epi_probe.py): the same K loop without the epilogue; results are not written
        if (operand_dtype != DT_F16) throw std::runtime_error("gemm_bt8_mixed: the no-epilogue diagnostic is built for F16 only");
        launch8<F16, 6, 2, EPI_OUT_F32, 2, 2>(p, true, st);
        return;
    }
    if (dbg == 9) {          // diagnostic (tools/gemm_timeline.py): per-tile time stamps into the buffer passed as `bias`
        p.ascale = p.bias; p.bias = nullptr;
        if (p.epi == EPI_RESADD_F32) launch8<F16, 6, 9, EPI_RESADD_F32, 2, 2>(p, true, st);
        else if (p.epi == EPI_OUT_OP) launch8<F16, 6, 9, EPI_OUT_OP, 2, 2>(p, true, st);
        else throw std::runtime_error("gemm_bt8_mixed: timeline diagnostic built for the residual-add and operand-out epilogues");
        return;
    }
    if (dbg >= 6 && dbg <= 8) {
        if (dbg == 6) launch8<F16, 6, 6, EPI_RESADD_F32, 2, 2>(p, true, st);
        else if (dbg == 7) launch8<F16, 6, 7, EPI_RESADD_F32, 2, 2>(p, true, st);
        else launch8<F16, 6, 8, EPI_RESADD_F32, 2, 2>(p, true, st);
        return;
    }
    if (operand_dtype == DT_F16) launch8_epi<F16, 6, 0, 2, 2>(p, true, st);
    else launch8_epi<BF16, 6, 0, 2, 2>(p, true, st);
}

#ifdef LR_FP6_AB
// Round 6 A/B (tools/fp6/, -DLR_FP6_AB=1 builds only): the split-operand GEMM with its residual pass in OCP MX FP6 (kernel form F8 == 3).
// p as for launch_gemm_bt8_mixed; the residual half of A and the rows of p.Wlo hold 96-byte FP6 K-tiles, p.aexp / p.wscale (bytes) the
// per-(row, 32 elements) scales in the kernel's slice order.  noepi: the K loop alone (results not written).
void launch_gemm_bt8_fp6ab(const GemmParams& p0, hipStream_t st, int noepi) {
    if (p0.M <= 0) return;
    GemmParams p = p0;
    if (p.kw <= 0 || p.kw % 128 || !p.Wlo || !p.aexp || !p.wscale || p.aexp2 || p.Wlo16 || p.A2 || p.oexp)
        throw std::runtime_error("gemm_bt8_fp6ab: needs kw % 128 == 0, FP6 twin rows and both scale arrays; exact weights, no adapter");
    build_segments(p, 2);
    if (noepi == 1) { launch8<F16, 6, 2, EPI_OUT_F32, 2, 3>(p, true, st); return; }
    if (noepi == 2) { launch8<F16, 6, 3, EPI_OUT_F32, 2, 3>(p, false, st); return; }      // in-kernel stamps -> the buffer passed as `bias` (FP6 residual tiles)
    if (noepi == 3) { p.wscale = nullptr; launch8<F16, 6, 3, EPI_OUT_F32, 2, 2>(p, false, st); return; }      // ... of the product's e4m3 form
    switch (p.epi) {
        case EPI_OUT_OP: launch8<F16, 6, 0, EPI_OUT_OP, 2, 3>(p, true, st); break;
        case EPI_OUT_F32: launch8<F16, 6, 0, EPI_OUT_F32, 2, 3>(p, true, st); break;
        case EPI_RESADD_F32: launch8<F16, 6, 0, EPI_RESADD_F32, 2, 3>(p, true, st); break;
        case EPI_SWIGLU_OP: launch8<F16, 6, 0, EPI_SWIGLU_OP, 2, 3>(p, true, st); break;
        default: throw std::runtime_error("gemm_bt8_fp6ab: epilogue not built");
    }
}
#endif

void launch_gemm_bt8(const GemmParams& p, int operand_dtype, int variant, hipStream_t st) {
    if (p.N % 8 || p.ldc % 8 || ((uintptr_t)p.C & 15) || (p.bias && ((uintptr_t)p.bias & 15)))
        throw std::runtime_error("gemm_bt8: N and ldc must be multiples of 8 and C/bias 16-byte aligned");
    if (p.kw > 0 && (p.kw % 64 || p.K != (p.Wlo ? 3 : 2) * p.kw)) throw std::runtime_error("gemm_bt8: split-operand mode needs K == 2 kw (3 kw with Wlo)");
    if (p.A2 && variant != 6) throw std::runtime_error("gemm_bt8: the K-extension exists in the product schedule only");
    if (p.epi == EPI_ROPE_OP && (!p.rope_cs || p.rope_hd % 16 || p.rope_cols % 256 || p.rope_cols % p.rope_hd || ((uintptr_t)p.rope_cs & 15)))
        throw std::runtime_error("gemm_bt8: bad RoPE epilogue parameters");
    if (operand_dtype == DT_F16) launch8_variant<F16>(p, variant, st);
    else launch8_variant<BF16>(p, variant, st);
}

#endif  // LR_GEMM8_NARROW_TU

}  // namespace lr
