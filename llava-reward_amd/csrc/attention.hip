// Fused softmax(Q K^T * scale + mask) V for the two towers of the scoring path:
//   * Phi-3 decoder self-attention, causal + padding mask, head_dim 96 (modeling_phi3_v.py:641-720;
//     additive mask semantics of :1453-1459: key j visible to query i iff j <= i and mask[j] == 1);
//   * CLIP encoder self-attention, dense, 577 tokens, head_dim 64 (transformers CLIPAttention,
//     scale head_dim^-0.5; call site modeling_phi3_v.py:212 / utils/utils.py:266-273).
// The S x S score matrix the reference's eager path materialises (0.9 GB fp32 per sample) never
// leaves registers: online softmax in fp32, operands in the 2-byte MFMA type.
//
// Work split: one workgroup = 4 waves = 128 query rows of one (sequence, head); each wave owns 32
// query rows and walks 64-key K/V tiles shared by the workgroup through LDS.
//   S^T = K Q^T   : A = K rows from LDS (ds_read_b128), B = Q^T from registers (each lane loaded its own
//                   query row once).  The 32x32 result has the QUERY on the lane and 16 keys in
//                   registers: row max / row sum are register ops + one exchange with lane^32.
//   O^T = V^T P^T : the score accumulators, converted pairwise to the operand type, ARE the B operand
//                   (cdna_hip_programming.md §3 "accumulator tile as the next MFMA's operand": k index
//                   of element j of lane half h = 16s + 8(j>>2) + 4h + (j&3)); A = V^T read straight
//                   from the row-major V image with ds_read_b64_tr_b16 (T10): per 16-lane group a
//                   4-key x 16-d block, lane i receives d-column i of 4 consecutive keys = elements
//                   j..j+3 of that fragment.  O^T has the query on the lane: rescale is lane-local.
// Data movement (the part that mattered): with register staging one tile per workgroup was in
// flight and each 64-key step cost ~5.8 us against ~0.4 us of MFMA work (latency-bound, 13 % MFMA
// utilisation).  K/V tiles now go HBM/L2 -> LDS by LDS-DMA into a 3-slot ring, two tiles ahead,
// retired with counted vmcnt + barrier (same discipline as gemm8.hip; the DMA is inline asm so hipcc
// does not drain it in front of every ds_read).  No ordinary global load lives in the loop: the key
// mask is turned into an LDS bitmask up front.  LDS-DMA writes lane-linear, so the images are
// unpadded and bank conflicts are removed by XOR-ing the 16-byte chunk index on the SOURCE side:
//   K (ds_read_b128, 16 rows per lane group):  HD 96: chunk ^ ((row>>2)&3);  HD 64: chunk ^ ((row>>1)&7);       HD 128: chunk ^ (row&15)
//   V (tr reads, 4 rows x 64 B per half-wave):  HD 96: none (192-B rows);     HD 64: chunk ^ (((row>>1)&1)<<2);  HD 128: chunk ^ ((row&3)<<2)
// Softmax diet: masks only on tiles that need them (wave-uniform), raw v_exp_f32 on log2-domain
// scores, packed conversions, O rescale only when a row maximum moved, dead diagonal sub-tiles skipped.
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace lr {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

template <typename OT> __device__ __forceinline__ unsigned pack2_fast(float lo, float hi);
template <> __device__ __forceinline__ unsigned pack2_fast<F16>(float lo, float hi) {      // values in [0,1]: no saturation needed
    const f32x2 x = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2));
}
template <> __device__ __forceinline__ unsigned pack2_fast<BF16>(float lo, float hi) {
    const f32x2 x = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}

// split2 (common.h) for values in [0, 1] (softmax weights): no saturation clamps, packed conversions -- 6 VALU operations per pair
// instead of ~18, bit-identical results (the clamps of Op::from_f32 never bind below 65504)
template <typename OT> __device__ __forceinline__ void split2_unit(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pack2_fast<OT>(a, b);
    lo = pack2_fast<OT>(a - Op<OT>::to_f32((unsigned short)(hi & 0xFFFFu)), b - Op<OT>::to_f32((unsigned short)(hi >> 16)));
}
// f16: the residual a - float(hi) as ONE v_fma_mix_f32 (fma(float(hi.half), -1, a): the f16 -> f32 widening rides in the instruction,
// the product is exact, so the value is the subtraction's, bit for bit) instead of a conversion and (half) a packed subtraction:
// 16 vector instructions fewer per 64-key tile and wave.  hipcc does not form it from the source above.
template <> __device__ __forceinline__ void split2_unit<F16>(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pack2_fast<F16>(a, b);
    lo = pack2_fast<F16>(sub_f16_lo_half(a, hi), sub_f16_hi_half(b, hi));
}

constexpr int ATT_MAX_S = 8192;        // keys covered by the LDS bitmask

// PREC (split-operand mode): Q, K, V rows carry their rounding residuals p.lo_off columns to the right; both
// contractions are evaluated as hi.hi + hi.lo + lo.hi (the lo.lo term is below 2^-22), P is split the same way in
// registers, and O leaves as [O_hi | O_lo].  K_lo / V_lo tiles ride in the same ring slot as K / V.
// NW = waves per workgroup: 4 (128 queries; two workgroups share a CU when LDS allows) or 8 (256 queries on one K/V ring: used
// by the split-operand mode, whose ring fills the LDS, so that every SIMD still holds two waves; waves 4-7 issue no DMA).
// PP = ping-pong schedule of an 8-wave workgroup (long sequences; see the comment at the loop).
template <typename OT, int HD, bool CAUSAL, bool PREC, int NW, bool PP = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void attn_kernel(AttnParams p) {   // <= 256 unified registers per wave, MFMA results stay in VGPRs
    constexpr int KT = 64;                 // keys per tile
    constexpr int KSTEPS = HD / 16;        // MFMA k-steps over the head dim
    constexpr int DT = HD / 32;            // 32-wide output tiles over the head dim
    constexpr int ROW = HD * 2;            // image row stride (bytes), unpadded (LDS-DMA is lane-linear)
    constexpr int CH = HD / 8;             // 16-byte chunks per row
    constexpr int NPO = KT * CH / 256;     // DMA instructions per thread per operand tile (3 or 2)
    constexpr int NOPS = PREC ? 4 : 2;     // operand tiles per ring slot: K, V (, K_lo, V_lo)
    constexpr int NPT = NOPS * NPO;        // DMA instructions per thread per slot
    constexpr int TILE = KT * ROW;         // bytes of one operand tile
    // 160 KB of LDS: 2 x 4 x 16 KB is all that fits at HD 128; HD 64 split-operand 4-wave form: 2 slots (64 KB) so that two workgroups share a CU
    constexpr int NSLOT = (PP && !PREC) ? 4 : (PREC && (HD == 128 || (HD == 64 && NW == 4))) ? 2 : 3;
    constexpr int PD = PP ? 2 : NSLOT - 1; // tiles in flight ahead of the one being consumed
    static_assert(!PP || (NW == 8 && NSLOT >= 3), "ping-pong schedule: 8 waves, at least 3 ring slots");
    constexpr int DMAG = (PP && PREC) ? 1 : 0;     // which half of an 8-wave workgroup issues the DMA
    static_assert(KT * CH % 256 == 0, "tile/threads mismatch");

    __shared__ __attribute__((aligned(16))) char smem[NSLOT * NOPS * TILE + ATT_MAX_S / 8];
    unsigned* sBits = (unsigned*)(smem + NSLOT * NOPS * TILE);  // bit k = key k visible (before the causal rule)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 31, lh = lane >> 5;
    int head = blockIdx.y, b = blockIdx.z, bx = blockIdx.x;
    if (p.lin_nqt > 0) {                           // XCD-aware order (AttnParams::lin_nqt): id = ((kvpair / 8) * (G nqt) + (g nqt + rank)) * 8 + kvpair % 8
        const int id = blockIdx.x, G = p.kv_group, per = G * p.lin_nqt;
        const int k = id >> 3, kvpair = (k / per) * 8 + (id & 7), r = k % per;
        const int kvheads = p.heads / G;
        if (kvpair >= kvheads * p.lin_batch) return;          // (the grid is padded to whole groups of 8 pairs)
        b = kvpair / kvheads;
        head = (kvpair % kvheads) * G + r / p.lin_nqt;
        bx = r % p.lin_nqt;
    }
    int S = p.S, qt, qsel = -1, qshift = 0;
    size_t rowbase;
    if (!CAUSAL && p.items) {                      // ragged mode: one segment per workgroup
        const int4 it = p.items[bx];
        rowbase = (size_t)it.x; S = it.y; qt = it.z;
    } else if (CAUSAL && p.qsel) {                 // gathered mode: the query tile that holds the wanted query of sequence b
        qsel = __builtin_amdgcn_readfirstlane(p.qsel_last ? S - 1 : min(max(p.qsel[b * p.qsel_stride], 0), S - 1));
        qt = qsel / (NW * 32);
        rowbase = (size_t)b * S;
    } else {
        const int nqt = (S + NW * 32 - 1) / (NW * 32);
        qt = nqt - 1 - bx;                         // heavy (late) causal tiles first
        rowbase = (size_t)b * S;
        // Causal: the query tiles are shifted towards the END of the sequence (by whole key tiles, so that a wave's diagonal stays
        // inside one key tile), which makes the partial workgroup the FIRST one (a few key tiles of work) instead of the last (all of
        // them): at S = 2642 with 256-query workgroups 242 instead of 262 tile periods per (sequence, head), and the heaviest
        // workgroup no longer runs with 5 of its 8 waves past the sequence.  Queries below 0 are dead lanes / waves.  A query's
        // arithmetic does not depend on the lane or tile that holds it (key tiles stay aligned to kbeg; rescales by alpha = 1 and
        // fully masked tiles are exact no-ops), so the outputs are bit-identical to the front-aligned tiling (tools/dbg/attn_equal.py).
        // Ping-pong kernels only (S >= 1024: one workgroup per CU, lock-stepped tile periods: -10 % at S = 2642 together with the
        // XCD-aware order, tools/dbg/attn_equal.py); the shorter-sequence forms, whose waves are not lock-stepped in pairs, measured
        // level to worse with it (S = 800 .. 960: +1 .. +9 %) and keep the front-aligned tiles.
        if (CAUSAL && PP && p.q_end_aligned) qshift = (nqt * (NW * 32) - S) / KT * KT;
    }
    const int q0 = qt * (NW * 32) - qshift + wave * 32;

    const unsigned short* Qp = (const unsigned short*)p.Q + p.qoff + head * HD;
    const int kvh = head / p.kv_group;                 // GQA: several query heads share one key/value head
    const unsigned short* Kp = (const unsigned short*)p.K + p.koff + kvh * HD;
    const unsigned short* Vp = (const unsigned short*)p.V + p.voff + kvh * HD;

    // ---- key range of this workgroup ----
    int kbeg = 0, kend = S;
    if (CAUSAL) {
        kend = min(S, qt * (NW * 32) - qshift + NW * 32);
        if (p.kmin) kbeg = (min(p.kmin[b * p.kmin_stride], S) / KT) * KT;
    }
    const int ntiles = kbeg < kend ? (kend - kbeg + KT - 1) / KT : 0;

    // ---- LDS-DMA: per-thread (row, swizzled chunk) of each piece; destination is lane-linear ----
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char*)smem;
    // Source address of a piece = a UNIFORM 64-bit base (operand, sequence, kv head, first key of the tile: SGPRs, rebuilt per tile
    // with scalar arithmetic) + a per-thread 32-bit byte offset ((row, swizzled chunk) of the piece inside a tile: 2 NPO registers,
    // constant over the kernel) -- the saddr form of the DMA instruction.  Until round 5 every piece's 64-bit per-lane address was
    // rebuilt with vector arithmetic at every issue (~20 VALU operations per piece): in-kernel stamps showed the 12 pieces of a
    // split-operand tile costing the issuing group 2200 cycles of its 4000-cycle vector segment, the critical path of the ping-pong
    // period once the matrix segments were pipelined.  The clamp that keeps a last, partial tile's rows inside the sequence
    // (min(key, S - 1): rows past it are masked scores, but must not be read past the buffer) is a per-tile uniform branch.
    unsigned offK[NPO], offV[NPO];
#pragma unroll
    for (int it = 0; it < NPO; ++it) {
        const int q = it * 256 + (tid & 255);
        const int r = q / CH, c = q - r * CH;
        const int dkc = (HD == 96 ? (c ^ ((r >> 2) & 3)) : HD == 64 ? (c ^ ((r >> 1) & 7)) : (c ^ (r & 15))) * 8;
        const int dvc = (HD == 96 ? c : HD == 64 ? (c ^ (((r >> 1) & 1) << 2)) : (c ^ ((r & 3) << 2))) * 8;
        offK[it] = (unsigned)(r * p.ldq + dkc) * 2u;
        offV[it] = (unsigned)(r * p.ldq + dvc) * 2u;
    }
    auto dma = [&](const unsigned short* base, unsigned off, unsigned dst) {       // base: uniform
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
    };
    auto uni = [&](const unsigned short* q) {          // a pointer the compiler may not have proved uniform -> SGPR pair
        const unsigned long long u = (unsigned long long)q;
        return (const unsigned short*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u));
    };
    // SPLIT_DMA (ping-pong, split operands; round 5): the early group issues the K / K_lo pieces of tile t + 2, the late group its
    // V / V_lo pieces, each in its own vector segment t and each retiring its pieces (vmcnt(0)) at the end of its matrix segment t.
    // Legal on the 3-slot ring: the K region of slot (t + 2) % 3 was last read by the late group's QK(t - 1), i.e. in its matrix
    // segment t - 2, which ended two barriers before the early group's vector segment t; the V region is still being read by the
    // late group's PV(t - 1) then, so only the late group (whose vector segment t follows that) may refill it -- as before.  Why:
    // a piece costs its issuer ~170 cycles of VMEM issue (stamps: 12 pieces = 2000 of the late group's 3100-cycle vector segment
    // against 1100 for the early group's); halved, both groups' vector segments fit under the partner's matrix segment.
#ifndef LR_ATT_SPLIT_DMA
#define LR_ATT_SPLIT_DMA 0
#endif
    constexpr bool SPLIT_DMA = PP && PREC && LR_ATT_SPLIT_DMA;
    auto issue = [&](int t, bool all = false) {            // tile t (keys kbeg + 64 t ..) -> slot t % NSLOT; all: the prologue's tiles
        const bool both = all || !SPLIT_DMA;
        if (both && NW > 4 && (wave >> 2) != DMAG) return;  // the DMA pieces are laid out for 256 threads
        const bool doK = both || (wave >> 2) == 0, doV = both || (wave >> 2) == 1;
        const int slot = t % NSLOT;
        const int k0 = kbeg + t * KT;
        const unsigned dstK = __builtin_amdgcn_readfirstlane(lds_base + slot * NOPS * TILE + (wave & 3) * 1024);
        const unsigned short* bK = uni(Kp + (rowbase + k0) * p.ldq);
        const unsigned short* bV = uni(Vp + (rowbase + k0) * p.ldq);
        if (k0 + KT <= S) {
#pragma unroll
            for (int it = 0; it < NPO; ++it) {
                const unsigned dk = dstK + it * 4096;
                if (doK) dma(bK, offK[it], dk);
                if (doV) dma(bV, offV[it], dk + TILE);
                if constexpr (PREC) {
                    if (doK) dma(bK + p.lo_off, offK[it], dk + 2 * TILE);
                    if (doV) dma(bV + p.lo_off, offV[it], dk + 3 * TILE);
                }
            }
        } else {                          // the sequence ends inside this tile: rows clamped to its last key
#pragma unroll
            for (int it = 0; it < NPO; ++it) {
                const int r = (it * 256 + (tid & 255)) / CH;
                const unsigned back = (unsigned)(max(k0 + r - (S - 1), 0) * p.ldq) * 2u;
                const unsigned dk = dstK + it * 4096;
                if (doK) dma(bK, offK[it] - back, dk);
                if (doV) dma(bV, offV[it] - back, dk + TILE);
                if constexpr (PREC) {
                    if (doK) dma(bK + p.lo_off, offK[it] - back, dk + 2 * TILE);
                    if (doV) dma(bV + p.lo_off, offV[it] - back, dk + 3 * TILE);
                }
            }
        }
    };

    // ---- prologue: the Q fragments requested FIRST, then the first two tiles by LDS-DMA, then the key bitmask ----
    // (round 6: the order used to be tiles, bitmask, Q.  The bitmask's mask load -- or, without a mask, the join behind its branch --
    //  carries a `vmcnt(0)`, which retires in order: the tiles had to LAND before the Q loads were even issued, and their latency
    //  followed: two exposed round trips per workgroup instead of one.  The 577-token CLIP workgroups and the ragged Qwen windows live
    //  for ten key tiles or fewer.)
    uint4 qf[KSTEPS];       // lane (c,h) holds Q[q0+c][16*ks + 8h .. +7]
    {
        const int qrow = max(min(q0 + lc, S - 1), 0);
        const unsigned short* src = Qp + (rowbase + qrow) * p.ldq + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *(const uint4*)(src + 16 * ks);
    }
    uint4 qfl[PREC ? KSTEPS : 1];       // residuals of the same Q elements
    if constexpr (PREC) {
        const int qrow = max(min(q0 + lc, S - 1), 0);
        const unsigned short* src = Qp + (rowbase + qrow) * p.ldq + p.lo_off + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qfl[ks] = *(const uint4*)(src + 16 * ks);
    }
    __builtin_amdgcn_sched_barrier(0);          // (the requests stay in front of the DMA issue)
    if (ntiles > 0) issue(0, true);
    if (PD > 1 && ntiles > 1) issue(1, true);
    {
        const int w0 = kbeg / KT, w1 = (kend + KT - 1) / KT;
        if (p.mask) {
            // A wave's mask words, eight key tiles at a time: all loads first (unconditional, addresses clamped), then the ballots.
            // One tile per iteration -- load, wait, ballot -- exposed a round trip per key tile: five or six per wave and workgroup at
            // S = 2642, ~4 us of a ~90 us workgroup (round 6).
            constexpr int J = 8;
            const int64_t* mrow = p.mask + (size_t)b * S;
            for (int w = w0 + wave; w < w1; w += NW * J) {
                int64_t mv[J];
#pragma unroll
                for (int j = 0; j < J; ++j) mv[j] = mrow[min(min(w + j * NW, w1 - 1) * KT + lane, S - 1)];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int ww = w + j * NW;
                    const unsigned long long bits = __ballot(ww * KT + lane < S && mv[j] != 0);
                    if (ww < w1 && lane == 0) { sBits[2 * ww] = (unsigned)bits; sBits[2 * ww + 1] = (unsigned)(bits >> 32); }
                }
            }
        } else {
            for (int w = w0 + wave; w < w1; w += NW) {
                const unsigned long long bits = __ballot(w * KT + lane < S);
                if (lane == 0) { sBits[2 * w] = (unsigned)bits; sBits[2 * w + 1] = (unsigned)(bits >> 32); }
            }
        }
    }
    // Retire the ordinary loads HERE (vmcnt(0), expcnt/lgkmcnt untouched).  Otherwise hipcc puts its waits
    // for the Q loads at their first use inside the loop, where a vmcnt(0) would drain the DMA ring on
    // every iteration.  Tiles 0 and 1 are needed before the first MFMA anyway.
    __builtin_amdgcn_s_waitcnt(0x0F70);

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;                // running max in log2 units, running denominator
    const float sc = p.scale * 1.4426950408889634f;
    const int qpos = q0 + lc;

    // fragment read offsets (bytes inside an operand tile)
    // K row of a lane = 32 kt + lc, 16-byte chunk 2 ks + lh, swizzled by XOR with m(row) (see the header): m does not depend on kt,
    // so the offsets of a lane are kt * 32 ROW + per-lane values that depend on the swizzled low bits of the chunk only.  HD 96: m
    // has 2 bits -> the chunk's bits above them (ks >> 1) are a compile-time 64-byte step and TWO per-lane values (ks even / odd)
    // serve all 12 reads (a 12-register table otherwise; the split-operand instantiation has none to spare).
    int kbase[HD == 96 ? 2 : KSTEPS];
    {
        const int m = HD == 96 ? ((lc >> 2) & 3) : HD == 64 ? ((lc >> 1) & 7) : (lc & 15);
#pragma unroll
        for (int i = 0; i < (HD == 96 ? 2 : KSTEPS); ++i) kbase[i] = lc * ROW + (((2 * i + lh) ^ m) << 4);
    }
    auto koff_of = [&](int kt, int ks) { return HD == 96 ? kbase[ks & 1] + kt * 32 * ROW + 64 * (ks >> 1) : kbase[ks] + kt * 32 * ROW; };
    // transposed V read: group g = lane>>4 covers keys 4h + q (q = (lane&15)>>2) and d columns
    // 16*(g&1) + 4*(lane&3) .. +3 of a 4-key x 16-d block; for HD 64 the 64-B window of rows 2,3 (mod 4) is swapped
    const int vq = (lane & 15) >> 2;
    int voff[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        const int row = 4 * lh + vq;                                  // + multiples of 8 keys: (row>>1)&1 unchanged
        const int win = HD == 96 ? d : HD == 64 ? (d ^ ((row >> 1) & 1)) : (d ^ (row & 3));
        voff[d] = row * ROW + win * 64 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    }

        f32x16 s[2];
        uint4 pf[4], pl[PREC ? 4 : 1];
        // Fragment reads run PIPE steps ahead of the MFMAs that consume them (round 5).  As compiled until then, every K / V
        // fragment of the matrix segment was requested, waited for (lgkmcnt(0)) and consumed: one LDS latency exposed per 1-3
        // MFMAs, 48 times per tile and wave -- 2100 of the segment's 3870 cycles (profiles/r4_attention_stamps.log: "no fragment
        // reads" 1770).  The LDS array itself was 14 % busy: latency, not bandwidth.  The steps are written in issue order and the
        // order is pinned with scheduling groups (reads of step n + PIPE, then the MFMAs of step n); the waits are the compiler's
        // own counted lgkmcnt (LDS returns in order).  Same MFMAs in the same order per accumulator: bit-identical results.
#ifndef LR_ATT_PIPE
#define LR_ATT_PIPE 3
#endif
#ifndef LR_ATT_CVT_IN_PV
#define LR_ATT_CVT_IN_PV 1
#endif
#ifndef LR_ATT_FUSE
#define LR_ATT_FUSE 0                  // PV(t) and QK(t + 1) of a ping-pong matrix segment as ONE pipeline of fragment reads
#endif
#ifndef LR_ATT_PRIO
#define LR_ATT_PRIO 1                  // s_setprio of the ping-pong loop: 1 = the matrix segment goes first, 2 = the vector segment, 0 = none
#endif
        constexpr int PIPE = PREC ? LR_ATT_PIPE : 2 * LR_ATT_PIPE;   // steps ahead; a step = 3 MFMAs (split operands) or 1
        // One software pipeline over the steps of PV(t) [if DO_PV] followed by QK(tq) [if DO_QK]: the fragment reads of step n + PIPE are
        // issued in front of the MFMAs of step n, ACROSS the seam between the two contractions (LR_ATT_FUSE: the first K fragments of
        // QK(t + 1) are requested under the last MFMAs of PV(t), instead of a second pipeline fill per matrix segment).
        // Declared below cvt_chunk (it converts P chunks on the way); qk / pv are its two halves alone.
        auto cvt_chunk = [&](int c) {          // keys 16 c .. 16 c + 15 of the tile: softmax weights in s -> MFMA operands pf / pl
            const int kt = c >> 1, st = c & 1;
            uint4& f = pf[c];
            if constexpr (PREC) {
                uint4& g = pl[c];
                split2_unit<OT>(s[kt][8 * st + 0], s[kt][8 * st + 1], f.x, g.x);
                split2_unit<OT>(s[kt][8 * st + 2], s[kt][8 * st + 3], f.y, g.y);
                split2_unit<OT>(s[kt][8 * st + 4], s[kt][8 * st + 5], f.z, g.z);
                split2_unit<OT>(s[kt][8 * st + 6], s[kt][8 * st + 7], f.w, g.w);
            } else {
                f.x = pack2_fast<OT>(s[kt][8 * st + 0], s[kt][8 * st + 1]);
                f.y = pack2_fast<OT>(s[kt][8 * st + 2], s[kt][8 * st + 3]);
                f.z = pack2_fast<OT>(s[kt][8 * st + 4], s[kt][8 * st + 5]);
                f.w = pack2_fast<OT>(s[kt][8 * st + 6], s[kt][8 * st + 7]);
            }
        };
        auto soft = [&](int t) {
#if defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 1)      // diagnostic build (tools/dbg): no softmax arithmetic, results invalid
            for (int kt = 0; kt < 2; ++kt)
                for (int st = 0; st < 2; ++st) {
                    uint4& f = pf[2 * kt + st];
                    f.x = pack2_fast<OT>(s[kt][8 * st + 0], s[kt][8 * st + 1]); f.y = pack2_fast<OT>(s[kt][8 * st + 2], s[kt][8 * st + 3]);
                    f.z = pack2_fast<OT>(s[kt][8 * st + 4], s[kt][8 * st + 5]); f.w = pack2_fast<OT>(s[kt][8 * st + 6], s[kt][8 * st + 7]);
                    if constexpr (PREC) pl[2 * kt + st] = f;
                }
            return;
#endif
            const int k0 = kbeg + t * KT;
#if !(defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 1024))       // diagnostic 1024: no key mask
            const unsigned blo = __builtin_amdgcn_readfirstlane(sBits[2 * (k0 / KT)]);
            const unsigned bhi = __builtin_amdgcn_readfirstlane(sBits[2 * (k0 / KT) + 1]);
            const bool need_causal = CAUSAL && (k0 + KT - 1 > q0);
            if (need_causal || (blo & bhi) != 0xFFFFFFFFu) {
                const unsigned wl = blo >> (4 * lh), wh = bhi >> (4 * lh);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kb = (r & 3) + 8 * (r >> 2);
                        bool ok = (((kt ? wh : wl) >> kb) & 1u) != 0;
                        if (CAUSAL) ok = ok && (k0 + kt * 32 + kb + 4 * lh <= qpos);
                        s[kt][r] = ok ? s[kt][r] : -INFINITY;
                    }
            }
#endif
#if defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 256)           // diagnostic 256: no row maximum, no rescale
            const float m_new = m_run;
#else
            float mx = s[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // Lazy reference maximum (split-operand forms; AttnParams::lazy_t, a RUN-TIME threshold in log2 units since round 6): a row's
            // reference moves only when its new maximum exceeds it by more than lazy_t, so the softmax weights of a tile are <= 2^lazy_t
            // instead of <= 1 (exact in the hi + lo operand pair and in the fp32 sums alike) and the rescale of the 16 DT output
            // registers -- taken for SOME row of the wave in most tiles of random data -- becomes rare.  Per query: a row's arithmetic
            // does not depend on the other rows of its wave (alpha == 1 exactly where nothing moved).  lazy_t == 0 IS the exact running
            // maximum, bit for bit (x > m ? x : m == fmaxf(m, x) for the finite values here): the strict form's passes run with it
            // (the yardstick keeps the reference's own arithmetic), the e4m3-residual default form with 8; the single-pass forms
            // (PREC false) always take the exact maximum.  What the threshold costs in accuracy is the exponent's argument: fma(s, sc,
            // -m_run) is rounded at ulp(lazy_t) instead of ulp(~0), i.e. up to 2^-21 absolute at 8 against 2^-24 -- fp32-level noise on
            // the softmax weights that an outlier-bearing model amplifies.
            const float mxs = mx * sc;
            const float m_new = PREC ? (mxs > m_run + p.lazy_t ? mxs : m_run) : fmaxf(m_run, mxs);
#endif
            if (!__all(m_new == m_run)) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#if defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 128)           // diagnostic 128: no exp
                    const float e = __builtin_fmaf(s[kt][r], sc, -m_run);
#else
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sc, -m_run));
#endif
                    s[kt][r] = e;
#if !(defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 512))        // diagnostic 512: no row sum
                    rs += e;
#endif
                }
#if !(defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 512))
            rs += __shfl_xor(rs, 32, 64);
#endif
            l_run += rs;
            // P -> operand type (hi, lo): chunk 0 here; chunks 1-3 inside the matrix segment, each under the MFMAs of the chunk before
            // it (LR_ATT_CVT_IN_PV, round 5: with the matrix segments pipelined the VECTOR segments set the period -- 3000 cycles
            // against 2550 -- and the matrix wave issues on 8 of every 32 cycles: the 20 conversions of a chunk ride there for free)
#pragma unroll
            for (int c = 0; c < (LR_ATT_CVT_IN_PV ? 1 : 4); ++c) cvt_chunk(c);
        };
        auto mat = [&](int t, int tq, auto do_pv, auto do_qk) {
            constexpr bool DO_PV = decltype(do_pv)::value, DO_QK = decltype(do_qk)::value;
            // LDS addresses as 32-bit integers: one VALU add per d block and tile, every other term (key step, +8 rows, the residual
            // tile) in the read's 16-bit offset field (through generic pointers the compiler spent two VALU operations per read: 96
            // per tile in the segment that should issue nothing but reads and MFMAs)
            constexpr int NP = DO_PV ? 4 * DT : 0;       // PV steps: (kt, st, d), d fastest
            constexpr int NQ = DO_QK ? 2 * KSTEPS : 0;   // QK steps: (kt, ks)
            constexpr int NS = NP + NQ;
            const unsigned vslot = __builtin_amdgcn_readfirstlane(lds_base + (t % NSLOT) * NOPS * TILE + TILE);
            const char* sK = smem + (tq % NSLOT) * NOPS * TILE;
            unsigned va[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) va[d] = vslot + voff[d];
            uint4 fh[NS], fl[PREC ? NS : 1];             // hi / lo fragment of every step (SSA values: the allocator keeps PIPE + 1 alive)
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            auto trd = [&](unsigned a, int off) {
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a + off));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a + off + 8 * ROW));
                const uint2 x = __builtin_bit_cast(uint2, v0), y = __builtin_bit_cast(uint2, v1);
                return make_uint4(x.x, x.y, y.x, y.y);
            };
            auto ld = [&](int n) {
#if defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 2)      // diagnostic 2: no LDS fragment reads (register stand-ins), results invalid
                fh[n] = qf[n % KSTEPS];
                if constexpr (PREC) fl[n] = qfl[n % KSTEPS];
                return;
#endif
                if (n < NP) {
                    const int kt = n / (2 * DT), st = (n / DT) & 1, d = n % DT;
                    fh[n] = trd(va[d], (kt * 32 + 16 * st) * ROW);
                    if constexpr (PREC) fl[n] = trd(va[d], (kt * 32 + 16 * st) * ROW + 2 * TILE);
                } else {
                    const int m = n - NP, kt = m / KSTEPS, ks = m % KSTEPS;
                    fh[n] = *(const uint4*)(sK + koff_of(kt, ks));
                    if constexpr (PREC) fl[n] = *(const uint4*)(sK + 2 * TILE + koff_of(kt, ks));
                }
            };
#pragma unroll
            for (int n = 0; n < PIPE && n < NS; ++n) ld(n);
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                if (n + PIPE < NS) ld(n + PIPE);
                if (n < NP) {
                    const int kt = n / (2 * DT), st = (n / DT) & 1, d = n % DT;
                    o[d] = Op<OT>::mfma32(fh[n], pf[2 * kt + st], o[d]);
                    if (LR_ATT_CVT_IN_PV && d == 0 && n / DT + 1 < 4) cvt_chunk(n / DT + 1);
                    if constexpr (PREC) {
                        o[d] = Op<OT>::mfma32(fh[n], pl[2 * kt + st], o[d]);
                        o[d] = Op<OT>::mfma32(fl[n], pf[2 * kt + st], o[d]);
                    }
                } else {
                    const int m = n - NP, kt = m / KSTEPS, ks = m % KSTEPS;
                    if (ks == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
                    }
                    s[kt] = Op<OT>::mfma32(fh[n], qf[ks], s[kt]);
                    if constexpr (PREC) {
                        s[kt] = Op<OT>::mfma32(fh[n], qfl[ks], s[kt]);
                        s[kt] = Op<OT>::mfma32(fl[n], qf[ks], s[kt]);
                    }
                }
            }
            // the issue order above, pinned as scheduling groups (reads per step: PV 2 tr reads per fragment, QK one b128; x2 with
            // split operands); the conversions of the next P chunk are left to the scheduler (VALU groups between the MFMAs made it drop
            // the whole read pipeline); the waits are the compiler's own counted lgkmcnt
            auto reads_of = [&](int n) {       // (the builtin wants literal counts: one call per case, folded once the loops are unrolled)
                if (n < NP) __builtin_amdgcn_sched_group_barrier(0x100, PREC ? 4 : 2, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, PREC ? 2 : 1, 0);
            };
#pragma unroll
            for (int n = 0; n < PIPE && n < NS; ++n) reads_of(n);
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                if (n + PIPE < NS) reads_of(n + PIPE);
                __builtin_amdgcn_sched_group_barrier(0x008, PREC ? 3 : 1, 0);
            }
        };
        auto qk = [&](int t) { mat(t, t, std::false_type{}, std::true_type{}); };
        auto pv = [&](int t) { mat(t, t, std::true_type{}, std::false_type{}); };
        auto pvqk = [&](int t) {                  // PV(t) then QK(t + 1)
            if constexpr (LR_ATT_FUSE) mat(t, t + 1, std::true_type{}, std::true_type{});
            else { pv(t); qk(t + 1); }
        };
    if constexpr (PP) {
        // Ping-pong schedule: waves w and w+4 share a SIMD and run half a tile apart, so that one is in its matrix segment
        // [O += V(t) P(t) ; S(t+1) = K(t+1) Q] while the other is in its vector segment [softmax(t)]; the two workgroup
        // barriers per tile are the hand-over points (the late group takes one extra barrier first, the early one at the end).
        // Single pass: waves 0-3 (the early group) issue the DMA: tile t+2 goes out in their vector segment t into the slot of
        // tile t-2 (4 slots: the late group reads V(t-1) during that very interval), and their counted wait for tile t+1
        // precedes the barrier that opens both groups' S(t+1).
        // Split operands (3 slots is all the LDS holds): waves 4-7 (the late group) issue tile t+2 in THEIR vector segment t,
        // which starts after the barrier that ends the last read of tile t-1, and retire it at the end of their matrix
        // segment t, one barrier before the early group's S(t+2); one tile period covers the latency.
        const int grp = wave >> 2;
#if defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 16)           // diagnostic 16 (tools/dbg/attn_stamps.py): cycles per segment of the ping-pong loop, summed
        unsigned long long sg[5] = {0, 0, 0, 0, 0};        // over the tiles: vector, barrier 1, matrix, barrier 2, tiles; workgroup 0, behind O
        auto stamp = [&]() -> unsigned long long {
            unsigned long long tt;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            return tt;
        };
#define LR_ATT_STAMP(i, a, b) sg[i] += (b) - (a)
#else
#define LR_ATT_STAMP(i, a, b)
        auto stamp = [&]() -> unsigned long long { return 0ull; };
#endif
        // act(t) is monotone (causal: the tiles up to the wave's diagonal; dense: all, or none for a wave wholly past the sequence), so
        // a wave's tiles are `nact` active ones followed by inactive ones in which it only keeps the DMA stream and the barriers going.
        // Two loops instead of `if (act(t))` around every segment: with the conditions inside one loop S and P were both loop-carried
        // (each keeps its old value on the untaken path), 64 registers where 32 are live -- the registers the fragment ring needs.
        int nact = 0;
        if (q0 < S && ntiles > 0) nact = CAUSAL ? max(0, min(ntiles, (q0 + 31 - kbeg) / KT + 1)) : ntiles;
        if (CAUSAL && q0 + 31 < kbeg) nact = 0;
        nact = __builtin_amdgcn_readfirstlane(nact);
        auto vec_dma = [&](int t) {
#if !(defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 8))          // diagnostic 8: no DMA in the loop (tiles 0, 1 only), results invalid
            if (t + 2 < ntiles) issue(t + 2);             // (the issuing group only)
#endif
        };
        auto vec_wait = [&](int t) {
            if (!PREC && grp == 0) {
                if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        };
        auto bar = [&]() {
#if !(defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 4))          // diagnostic 4: no barriers in the loop, results invalid
            __syncthreads();
#endif
        };
        if (ntiles > 0) {
            __syncthreads();                              // tiles 0 and 1 landed (the wait above retired them), sBits written
            if (grp) __syncthreads();                     // the late group starts one interval behind
            if (nact > 0) qk(0);
            else asm volatile("" : "=v"(s[0]), "=v"(s[1]));
            for (int t = 0; t < nact; ++t) {
                const unsigned long long ts0 = stamp();
                // ---- vector segment ----
                vec_dma(t);
                soft(t);
                vec_wait(t);
                const unsigned long long ts1 = stamp();
                bar();
                const unsigned long long ts2 = stamp();
                // ---- matrix segment ----
                if (LR_ATT_PRIO == 1) __builtin_amdgcn_s_setprio(1);            // the wave in its matrix segment goes first (2-4 % on every shape)
                if (LR_ATT_PRIO == 2) __builtin_amdgcn_s_setprio(0);
                if (t + 1 < nact) pvqk(t);
                else {
                    pv(t);
                    asm volatile("" : "=v"(s[0]), "=v"(s[1]));       // (S is dead: no value is carried round the loop on this path)
                }
                if (LR_ATT_PRIO == 1) __builtin_amdgcn_s_setprio(0);
                if (LR_ATT_PRIO == 2) __builtin_amdgcn_s_setprio(1);
                if (PREC && (grp == 1 || SPLIT_DMA)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long ts3 = stamp();
                if (t + 1 < ntiles || grp == 0) bar();
                const unsigned long long ts4 = stamp();
                LR_ATT_STAMP(0, ts0, ts1); LR_ATT_STAMP(1, ts1, ts2); LR_ATT_STAMP(2, ts2, ts3); LR_ATT_STAMP(3, ts3, ts4);
                (void)ts0; (void)ts1; (void)ts2; (void)ts3; (void)ts4;
            }
            for (int t = nact; t < ntiles; ++t) {         // past this wave's diagonal: the DMA stream and the barriers only
                vec_dma(t);
                vec_wait(t);
                bar();
                if (PREC && (grp == 1 || SPLIT_DMA)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (t + 1 < ntiles || grp == 0) bar();
            }
        }
#if defined(LR_ATT_DIAG) && (LR_ATT_DIAG & 16)
        if (blockIdx.x == 0 && lane == 0 && p.lin_nqt > 0) {
            unsigned long long* dbg = (unsigned long long*)((unsigned short*)p.O + (size_t)p.lin_batch * S * p.ldo) + wave * 8;
            sg[4] = (unsigned long long)ntiles;
            for (int i = 0; i < 5; ++i) dbg[i] = sg[i];
        }
#endif
    } else {
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = kbeg + t * KT;
        if (t + PD < ntiles) {
            issue(t + PD);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD * NPT) : "memory");
        } else if (PD > 1 && t + 1 < ntiles) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPT) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                  // tile t landed for every wave (and sBits on the first pass)
        // waves whose 32 queries all precede this tile have nothing to do (diagonal workgroup tiles); nor have waves whose queries
        // all lie beyond the sequence (577 tokens = 4.5 x 128 queries: the last workgroup's fourth wave).
        // (Skipping the 32-key score halves / 16-key PV steps of a last key tile that holds no key -- 577 = 9 x 64 + 1 -- was built
        //  and measured level: the branches cost what the skipped MFMAs save.)
        const bool active = (!CAUSAL || (k0 <= q0 + 31)) && q0 < S;
        if (active) {
            qk(t);            // S^T tiles: keys kt*32 + [(r&3) + 8(r>>2) + 4h], query lc
            soft(t);          // masks where the tile needs them, online softmax in log2 units, P -> operand type (chunk 0)
            pv(t);            // O^T += V^T P^T (chunks 1-3 converted under the MFMAs)
        }
        __syncthreads();                                  // every wave is done with slot t%3 before tile t+3 is issued into it
    }

    }

    // ---- epilogue: O[q][d], d = dt*32 + (r&3) + 8(r>>2) + 4h : 4 consecutive d per register quad ----
    if (qsel >= 0 ? qpos == qsel : (qpos < S && qpos >= 0)) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        unsigned short* dst = (unsigned short*)p.O + (qsel >= 0 ? (size_t)b : rowbase + qpos) * p.ldo + head * HD + 4 * lh;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w, wl;
                split2<OT>(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv, w.x, wl.x);
                split2<OT>(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv, w.y, wl.y);
                *(uint2*)(dst + d * 32 + 8 * g) = w;
                if (p.o_split > 0) *(uint2*)(dst + p.o_split + d * 32 + 8 * g) = wl;
            }
    }
}

template <typename OT, int HD, bool CAUSAL>
static void launch_one(const AttnParams& p0, int batch, hipStream_t st) {
    AttnParams p = p0;
    const int nqt = p.items ? p.n_items : p.qsel ? 1 : (p.S + 127) / 128;
    const int nq8 = p.qsel ? 1 : (p.S + 255) / 256;        // gathered mode: one query tile per (sequence, head)
    // dense launches: 1-D grid in the XCD-aware order (AttnParams::lin_nqt), padded to whole groups of 8 (sequence, kv head) pairs
    const char* xe = getenv("LR_ATT_XCD_ORDER");            // A/B switch, read per launch: 0 = the 3-D grid
    const bool xcd_order = !xe || atoi(xe) != 0;
    const char* qe = getenv("LR_ATT_QSHIFT");                // A/B switch: 0 = query tiles aligned to the start of the sequence
    p.q_end_aligned = (!qe || atoi(qe) != 0) ? 1 : 0;
    const char* pe = getenv("LR_ATT_PINGPONG");              // A/B switch: 0 = the plain per-tile loop for long sequences too
    const bool pingpong = !pe || atoi(pe) != 0;
    const char* pm = getenv("LR_ATT_PP_MIN_S");              // A/B switch: shortest sequence that takes the ping-pong schedule (1024)
    const int pp_min = pm ? atoi(pm) : 1024;
    auto grid = [&](int nq) {
        if (p.items || !xcd_order) { p.lin_nqt = 0; p.lin_batch = 0; return dim3(nq, p.heads, batch); }
        p.lin_nqt = nq; p.lin_batch = batch;
        const int kvpairs = p.heads / p.kv_group * batch;
        return dim3((unsigned)((kvpairs + 7) / 8 * 8 * p.kv_group * nq), 1, 1);
    };
    // Long sequences: 256-query workgroups on the ping-pong schedule (measured at B=32: HD 128 1.47x, where the 4-wave form
    // fits one workgroup per CU; HD 96 1.03-1.07x; split operands 1.01-1.03x; below ~1k keys the 128-query grid fills better).
    if (p.lo_off == 0 && !p.items && p.S >= pp_min && pingpong) {
        const dim3 g = grid(nq8);
        hipLaunchKernelGGL((attn_kernel<OT, HD, CAUSAL, false, 8, true>), g, dim3(512), 0, st, p);
    } else if (p.lo_off > 0 && !p.items && p.S >= pp_min && HD != 128 && pingpong) {
        if constexpr (HD != 128) {
            const dim3 g = grid(nq8);
            hipLaunchKernelGGL((attn_kernel<OT, HD, CAUSAL, true, 8, true>), g, dim3(512), 0, st, p);
        }
    } else if (p.lo_off > 0 && !p.items && HD == 64 && p.S > 128) {
        // head_dim 64 (CLIP, 577 tokens): a 2-slot ring lets two 4-wave workgroups share a CU, and 128-query workgroups waste 10 % of
        // their query slots on 577 tokens where 256-query ones waste 25 %: 4.23 -> 3.53 ms at 544 x 577 x 16 heads, bit-identical
        const dim3 g = grid(nqt);
        hipLaunchKernelGGL((attn_kernel<OT, HD, CAUSAL, true, 4>), g, dim3(256), 0, st, p);
    } else if (p.lo_off > 0 && !p.items && p.S > 128) {      // split-operand mode: 8 waves per workgroup (the ring fills the LDS)
        const dim3 g = grid(nq8);
        hipLaunchKernelGGL((attn_kernel<OT, HD, CAUSAL, true, 8>), g, dim3(512), 0, st, p);
    } else if (p.lo_off > 0) {
        const dim3 g = grid(nqt);
        hipLaunchKernelGGL((attn_kernel<OT, HD, CAUSAL, true, 4>), g, dim3(256), 0, st, p);
    } else {
        const dim3 g = grid(nqt);
        hipLaunchKernelGGL((attn_kernel<OT, HD, CAUSAL, false, 4>), g, dim3(256), 0, st, p);
    }
}

void launch_attention(const AttnParams& p, int batch, int head_dim, bool causal, int operand_dtype, hipStream_t st) {
    if (batch <= 0) return;
    if (p.ldq % 8 || p.qoff % 8 || p.koff % 8 || p.voff % 8 || p.ldo % 4)
        throw std::runtime_error("attention: operand rows must be 16-byte aligned");
    if (p.lo_off < 0 || p.lo_off % 8 || p.o_split < 0 || p.o_split % 4) throw std::runtime_error("attention: bad split-operand offsets");
    if (p.items && (causal || p.mask || batch != 1 || p.n_items < 1)) throw std::runtime_error("attention: ragged mode is dense, unmasked, batch 1");
    if (p.qsel && (!causal || p.items)) throw std::runtime_error("attention: gathered mode is causal and batched");
    if (p.S > ATT_MAX_S) throw std::runtime_error("attention: sequence length above 8192 is not supported");
    if (p.kv_group < 1 || p.heads % p.kv_group) throw std::runtime_error("attention: heads must be a multiple of kv_group");
    // (2^lazy_t must fit the f16 hi half of a softmax weight, and the fp32 sums must not lose the small weights: 15 is the ceiling)
    if (!(p.lazy_t >= 0.f && p.lazy_t <= 15.f)) throw std::runtime_error("attention: lazy_t (reference-maximum threshold, log2 units) must be in [0, 15]");
    const bool f16 = operand_dtype == DT_F16;
    if (head_dim == 96 && causal) { f16 ? launch_one<F16, 96, true>(p, batch, st) : launch_one<BF16, 96, true>(p, batch, st); }
    else if (head_dim == 64 && !causal) { f16 ? launch_one<F16, 64, false>(p, batch, st) : launch_one<BF16, 64, false>(p, batch, st); }
    else if (head_dim == 64 && causal) { f16 ? launch_one<F16, 64, true>(p, batch, st) : launch_one<BF16, 64, true>(p, batch, st); }
    else if (head_dim == 96 && !causal) { f16 ? launch_one<F16, 96, false>(p, batch, st) : launch_one<BF16, 96, false>(p, batch, st); }
    else if (head_dim == 128 && causal) { f16 ? launch_one<F16, 128, true>(p, batch, st) : launch_one<BF16, 128, true>(p, batch, st); }
    else throw std::runtime_error("attention: head_dim must be 64, 96 or 128 (128: causal only)");
}

}  // namespace lr
