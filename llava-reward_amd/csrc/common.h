// Shared device/host helpers for the gfx950 reward-scoring kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace lr {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// Operand element types of the MFMA contractions (2 bytes each).  bf16 is the reference's
// checkpoint dtype; f16 has 3 more mantissa bits at the same MFMA rate.
struct BF16 {};
struct F16 {};

template <typename OT> struct Op;
template <> struct Op<BF16> {
    typedef bf16x8 vec8;
    static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    template <typename V> static __device__ __forceinline__ f32x4 mfma16(V a, V b, f32x4 c) {      // V: 16 bytes (uint4, or a native 4 x u32 vector)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned short from_f32(float x) {
        return __builtin_bit_cast(unsigned short, (__bf16)x);
    }
    static __device__ __forceinline__ float to_f32(unsigned short u) {
        return __builtin_bit_cast(float, ((unsigned)u) << 16);
    }
};
template <> struct Op<F16> {
    typedef f16x8 vec8;
    static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    template <typename V> static __device__ __forceinline__ f32x4 mfma16(V a, V b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned short from_f32(float x) {
        // saturate instead of overflowing to inf (f16 max 65504); NaN stays NaN
        x = __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f);
        return __builtin_bit_cast(unsigned short, (_Float16)x);
    }
    static __device__ __forceinline__ float to_f32(unsigned short u) {
        return (float)__builtin_bit_cast(_Float16, u);
    }
};

// One-byte residuals of split operands (default parity mode): the residual half of an operand row is stored as e4m3
// with one power-of-two scale (E8M0 byte) per (row, 128-column block) -- the block-scaled form the CDNA4 scaled matrix instruction
// takes natively.  A block is row-local and at most one producer tile wide, so every producer (norms, GEMM epilogues) can encode
// in its own epilogue, and a row's bytes never depend on the other rows of the batch.
// Scale array of an operand with M rows and K columns, laid out for the consuming GEMM: per group of 4 K-tiles (128 columns each) and
// 256-row tile one KB = 4 slices of 256 bytes (what one LDS-DMA instruction moves); inside a slice the byte of row r sits at
// lo8_sidx(r): 8 bytes per lane of the consumer = its 4 row tiles of A half 0, then of A half 1.
__host__ __device__ inline size_t lo8_scale_bytes(int M, int K) { return (size_t)((M + 255) / 256) * 1024 * (size_t)((K / 128 + 3) / 4); }
__host__ __device__ inline int lo8_sidx(int r) { return ((((r >> 6) & 1) * 16 + (r & 15)) * 2 + (r >> 7)) * 4 + ((r >> 4) & 3); }     // r = row % 256
__host__ __device__ inline size_t lo8_scale_at(int row, int kblock, int M) {
    return ((size_t)(kblock >> 2) * ((M + 255) / 256) + (row >> 8)) * 1024 + (kblock & 3) * 256 + lo8_sidx(row & 255);
}
#if defined(__HIPCC__)
// maximum over the 16 lanes of a DPP row (lanes 16g .. 16g + 15); m >= 0.  Call in uniform control flow.
__device__ __forceinline__ float row16_max(float m) {
    int v = __builtin_bit_cast(int, m);                 // non-negative floats order like their bit patterns
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));    // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));    // row_mirror
    return __builtin_bit_cast(float, v);
}
// DIAGNOSTIC ONLY (-DLR_EMU_FP6=1, tools/fp6/; never in the product build): what the default form would lose if its residual pass ran
// in OCP MX FP6 -- e2m3 elements {0, 0.125 .. 7.5}, one power-of-two scale per 32 elements of a row, for A_lo AND for the weight twin
// (v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz / blgp = 2 issues at twice the e4m3 rate, MI355X_MICROARCH.md).  The residuals are
// snapped to that grid and THEN encoded in the product's e4m3 / 128-block form, which holds every snapped value exactly (3 mantissa
// bits either way; a 32-block whose maximum is below 2^-9 of its 128-block's would lose bits, by less than 2^-10 of that maximum), so
// the unchanged e4m3 kernels compute the products an fp6 pass would -- its numerics without its kernel.
#ifndef LR_EMU_FP6
#define LR_EMU_FP6 0
#endif
__device__ __forceinline__ float emu_e2m3(float v, float bmax) {     // bmax = max |.| over the 32-element block that holds v
    if (!(bmax > 0.f)) return v;
    const int e = ilogbf(bmax) - 2;                // MX: scale = 2^(floor(log2 amax) - emax), emax(e2m3) = 2: amax / scale in [4, 8)
    const float aq = fminf(fabsf(ldexpf(v, -e)), 7.5f);
    const int ex = aq >= 1.f ? ilogbf(aq) : 0;     // sub-normals share the grid of [1, 2): 0.125
    const float ulp = ldexpf(1.f, ex - 3);
    return copysignf(ldexpf(fminf(rintf(aq / ulp) * ulp, 7.5f), e), v);
}
// maximum over the 4 lanes of a quad (8 columns per lane: one 32-column block); m >= 0.  Call in uniform control flow.
__device__ __forceinline__ float quad_max(float m) {
    int v = __builtin_bit_cast(int, m);
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
    return __builtin_bit_cast(float, v);
}
// 2^(E-127) puts amax into [128, 256) <= 448 (e4m3's largest finite value); E = 127 for an all-zero block
__device__ __forceinline__ int e8m0_of_amax(float amax) { return amax > 0.f ? min(max(127 + (ilogbf(amax) - 7), 1), 253) : 127; }
__device__ __forceinline__ float e8m0_inv_scale(int E) { return __builtin_bit_cast(float, (unsigned)(254 - E) << 23); }   // 2^(127 - E)
#endif

template <typename OT> __device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return (unsigned)Op<OT>::from_f32(lo) | ((unsigned)Op<OT>::from_f32(hi) << 16);
}

// Split-operand ("precise") mode: an fp32 value leaves as hi = round(x) plus lo = round(x - hi), both in the operand type.
// With f16 the pair carries 22 mantissa bits; weights are bf16-valued and therefore exact in f16, so a GEMM over
// [A_hi | A_lo] x [W | W] reproduces the fp32 product to ~2^-22 at twice the MFMA work.
template <typename OT> __device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const unsigned short ha = Op<OT>::from_f32(a), hb = Op<OT>::from_f32(b);
    hi = (unsigned)ha | ((unsigned)hb << 16);
    lo = pack2<OT>(a - Op<OT>::to_f32(ha), b - Op<OT>::to_f32(hb));
}
// The same results with packed conversions (one v_cvt_pk per pair instead of two conversions and a pack): for the GEMM epilogues,
// whose read + store loops are VALU-bound (~10 operations per element before this).  round_pair = Op::from_f32 of both, packed.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <typename OT> __device__ __forceinline__ unsigned round_pair(float a, float b);
template <> __device__ __forceinline__ unsigned round_pair<F16>(float a, float b) {
    const f32x2_t x = {__builtin_fminf(__builtin_fmaxf(a, -65504.f), 65504.f), __builtin_fminf(__builtin_fmaxf(b, -65504.f), 65504.f)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2_t));
}
template <> __device__ __forceinline__ unsigned round_pair<BF16>(float a, float b) {      // (the scalar conversions: the packed bf16 form
    return pack2<BF16>(a, b);                                                               //  does not round every value the same way)
}
template <typename OT> __device__ __forceinline__ void split2p(float a, float b, unsigned& hi, unsigned& lo) {
    hi = round_pair<OT>(a, b);
    lo = round_pair<OT>(a - Op<OT>::to_f32((unsigned short)(hi & 0xFFFFu)), b - Op<OT>::to_f32((unsigned short)(hi >> 16)));
}
#if defined(__HIPCC__)
// a - float(h) for the low / high f16 half h of `packed`, as ONE v_fma_mix_f32 (fma(float(h), -1, a): the widening rides in the
// instruction and the product is exact, so this IS the subtraction's correctly rounded value, bit for bit) instead of a conversion
// and a subtraction.  hipcc does not form it from the plain expression.  The attention kernels' P split uses it (round 5).
__device__ __forceinline__ float sub_f16_lo_half(float a, unsigned packed) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(a));
    return r;
}
__device__ __forceinline__ float sub_f16_hi_half(float a, unsigned packed) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(a));
    return r;
}
// (Tried in the GEMM epilogues too -- split2p / store_hi_lo8 -- and measured level on every shape, +-0.3 %: their read + store loops
//  are not bound by the vector-instruction count.  Only the attention kernels use it.)
#endif

// x * sigmoid(a*x) with the hardware exp2/rcp (1 ulp each; the result is rounded to a 16-bit operand anyway)
__device__ __forceinline__ float x_sigmoid_fast(float x, float a) {
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * a * x);
    return x * __builtin_amdgcn_rcpf(1.f + e);
}

// One RoPE pair, (x, y) = (q[i], q[i + hd/2]) -> (x cos - y sin, y cos + x sin), in the REFERENCE's arithmetic
// (modeling_phi3_v.py:521-553, q * cos + rotate_half(q) * sin in fp32): two rounded products and one rounded sum per output, no fused
// multiply-add.  Contraction is switched off for these four lines on purpose (round 6): left as a * b - c * d, which of the two products the
// compiler fuses depends on the code around the expression -- it differed between instantiations of one GEMM kernel -- and the
// rewards move with it (1 ulp of the operand type on a few elements per launch).
// (the pragma, not __fmul_rn / __fadd_rn: HIP defines those as the plain operators, which contract like any others)
__device__ __forceinline__ void rope_pair(float x, float y, float c, float sn, float& ox, float& oy) {
#pragma clang fp contract(off)
    ox = x * c - y * sn;
    oy = y * c + x * sn;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short u) {
    return __builtin_bit_cast(float, ((unsigned)u) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- epilogue selectors of gemm_bt ----
enum : int {
    EPI_OUT_OP = 0,      // C_op[m][n]   = act(acc + bias)
    EPI_OUT_F32 = 1,     // C_f32[m][n]  = acc + bias
    EPI_RESADD_F32 = 2,  // C_f32[m][n] += acc + bias
    EPI_SWIGLU_OP = 3,   // C_op[m][n/2] = silu(gate) * up, weight rows interleaved in blocks of 32
    EPI_ROPE_OP = 4,     // C_op[m][n]   = RoPE(acc) for n < rope_cols (pair-interleaved head dims), acc elsewhere
};
enum : int { ACT_NONE = 0, ACT_QUICK_GELU = 1, ACT_GELU_ERF = 2 };
enum : int { DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2 };

struct GemmParams {
    const void* A;      // [M, lda] operand dtype, row-major
    const void* W;      // [N, ldw] operand dtype, row-major (C = A * W^T)
    void* C;            // see epilogue
    const float* bias;  // [N] or null
    int M, N, K;        // K multiple of 64, N multiple of the block's BN
    int lda, ldw, ldc;  // in elements
    int epi, act;
    // EPI_ROPE_OP: cos/sin table [M][rope_hd/2][2] and the number of leading columns (q and k sections) to rotate
    const float* rope_cs;
    int rope_cols, rope_hd;
    // split-operand mode.  kw > 0: W has only kw columns of K and is re-read from column 0 when k reaches kw
    // (A = [A_hi | A_lo], K = 2 kw).  split > 0: epilogues that emit operands also store the rounding residual of
    // every element `split` columns to the right of it (C = [C_hi | C_lo], ldc covers both halves).
    int kw, split;
    // split-operand mode with weights that are NOT exact in the operand type (e.g. LoRA-merged fp32 weights): Wlo holds the
    // rounding residuals of W (same layout) and K = 3 kw runs A = [hi | lo | hi] against [W | W | Wlo]
    const void* Wlo;
    // W8A8 mode (launch_gemm_bt8_fp8): dequantisation scales, one per row of A and one per row of W
    const float* ascale;
    const float* wscale;
    // split-operand mode with an e4m3 residual pass (launch_gemm_bt8_mixed): block scales of A's residual half (lo8_scale_bytes(M, kw)
    // bytes, see lo8_scale_at) and the tensor scale of W8
    const unsigned char* aexp;
    int wexp;
    // EPI_OUT_OP / EPI_SWIGLU_OP in split-operand mode: write the residual half of C in that one-byte form too (the GEMM that reads C
    // takes the e4m3 residual pass): oexp = C's scale array (lo8_scale_at over M rows), null = 16-bit residuals.  Output columns % 128 == 0.
    unsigned char* oexp;
    // ... with weights that are not exact in the operand type: a third segment, A_hi as e4m3 (row exponents aexp2, bytes behind the
    // residual bytes) against e4m3(W_lo) (tensor exponent wexp2, bytes behind W8 in the same rows); K = 2 kw
    const int* aexp2;
    int wexp2;
    // K-extension (an un-merged LoRA adapter): y = x W^T + t B^T with t = s x A^T computed by a GEMM of its own.
    // A2 = t rows [M, lda2] = [t_hi (k2) | t_lo (k2)] in split-operand mode, [t (k2)] otherwise; W2 = B [N, ldw2] (k2 columns used),
    // W2lo = its rounding residuals when B is not exact in the operand type (else null).  k2 % 64 == 0.  All 16-bit segments.
    const void* A2;
    const void* W2;
    const void* W2lo;
    int lda2, ldw2, k2;
    // e4m3-residual form only: 16-bit residual rows of W for an f16 segment x_hi x W_lo^T (weights inexact in the operand type whose
    // e4m3 twin lives in a buffer of its own: the adapter's A matrix), instead of the e4m3 third segment
    const void* Wlo16;
    // ---- filled by the launchers of gemm8.hip (build_segments), not by callers: the K loop as a list of segments ----
    // A K-tile (64 two-byte units of a row) of segment s comes from A (or A2) at column a_col + 64 i and from W / Wlo / Wlo16 / W2 /
    // W2lo at column w_col + 64 i.  Segments are ordered 16-bit first (K-tiles [0, nk_f16)), then the e4m3 residual tiles
    // [nk_f16, nk_e1) scaled by aexp / wexp, then the e4m3 tiles [nk_e1, nk) scaled by aexp2 / wexp2.
    struct KSeg { int kt_end, src, a_col, w_col; };
    enum : int { SRC_A2 = 1, SRC_W2 = 2, SRC_LO = 4, SRC_LO16 = 8, MAX_SEG = 12 };
    int nseg, nk_f16, nk_e1, nk;
    KSeg seg[MAX_SEG];
    // persistent launches: tile scheduler words (launch8), 8 per-XCD claim counters + 1 count of finished workgroups, all zero
    // between launches; null = every workgroup walks a fixed list of tiles
    int* sched;
    // ---- caller-provided storage for `sched`: 16 ints of DEVICE memory on the launch device, zero, used by one launch at a time
    // (an engine passes its own; null = the launcher keeps one set per (device, stream)) ----
    int* sched_mem;
    // rows of tiles per band of the tile walk (launch8 sets it: 4 unless LR_GEMM_GM says otherwise): an XCD's 32 concurrent tiles form a gm x (32 / gm) patch
    int gm;
    int band_chunks;      // dynamic walk: the XCDs' chunks are cut at band boundaries (launch8)
};

struct AttnParams {
    const void* Q;  // operand dtype rows [b*S + t][ldq], head h at column qoff + h*HD
    const void* K;
    const void* V;
    void* O;        // operand dtype [b*S + t][ldo], head h at column h*HD
    const int64_t* mask;   // [B, S] or null (non-causal towers)
    const int* kmin;       // first valid key of sequence b at kmin[b*kmin_stride] (or null)
    int kmin_stride;
    int ldq, ldo;
    int qoff, koff, voff;
    int S;          // tokens per sequence
    int heads;
    float scale;    // 1/sqrt(HD)
    int kv_group;   // query heads per key/value head (GQA); 1 = MHA
    // ragged (block-diagonal) mode, dense only: workgroup x handles query tile items[x].z of the segment that starts at
    // row items[x].x and is items[x].y rows long; S, mask and kmin are ignored
    const int4* items;
    int n_items;
    // split-operand mode: lo_off > 0 = Q/K/V rows carry their rounding residuals lo_off columns to the right and the
    // kernel evaluates hi.hi + hi.lo + lo.hi for both contractions; o_split > 0 = O is stored as [O_hi | O_lo]
    int lo_off, o_split;
    // gathered mode (causal, batched): ONE query per sequence -- position qsel_last ? S - 1 : qsel[b * qsel_stride] -- is wanted
    // (the row the reward is read from, rw_model:420-421, in the last decoder layer).  One workgroup per (sequence, head) runs the
    // query tile that holds it, with the arithmetic of the full launch (bit-identical), and stores that query's row to O[b].
    const int* qsel;
    int qsel_stride, qsel_last;
    // ---- filled by launch_attention, not by callers ----
    // XCD-aware workgroup order of the dense launches (1-D grid): workgroup ids go round-robin over the 8 XCDs, so id & 7 names
    // the XCD; all query tiles of a (sequence, key/value head) -- its kv_group query heads included -- run on ONE XCD, back to
    // back, heaviest first, and that XCD's L2 holds their K / V rows.  nqt = query tiles per (sequence, head), batch = sequences;
    // 0 = the 3-D grid (ragged mode).
    int lin_nqt, lin_batch;
    int q_end_aligned;      // causal: query tiles shifted towards the end of the sequence by whole key tiles (see attn_kernel)
    // ---- set by callers (0 = the exact running maximum) ----
    // split-operand forms: the softmax reference maximum of a row moves only when the new maximum exceeds it by more than lazy_t
    // (log2 units, 0 .. 15).  The engine passes ATT_LAZY_T_DEFAULT in stages that run the e4m3-residual default form and 0 (exact,
    // the reference's own arithmetic) in stages that run the strict form.
    float lazy_t;
};
constexpr float ATT_LAZY_T_DEFAULT = 8.f;

}  // namespace lr

#define LR_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e));     \
        }                                                                                    \
    } while (0)
