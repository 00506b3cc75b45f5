// Fused softmax(Q K^T * scale + mask) V for the two towers of the scoring path:
//   * Phi-3 decoder self-attention, causal + padding mask, head_dim 96 (modeling_phi3_v.py:641-720;
//     additive mask semantics of :1453-1459: key j visible to query i iff j <= i and mask[j] == 1);
//   * CLIP encoder self-attention, dense, 577 tokens, head_dim 64 (transformers CLIPAttention,
//     scale head_dim^-0.5; call site modeling_phi3_v.py:212 / utils/utils.py:266-273).
// The S x S score matrix the reference's eager path materialises (0.9 GB fp32 per sample) never
// leaves registers: online softmax in fp32, operands in the 2-byte MFMA type.
//
// Work split: one workgroup = 4 waves = 128 query rows of one (sequence, head); each wave owns 32
// query rows and walks the key/value tiles (64 keys) staged once per workgroup in LDS.
//   S^T = K Q^T   : A = K rows from LDS (ds_read_b128, rows padded by 16 B: conflict-free),
//                   B = Q^T from registers (each lane loaded its own query row from HBM once).
//                   The 32x32 result has the QUERY on the lane and 16 keys in registers, so
//                   row max / row sum are 16 register ops + one exchange with lane^32.
//   O^T = V^T P^T : the score accumulators, converted pairwise to the operand type, ARE the B
//                   operand (cdna_hip_programming.md §3 "accumulator tile as the next MFMA's
//                   operand": k index of element j of lane half h = 16s + 8(j>>2) + 4h + (j&3));
//                   A = V^T from an LDS image transposed while staging (2 x ds_read_b64 per step).
//                   O^T again has the query on the lane, so the online-softmax rescale is lane-local.
// Global loads of the next K/V tile are issued before the current tile's MFMAs and written to LDS
// after them (issue-early / write-late, T14).
#include "common.h"
#include "kernels.h"

namespace lr {

template <typename OT, int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_kernel(AttnParams p) {
    constexpr int KT = 64;                 // keys per tile
    constexpr int KSTEPS = HD / 16;        // MFMA k-steps over the head dim
    constexpr int DT = HD / 32;            // 32-wide output tiles over the head dim
    constexpr int KROW = HD * 2 + 16;      // K image row stride (bytes)
    constexpr int VROW = KT * 2 + 8;       // V^T image row stride (bytes)
    constexpr int CH = HD / 8;             // 16-byte chunks per row
    constexpr int LD_PER_T = KT * CH / 256;  // 16-byte loads per thread per operand tile (2 or 3)
    static_assert(KT * CH % 256 == 0, "tile/threads mismatch");

    __shared__ __attribute__((aligned(16))) char smem[KT * KROW + HD * VROW + KT * 4];
    char* sK = smem;
    char* sV = smem + KT * KROW;
    float* sM = (float*)(smem + KT * KROW + HD * VROW);   // additive key mask (0 or -inf)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lc = lane & 31, lh = lane >> 5;
    const int nqt = (p.S + 127) / 128;
    const int qt = nqt - 1 - (int)blockIdx.x;      // heavy (late) causal tiles first
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = qt * 128 + wave * 32;
    const size_t rowbase = (size_t)b * p.S;

    const unsigned short* Qp = (const unsigned short*)p.Q + p.qoff + head * HD;
    const unsigned short* Kp = (const unsigned short*)p.K + p.koff + head * HD;
    const unsigned short* Vp = (const unsigned short*)p.V + p.voff + head * HD;

    // ---- Q fragments: lane (c,h) holds Q[q0+c][16*ks + 8h .. +7] ----
    uint4 qf[KSTEPS];
    {
        const int qrow = min(q0 + lc, p.S - 1);
        const unsigned short* src = Qp + (rowbase + qrow) * p.ldq + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *(const uint4*)(src + 16 * ks);
    }

    // ---- key range of this workgroup ----
    int kbeg = 0, kend = p.S;
    if (CAUSAL) {
        kend = min(p.S, qt * 128 + 128);
        if (p.kmin) kbeg = (min(p.kmin[b * p.kmin_stride], p.S) / KT) * KT;
    }

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float sc = p.scale * 1.4426950408889634f;   // scores kept in log2 units
    const int qpos = q0 + lc;

    // staging assignment: thread -> (key = tid & 63, chunk = (tid >> 6) + 4*i)
    const int skey = tid & 63;
    uint4 kreg[LD_PER_T], vreg[LD_PER_T];
    float mreg = 0.f;
    auto gload = [&](int k0) {
        const int key = min(k0 + skey, p.S - 1);
        const unsigned short* ks_ = Kp + (rowbase + key) * p.ldq;
        const unsigned short* vs_ = Vp + (rowbase + key) * p.ldq;
#pragma unroll
        for (int i = 0; i < LD_PER_T; ++i) {
            const int ch = (tid >> 6) + 4 * i;
            kreg[i] = *(const uint4*)(ks_ + ch * 8);
            vreg[i] = *(const uint4*)(vs_ + ch * 8);
        }
        if (tid < KT) {
            const int kk = k0 + tid;
            bool ok = kk < p.S;
            if (ok && p.mask) ok = p.mask[(size_t)b * p.S + kk] != 0;
            mreg = ok ? 0.f : -INFINITY;
        }
    };
    auto lwrite = [&]() {
#pragma unroll
        for (int i = 0; i < LD_PER_T; ++i) {
            const int ch = (tid >> 6) + 4 * i;
            *(uint4*)(sK + skey * KROW + ch * 16) = kreg[i];
            const unsigned short* e = (const unsigned short*)&vreg[i];
#pragma unroll
            for (int j = 0; j < 8; ++j) *(unsigned short*)(sV + (ch * 8 + j) * VROW + skey * 2) = e[j];
        }
        if (tid < KT) sM[tid] = mreg;
    };

    if (kbeg < kend) gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += KT) {
        lwrite();
        __syncthreads();
        if (k0 + KT < kend) gload(k0 + KT);

        // ---- S^T tiles: keys kt*32 + [(r&3) + 8(r>>2) + 4h], query lc ----
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const uint4 kf = *(const uint4*)(sK + (kt * 32 + lc) * KROW + (2 * ks + lh) * 16);
                s[kt] = Op<OT>::mfma32(kf, qf[ks], s[kt]);
            }
        }
        // ---- mask + online softmax (lane-local: this lane's query is lc) ----
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float t = s[kt][r] * sc + sM[kl];
                if (CAUSAL && (k0 + kl > qpos)) t = -INFINITY;
                s[kt][r] = t;
                mx = fmaxf(mx, t);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        float rs = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = exp2f(s[kt][r] - m_new);
                s[kt][r] = e;
                rs += e;
            }
        rs += __shfl_xor(rs, 32, 64);
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uint4 pf;
                pf.x = pack2<OT>(s[kt][8 * st + 0], s[kt][8 * st + 1]);
                pf.y = pack2<OT>(s[kt][8 * st + 2], s[kt][8 * st + 3]);
                pf.z = pack2<OT>(s[kt][8 * st + 4], s[kt][8 * st + 5]);
                pf.w = pack2<OT>(s[kt][8 * st + 6], s[kt][8 * st + 7]);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const char* vrow = sV + (d * 32 + lc) * VROW + (kt * 32 + 16 * st + 4 * lh) * 2;
                    const uint2 v0 = *(const uint2*)(vrow);
                    const uint2 v1 = *(const uint2*)(vrow + 16);
                    uint4 vf;
                    vf.x = v0.x; vf.y = v0.y; vf.z = v1.x; vf.w = v1.y;
                    o[d] = Op<OT>::mfma32(vf, pf, o[d]);
                }
            }
        __syncthreads();
    }

    // ---- epilogue: O[q][d], d = dt*32 + (r&3) + 8(r>>2) + 4h : 4 consecutive d per register quad ----
    if (qpos < p.S) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        unsigned short* dst = (unsigned short*)p.O + (rowbase + qpos) * p.ldo + head * HD + 4 * lh;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 w;
                w.x = pack2<OT>(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
                w.y = pack2<OT>(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
                *(uint2*)(dst + d * 32 + 8 * g) = w;
            }
    }
}

template <typename OT, int HD, bool CAUSAL>
static void launch_one(const AttnParams& p, int batch, hipStream_t st) {
    const int nqt = (p.S + 127) / 128;
    hipLaunchKernelGGL((attn_kernel<OT, HD, CAUSAL>), dim3(nqt, p.heads, batch), dim3(256), 0, st, p);
}

void launch_attention(const AttnParams& p, int batch, int head_dim, bool causal, int operand_dtype, hipStream_t st) {
    if (batch <= 0) return;
    if (p.ldq % 8 || p.qoff % 8 || p.koff % 8 || p.voff % 8 || p.ldo % 4)
        throw std::runtime_error("attention: operand rows must be 16-byte aligned");
    const bool f16 = operand_dtype == DT_F16;
    if (head_dim == 96 && causal) { f16 ? launch_one<F16, 96, true>(p, batch, st) : launch_one<BF16, 96, true>(p, batch, st); }
    else if (head_dim == 64 && !causal) { f16 ? launch_one<F16, 64, false>(p, batch, st) : launch_one<BF16, 64, false>(p, batch, st); }
    else if (head_dim == 64 && causal) { f16 ? launch_one<F16, 64, true>(p, batch, st) : launch_one<BF16, 64, true>(p, batch, st); }
    else if (head_dim == 96 && !causal) { f16 ? launch_one<F16, 96, false>(p, batch, st) : launch_one<BF16, 96, false>(p, batch, st); }
    else throw std::runtime_error("attention: head_dim must be 64 or 96");
}

}  // namespace lr
