// Do the vector instructions of one wave execute under the matrix instructions of the OTHER wave on its SIMD?  (round 5: the ping-pong
// attention kernel assumes they do; in-kernel stamps say a wave's softmax segment takes 2.7x longer beside a partner that issues
// back-to-back MFMAs than alone.)  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a loop of v_mfma_f32_32x32x16_f16,
// waves 4-7 (their SIMD partners) a loop of vector instructions; each group is timed with s_memtime, alone and beside the other.
//   MF: 1 = one accumulation chain, 3 = three accumulators in rotation
//   VK: 0 = dependent v_fma chain, 1 = 8 independent v_fma chains, 2 = v_exp chains, 3 = v_cvt_pk + v_fma mix
//   hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MF, int VK, int PRIO>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int n_m, int n_v, int run_m, int run_v) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float acc = 0.f;
    __syncthreads();
    if (wave < 4) {
        if (run_m) {
            v8h a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * ((lane + i) & 15)); b[i] = (_Float16)(0.02f * ((lane * 3 + i) & 7)); }
            v16f c[3];
            for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
            if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
            for (int i = 0; i < n_m; ++i) {
#pragma unroll
                for (int j = 0; j < 6; ++j) c[MF == 1 ? 0 : j % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[MF == 1 ? 0 : j % 3], 0, 0, 0);
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
            for (int j = 0; j < 3; ++j) for (int r = 0; r < 16; ++r) acc += c[j][r];
        }
    } else if (run_v) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = 0.001f * (lane + j);
        const float m = 0.999f, d = 1e-3f;
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        for (int i = 0; i < n_v; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (VK == 0) x[0] = __builtin_fmaf(x[0], m, d);
                else if (VK == 1) x[j] = __builtin_fmaf(x[j], m, d);
                else if (VK == 2) x[j] = __builtin_amdgcn_exp2f(x[j] * 0.5f);
                else { const auto h = __builtin_convertvector((__attribute__((ext_vector_type(2))) float){x[j], x[(j + 1) & 7]}, __attribute__((ext_vector_type(2))) _Float16); x[j] = __builtin_fmaf((float)h[0], m, (float)h[1]); }
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        for (int j = 0; j < 8; ++j) acc += x[j];
    }
    if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MF, int VK, int PRIO> static void run(const char* name) {
    unsigned long long *out, h[8]; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4);
    const int n_m = 2000, n_v = 4000;          // 12000 MFMAs; 32000 vector instructions (VK 3: ~3x)
    double res[3][2];
    for (int mode = 0; mode < 3; ++mode) {     // 0 = MFMA alone, 1 = vector alone, 2 = both
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(out, 0, 64);
            k<MF, VK, PRIO><<<256, 512>>>(out, sink, n_m, n_v, mode != 1, mode != 0);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        res[mode][0] = (double)h[0] / (n_m * 6.0); res[mode][1] = (double)h[4] / (n_v * 8.0);
    }
    printf("%-64s MFMA %6.2f ticks alone, %6.2f beside | vector op %5.2f ticks alone, %5.2f beside (x%.2f) | sum-of-alone/both-wall %.2f\n", name, res[0][0], res[2][0],
           res[1][1], res[2][1], res[2][1] / res[1][1],
           (res[0][0] * n_m * 6 + res[1][1] * n_v * 8) / fmax(res[2][0] * n_m * 6, res[2][1] * n_v * 8));
}

int main() {
    run<1, 0, 0>("one MFMA chain | dependent v_fma chain");
    run<1, 1, 0>("one MFMA chain | 8 independent v_fma chains");
    run<3, 1, 0>("3 accumulators  | 8 independent v_fma chains");
    run<1, 2, 0>("one MFMA chain | v_exp chains");
    run<3, 2, 0>("3 accumulators  | v_exp chains");
    run<1, 3, 0>("one MFMA chain | cvt_pk + fma mix");
    run<1, 1, 1>("one MFMA chain | 8 fma chains, MFMA wave prio 1");
    run<1, 1, 2>("one MFMA chain | 8 fma chains, vector wave prio 1");
    run<3, 1, 2>("3 accumulators  | 8 fma chains, vector wave prio 1");
    return 0;
}
