// Full-chip issue rate of the f16 and e4m3 matrix instructions (8 independent accumulators per wave, 2 waves per SIMD):
// what the power-limited chip sustains, not the nominal peak.  hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void rate(float* out, int iters, unsigned seed) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)(seed * (threadIdx.x + 1) * 2654435761u + i * 40503u) & 0x3F3F3F3F; b[i] = a[i] ^ 0x11111111; }
    v8h ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)((a[i] & 255) * 0.01f); bh[i] = (_Float16)((b[i] & 255) * 0.01f); }
    v4f c[8];
    for (int j = 0; j < 8; ++j) c[j] = v4f{0, 0, 0, 0};
    for (int i = 0; i < (KIND >= 6 ? 0 : iters); ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c[j], 0, 0, 0);
            else if (KIND == 1) c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 0, 0, 0, 127, 0, 127);
            else if (KIND == 2) c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 2, 2, 0, 127, 0, 127);      // fp6 e2m3 x fp6 e2m3
            else if (KIND == 3) c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 4, 4, 0, 127, 0, 127);      // fp4 e2m1 x fp4 e2m1
            else if (KIND == 4) c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 2, 0, 0, 127, 0, 127);      // fp6 x e4m3
            else if (KIND == 5) c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 2, 4, 0, 127, 0, 127);      // fp6 x fp4
        }
    }
    float s = 0;
    if (KIND == 6 || KIND == 7) {          // 32x32 shapes: 4 accumulators of 16 registers, the same flops per loop iteration as 8 x 16x16
        v16f d[4];
        for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) d[j][k] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (KIND == 6) d[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, d[j], 0, 0, 0);
                else d[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, d[j], 0, 0, 0, 127, 0, 127);
            }
        }
        for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) s += d[j][k];
    }
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND> static void run(const char* name, double flop_per_mfma) {
    float* out; hipMalloc(&out, 4);
    const int iters = 20000, blocks = 256;
    rate<KIND><<<blocks, 512>>>(out, 100, 1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    rate<KIND><<<blocks, 512>>>(out, iters, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 8 * iters * (KIND >= 6 ? 4 : 8);          // waves x mfma
    printf("%-28s %8.3f ms  %7.0f TFLOP/s  (%.1f ns per MFMA per wave slot)\n", name, ms, n * flop_per_mfma / ms / 1e9, ms * 1e6 / (iters * 8.0 * 2));
}

int main() {
    run<0>("f16 16x16x32", 2.0 * 16 * 16 * 32);
    run<1>("e4m3 16x16x128 (scale 2^0)", 2.0 * 16 * 16 * 128);
    run<6>("f16 32x32x16", 2.0 * 32 * 32 * 16);
    run<7>("e4m3 32x32x64", 2.0 * 32 * 32 * 64);
    run<0>("f16 16x16x32", 2.0 * 16 * 16 * 32);
    run<6>("f16 32x32x16", 2.0 * 32 * 32 * 16);
    run<1>("e4m3 16x16x128 (scale 2^0)", 2.0 * 16 * 16 * 128);
    run<7>("e4m3 32x32x64", 2.0 * 32 * 32 * 64);
    run<2>("fp6 e2m3 16x16x128", 2.0 * 16 * 16 * 128);
    run<3>("fp4 e2m1 16x16x128", 2.0 * 16 * 16 * 128);
    run<4>("fp6 x e4m3 16x16x128", 2.0 * 16 * 16 * 128);
    run<5>("fp6 x fp4 16x16x128", 2.0 * 16 * 16 * 128);
    run<2>("fp6 e2m3 16x16x128", 2.0 * 16 * 16 * 128);
    run<1>("e4m3 16x16x128 (scale 2^0)", 2.0 * 16 * 16 * 128);
    return 0;
}
