// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands): operand lane map, scale semantics, issue rate.
// Groundwork for an MX-fp8 "lo" pass of the split-operand GEMM; not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__global__ void probe(const unsigned char* A, const unsigned char* B, float* C, int sa, int sb) {
    // A, B: [64 lanes][32 bytes] already in per-lane order
    const int lane = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + lane * 32))[i]; b[i] = ((const int*)(B + lane * 32))[i]; }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) C[lane * 4 + r] = c[r];
}

__global__ void rate(float* out, int iters) {
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
        c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0) / (4.0f * iters); out[1] = c0[0] + c1[0] + c2[0] + c3[0]; }
}

static unsigned char enc(int v) {      // e4m3fn of small integers
    switch (v) { case 0: return 0x00; case 1: return 0x38; case -1: return 0xB8; case 2: return 0x40; case -2: return 0xC0;
                 case 3: return 0x44; case -3: return 0xC4; case 4: return 0x48; case -4: return 0xC8; }
    return 0;
}

int main() {
    const int M = 16, K = 128;
    std::vector<int> a(M * K), b(M * K);
    srand(1);
    for (auto& x : a) x = rand() % 9 - 4;
    for (auto& x : b) x = rand() % 9 - 4;
    std::vector<float> ref(M * M, 0.f);
    for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) { int s = 0; for (int k = 0; k < K; ++k) s += a[i * K + k] * b[j * K + k]; ref[i * M + j] = (float)s; }
    unsigned char *dA, *dB; float* dC;
    hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dC, 256 * 4);
    // hypotheses for k(lane q = lane>>4, byte p): 0: k = 32q + p;  1: k = 16q + (p & 15) + 64 (p >> 4);  2: k = 8q + (p & 7) + 32 (p >> 3)
    for (int hyp = 0; hyp < 3; ++hyp) {
        std::vector<unsigned char> pa(64 * 32), pb(64 * 32);
        for (int lane = 0; lane < 64; ++lane)
            for (int p = 0; p < 32; ++p) {
                const int r = lane & 15, q = lane >> 4;
                const int k = hyp == 0 ? 32 * q + p : hyp == 1 ? 16 * q + (p & 15) + 64 * (p >> 4) : 8 * q + (p & 7) + 32 * (p >> 3);
                pa[lane * 32 + p] = enc(a[r * K + k]);
                pb[lane * 32 + p] = enc(b[r * K + k]);
            }
        hipMemcpy(dA, pa.data(), 64 * 32, hipMemcpyHostToDevice);
        hipMemcpy(dB, pb.data(), 64 * 32, hipMemcpyHostToDevice);
        for (int sc = 0; sc < 3; ++sc) {
            const int sa = sc == 0 ? 127 : sc == 1 ? 128 : 0x7F7F7F80, sb = 127;      // 2^0 | 2^1 | byte0 = 0x80 (2^1), other bytes 0x7F
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb);
            std::vector<float> c(256);
            hipMemcpy(c.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
            // C/D map of 16x16: col = lane & 15, row = 4 (lane >> 4) + r
            double err1 = 0, err2 = 0;
            for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
                const int col = lane & 15, row = 4 * (lane >> 4) + r;
                err1 = fmax(err1, fabs(c[lane * 4 + r] - ref[row * M + col]));
                err2 = fmax(err2, fabs(c[lane * 4 + r] - 2 * ref[row * M + col]));
            }
            printf("hyp %d scale_a %#x: max|C - ref| = %g   max|C - 2 ref| = %g   (C[0] = %g, ref = %g)\n", hyp, sa, err1, err2, c[0], ref[0]);
        }
    }
    float* dR; hipMalloc(&dR, 8);
    hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, dR, 10000);
    hipLaunchKernelGGL(rate, dim3(1), dim3(64), 0, 0, dR, 100000);
    float r[2]; hipMemcpy(r, dR, 8, hipMemcpyDeviceToHost);
    printf("issue interval: %.1f clock64 ticks per 16x16x128 scaled MFMA (one wave)\n", r[0]);
    return 0;
}
