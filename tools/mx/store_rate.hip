// Per-CU store rate: G workgroups of 512 threads (one per CU: 160 KB of LDS requested) each write `kb` KB, 16 bytes per lane, as
// whole 1-KB rows per wave-instruction (form 0) or as two 512-byte row segments 4 KB apart (form 1: an operand-out epilogue's
// [hi | lo] rows).  hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FORM>
__global__ __launch_bounds__(512) void st(float4* out, int kb, int reps) {
    extern __shared__ char smem[];
    if (FORM == 4 && (blockIdx.x & 7) != 0) return;          // form 4: only the workgroups of XCD 0 (round-robin placement) store
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4 v = make_float4(threadIdx.x, blockIdx.x, 1.f, 2.f);
    float4* base = out + (size_t)blockIdx.x * (kb * 64);            // kb KB = kb * 64 float4
    for (int r = 0; r < reps; ++r) {
        if (FORM == 3 || FORM == 4) base = out + ((size_t)r * gridDim.x + blockIdx.x) * (kb * 64);      // streaming: a fresh region every repetition
        for (int i = wave; i < kb; i += 8) {                          // one KB per wave-instruction
            if (FORM == 2) {         // tile-like: this workgroup owns a 256-row x 1-KB tile of a [rows, 12288 B] matrix (a residual-add GEMM's output)
                float4* t = out + (size_t)(blockIdx.x >> 2) * 256 * 768 + (blockIdx.x & 3) * 64;
                t[(size_t)(i & 255) * 768 + lane] = v;
            } else if (FORM == 0 || FORM == 3 || FORM == 4) base[i * 64 + lane] = v;
            else base[(i >> 1) * 128 + (i & 1) * 32 + (lane >> 5) * 64 + (lane & 31)] = v;      // 2 x 512 B, 1 KB apart
        }
        v.x += 1.f;
    }
    if (smem[threadIdx.x] == 77) out[0] = v;
}
template <int FORM> void run(int G, int kb) {
    float4* out; hipMalloc(&out, (size_t)256 * kb * 1024 * 52);
    hipFuncSetAttribute((const void*)st<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    st<FORM><<<G, 512, 160 * 1024>>>(out, kb, 2);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 50;
    hipEventRecord(e0);
    st<FORM><<<G, 512, 160 * 1024>>>(out, kb, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("form %d  %3d workgroups x %4d KB: %7.1f GB/s per CU, %6.2f TB/s total, %6.1f us per %d KB\n", FORM, G, kb,
           (double)kb * 1024 * reps / (ms * 1e-3) / 1e9, (double)G * kb * 1024 * reps / (ms * 1e-3) / 1e12, ms * 1e3 / reps, kb);
    hipFree(out);
}
int main() {
    for (int G : {1, 8, 32, 64, 256}) { run<0>(G, 256); run<1>(G, 256); }
    for (int G : {1, 8, 32, 64, 256}) { run<3>(G, 256); }
    for (int G : {64, 128, 256}) { run<4>(G, 256); }           // G / 8 CUs of ONE XCD storing: per-CU rate printed is right, totals are x 1/8
    return 0;
}
