// What the matrix pipe sustains on this chip with nothing else going on: 512 workgroups x 4 waves (two waves per SIMD), every wave
// a loop of 32 independent v_mfma_f32_16x16x32_bf16 (the accumulator set of a 128 x 64 wave tile), no memory traffic in the loop.
// Launch lengths from 0.1 ms to 20 ms show what the clocks do under sustained matrix load.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_ceiling.hip -o build_variants/mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, int iters) {
    f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[4], b[8];
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 8; ++e) a[k][e] = (__bf16)(float)((threadIdx.x + k + e) & 7);
    for (int k = 0; k < 8; ++k)
        for (int e = 0; e < 8; ++e) b[k][e] = (__bf16)(float)((threadIdx.x * 3 + k + e) & 3);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
static void run(const char* name, float* d, int iters, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = (double)blocks * 4 * iters * NACC * 16384.0;
    printf("%-28s %d workgroups, %6d iterations: %8.3f ms per launch, %7.1f TFLOP/s\n", name, blocks, iters, ms, flop / ms / 1e9);
}

int main() {
    float* d;
    hipMalloc(&d, 2048 * 256 * 4);
    for (int iters : {200, 2000, 20000, 200000}) run<32>("32 accumulators / wave", d, iters, 512);
    for (int iters : {2000, 20000}) run<16>("16 accumulators / wave", d, iters * 2, 512);
    for (int iters : {2000, 20000}) run<32>("32 acc, one wave per SIMD", d, iters, 256);
    for (int iters : {2000, 20000}) run<8>("8 accumulators / wave", d, iters * 4, 512);
    return 0;
}
