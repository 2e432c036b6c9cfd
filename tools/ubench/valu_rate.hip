// Microbenchmark: issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950 (how many cycles does a SIMD spend per wave64 VALU instruction?).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o build/valu_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NACC = 16, ITERS = 65536;
#pragma clang diagnostic ignored "-Wunused-value"
__device__ long long g_cyc;
#define T0 long long t0 = clock64();
#define T1 if (blockIdx.x == 0 && threadIdx.x == 0) g_cyc = clock64() - t0;
__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
  float x[NACC];
  for (int i = 0; i < NACC; i++) x[i] = threadIdx.x + i;
  T0
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
  }
  T1
  float s = 0; for (int i = 0; i < NACC; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pk(float* out, float a, float b) {
  f2 x[NACC]; f2 av = {a, a}, bv = {b, b};
  for (int i = 0; i < NACC; i++) x[i] = f2{(float)threadIdx.x + i, (float)i};
  T0
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
  }
  T1
  f2 s = {0, 0}; for (int i = 0; i < NACC; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ __launch_bounds__(256) void k_dep(float* out, float a, float b) {  // one dependent chain per wave
  float x = threadIdx.x;
  T0
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
  }
  T1
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void k_dpp(float* out, float a, float b) {  // row_shr DPP adds (the row sums of the cooperative kernels)
  float x[NACC];
  for (int i = 0; i < NACC; i++) x[i] = threadIdx.x + i;
  T0
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
  }
  T1
  float s = 0; for (int i = 0; i < NACC; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K> void run(const char* name, K k, int wavesPerSimd, float* out) {
  int cus = 256; hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); cus = p.multiProcessorCount;
  int blocks = cus * wavesPerSimd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_simd = (double)wavesPerSimd * NACC * ITERS;
  double clk = p.clockRate * 1e3;  // Hz
  long long cyc; hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
  printf("%-8s waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD at the nominal %.2f GHz; clock64(): %.2f ticks per instruction of one wave (%.1f MHz tick rate)\n", name, wavesPerSimd, ms, ms * 1e-3 * clk / instr_per_simd, clk * 1e-9, (double)cyc / (NACC * (double)ITERS), cyc / (ms * 1e3));
}
int main() {
  float* out; hipMalloc(&out, 256 * 256 * 8 * sizeof(float) * 4);
  for (int w : {1, 2, 4, 8}) { run("fma", k_fma, w, out); run("pk_fma", k_pk, w, out); run("dep_fma", k_dep, w, out); run("dpp_add", k_dpp, w, out); }
  return 0;
}
