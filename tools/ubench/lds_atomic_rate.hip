// Microbenchmark: LDS cycles of ds_add_f32 (no return) against ds_write_b32 / ds_read_b32 on gfx950, at the leap kernel's occupancy (8 waves per CU) and lane patterns.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic_rate.hip -o variants/lds_atomic_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 4096, NOPS = 32;
__device__ long long g_cyc;
// mode 0: ds_add_f32, 1: ds_write_b32, 2: ds_read_b32.  pattern 0: 64 distinct dwords per instruction; 1: lanes l of a 16-lane row with l >= nact masked off; 2: every active lane of a row hits
// the SAME dword (four addresses per instruction); 3: as 1, two lanes of a row share a dword
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int pattern, int nact, int rowstride) {
  __shared__ float s[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) s[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l = lane & 15, r = lane >> 4;
  int idx = wv * 4096 + r * rowstride + (pattern == 2 ? 0 : (pattern == 3 ? l >> 1 : l));
  const bool on = l < nact;
  float v = 1.f + lane, acc = 0.f;
  const unsigned a = (unsigned)(size_t)(s + idx) ;  // LDS byte address (low 32 bits of the generic-to-local cast)
  unsigned addr = (unsigned)((char*)(s + idx) - (char*)s);
  long long t0 = clock64();
  for (int it = 0; it < ITERS; it++) {
    if (on) {
#pragma unroll
      for (int i = 0; i < NOPS; i++) {
        if (MODE == 0) asm volatile("ds_add_f32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(i * 64) : "memory");
        else if (MODE == 1) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(i * 64) : "memory");
        else { float t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(i * 64) : "memory"); asm volatile("s_waitcnt lgkmcnt(15)"); acc += 0.f * 0.f; (void)t; }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) g_cyc = t1 - t0;
  (void)a;
  out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x] + acc;
}
template <int MODE> void run(const char* name, int pattern, int nact, int rowstride, int wgPerCu, float* out) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int blocks = p.multiProcessorCount * wgPerCu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, pattern, nact, rowstride); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, pattern, nact, rowstride); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cyc; hipMemcpyFromSymbol(&cyc, HIP_SYMBOL(g_cyc), sizeof(cyc));
  const double inst_per_cu = (double)ITERS * NOPS * 4 * wgPerCu;
  printf("%-12s pattern %d nact %2d rowstride %4d  wg/CU %d: %.3f ms, %.2f clock64 ticks per wave-instruction per CU (one wave: %.1f ticks per instruction)\n", name, pattern, nact, rowstride, wgPerCu, ms,
         (double)cyc / inst_per_cu, (double)cyc / ((double)ITERS * NOPS));
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  for (int wg : {1, 2}) {
    run<0>("ds_add_f32", 0, 16, 16, wg, out);
    run<0>("ds_add_f32", 1, 5, 16, wg, out);
    run<0>("ds_add_f32", 1, 5, 1040, wg, out);   // rows 16 banks apart (the kernel's record stride)
    run<0>("ds_add_f32", 1, 1, 1040, wg, out);
    run<0>("ds_add_f32", 2, 16, 1040, wg, out);
    run<0>("ds_add_f32", 2, 5, 1040, wg, out);
    run<0>("ds_add_f32", 3, 6, 1040, wg, out);
    run<1>("ds_write_b32", 0, 16, 16, wg, out);
    run<1>("ds_write_b32", 1, 5, 1040, wg, out);
    run<2>("ds_read_b32", 0, 16, 16, wg, out);
    run<2>("ds_read_b32", 1, 5, 1040, wg, out);
  }
  return 0;
}
