// tools/experiments/dp_rate_bench.hip -- r06: what an fp64 vector instruction costs on gfx950 (MI355X) as a function of the waves per SIMD and of
// the independent chains per wave, and whether fp32 / integer instructions issued between them cost anything.
// hipcc --offload-arch=gfx950 -O3 dp_rate_bench.hip -o dp_rate_bench && ./dp_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double sc(double c) { asm("" : "+s"(c)); return c; }
__device__ __forceinline__ float scf(float c) { asm("" : "+s"(c)); return c; }
// MODE 0: CH chains of v_fma_f64; 1: + CH chains of v_fma_f32 beside them; 2: + CH v_add_u32 / v_xor chains; 3: v_mul_f64; 4: v_add_f64; 5: only the fp32 chains
template <int MODE, int CH>
__global__ __launch_bounds__(256) void k(double *y, int n, double seed) {
  extern __shared__ double lds[];
  double p[CH]; float q[CH]; unsigned u[CH];
  const double c1 = sc(0.99999), c2 = sc(1e-5);
  const float f1 = scf(0.99999f), f2 = scf(1e-5f);
#pragma unroll
  for (int j = 0; j < CH; ++j) { p[j] = seed + j + threadIdx.x * 1e-9; q[j] = (float)p[j]; u[j] = threadIdx.x + j; }
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (MODE <= 2) p[j] = fma(p[j], c1, c2);
        if (MODE == 3) p[j] = p[j] * c1;
        if (MODE == 4) p[j] = p[j] + c2;
        if (MODE == 1 || MODE == 5) q[j] = fmaf(q[j], f1, f2);
        if (MODE == 2) { u[j] ^= u[j] << 3; }      // (two integer instructions a step: v_lshlrev + v_xor)
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < CH; ++j) s += p[j] + q[j] + u[j];
  y[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (n < 0) lds[threadIdx.x] = s;
}
template <int MODE, int CH> void run(double *y, int occ, const char *what) {
  const int n = 4000, blocks = 256 * occ * 2;        // two resident rounds
  const size_t lds = (size_t)(160 * 1024 / occ) - 1024;     // 160 KB of LDS a CU: `occ` blocks of 4 waves = occ waves per SIMD
  hipFuncSetAttribute((const void *)k<MODE, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, CH>), dim3(blocks), dim3(256), lds, 0, y, n, 0.25);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  // wave-instructions of the counted kind per SIMD: blocks * 4 waves / 1024 SIMDs * n * 8 * CH
  const double per_simd = (double)blocks * 4 / 1024 * n * 8 * CH;
  printf("%-34s waves/SIMD %d chains %d: %7.3f ms  %.2f cycles per counted wave-instruction and SIMD at 2.4 GHz\n", what, occ, CH, best, best * 1e-3 * 2.4e9 / per_simd);
}
int main() {
  double *y; hipMalloc(&y, 8192 * 256 * 8);
  for (int occ : {1, 2, 3, 4, 6, 8}) { run<0, 1>(y, occ, "v_fma_f64"); run<0, 2>(y, occ, "v_fma_f64"); run<0, 4>(y, occ, "v_fma_f64"); run<0, 8>(y, occ, "v_fma_f64"); }
  for (int occ : {1, 3, 8}) { run<3, 4>(y, occ, "v_mul_f64"); run<4, 4>(y, occ, "v_add_f64"); run<5, 4>(y, occ, "v_fma_f32 alone"); run<1, 4>(y, occ, "v_fma_f64 + as many v_fma_f32"); run<2, 4>(y, occ, "v_fma_f64 + as many v_mad_u32"); }
  return 0;
}
