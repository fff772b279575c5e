// tools/experiments/sgpr_const_bench.hip -- r06: what does a Horner step cost on gfx950 when its fp64 coefficient is (A) v_mov'ed into a VGPR pair and
// accumulated with v_fmac_f64 (what hipcc emits: 3 VALU a step) or (B) held in an SGPR pair and used as the addend of v_fma_f64 (1 VALU a step)?
// hipcc --offload-arch=gfx950 -O3 -mllvm -disable-machine-licm sgpr_const_bench.hip -o sgpr_const_bench && ./sgpr_const_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double sc(double c) { asm("" : "+s"(c)); return c; }
template <int MODE>
__global__ __launch_bounds__(256, 3) void k(double *y, int n, double seed) {
  double z0 = seed + threadIdx.x * 1e-9, z1 = z0 + 0.1, z2 = z0 + 0.2, z3 = z0 + 0.3;
  for (int it = 0; it < n; ++it) {
// (four chains with four different coefficient sets: a coefficient used once is what the compiler moves into the accumulator)
#define STEP(C) if (MODE == 0) { p0 = fma(p0, z0, C); p1 = fma(p1, z1, C * 1.0000001); p2 = fma(p2, z2, C * 1.0000002); p3 = fma(p3, z3, C * 1.0000003); } \
                else { p0 = fma(p0, z0, sc(C)); p1 = fma(p1, z1, sc(C * 1.0000001)); p2 = fma(p2, z2, sc(C * 1.0000002)); p3 = fma(p3, z3, sc(C * 1.0000003)); }
    double p0 = 1.58969099521155010221e-10, p1 = p0, p2 = p0, p3 = p0;
    STEP(-2.50507602534068634195e-08) STEP(2.75573137070700676789e-06) STEP(-1.98412698298579493134e-04) STEP(8.33333333332248946124e-03)
    STEP(-1.66666666666666324348e-01) STEP(4.16666666666666019037e-02) STEP(-1.38888888888741095749e-03) STEP(2.48015872894767294178e-05)
    z0 = p0 * 1e-3 + 0.3; z1 = p1 * 1e-3 + 0.4; z2 = p2 * 1e-3 + 0.5; z3 = p3 * 1e-3 + 0.6;
  }
  y[blockIdx.x * blockDim.x + threadIdx.x] = z0 + z1 + z2 + z3;
}
int main() {
  double *y; hipMalloc(&y, 4096 * 256 * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(3072), dim3(256), 0, 0, y, 20000, 0.25);
      else hipLaunchKernelGGL(k<1>, dim3(3072), dim3(256), 0, 0, y, 20000, 0.25);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      // 32 Horner fma + 4 fma (z update) per iteration and lane
      printf("mode %d (%s): %.2f ms, %.2f fp64 TFLOP/s counted on the 36 fma\n", mode, mode ? "SGPR coefficient, v_fma_f64" : "as compiled: v_mov x2 + v_fmac_f64", ms,
             3072.0 * 256 * 20000 * 36 * 2 / (ms * 1e-3) / 1e12);
    }
  return 0;
}
