// tools/exec_skip_bench.hip: does a wave64 fp64 VALU instruction cost less when whole 16-lane groups of the wave are masked off?
// (r04: grt_geo_kernel's partial top chunk -- 10 of 64 lanes -- turned out to cost about a third of a full pass, not a full one.)
// Eight independent FMA chains per lane, 4096 trips, under a divergent `if` that leaves `mask` lanes active; 8192 waves.
//   hipcc --offload-arch=gfx950 -O3 tools/exec_skip_bench.hip -o build/exp/exec_skip_bench && build/exp/exec_skip_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long mask, int trips, int f32) {
  const int lane = threadIdx.x & 63;
  double a0 = 1.0 + lane, a1 = 2.0, a2 = 3.0, a3 = 4.0, a4 = 5.0, a5 = 6.0, a6 = 7.0, a7 = 8.0;
  float b0 = 1.f + lane, b1 = 2.f, b2 = 3.f, b3 = 4.f, b4 = 5.f, b5 = 6.f, b6 = 7.f, b7 = 8.f;
  const double m = 1.0000001, c = 1e-9;
  if ((mask >> lane) & 1ull) {
    if (!f32) {
#pragma unroll 1
      for (int i = 0; i < trips; ++i) {
        a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
        a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
      }
    } else {
#pragma unroll 1
      for (int i = 0; i < trips; ++i) {
        b0 = fmaf(b0, 1.0000001f, 1e-9f); b1 = fmaf(b1, 1.0000001f, 1e-9f); b2 = fmaf(b2, 1.0000001f, 1e-9f); b3 = fmaf(b3, 1.0000001f, 1e-9f);
        b4 = fmaf(b4, 1.0000001f, 1e-9f); b5 = fmaf(b5, 1.0000001f, 1e-9f); b6 = fmaf(b6, 1.0000001f, 1e-9f); b7 = fmaf(b7, 1.0000001f, 1e-9f);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7);
}
int main() {
  const int blocks = 2048 * 4, trips = 4096;
  double *out;
  if (hipMalloc(&out, (size_t)blocks * 256 * 8) != hipSuccess) return 2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  struct { const char *name; unsigned long long mask; } cases[] = {
      {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lanes 0-9", 0x3ffull}, {"lanes 0-59", 0x0fffffffffffffffull},
      {"one lane of each 16-lane group", 0x0001000100010001ull}, {"lanes 16-31", 0xffff0000ull}, {"lanes 0-15 and 48-63", 0xffff00000000ffffull}};
  for (int f32 = 0; f32 < 2; ++f32)
    for (auto &cs : cases) {
      k<<<blocks, 256>>>(out, cs.mask, 16, f32);
      hipDeviceSynchronize();
      float best = 1e30f;
      for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        k<<<blocks, 256>>>(out, cs.mask, trips, f32);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double insts = (double)blocks * 4 * trips * 8;      // wave-level FMA instructions
      printf("%s %-34s %8.3f ms  %.3f wave-FMA per ns  (%.2f cycles per wave instruction per SIMD at 2.4 GHz, 1024 SIMDs)\n", f32 ? "f32" : "f64", cs.name, best,
             insts / best / 1e6, best * 1e-3 * 2.4e9 * 1024 / insts);
    }
  return 0;
}
