// tests/fp32_island_check.hip: are the REAL(4) building blocks of the fluid samplers bit-identical on gfx950 and on the host?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast-honor-pragmas -I grtrans_amd/csrc tests/fp32_island_check.hip -o tests/fp32_island_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "grt_dev_phys.hpp"
using namespace grtd;
#define NOUT 18
template <class T> GRT_HD void stages(float x, float y, T *o) {
  GRT_NO_CONTRACT
  o[0] = cosf_cr(x * 3.f); o[1] = sinf_cr(x * 3.f); o[2] = acosf_cr(fabsf(o[0])); o[3] = logf_cr(x * 20.f + 1.f);
  o[4] = x / (y + 0.5f); o[5] = sqrtf(x + y); o[6] = powf_cr(x + 0.1f, y * 3.f);
  float v[4] = {x, y, x * y, x + y};
  o[7] = bilin(v, x, y);
  o[8] = (float)((double)x * (double)y + 1.0);
  Metric m = blmetric_cov_real(x * 20.f + 2.f, y * 3.f, 0.9f);
  o[9] = (float)m.g[0]; o[10] = (float)m.g[3]; o[11] = (float)m.g[9];
  float v1, v2, v3;
  lnrf_frame_inv_real(x * 0.5f, y * 0.3f, x * y * 0.4f, x * 20.f + 2.f, 0.9f, y * 3.f, v1, v2, v3);
  o[12] = v1; o[13] = v2; o[14] = v3;
  // r06: the REAL(4) quotient through a shared fp64 reciprocal (grt_dev_phys.hpp:f32_div) against the division itself: O(1) operands, operands
  // 30 decades apart, and a denominator that is 0 for some inputs
  const float big = ldexpf(x + 0.5f, (int)(y * 200.f) - 100);
  o[15] = f32_div(x, f32_rcp(y + 0.5f)) - x / (y + 0.5f);
  o[16] = f32_div(y + 0.25f, f32_rcp(big)) - (y + 0.25f) / big;
  const float den = (x < 0.001f) ? 0.f : x - 0.5f;
  const float q1 = f32_div(y, f32_rcp(den)), q2 = y / den;
  o[17] = (q1 == q2 || (q1 != q1 && q2 != q2)) ? 0.f : 1.f;
}
__global__ void kern(const float *x, const float *y, float *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stages(x[i], y[i], o + (size_t)i * NOUT);
}
int main() {
  const int n = 1 << 20;
  std::vector<float> x(n), y(n), od((size_t)n * NOUT), oh(NOUT);
  srand(5);
  for (int i = 0; i < n; ++i) { x[i] = (float)rand() / RAND_MAX; y[i] = (float)rand() / RAND_MAX; }
  float *dx, *dy, *dout;
  if (hipMalloc(&dx, n * 4) != hipSuccess || hipMalloc(&dy, n * 4) != hipSuccess || hipMalloc(&dout, (size_t)n * NOUT * 4) != hipSuccess) return 2;
  (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
  kern<<<n / 256, 256>>>(dx, dy, dout, n);
  if (hipMemcpy(od.data(), dout, (size_t)n * NOUT * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  long bad[NOUT] = {0};
  for (int i = 0; i < n; ++i) {
    stages(x[i], y[i], oh.data());
    for (int k = 0; k < NOUT; ++k) if (std::memcmp(&oh[k], &od[(size_t)i * NOUT + k], 4)) ++bad[k];
  }
  const char *nm[NOUT] = {"cosf_cr", "sinf_cr", "acosf_cr", "logf_cr", "div", "sqrtf", "powf_cr", "bilin", "d2f", "g_tt", "g_tphi", "g_phiphi", "lnrf_vr", "lnrf_vth", "lnrf_vph", "f32_div", "f32_div_w", "f32_div_0"};
  long tot = 0;
  for (int k = 0; k < NOUT; ++k) { printf("%-10s differing: %ld of %d\n", nm[k], bad[k], n); tot += bad[k]; }
  return tot ? 1 : 0;
}
