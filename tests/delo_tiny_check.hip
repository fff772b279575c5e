// tests/delo_tiny_check.hip: the DELO step of the transfer kernel (grt_dev_phys.hpp:delo_apply, closed form) ON THE DEVICE against the
// reference-literal affine map (delo_map: radtrans_integrate.f90:746-837 with IEEE divisions and the cofactor inverse) on the host, for steps
// between a near-vacuum zone -- alpha_I at and below the smallest normal number, 1e-305 ... 1e-312 -- and an emitting one, in both orders,
// mixed in one wave with ordinary steps.  Prints the worst relative difference per class and the number of non-finite device results.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast-honor-pragmas -I grtrans_amd/csrc tests/delo_tiny_check.hip -o tests/delo_tiny_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "grt_dev_phys.hpp"
using namespace grtd;
struct Step { double j0[4], K0[7], j1[4], K1[7], dx, delta, I[4]; };
__global__ void kern(const Step *s, double *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) delo_apply(s[i].j0, s[i].K0, s[i].j1, s[i].K1, s[i].dx, s[i].delta, s[i].I, o + 4 * (size_t)i);
}
static double u01() { return (double)rand() / RAND_MAX; }
static void coef(double aI, double S, double *j, double *K) {      // source function S, polarised fractions < 1, Faraday terms up to 30 alpha_I
  K[0] = aI;
  for (int c = 1; c < 4; ++c) K[c] = aI * (u01() - 0.5) * 0.9;
  for (int c = 4; c < 7; ++c) K[c] = aI * (u01() - 0.5) * 60.0;
  j[0] = aI * S;
  for (int c = 1; c < 4; ++c) j[c] = j[0] * (u01() - 0.5) * 0.9;
}
int main() {
  const int n = 1 << 14;
  std::vector<Step> st(n);
  std::vector<int> cls(n);
  srand(7);
  const double tinyv[6] = {1e-305, 3e-308, 1e-308, 4e-309, 1e-310, 1e-312};
  for (int i = 0; i < n; ++i) {
    Step &s = st[i];
    const int c = i % 4;                       // 0: ordinary step, 1: far end in vacuum, 2: near end in vacuum, 3: both
    cls[i] = c;
    const double t = tinyv[rand() % 6] * (0.5 + u01()), big = pow(10.0, -2.0 + 3.0 * u01());
    const double a0 = (c == 1 || c == 3) ? t : big, a1 = (c == 2 || c == 3) ? t * (0.5 + u01()) : big * (0.5 + u01());
    coef(a0, 1e4 * (0.5 + u01()), s.j0, s.K0);
    coef(a1, 1e4 * (0.5 + u01()), s.j1, s.K1);
    s.delta = (c == 3) ? 0.05 + u01() : pow(10.0, -1.5 + 2.5 * u01());     // (both ends in vacuum cannot be thick in a real run; the algebra still holds)
    s.dx = (c == 3) ? 1.0 : 2.0 * s.delta / (a0 + a1);
    for (int k = 0; k < 4; ++k) s.I[k] = (k ? (u01() - 0.5) * 0.5 : 1.0) * 1e3;
  }
  Step *ds; double *dout;
  if (hipMalloc(&ds, n * sizeof(Step)) != hipSuccess || hipMalloc(&dout, (size_t)n * 32) != hipSuccess) return 2;
  (void)hipMemcpy(ds, st.data(), n * sizeof(Step), hipMemcpyHostToDevice);
  kern<<<n / 256, 256>>>(ds, dout, n);
  std::vector<double> od((size_t)n * 4);
  if (hipMemcpy(od.data(), dout, (size_t)n * 32, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  double worst[4] = {0, 0, 0, 0};
  long nonfinite = 0, refbad = 0;
  for (int i = 0; i < n; ++i) {
    Affine m;
    delo_map(st[i].j0, st[i].K0, st[i].j1, st[i].K1, st[i].dx, st[i].delta, m);
    double want[4], sc = 0.0;
    bool fin = true;
    for (int r = 0; r < 4; ++r) {
      want[r] = m.b[r];
      for (int c = 0; c < 4; ++c) want[r] += m.A[r][c] * st[i].I[c];
      sc = fmax(sc, fabs(want[r]));
      fin = fin && std::isfinite(want[r]);
    }
    if (!fin) { ++refbad; continue; }
    for (int r = 0; r < 4; ++r) {
      const double d = od[4 * (size_t)i + r];
      if (!std::isfinite(d)) { ++nonfinite; break; }
      worst[cls[i]] = fmax(worst[cls[i]], fabs(d - want[r]) / sc);
    }
  }
  printf("ordinary %.3e far_vacuum %.3e near_vacuum %.3e both %.3e nonfinite %ld ref_nonfinite %ld of %d\n", worst[0], worst[1], worst[2], worst[3], nonfinite, refbad, n);
  return 0;
}
