// tests/sqrt_square_check.hip (test infrastructure; built by __graft_entry__.build(), run by tests/test_gpu_parity.py) -- r06: the property the formal step leans on (grt_dev_phys.hpp:formal_apply: rt = fsqrt(h h + ap ap), lam2 = fsqrt(rt - h)):
// fsqrt(fl(h^2)) >= |h|, and fsqrt correctly rounded, for the product's fsqrt (rsq seed, one coupled Goldschmidt step, ONE correction since r06) on the device.
// make -C grtrans_amd/csrc -f Makefile -f ../../tests/checks.mk ../../tests/sqrt_square_check   (the product Makefile's compiler and flags)
// (r06 tried a cheaper root -- one third-order step, 0.75 ulp -- as the general fsqrt: sqrt(h^2) came out one ulp below |h| and the formal step returned NaN: DESIGN 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "grt_dev_special.hpp"
using namespace grtd;
__global__ void k(const double *h, const double *ap, unsigned long long *below, unsigned long long *notcr, int n, int reps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  unsigned long long b = 0, c = 0;
  double x = h[i], a = ap[i];
  for (int r = 0; r < reps; ++r) {
    const double hd = x * (1.0 + r * 1e-7), apv = (r & 1) ? 0.0 : a * hd;      // every other trial with ap = 0 exactly
    const double rad = hd * hd + apv * apv;
    const double rt = fsqrt(rad);
    if (rt < fabs(hd)) ++b;
    if (rt != __dsqrt_rn(rad)) ++c;
  }
  atomicAdd(below, b); atomicAdd(notcr, c);
}
int main() {
  const int n = 1 << 22, reps = 64;
  std::vector<double> h(n), ap(n);
  srand(9);
  for (int i = 0; i < n; ++i) { h[i] = ldexp(1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX * 1e-9, rand() % 120 - 90) * ((rand() & 1) ? -1 : 1); ap[i] = ldexp((double)rand() / RAND_MAX, -(rand() % 60)); }
  double *dh, *da; unsigned long long *db, *dc, hb = 0, hc = 0;
  hipMalloc(&dh, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, 8); hipMalloc(&dc, 8);
  hipMemcpy(dh, h.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(da, ap.data(), n * 8, hipMemcpyHostToDevice);
  hipMemset(db, 0, 8); hipMemset(dc, 0, 8);
  k<<<n / 256, 256>>>(dh, da, db, dc, n, reps);
  hipMemcpy(&hb, db, 8, hipMemcpyDeviceToHost); hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
  printf("fsqrt(h^2 + ap^2) over %.3g (h, ap) pairs (half of them with ap = 0): below |h|: %llu; different from the correctly rounded root: %llu\n", (double)n * reps, hb, hc);
  return (hb != 0 || hc != 0) ? 1 : 0;
}
