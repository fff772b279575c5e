// tools/experiments/seed_accuracy.hip -- r06: (1) relative error of gfx950's v_rcp_f64 / v_rsq_f64 seeds, of the two-Newton-step frcp / frsqrt the
// kernels use, and of ONE third-order step (r (1 + e + e^2); y (1 + e/2 + 3 e^2/8)); (2) is (float)((double)a * rcp64(b)) the correctly
// rounded REAL(4) quotient a / b (bit-equal to the IEEE division) on random and special operands?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off seed_accuracy.hip -o seed_accuracy && ./seed_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
__device__ double rcp2(double b) { double r = __builtin_amdgcn_rcp(b); r = fma(fma(-b, r, 1.0), r, r); r = fma(fma(-b, r, 1.0), r, r); return r; }
__device__ double rcp3(double b) { double r = __builtin_amdgcn_rcp(b); const double e = fma(-b, r, 1.0); return fma(r, fma(e, e, e), r); }
__device__ double rsq2(double x) { double y = __builtin_amdgcn_rsq(x); double e = fma(-(x * y), y, 1.0); y = fma(0.5 * y, e, y); e = fma(-(x * y), y, 1.0); y = fma(0.5 * y, e, y); return y; }
__device__ double rsq3(double x) { double y = __builtin_amdgcn_rsq(x); const double e = fma(-(x * y), y, 1.0); return fma(y, e * fma(e, 0.375, 0.5), y); }
__global__ void k64(const double *x, double *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  o[6 * i] = __builtin_amdgcn_rcp(x[i]); o[6 * i + 1] = rcp2(x[i]); o[6 * i + 2] = rcp3(x[i]);
  o[6 * i + 3] = __builtin_amdgcn_rsq(x[i]); o[6 * i + 4] = rsq2(x[i]); o[6 * i + 5] = rsq3(x[i]);
}
__global__ void k32(const float *a, const float *b, float *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  o[3 * i] = a[i] / b[i];
  o[3 * i + 1] = (float)((double)a[i] * rcp2((double)b[i]));
  o[3 * i + 2] = (float)((double)a[i] * rcp3((double)b[i]));
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), o((size_t)6 * n);
  srand(7);
  for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX * 1e-9, rand() % 200 - 100);
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, (size_t)6 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k64<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(o.data(), dout, (size_t)6 * n * 8, hipMemcpyDeviceToHost);
  double mx[6] = {0};
  for (int i = 0; i < n; ++i) {
    const long double r = 1.0L / (long double)x[i], q = 1.0L / sqrtl((long double)x[i]);
    for (int k = 0; k < 6; ++k) { const long double w = k < 3 ? r : q; const double e = (double)fabsl(((long double)o[6 * (size_t)i + k] - w) / w); if (e > mx[k]) mx[k] = e; }
  }
  const char *nm[6] = {"v_rcp_f64 seed", "rcp + 2 Newton steps (frcp)", "rcp + 1 third-order step", "v_rsq_f64 seed", "rsq + 2 Newton steps (frsqrt)", "rsq + 1 third-order step"};
  for (int k = 0; k < 6; ++k) printf("%-32s max relative error %.3e = 2^%.1f (%.2f ulp)\n", nm[k], mx[k], log2(mx[k]), mx[k] / 1.11e-16);
  // fp32 quotients
  std::vector<float> a(n), b(n), of((size_t)3 * n);
  for (int i = 0; i < n; ++i) {
    unsigned ua = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31), ub = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    if (i % 4 == 0) { a[i] = ldexpf(1.f + (float)rand() / RAND_MAX, rand() % 40 - 20); b[i] = ldexpf(1.f + (float)rand() / RAND_MAX, rand() % 40 - 20); }
    else { memcpy(&a[i], &ua, 4); memcpy(&b[i], &ub, 4); }      // random bit patterns: subnormals, infinities, NaNs, zeros among them
  }
  a[1] = 0.f; b[1] = 0.f; a[2] = 1.f; b[2] = 0.f; a[3] = 1.f; b[3] = INFINITY; a[5] = INFINITY; b[5] = INFINITY; a[6] = 1e-38f; b[6] = 1e10f; a[7] = 3e38f; b[7] = 1e-10f;
  float *da, *db, *dof; hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dof, (size_t)3 * n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  k32<<<n / 256, 256>>>(da, db, dof, n);
  hipMemcpy(of.data(), dof, (size_t)3 * n * 4, hipMemcpyDeviceToHost);
  long bad2 = 0, bad3 = 0, badh = 0, bad2o = 0, nord = 0;
  for (int i = 0; i < n; ++i) {
    const float h = a[i] / b[i];
    const bool nan0 = of[3 * (size_t)i] != of[3 * (size_t)i];
    const bool ordinary = std::isfinite(a[i]) && std::isfinite(b[i]) && b[i] != 0.f && fabsf(b[i]) > 1e-37f && std::isfinite(h) && (h == 0.f || fabsf(h) > 1.2e-38f);
    nord += ordinary;
    auto same = [&](float u, float v) { return (u != u && v != v) || memcmp(&u, &v, 4) == 0; };
    if (!same(of[3 * (size_t)i], h)) ++badh;
    if (!same(of[3 * (size_t)i + 1], of[3 * (size_t)i])) { ++bad2; if (ordinary) ++bad2o; }
    if (!same(of[3 * (size_t)i + 2], of[3 * (size_t)i])) ++bad3;
    (void)nan0;
  }
  printf("REAL(4) quotients, %d operand pairs (3/4 random bit patterns): device '/' vs host '/': %ld differ; via frcp: %ld differ (%ld among the %ld ordinary pairs); via third-order rcp: %ld differ\n",
         n, badh, bad2, bad2o, nord, bad3);
  return 0;
}
