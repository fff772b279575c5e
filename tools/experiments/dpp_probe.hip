#include <hip/hip_runtime.h>
__device__ __forceinline__ double wave_up1(double x, double carry) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  int clo = __double2loint(carry), chi = __double2hiint(carry);
  lo = __builtin_amdgcn_update_dpp(clo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(chi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_up(double x, int ctrl) { return 0; }
template <int CTRL> __device__ __forceinline__ double dppz(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ void k(double *o, const double *in) {
  int lane = threadIdx.x;
  double x = in[lane];
  o[lane] = wave_up1(x, -1.0);
  // in-row suffix sums
  double s = x;
  s += dppz<0x101>(s);
  s += dppz<0x102>(s);
  s += dppz<0x104>(s);
  s += dppz<0x108>(s);
  o[64 + lane] = s;
}
int main() {
  double *d, h[128], in[64];
  for (int i = 0; i < 64; ++i) in[i] = i + 1;
  hipMalloc(&d, 192 * 8);
  hipMemcpy(d + 128, in, 64 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, d + 128);
  hipMemcpy(h, d, 128 * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%d: up1 %g rowsuffix %g\n", i, h[i], h[64 + i]);
  return 0;
}
