// grtrans_amd/csrc/grt_pl_pack.hpp -- host-side preparation of the polsynchpl tables for the device.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include "grt_types.h"

// h = [2020 table values][sgy 543][sya, syp, syv, syap, syav 5 x 243]: log-log slopes between neighbouring
// abscissae, computed once on upload (interp_gxp, polsynchemis.f90:82-92, recomputes them per call)
inline void grt_pl_pack_tables(const double *t, std::vector<double> &h) {
  h.assign(2020 + 543 + 5 * 243, 0.0);
  std::memcpy(h.data(), t, 2020 * sizeof(double));
  auto slopes = [&](int xo, int nx, int yo, int so) {
    for (int iy = 0; iy < 3; ++iy)
      for (int i = 0; i < nx; ++i) {
        int j = (i < nx - 1) ? i : nx - 2;
        h[so + iy * nx + i] = (h[yo + iy * nx + j + 1] - h[yo + iy * nx + j]) / (h[xo + j + 1] - h[xo + j]);
      }
  };
  slopes(0, 181, 181, 2020);
  const int yo[5] = {805, 1048, 1291, 1534, 1777};
  for (int k = 0; k < 5; ++k) slopes(724, 81, yo[k], 2020 + 543 + k * 243);
}

inline void grt_pl_point_tables(const double *d, GrtTables &T) {
  T.gx = d; T.gy = d + 181; T.xv = d + 724; T.ya = d + 805;
  T.yp = d + 1048; T.yv = d + 1291; T.yap = d + 1534; T.yav = d + 1777;
  T.sgy = d + 2020; T.sya = d + 2563; T.syp = d + 2806; T.syv = d + 3049;
  T.syap = d + 3292; T.syav = d + 3535;
}

// ray- and point-independent factors of polsynchpl (polsynchemis.f90:557-591)
inline void grt_pl_const(const double *tables2020, double p, double gmin, double gmax, GrtPLConst &plc) {
  const double al = (p - 1.0) / 2.0;
  plc.p = p; plc.gmin = gmin; plc.gmax = gmax; plc.alpha = al;
  plc.A0 = (p - 1.0) / (std::pow(gmin, 1.0 - p) - std::pow(gmax, 1.0 - p));
  plc.lmin = std::log(1.5 * gmin * gmin); plc.lmax = std::log(1.5 * gmax * gmax);
  plc.kq = std::pow(gmin, -2.0 * al + 1.0) / (al - 0.5);
  plc.kv = 2.0 * (al + 1.5) / (al + 1.0) * std::log(gmin) * std::pow(gmin, -2.0 * (al + 1.0));
  plc.l2g = 2.0 * std::log(gmin);
  plc.kqe = std::exp((al - 0.5) * (plc.l2g - std::log(1.5)));
  plc.gx0 = tables2020[0]; plc.gxs = 181.0 / (tables2020[180] - tables2020[0]);
  plc.xv0 = tables2020[724]; plc.xvs = 81.0 / (tables2020[804] - tables2020[724]);
  plc.iy = (p < 5.0) ? ((p > 3.2) ? 1 : 0) : 2;      // find_inds_gxp, polsynchemis.f90:74
}
