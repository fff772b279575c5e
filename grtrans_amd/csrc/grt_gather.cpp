// grtrans_amd/csrc/grt_gather.cpp -- multi-GPU assembly of the image.
// The reference has no communication at all: pixels are independent and ivals is
// assembled in one process (pgrtrans.f90:221-242).  With one process per GPU each rank traces a
// contiguous i1..i2 strip; the only exchange is this gather of the fp32 pixel tiles to the root
// over xGMI (RCCL point-to-point inside one group: ragged tiles, no padding, no all-reduce).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "grt_types.h"

extern "C" int grt_gather(const float *d_tile, int64_t count, float *d_full, const int64_t *counts, int nranks,
                          int rank, int root, void *comm_, void *stream_) {
  if (!comm_ || !counts || nranks < 1 || rank < 0 || rank >= nranks) return GRT_E_BADARG;
  ncclComm_t comm = (ncclComm_t)comm_;
  hipStream_t stream = (hipStream_t)stream_;
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return 1000 + (int)r;
  if (rank == root) {
    int64_t off = 0;
    for (int p = 0; p < nranks; ++p) {
      if (p == root) {
        if (d_full + off != d_tile)
          if (hipMemcpyAsync(d_full + off, d_tile, counts[p] * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return GRT_E_BADARG;
      } else {
        r = ncclRecv(d_full + off, (size_t)counts[p], ncclFloat, p, comm, stream);
        if (r != ncclSuccess) return 1000 + (int)r;
      }
      off += counts[p];
    }
  } else {
    r = ncclSend(d_tile, (size_t)count, ncclFloat, root, comm, stream);
    if (r != ncclSuccess) return 1000 + (int)r;
  }
  r = ncclGroupEnd();
  return r == ncclSuccess ? 0 : 1000 + (int)r;
}
