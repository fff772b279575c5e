// grtrans_amd/csrc/grt_gather.cpp -- multi-GPU assembly of the image (one process per GPU, RCCL over xGMI).
// The reference has no communication at all: pixels are independent and ivals is assembled in one process
// (pgrtrans.f90:221-242).  Here every rank traces its pixel ranges (the reference's own i1..i2 key, contiguous strips or
// interleaved column tiles), keeps its cameras in a rank-level device image and the only exchange is ONE gather of those
// fp32 images to the root: point-to-point sends inside RCCL groups (ragged per-rank sizes, no padding, no all-reduce on
// the data path).  The communicator is owned by the library (grt_comm_init*) or by the caller (grt_gather).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "grt_types.h"

extern "C" int grt_resident_pointers(float **d_pixvals, int32_t **d_ray_status);

namespace {
struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  hipStream_t stream = nullptr;
  // rank-level image [ncams][npix_rank][nvals]
  float *d_stage = nullptr;
  size_t cap_stage = 0;
  int ncams = 0, nvals = 0;
  long long npix_rank = 0;
  float *d_full = nullptr;        // root: [ncams][sum npix][nvals]
  size_t cap_full = 0;
  double *d_red = nullptr;
};
Comm M;
int nccl_rc(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }
}  // namespace

extern "C" {

int grt_comm_unique_id(char *id128) {
  if (!id128) return GRT_E_BADARG;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return nccl_rc(r);
  std::memcpy(id128, &id, 128);
  return 0;
}

int grt_comm_free(void) {
  if (M.stream) (void)hipStreamSynchronize(M.stream);
  if (M.comm) (void)ncclCommDestroy(M.comm);
  if (M.d_stage) (void)hipFree(M.d_stage);
  if (M.d_full) (void)hipFree(M.d_full);
  if (M.d_red) (void)hipFree(M.d_red);
  if (M.stream) (void)hipStreamDestroy(M.stream);
  M = Comm();
  return 0;
}

// ncclCommInitRank blocks until every rank has joined and has no timeout of its own: a rank that never arrives (crashed before
// the rendezvous, stale id) would hang the host silently.  The call stays on the caller's thread (the communicator belongs to
// it); a watchdog thread ends the PROCESS with exit code 98 and a message if the call has not returned within timeout_s -- a
// rank stuck in the rendezvous cannot be recovered, and every launcher of this package treats a non-zero exit as the job's failure.
static int comm_init_timed(int rank, int nranks, const ncclUniqueId &id, double timeout_s) {
  struct Box { std::mutex m; std::condition_variable cv; bool done = false; };
  auto box = std::make_shared<Box>();
  std::thread dog;
  if (timeout_s > 0)
    dog = std::thread([box, rank, nranks, timeout_s]() {
      std::unique_lock<std::mutex> lk(box->m);
      if (!box->cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return box->done; })) {
        fprintf(stderr, "grtrans_amd: ncclCommInitRank (rank %d of %d) did not return within %.0f s: a rank is missing or the id is stale; "
                        "ending this process (exit code 98)\n", rank, nranks, timeout_s);
        fflush(stderr);
        _exit(98);
      }
    });
  ncclComm_t c = nullptr;
  const ncclResult_t r = ncclCommInitRank(&c, nranks, id, rank);
  {
    std::lock_guard<std::mutex> g(box->m);
    box->done = true;
  }
  box->cv.notify_all();
  if (dog.joinable()) dog.join();
  if (r != ncclSuccess) return nccl_rc(r);
  M.comm = c;
  return 0;
}

static int comm_init_impl(int rank, int nranks, const char *id128, double timeout_s) {
  if (nranks < 1 || rank < 0 || rank >= nranks || !id128) return GRT_E_BADARG;
  if (M.comm) { (void)ncclCommDestroy(M.comm); M.comm = nullptr; }       // (the staging buffers and the stream are kept)
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  if (!M.stream && hipStreamCreateWithFlags(&M.stream, hipStreamNonBlocking) != hipSuccess) return GRT_E_NODEVICE;
  int rc = comm_init_timed(rank, nranks, id, timeout_s);
  if (rc) { M.comm = nullptr; return rc; }
  M.rank = rank; M.nranks = nranks;
  return 0;
}

// the calling process has selected its device already (grt_init); every rank calls with the same 128 id bytes
int grt_comm_init(int rank, int nranks, const char *id128) {
  const char *e = std::getenv("GRT_COMM_INIT_TIMEOUT");
  return comm_init_impl(rank, nranks, id128, e ? std::atof(e) : 600.0);
}

// Rendezvous through a file for hosts that have no launcher-provided channel (the Fortran executable): rank 0 removes whatever
// sits at `path`, writes [8-byte magic][64-byte run nonce][128-byte id] to `path`.tmp and renames it; the other ranks poll `path`
// and accept it only when the nonce is THEIR run's (a file left by a crashed run is ignored, not joined).  The nonce is
// GRT_COMM_NONCE when the launcher sets it, else the parent process id (ranks of one launcher share a parent).  Rank 0 unlinks
// the file once ncclCommInitRank has returned: every rank has joined, so every rank has read it.
int grt_comm_init_file(int rank, int nranks, const char *path, double timeout_s) {
  if (!path || !*path || nranks < 1 || rank < 0 || rank >= nranks) return GRT_E_BADARG;
  char rec[8 + 64 + 128];
  char want[8 + 64];
  std::memset(want, 0, sizeof want);
  std::memcpy(want, "GRTCOMM2", 8);
  {
    const char *nz = std::getenv("GRT_COMM_NONCE");
    char buf[64];
    if (!nz || !*nz) { std::snprintf(buf, sizeof buf, "ppid=%ld", (long)getppid()); nz = buf; }
    std::strncpy(want + 8, nz, 63);
  }
  char id[128];
  if (rank == 0) {
    int rc = grt_comm_unique_id(id);
    if (rc) return rc;
    (void)std::remove(path);                     // never leave a previous run's id where the other ranks could find it
    std::memcpy(rec, want, sizeof want);
    std::memcpy(rec + sizeof want, id, 128);
    std::string tmp = std::string(path) + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return GRT_E_IO;
    const bool ok = std::fwrite(rec, 1, sizeof rec, f) == sizeof rec;
    std::fclose(f);
    if (!ok || std::rename(tmp.c_str(), path) != 0) return GRT_E_IO;
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      FILE *f = std::fopen(path, "rb");
      if (f) {
        const size_t n = std::fread(rec, 1, sizeof rec, f);
        std::fclose(f);
        if (n == sizeof rec && std::memcmp(rec, want, sizeof want) == 0) { std::memcpy(id, rec + sizeof want, 128); break; }
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
        fprintf(stderr, "grtrans_amd: rank %d found no rendezvous file of this run at %s within %.0f s\n", rank, path, timeout_s);
        return GRT_E_IO;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  const int rc = comm_init_impl(rank, nranks, id, timeout_s);
  if (rank == 0) (void)std::remove(path);
  return rc;
}

int grt_comm_rank(int *rank, int *nranks) {
  if (rank) *rank = M.rank;
  if (nranks) *nranks = M.nranks;
  return M.comm ? 0 : GRT_E_NOTINIT;
}

// the library communicator and its stream, for callers that drive grt_gather themselves
int grt_comm_handle(void **comm, void **stream) {
  if (!M.comm) return GRT_E_NOTINIT;
  if (comm) *comm = (void *)M.comm;
  if (stream) *stream = (void *)M.stream;
  return 0;
}

// sum (op 0) or max (op 1) of n doubles over the ranks, in place on the host: run statistics only (ray-steps, times)
int grt_comm_allreduce(double *vals, int n, int op) {
  if (!M.comm) return GRT_E_NOTINIT;
  if (!vals || n < 1 || n > 64) return GRT_E_BADARG;
  if (!M.d_red && hipMalloc((void **)&M.d_red, 64 * sizeof(double)) != hipSuccess) return GRT_E_NODEVICE;
  if (hipMemcpyAsync(M.d_red, vals, n * sizeof(double), hipMemcpyHostToDevice, M.stream) != hipSuccess) return GRT_E_NODEVICE;
  ncclResult_t r = ncclAllReduce(M.d_red, M.d_red, (size_t)n, ncclDouble, op == 1 ? ncclMax : ncclSum, M.comm, M.stream);
  if (r != ncclSuccess) return nccl_rc(r);
  if (hipMemcpyAsync(vals, M.d_red, n * sizeof(double), hipMemcpyDeviceToHost, M.stream) != hipSuccess) return GRT_E_NODEVICE;
  return hipStreamSynchronize(M.stream) == hipSuccess ? 0 : GRT_E_NODEVICE;
}

// rank-level device image [ncams][npix_rank][nvals], zeroed (camera.f90:60-62: pixels never traced stay zero)
int grt_stage_reset(int ncams, int nvals, int64_t npix_rank) {
  if (ncams < 1 || nvals < 1 || npix_rank < 0) return GRT_E_BADARG;
  if (!M.stream && hipStreamCreateWithFlags(&M.stream, hipStreamNonBlocking) != hipSuccess) return GRT_E_NODEVICE;
  const size_t need = (size_t)ncams * (size_t)npix_rank * nvals;
  if (need > M.cap_stage || !M.d_stage) {
    if (M.d_stage) (void)hipFree(M.d_stage);
    M.d_stage = nullptr; M.cap_stage = 0;
    if (hipMalloc((void **)&M.d_stage, (need ? need : 1) * sizeof(float)) != hipSuccess) return GRT_E_NODEVICE;
    M.cap_stage = need;
  }
  M.ncams = ncams; M.nvals = nvals; M.npix_rank = npix_rank;
  if (need && hipMemsetAsync(M.d_stage, 0, need * sizeof(float), M.stream) != hipSuccess) return GRT_E_NODEVICE;
  return hipStreamSynchronize(M.stream) == hipSuccess ? 0 : GRT_E_NODEVICE;
}

// the tile of the last grt_trace_tile[_resident] call, [nfreq][npix_tile][nvals] in HBM, into cameras cam0 .. cam0+nfreq-1 of
// the rank-level image at pixel offset pix_offset (device-to-device)
int grt_stage_resident(int cam0, int nfreq, int64_t npix_tile, int64_t pix_offset) {
  float *src = nullptr;
  int rc = grt_resident_pointers(&src, nullptr);
  if (rc) return rc;
  if (!M.d_stage || !src) return GRT_E_NOTINIT;
  if (cam0 < 0 || nfreq < 1 || cam0 + nfreq > M.ncams || pix_offset < 0 || npix_tile < 0 || pix_offset + npix_tile > M.npix_rank) return GRT_E_BADARG;
  const size_t row = (size_t)npix_tile * M.nvals * sizeof(float);
  if (row == 0) return 0;
  if (hipMemcpy2DAsync(M.d_stage + ((size_t)cam0 * M.npix_rank + pix_offset) * M.nvals, (size_t)M.npix_rank * M.nvals * sizeof(float), src, row, row,
                       (size_t)nfreq, hipMemcpyDeviceToDevice, M.stream) != hipSuccess)
    return GRT_E_BADARG;
  return hipStreamSynchronize(M.stream) == hipSuccess ? 0 : GRT_E_NODEVICE;
}

// Caller-owned communicator and buffers: every rank's flat tile to `root`, rank after rank.  (A tile is
// [ncams][npix_tile][nvals]: for ncams > 1 the concatenation is rank-major, not an image -- grt_gather_stage is the
// per-camera form.)  The group is closed on every exit path.
int grt_gather(const float *d_tile, int64_t count, float *d_full, const int64_t *counts, int nranks, int rank, int root,
               void *comm_, void *stream_) {
  if (!comm_ || !counts || nranks < 1 || rank < 0 || rank >= nranks || root < 0 || root >= nranks) return GRT_E_BADARG;
  ncclComm_t comm = (ncclComm_t)comm_;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = 0;
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return nccl_rc(r);
  if (rank == root) {
    int64_t off = 0;
    for (int p = 0; p < nranks && rc == 0; ++p) {
      if (p == root) {
        if (counts[p] > 0 && d_full + off != d_tile &&
            hipMemcpyAsync(d_full + off, d_tile, (size_t)counts[p] * sizeof(float), hipMemcpyDeviceToDevice, stream) != hipSuccess)
          rc = GRT_E_BADARG;
      } else if (counts[p] > 0) {
        rc = nccl_rc(ncclRecv(d_full + off, (size_t)counts[p], ncclFloat, p, comm, stream));
      }
      off += counts[p];
    }
  } else if (count > 0) {
    rc = nccl_rc(ncclSend(d_tile, (size_t)count, ncclFloat, root, comm, stream));
  }
  r = ncclGroupEnd();      // always: an open group would swallow the thread's next RCCL call
  return rc ? rc : nccl_rc(r);
}

// Library communicator: the rank-level images to the root, camera by camera.  npix_of_rank[nranks]; on the root host_full
// receives [ncams][sum npix][nvals] with a camera's pixels in rank order (rank 0's, rank 1's, ...); other ranks pass NULL.
// One RCCL group per camera (nranks - 1 receives on the root, one send elsewhere).
int grt_gather_stage(const int64_t *npix_of_rank, int root, float *host_full) {
  if (!npix_of_rank || root < 0 || root >= M.nranks) return GRT_E_BADARG;
  if (!M.d_stage) return GRT_E_NOTINIT;
  if (M.nranks > 1 && !M.comm) return GRT_E_NOTINIT;
  if (npix_of_rank[M.rank] != M.npix_rank) return GRT_E_BADARG;
  long long total = 0;
  std::vector<long long> pre((size_t)M.nranks);
  for (int p = 0; p < M.nranks; ++p) { pre[p] = total; total += npix_of_rank[p]; }
  const size_t nv = (size_t)M.nvals;
  if (M.rank == root) {
    if (!host_full) return GRT_E_BADARG;
    const size_t need = (size_t)M.ncams * (size_t)total * nv;
    if (need > M.cap_full || !M.d_full) {
      if (M.d_full) (void)hipFree(M.d_full);
      M.d_full = nullptr; M.cap_full = 0;
      if (hipMalloc((void **)&M.d_full, (need ? need : 1) * sizeof(float)) != hipSuccess) return GRT_E_NODEVICE;
      M.cap_full = need;
    }
  }
  for (int c = 0; c < M.ncams; ++c) {
    const float *mine = M.d_stage + (size_t)c * M.npix_rank * nv;
    const int64_t cnt = (int64_t)M.npix_rank * (int64_t)nv;
    if (M.nranks == 1) {
      if (cnt && hipMemcpyAsync(M.d_full + (size_t)c * total * nv, mine, (size_t)cnt * sizeof(float), hipMemcpyDeviceToDevice, M.stream) != hipSuccess)
        return GRT_E_NODEVICE;
      continue;
    }
    std::vector<int64_t> counts((size_t)M.nranks);
    for (int p = 0; p < M.nranks; ++p) counts[p] = npix_of_rank[p] * (int64_t)nv;
    int rc = grt_gather(mine, cnt, M.rank == root ? M.d_full + (size_t)c * total * nv : nullptr, counts.data(), M.nranks, M.rank, root,
                        (void *)M.comm, (void *)M.stream);
    if (rc) return rc;
  }
  if (M.rank == root && (size_t)M.ncams * total * nv > 0 &&
      hipMemcpyAsync(host_full, M.d_full, (size_t)M.ncams * (size_t)total * nv * sizeof(float), hipMemcpyDeviceToHost, M.stream) != hipSuccess)
    return GRT_E_NODEVICE;
  return hipStreamSynchronize(M.stream) == hipSuccess ? 0 : GRT_E_NODEVICE;
}

}  // extern "C"
