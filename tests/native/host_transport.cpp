// host_transport.cpp — TEST transport for carmel_hip_comm_create_custom: the three collectives staged through a POSIX
// shared-memory segment and summed on the host, rank by rank in a fixed order (every rank gets the same bits).  It exists
// so that the N > 1 paths of the library, the front end and bench.py can run -- and be tested -- with several ranks on ONE
// GPU (RCCL refuses two ranks on a device).  Not part of the product library; built into tests/native/ by tests/native/Makefile.
//
//   int carmel_hip_transport_open(const char* session, int rank, int world, int device, carmel_hip_transport* out);
//   int carmel_hip_transport_sendrecv(void* ctx, const carmel_hip_p2p* ops, uint32_t n_ops, void* stream);   (carmel_hip_comm_set_sendrecv)
//
// `session` names the segment (the same string on every rank; rank 0 creates it).  Every wait has a deadline: a rank
// that never arrives makes the others fail with an error instead of spinning for ever.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/carmel_hip.h"

namespace {
struct Ring {
  std::string name;
  int rank = 0, world = 1;
  size_t cap = 0;  // doubles per slot
  char* base = nullptr;
  size_t bytes = 0;
  uint64_t round = 0;
  double deadline_s = 120.0;
  std::vector<double> host;
  std::atomic<uint64_t>* arrive() { return (std::atomic<uint64_t>*)base; }
  std::atomic<uint64_t>* leave() { return (std::atomic<uint64_t>*)(base + 64); }
  std::atomic<uint64_t>* mapped() { return (std::atomic<uint64_t>*)(base + 128); }
  double* slot(int r) { return (double*)(base + 256) + (size_t)r * cap; }
  bool wait(std::atomic<uint64_t>* c, uint64_t target) {
    const auto t0 = std::chrono::steady_clock::now();
    while (c->load(std::memory_order_acquire) < target) {
      sched_yield();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadline_s) return false;
    }
    return true;
  }
  // everybody has written its slot / everybody has read what it needs
  bool barrier_in() {
    arrive()->fetch_add(1, std::memory_order_acq_rel);
    return wait(arrive(), (round + 1) * (uint64_t)world);
  }
  bool barrier_out() {
    leave()->fetch_add(1, std::memory_order_acq_rel);
    const bool ok = wait(leave(), (round + 1) * (uint64_t)world);
    ++round;
    return ok;
  }
};

#define HT_HIP(x)                                                                            \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "host transport: %s: %s\n", #x, hipGetErrorString(e_));           \
      return -2;                                                                             \
    }                                                                                        \
  } while (0)

int ht_allreduce_piece(Ring* R, double* dev, size_t n, int op, hipStream_t s) {
  HT_HIP(hipMemcpyAsync(R->slot(R->rank), dev, n * sizeof(double), hipMemcpyDeviceToHost, s));
  HT_HIP(hipStreamSynchronize(s));
  if (!R->barrier_in()) return -3;
  R->host.assign(R->slot(0), R->slot(0) + n);
  for (int r = 1; r < R->world; ++r) {  // the same order on every rank: bit-identical results
    const double* q = R->slot(r);
    if (op == 1)
      for (size_t k = 0; k < n; ++k) R->host[k] = q[k] > R->host[k] ? q[k] : R->host[k];
    else
      for (size_t k = 0; k < n; ++k) R->host[k] += q[k];
  }
  HT_HIP(hipMemcpyAsync(dev, R->host.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
  HT_HIP(hipStreamSynchronize(s));
  return R->barrier_out() ? 0 : -3;
}
int ht_allreduce(void* ctx, double* dev, uint64_t n, int op, void* stream) {
  Ring* R = (Ring*)ctx;
  for (uint64_t off = 0; off < n; off += R->cap) {
    const int rc = ht_allreduce_piece(R, dev + off, (size_t)std::min<uint64_t>(R->cap, n - off), op, (hipStream_t)stream);
    if (rc) return rc;
  }
  return 0;
}
int ht_reduce_scatter(void* ctx, double* buf, uint64_t count, void* stream) {
  Ring* R = (Ring*)ctx;
  hipStream_t s = (hipStream_t)stream;
  const uint64_t per = R->cap / (uint64_t)R->world;  // doubles of every rank's piece per round
  if (!per) return -1;
  for (uint64_t off = 0; off < count; off += per) {
    const size_t n = (size_t)std::min<uint64_t>(per, count - off);
    // slot layout of a round: piece q's stretch [off, off + n) at q * n
    for (int q = 0; q < R->world; ++q)
      HT_HIP(hipMemcpyAsync(R->slot(R->rank) + (size_t)q * n, buf + (size_t)q * count + off, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HT_HIP(hipStreamSynchronize(s));
    if (!R->barrier_in()) return -3;
    R->host.assign(R->slot(0) + (size_t)R->rank * n, R->slot(0) + (size_t)R->rank * n + n);
    for (int r = 1; r < R->world; ++r) {
      const double* q = R->slot(r) + (size_t)R->rank * n;
      for (size_t k = 0; k < n; ++k) R->host[k] += q[k];
    }
    HT_HIP(hipMemcpyAsync(buf + (size_t)R->rank * count + off, R->host.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
    HT_HIP(hipStreamSynchronize(s));
    if (!R->barrier_out()) return -3;
  }
  return 0;
}
int ht_all_gather(void* ctx, double* buf, uint64_t count, void* stream) {
  Ring* R = (Ring*)ctx;
  hipStream_t s = (hipStream_t)stream;
  for (uint64_t off = 0; off < count; off += R->cap) {
    const size_t n = (size_t)std::min<uint64_t>(R->cap, count - off);
    HT_HIP(hipMemcpyAsync(R->slot(R->rank), buf + (size_t)R->rank * count + off, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HT_HIP(hipStreamSynchronize(s));
    if (!R->barrier_in()) return -3;
    for (int r = 0; r < R->world; ++r)
      if (r != R->rank) HT_HIP(hipMemcpyAsync(buf + (size_t)r * count + off, R->slot(r), n * sizeof(double), hipMemcpyHostToDevice, s));
    HT_HIP(hipStreamSynchronize(s));
    if (!R->barrier_out()) return -3;
  }
  return 0;
}
// one group of sends and receives.  Rounds: every rank packs as much of its outgoing data as its slot holds as records
// [destination, n, n doubles] behind a header [records, more to come]; after the barrier every rank picks the records
// addressed to it out of the others' slots, in order, into its receives from that rank, in order.
int ht_sendrecv_impl(Ring* R, const carmel_hip_p2p* ops, uint32_t n_ops, hipStream_t s) {
  std::vector<uint32_t> sends;
  std::vector<std::vector<uint32_t>> recvs(R->world);
  for (uint32_t k = 0; k < n_ops; ++k) {
    if (ops[k].peer < 0 || ops[k].peer >= R->world || ops[k].peer == R->rank) return -1;
    if (ops[k].send)
      sends.push_back(k);
    else
      recvs[ops[k].peer].push_back(k);
  }
  size_t si = 0;
  uint64_t so = 0;  // next send op, doubles of it already out
  std::vector<size_t> ri(R->world, 0);
  std::vector<uint64_t> ro(R->world, 0);
  if (R->cap < 8) return -1;
  for (;;) {
    double* mine = R->slot(R->rank);
    size_t pos = 2, nrec = 0;
    while (si < sends.size() && pos + 3 <= R->cap) {
      const carmel_hip_p2p& o = ops[sends[si]];
      const uint64_t take = std::min<uint64_t>(o.n - so, R->cap - pos - 2);
      mine[pos] = (double)o.peer;
      mine[pos + 1] = (double)take;
      if (take) HT_HIP(hipMemcpyAsync(mine + pos + 2, o.dev_buf + so, take * sizeof(double), hipMemcpyDeviceToHost, s));
      pos += 2 + take;
      so += take;
      ++nrec;
      if (so == o.n) {
        ++si;
        so = 0;
      }
    }
    mine[0] = (double)nrec;
    mine[1] = si < sends.size() ? 1.0 : 0.0;
    HT_HIP(hipStreamSynchronize(s));
    if (!R->barrier_in()) return -3;
    bool more = false;
    for (int r = 0; r < R->world; ++r) {
      const double* q = R->slot(r);
      more = more || q[1] != 0.0;
      if (r == R->rank) continue;
      size_t p = 2;
      for (size_t k = 0, n = (size_t)q[0]; k < n; ++k) {
        const int dst = (int)q[p];
        uint64_t len = (uint64_t)q[p + 1];
        const double* data = q + p + 2;
        p += 2 + len;
        if (dst != R->rank) continue;
        while (len || (ri[r] < recvs[r].size() && ops[recvs[r][ri[r]]].n == 0)) {  // (a record never spans two sends)
          if (ri[r] >= recvs[r].size()) return -5;  // more arrives than this rank receives
          const carmel_hip_p2p& o = ops[recvs[r][ri[r]]];
          const uint64_t put = std::min<uint64_t>(len, o.n - ro[r]);
          if (put) HT_HIP(hipMemcpyAsync(o.dev_buf + ro[r], data, put * sizeof(double), hipMemcpyHostToDevice, s));
          if (put < len) return -5;  // lengths of a send and its receive differ
          data += put;
          len -= put;
          ro[r] += put;
          if (ro[r] == o.n) {
            ++ri[r];
            ro[r] = 0;
          }
        }
      }
    }
    HT_HIP(hipStreamSynchronize(s));
    if (!R->barrier_out()) return -3;
    if (!more) break;
  }
  for (int r = 0; r < R->world; ++r)
    if (ri[r] != recvs[r].size()) return -5;  // a receive nobody sent
  return 0;
}
void ht_destroy(void* ctx) {
  Ring* R = (Ring*)ctx;
  if (R->base) munmap(R->base, R->bytes);
  delete R;
}
}  // namespace

extern "C" int carmel_hip_transport_sendrecv(void* ctx, const carmel_hip_p2p* ops, uint32_t n_ops, void* stream) {
  return ht_sendrecv_impl((Ring*)ctx, ops, n_ops, (hipStream_t)stream);
}

extern "C" int carmel_hip_transport_open(const char* session, int rank, int world, int device, carmel_hip_transport* out) {
  if (!session || !out || world < 1 || rank < 0 || rank >= world) return -1;
  if (hipSetDevice(device) != hipSuccess) return -2;
  Ring* R = new Ring();
  R->name = std::string("/") + session;
  R->rank = rank;
  R->world = world;
  R->cap = (size_t)1 << 20;  // doubles per rank (8 MB)
  if (const char* e = getenv("CARMEL_HOST_TRANSPORT_CAP")) R->cap = (size_t)std::max(64, atoi(e));
  if (const char* e = getenv("CARMEL_HOST_TRANSPORT_DEADLINE")) R->deadline_s = atof(e);
  R->bytes = 256 + (size_t)world * R->cap * sizeof(double);
  int fd = -1;
  const auto t0 = std::chrono::steady_clock::now();
  auto late = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > R->deadline_s; };
  if (rank == 0) {
    fd = shm_open(R->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);  // a stale segment of the same name is an error, not reused
    if (fd < 0 || ftruncate(fd, (off_t)R->bytes) != 0) {
      if (fd >= 0) close(fd);
      delete R;
      return -4;
    }
  } else {
    while (fd < 0) {  // until rank 0 has created and sized the segment
      fd = shm_open(R->name.c_str(), O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < R->bytes)) {
        close(fd);
        fd = -1;
      }
      if (fd < 0) {
        if (late()) {
          delete R;
          return -3;
        }
        usleep(1000);
      }
    }
  }
  R->base = (char*)mmap(nullptr, R->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (R->base == (char*)MAP_FAILED) {
    R->base = nullptr;
    if (rank == 0) shm_unlink(R->name.c_str());
    delete R;
    return -4;
  }
  // once every rank has mapped the segment its name can go: nothing is left in /dev/shm after a crash or a kill
  R->mapped()->fetch_add(1, std::memory_order_acq_rel);
  if (!R->wait(R->mapped(), (uint64_t)world)) {
    if (rank == 0) shm_unlink(R->name.c_str());
    ht_destroy(R);
    return -3;
  }
  if (rank == 0) shm_unlink(R->name.c_str());
  out->ctx = R;
  out->allreduce = ht_allreduce;
  out->reduce_scatter = ht_reduce_scatter;
  out->all_gather = ht_all_gather;
  out->destroy = ht_destroy;
  out->name = "host-staged test transport (shared memory)";
  return 0;
}
