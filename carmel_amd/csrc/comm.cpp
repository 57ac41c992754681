// comm.cpp — corpus-sharded EM across the GPUs of one node: the per-iteration all-reduce (sum) of the arc count vector
// counts[n_arcs + 4] over RCCL / xGMI, enqueued on the trainer's own stream between the count pass and the M-step, so
// that an iteration is   estimate_async -> allreduce_counts -> maximize   with no host synchronisation in between.
//
// The reference has no counterpart (it is single-process); what is exchanged is what forward_backward::estimate leaves
// in arc_counts::counts plus the corpus scalars of train.cc:326-332.  RCCL is looked up at run time
// (dlopen) so that single-GPU users -- and the `carmel` binary on a box without RCCL -- never need it, and so that a
// process that already has RCCL loaded (PyTorch) shares that instance.
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cstring>
#include <string>
#include "engine.hpp"

namespace {
// the few RCCL entry points used (rccl.h:40-43, 187, 220, 448-467, 611)
typedef struct {
  char internal[128];
} rccl_unique_id;
typedef void* rccl_comm_t;
typedef int (*fn_get_unique_id)(rccl_unique_id*);
typedef int (*fn_comm_init_rank)(rccl_comm_t*, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(rccl_comm_t);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, rccl_comm_t, hipStream_t);
typedef const char* (*fn_error_string)(int);
enum { RCCL_SUM = 0, RCCL_MAX = 2, RCCL_FLOAT64 = 8 };
struct Rccl {
  void* h = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_error_string error_string = nullptr;
  std::string err;
  bool load() {
    if (h) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
      err = std::string("RCCL not found: ") + dlerror();
      return false;
    }
    get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
    comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
    comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
    error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
    if (!get_unique_id || !comm_init_rank || !comm_destroy || !all_reduce) {
      err = "RCCL lacks an expected symbol";
      h = nullptr;
      return false;
    }
    return true;
  }
  std::string what(int rc) { return error_string ? error_string(rc) : std::to_string(rc); }
};
Rccl g_rccl;
}  // namespace

// Host-staged transport (CARMEL_HIP_COMM=host when the id is made): every rank copies its vector to a shared-memory
// slot, waits for the others, and sums all slots ON ITS GPU.  RCCL refuses two ranks on one device, so this is what lets
// the N > 1 path of the front end run -- and be tested -- on a box with a single GPU; it is not a production transport.
struct HostRing {
  std::string name;
  size_t cap = 0;  // doubles per slot
  char* base = nullptr;
  size_t bytes = 0;
  std::atomic<uint64_t>* arrive() { return (std::atomic<uint64_t>*)base; }
  std::atomic<uint64_t>* leave() { return (std::atomic<uint64_t>*)(base + 64); }
  double* slot(int r) { return (double*)(base + 128) + (size_t)r * cap; }
};

struct carmel_hip_comm {
  rccl_comm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  DevBuf<double> scratch;  // small host-vector reductions
  bool host = false;
  HostRing ring;
  uint64_t round = 0;
  std::vector<double> stage;
  DevBuf<double> dstage;
  ~carmel_hip_comm() {
    if (ring.base) munmap(ring.base, ring.bytes);
    if (host && rank == 0 && !ring.name.empty()) shm_unlink(ring.name.c_str());
  }
};

static const size_t HOST_RING_CAP = 1u << 21;  // doubles per rank (16 MB): the host transport is for small models

// all ranks: v[0..n) := sum over ranks (device pointer), on stream s, which is synchronised
static int host_allreduce(carmel_hip_comm* c, double* dev, size_t n, hipStream_t s, bool op_max) {
  if (n > c->ring.cap) return fail(CARMEL_HIP_ERR_UNSUPPORTED, "host-staged transport: vector too long (use RCCL)");
  HostRing& R = c->ring;
  HIPCHK(hipMemcpyAsync(R.slot(c->rank), dev, n * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const uint64_t target = (c->round + 1) * (uint64_t)c->world;
  R.arrive()->fetch_add(1, std::memory_order_acq_rel);
  while (R.arrive()->load(std::memory_order_acquire) < target) sched_yield();
  if (c->dstage.n < n) HIPCHK(c->dstage.alloc(n));
  if (op_max) {  // a handful of scalars: on the host
    c->stage.assign(R.slot(0), R.slot(0) + n);
    for (int r = 1; r < c->world; ++r)
      for (size_t k = 0; k < n; ++k) c->stage[k] = std::max(c->stage[k], R.slot(r)[k]);
    HIPCHK(hipMemcpyAsync(dev, c->stage.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
  } else {
    HIPCHK(hipMemcpyAsync(dev, R.slot(0), n * sizeof(double), hipMemcpyHostToDevice, s));
    for (int r = 1; r < c->world; ++r) {  // the same order on every rank: bit-identical sums
      HIPCHK(hipMemcpyAsync(c->dstage.p, R.slot(r), n * sizeof(double), hipMemcpyHostToDevice, s));
      HIPCHK(launch_add(dev, c->dstage.p, n, s));
    }
  }
  HIPCHK(hipStreamSynchronize(s));
  R.leave()->fetch_add(1, std::memory_order_acq_rel);  // nobody overwrites a slot before everybody has read it
  while (R.leave()->load(std::memory_order_acquire) < target) sched_yield();
  ++c->round;
  return CARMEL_HIP_OK;
}

extern "C" {

int carmel_hip_comm_unique_id(void* id128) {
  if (!id128) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (const char* e = getenv("CARMEL_HIP_COMM"))
    if (std::string(e) == "host") {  // the id names a shared-memory segment
      std::memset(id128, 0, 128);
      std::snprintf((char*)id128, 128, "HOST/carmel_hip_%d_%ld", (int)getpid(), (long)time(nullptr));
      return CARMEL_HIP_OK;
    }
  if (!g_rccl.load()) return fail(CARMEL_HIP_ERR_UNSUPPORTED, g_rccl.err);
  rccl_unique_id id;
  const int rc = g_rccl.get_unique_id(&id);
  if (rc) return fail(CARMEL_HIP_ERR_HIP, "ncclGetUniqueId: " + g_rccl.what(rc));
  std::memcpy(id128, id.internal, 128);
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_create(carmel_hip_comm** out, int device, int rank, int world, const void* id128) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(CARMEL_HIP_ERR_ARG, "bad argument");
  if (std::memcmp(id128, "HOST/", 5) == 0) {
    HIPCHK(hipSetDevice(device));
    carmel_hip_comm* c = new carmel_hip_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->host = true;
    c->ring.name = std::string("/") + ((const char*)id128 + 5);
    c->ring.cap = HOST_RING_CAP;
    c->ring.bytes = 128 + (size_t)world * HOST_RING_CAP * sizeof(double);
    int fd = -1;
    for (int tries = 0; tries < 20000 && fd < 0; ++tries) {  // rank 0 creates, the others wait for it
      fd = rank == 0 ? shm_open(c->ring.name.c_str(), O_CREAT | O_RDWR, 0600) : shm_open(c->ring.name.c_str(), O_RDWR, 0600);
      if (fd < 0) usleep(1000);
    }
    if (fd < 0 || (rank == 0 && ftruncate(fd, (off_t)c->ring.bytes) != 0)) {
      delete c;
      return fail(CARMEL_HIP_ERR_HIP, "host-staged transport: shared memory unavailable");
    }
    if (rank != 0) {  // until rank 0 has sized the segment
      struct stat st;
      for (int tries = 0; tries < 20000; ++tries) {
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->ring.bytes) break;
        usleep(1000);
      }
    }
    c->ring.base = (char*)mmap(nullptr, c->ring.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->ring.base == (char*)MAP_FAILED) {
      c->ring.base = nullptr;
      delete c;
      return fail(CARMEL_HIP_ERR_HIP, "host-staged transport: mmap failed");
    }
    *out = c;
    return CARMEL_HIP_OK;
  }
  if (!g_rccl.load()) return fail(CARMEL_HIP_ERR_UNSUPPORTED, g_rccl.err);
  HIPCHK(hipSetDevice(device));
  rccl_unique_id id;
  std::memcpy(id.internal, id128, 128);
  carmel_hip_comm* c = new carmel_hip_comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  const int rc = g_rccl.comm_init_rank(&c->comm, world, id, rank);
  if (rc) {
    delete c;
    return fail(CARMEL_HIP_ERR_HIP, "ncclCommInitRank: " + g_rccl.what(rc));
  }
  *out = c;
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_destroy(carmel_hip_comm* c) {
  if (!c) return CARMEL_HIP_OK;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)g_rccl.comm_destroy(c->comm);
  delete c;
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_rank(carmel_hip_comm* c) { return c ? c->rank : 0; }
int carmel_hip_comm_world(carmel_hip_comm* c) { return c ? c->world : 1; }

int carmel_hip_allreduce_counts(carmel_hip_trainer* t, carmel_hip_comm* c) {
  if (!t || !c) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (t->device != c->device) return fail(CARMEL_HIP_ERR_ARG, "trainer and communicator live on different devices");
  HIPCHK(hipSetDevice(t->device));
  if (c->host) return host_allreduce(c, t->counts_ptr(), t->w.n_arcs + 4, t->stream, false);
  // the unrolled cascade sweep keeps per-PARAMETER sums in the same buffer: its first u_n_slots entries are what counts
  // there (everything is a sum over pairs either way, so the reduction is the same plain sum); the scalars follow at
  // n_arcs.  Reducing the whole buffer keeps one collective per iteration.
  const int rc = g_rccl.all_reduce(t->counts_ptr(), t->counts_ptr(), t->w.n_arcs + 4, RCCL_FLOAT64, RCCL_SUM, c->comm, t->stream);
  if (rc) return fail(CARMEL_HIP_ERR_HIP, "ncclAllReduce: " + g_rccl.what(rc));
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_allreduce_host(carmel_hip_comm* c, double* v, uint32_t n, int op_max) {
  if (!c || !v) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->scratch.n < n) HIPCHK(c->scratch.alloc(n));
  HIPCHK(hipMemcpy(c->scratch.p, v, n * sizeof(double), hipMemcpyHostToDevice));
  if (c->host) {
    int rc = host_allreduce(c, c->scratch.p, n, nullptr, op_max != 0);
    if (rc) return rc;
    HIPCHK(hipMemcpy(v, c->scratch.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return CARMEL_HIP_OK;
  }
  const int rc = g_rccl.all_reduce(c->scratch.p, c->scratch.p, n, RCCL_FLOAT64, op_max ? RCCL_MAX : RCCL_SUM, c->comm, nullptr);
  if (rc) return fail(CARMEL_HIP_ERR_HIP, "ncclAllReduce: " + g_rccl.what(rc));
  HIPCHK(hipStreamSynchronize(nullptr));
  HIPCHK(hipMemcpy(v, c->scratch.p, n * sizeof(double), hipMemcpyDeviceToHost));
  return CARMEL_HIP_OK;
}

}  // extern "C"
