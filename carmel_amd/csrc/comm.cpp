// comm.cpp — corpus-sharded EM across the GPUs of one node: the communicator (RCCL over xGMI, or a caller-supplied
// transport) and its three collectives on f64 device buffers.  What is exchanged with them, and when, is exchange.cpp.
//
// The reference has no counterpart (it is single-process); what is exchanged is what forward_backward::estimate leaves
// in arc_counts::counts plus the corpus scalars of train.cc:326-332.  RCCL's interface comes from
// <rccl/rccl.h>; the library itself is looked up at run time (dlopen) so that single-GPU users -- and the `carmel`
// binary on a box without RCCL -- never need it, and so that a process that already has RCCL loaded (PyTorch) shares that
// instance.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstring>
#include <string>
#include "comm.hpp"

namespace {
struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclCommAbort) comm_abort = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclReduceScatter) reduce_scatter = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
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
    get_unique_id = (decltype(get_unique_id))dlsym(h, "ncclGetUniqueId");
    comm_init_rank = (decltype(comm_init_rank))dlsym(h, "ncclCommInitRank");
    comm_destroy = (decltype(comm_destroy))dlsym(h, "ncclCommDestroy");
    comm_abort = (decltype(comm_abort))dlsym(h, "ncclCommAbort");
    all_reduce = (decltype(all_reduce))dlsym(h, "ncclAllReduce");
    reduce_scatter = (decltype(reduce_scatter))dlsym(h, "ncclReduceScatter");
    all_gather = (decltype(all_gather))dlsym(h, "ncclAllGather");
    error_string = (decltype(error_string))dlsym(h, "ncclGetErrorString");
    send = (decltype(send))dlsym(h, "ncclSend");
    recv = (decltype(recv))dlsym(h, "ncclRecv");
    group_start = (decltype(group_start))dlsym(h, "ncclGroupStart");
    group_end = (decltype(group_end))dlsym(h, "ncclGroupEnd");
    if (!get_unique_id || !comm_init_rank || !comm_destroy || !all_reduce || !reduce_scatter || !all_gather) {
      err = "RCCL lacks an expected symbol";
      h = nullptr;
      return false;
    }
    return true;
  }
  std::string what(ncclResult_t rc) { return error_string ? error_string(rc) : std::to_string((int)rc); }
};
Rccl g_rccl;
static_assert(sizeof(ncclUniqueId) == 128, "the 128-byte communicator id of the C-ABI is ncclUniqueId");
}  // namespace

carmel_hip_comm::~carmel_hip_comm() {
  if (custom && tr.destroy) tr.destroy(tr.ctx);
  if (xstream) (void)hipStreamDestroy(xstream);
}

int comm_allreduce(carmel_hip_comm* c, double* dev, size_t n, bool op_max, hipStream_t s) {
  if (!n) return CARMEL_HIP_OK;
  if (c->custom) {
    const int rc = c->tr.allreduce(c->tr.ctx, dev, n, op_max ? 1 : 0, (void*)s);
    return rc ? fail(CARMEL_HIP_ERR_HIP, c->what + ": allreduce failed (" + std::to_string(rc) + ")") : CARMEL_HIP_OK;
  }
  const ncclResult_t rc = g_rccl.all_reduce(dev, dev, n, ncclDouble, op_max ? ncclMax : ncclSum, (ncclComm_t)c->rccl, s);
  return rc ? fail(CARMEL_HIP_ERR_HIP, "ncclAllReduce: " + g_rccl.what(rc)) : CARMEL_HIP_OK;
}

int comm_reduce_scatter(carmel_hip_comm* c, double* buf, size_t count, hipStream_t s) {
  if (!count) return CARMEL_HIP_OK;
  if (c->custom) {
    if (!c->tr.reduce_scatter) return comm_allreduce(c, buf, count * (size_t)c->world, false, s);  // more traffic, same sums
    const int rc = c->tr.reduce_scatter(c->tr.ctx, buf, count, (void*)s);
    return rc ? fail(CARMEL_HIP_ERR_HIP, c->what + ": reduce_scatter failed (" + std::to_string(rc) + ")") : CARMEL_HIP_OK;
  }
  // in place: the receive buffer is this rank's piece of the send buffer
  const ncclResult_t rc = g_rccl.reduce_scatter(buf, buf + (size_t)c->rank * count, count, ncclDouble, ncclSum, (ncclComm_t)c->rccl, s);
  return rc ? fail(CARMEL_HIP_ERR_HIP, "ncclReduceScatter: " + g_rccl.what(rc)) : CARMEL_HIP_OK;
}

int comm_all_gather(carmel_hip_comm* c, double* buf, size_t count, hipStream_t s) {
  if (!count) return CARMEL_HIP_OK;
  if (c->custom) {
    if (!c->tr.all_gather) {  // the others' pieces set to -0.0, then a sum: x + -0.0 == x bit for bit, for -0.0 and +0.0 too
      if (c->rank > 0) HIPCHK(launch_fill(buf, -0.0, (uint64_t)c->rank * count, s));
      if (c->rank + 1 < c->world)
        HIPCHK(launch_fill(buf + (size_t)(c->rank + 1) * count, -0.0, (uint64_t)(c->world - 1 - c->rank) * count, s));
      return comm_allreduce(c, buf, count * (size_t)c->world, false, s);
    }
    const int rc = c->tr.all_gather(c->tr.ctx, buf, count, (void*)s);
    return rc ? fail(CARMEL_HIP_ERR_HIP, c->what + ": all_gather failed (" + std::to_string(rc) + ")") : CARMEL_HIP_OK;
  }
  const ncclResult_t rc = g_rccl.all_gather(buf + (size_t)c->rank * count, buf, count, ncclDouble, (ncclComm_t)c->rccl, s);
  return rc ? fail(CARMEL_HIP_ERR_HIP, "ncclAllGather: " + g_rccl.what(rc)) : CARMEL_HIP_OK;
}

bool comm_has_p2p(const carmel_hip_comm* c) {
  return c->custom ? c->tr_sendrecv != nullptr : (g_rccl.send && g_rccl.recv && g_rccl.group_start && g_rccl.group_end);
}

// the exchange's direct form: every transfer of one chunk in one group (RCCL fuses a group's sends and receives into one launch)
int comm_p2p(carmel_hip_comm* c, const carmel_hip_p2p* ops, uint32_t n_ops, hipStream_t s) {
  if (!n_ops) return CARMEL_HIP_OK;
  if (c->custom) {
    if (!c->tr_sendrecv) return fail(CARMEL_HIP_ERR_UNSUPPORTED, c->what + ": no point-to-point transfers");
    const int rc = c->tr_sendrecv(c->tr.ctx, ops, n_ops, (void*)s);
    return rc ? fail(CARMEL_HIP_ERR_HIP, c->what + ": sendrecv failed (" + std::to_string(rc) + ")") : CARMEL_HIP_OK;
  }
  ncclResult_t rc = g_rccl.group_start();
  if (rc) return fail(CARMEL_HIP_ERR_HIP, "ncclGroupStart: " + g_rccl.what(rc));
  ncclResult_t bad = ncclSuccess;
  for (uint32_t k = 0; k < n_ops && !bad; ++k) {
    const carmel_hip_p2p& o = ops[k];
    bad = o.send ? g_rccl.send(o.dev_buf, o.n, ncclDouble, o.peer, (ncclComm_t)c->rccl, s)
                 : g_rccl.recv(o.dev_buf, o.n, ncclDouble, o.peer, (ncclComm_t)c->rccl, s);
  }
  rc = g_rccl.group_end();  // (always: an open group would swallow every later call)
  if (bad) return fail(CARMEL_HIP_ERR_HIP, "ncclSend / ncclRecv: " + g_rccl.what(bad));
  return rc ? fail(CARMEL_HIP_ERR_HIP, "ncclGroupEnd: " + g_rccl.what(rc)) : CARMEL_HIP_OK;
}

extern "C" {

int carmel_hip_comm_set_sendrecv(carmel_hip_comm* c, carmel_hip_sendrecv_fn fn) {
  if (!c) return fail(CARMEL_HIP_ERR_ARG, "null communicator");
  if (!c->custom) return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_comm_set_sendrecv: the communicator runs over RCCL");
  if (!c->planned.empty()) return fail(CARMEL_HIP_ERR_STATE, "carmel_hip_comm_set_sendrecv: before any exchange is planned on it");
  c->tr_sendrecv = fn;
  return CARMEL_HIP_OK;
}

// every rank sends every rank (itself included, as RCCL allows inside a group) n doubles of a pattern naming sender and
// receiver, and checks what arrives: the point-to-point groups of the exchange's direct form on this transport, before a
// training run depends on them
int carmel_hip_comm_selftest(carmel_hip_comm* c, uint32_t n) {
  if (!c) return fail(CARMEL_HIP_ERR_ARG, "null communicator");
  if (!comm_has_p2p(c)) return fail(CARMEL_HIP_ERR_UNSUPPORTED, c->what + ": no point-to-point transfers");
  if (!n) n = 1024;
  HIPCHK(hipSetDevice(c->device));
  const uint32_t W = (uint32_t)c->world, me = (uint32_t)c->rank;
  const bool self = !c->custom;  // (a custom transport's contract excludes the caller itself)
  std::vector<double> h((size_t)2 * W * n, -1.0);
  for (uint32_t q = 0; q < W; ++q)
    for (uint32_t i = 0; i < n; ++i) h[(size_t)q * n + i] = (double)me * 1e6 + (double)q * 1e3 + (double)(i % 997);
  DevBuf<double> d;
  HIPCHK(d.alloc(h.size()));
  HIPCHK(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, c->xstream));
  std::vector<carmel_hip_p2p> ops;
  for (uint32_t q = 0; q < W; ++q) {
    if (q == me && !self) continue;
    ops.push_back({(int32_t)q, 1, d.p + (size_t)q * n, n});
    ops.push_back({(int32_t)q, 0, d.p + (size_t)(W + q) * n, n});
  }
  int rc = comm_p2p(c, ops.data(), (uint32_t)ops.size(), c->xstream);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h.data(), d.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->xstream));
  HIPCHK(hipStreamSynchronize(c->xstream));
  for (uint32_t q = 0; q < W; ++q) {
    if (q == me && !self) continue;
    for (uint32_t i = 0; i < n; ++i)
      if (h[(size_t)(W + q) * n + i] != (double)q * 1e6 + (double)me * 1e3 + (double)(i % 997))
        return fail(CARMEL_HIP_ERR_HIP, c->what + ": point-to-point self-test: rank " + std::to_string(me) + " got a wrong value from rank " +
                                            std::to_string(q) + " at " + std::to_string(i));
  }
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_unique_id(void* id128) {
  if (!id128) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  if (!g_rccl.load()) return fail(CARMEL_HIP_ERR_UNSUPPORTED, g_rccl.err);
  ncclUniqueId id;
  const ncclResult_t rc = g_rccl.get_unique_id(&id);
  if (rc) return fail(CARMEL_HIP_ERR_HIP, "ncclGetUniqueId: " + g_rccl.what(rc));
  std::memcpy(id128, id.internal, 128);
  return CARMEL_HIP_OK;
}

static int comm_common(carmel_hip_comm* c, int device, int rank, int world) {
  c->rank = rank;
  c->world = world;
  c->device = device;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_create(carmel_hip_comm** out, int device, int rank, int world, const void* id128) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return fail(CARMEL_HIP_ERR_ARG, "bad argument");
  if (!g_rccl.load()) return fail(CARMEL_HIP_ERR_UNSUPPORTED, g_rccl.err);
  carmel_hip_comm* c = new carmel_hip_comm();
  int rc0 = comm_common(c, device, rank, world);
  if (rc0) {
    delete c;
    return rc0;
  }
  c->what = "RCCL";
  ncclUniqueId id;
  std::memcpy(id.internal, id128, 128);
  ncclComm_t nc = nullptr;
  const ncclResult_t rc = g_rccl.comm_init_rank(&nc, world, id, rank);
  if (rc) {
    delete c;
    return fail(CARMEL_HIP_ERR_HIP, "ncclCommInitRank: " + g_rccl.what(rc));
  }
  c->rccl = (void*)nc;
  *out = c;
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_create_custom(carmel_hip_comm** out, int device, int rank, int world, const carmel_hip_transport* tr) {
  if (!out || !tr || !tr->allreduce || world < 1 || rank < 0 || rank >= world) return fail(CARMEL_HIP_ERR_ARG, "bad argument");
  carmel_hip_comm* c = new carmel_hip_comm();
  int rc0 = comm_common(c, device, rank, world);
  if (rc0) {
    delete c;
    return rc0;
  }
  c->custom = true;
  c->tr = *tr;
  c->what = tr->name ? tr->name : "custom transport";
  *out = c;
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_destroy(carmel_hip_comm* c) {
  if (!c) return CARMEL_HIP_OK;
  (void)hipSetDevice(c->device);
  exchange_comm_gone(c);  // plans first: they hold events on this communicator's stream
  if (c->rccl) (void)g_rccl.comm_destroy((ncclComm_t)c->rccl);
  delete c;
  return CARMEL_HIP_OK;
}

// after a failed collective on some rank: give up on whatever is still enqueued instead of waiting for it
// (ncclCommAbort; a custom transport is simply destroyed)
int carmel_hip_comm_abort(carmel_hip_comm* c) {
  if (!c) return CARMEL_HIP_OK;
  (void)hipSetDevice(c->device);
  if (c->rccl) {
    if (g_rccl.comm_abort)
      (void)g_rccl.comm_abort((ncclComm_t)c->rccl);
    else
      (void)g_rccl.comm_destroy((ncclComm_t)c->rccl);
  }
  c->rccl = nullptr;
  exchange_comm_gone(c);  // after the abort: what was enqueued has been given up, the streams drain
  delete c;
  return CARMEL_HIP_OK;
}

int carmel_hip_comm_rank(carmel_hip_comm* c) { return c ? c->rank : 0; }
int carmel_hip_comm_world(carmel_hip_comm* c) { return c ? c->world : 1; }
const char* carmel_hip_comm_transport_name(carmel_hip_comm* c) { return c ? c->what.c_str() : ""; }

int carmel_hip_comm_allreduce_host(carmel_hip_comm* c, double* v, uint32_t n, int op_max) {
  if (!c || !v) return fail(CARMEL_HIP_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(c->device));
  if (c->scratch.n < n) HIPCHK(c->scratch.alloc(n));
  HIPCHK(hipMemcpyAsync(c->scratch.p, v, n * sizeof(double), hipMemcpyHostToDevice, c->xstream));
  const int rc = comm_allreduce(c, c->scratch.p, n, op_max != 0, c->xstream);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(v, c->scratch.p, n * sizeof(double), hipMemcpyDeviceToHost, c->xstream));
  HIPCHK(hipStreamSynchronize(c->xstream));
  return CARMEL_HIP_OK;
}

}  // extern "C"
