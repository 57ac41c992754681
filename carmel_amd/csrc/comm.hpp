// comm.hpp — the communicator of corpus-sharded EM (comm.cpp) as the exchange code sees it (exchange.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "engine.hpp"

struct carmel_hip_comm {
  void* rccl = nullptr;  // ncclComm_t
  int rank = 0, world = 1, device = 0;
  bool custom = false;   // a caller-supplied transport (carmel_hip_comm_create_custom) instead of RCCL
  carmel_hip_transport tr{};
  carmel_hip_sendrecv_fn tr_sendrecv = nullptr;  // custom transport: its point-to-point groups (carmel_hip_comm_set_sendrecv)
  hipStream_t xstream = nullptr;  // the exchange's own stream: collectives run here beside the trainer's kernels
  DevBuf<double> scratch;         // small host-vector reductions
  std::string what;               // "RCCL" or the transport's name
  std::vector<carmel_hip_trainer*> planned;  // trainers whose exchange plan points at this communicator: destroying or
                                             // aborting it drops their plans first (exchange_comm_gone), so no plan outlives it
  ~carmel_hip_comm();
};

// dev[0 .. n) := sum (op_max: max) over the ranks, ordered on stream s
int comm_allreduce(carmel_hip_comm* c, double* dev, size_t n, bool op_max, hipStream_t s);
// buf holds world * count doubles: afterwards this rank's piece buf[rank * count ..) is the sum over the ranks of that piece
int comm_reduce_scatter(carmel_hip_comm* c, double* buf, size_t count, hipStream_t s);
// ... every rank's piece is copied to all ranks
int comm_all_gather(carmel_hip_comm* c, double* buf, size_t count, hipStream_t s);

// one group of sends and receives (carmel_hip_p2p), ordered on stream s; comm_has_p2p: the transport can
bool comm_has_p2p(const carmel_hip_comm* c);
int comm_p2p(carmel_hip_comm* c, const carmel_hip_p2p* ops, uint32_t n_ops, hipStream_t s);

// the communicator is going away (destroy: after its stream has drained; abort: whatever was enqueued is given up): every
// trainer planned on it goes back to having no plan -- a replicated M-step on whatever counts it holds
void exchange_comm_gone(carmel_hip_comm* c);
