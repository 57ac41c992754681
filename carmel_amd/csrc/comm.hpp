// comm.hpp — the communicator of corpus-sharded EM (comm.cpp) as the exchange code sees it (exchange.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "engine.hpp"

struct carmel_hip_comm {
  void* rccl = nullptr;  // ncclComm_t
  int rank = 0, world = 1, device = 0;
  bool custom = false;   // a caller-supplied transport (carmel_hip_comm_create_custom) instead of RCCL
  carmel_hip_transport tr{};
  hipStream_t xstream = nullptr;  // the exchange's own stream: collectives run here beside the trainer's kernels
  DevBuf<double> scratch;         // small host-vector reductions
  std::string what;               // "RCCL" or the transport's name
  ~carmel_hip_comm();
};

// dev[0 .. n) := sum (op_max: max) over the ranks, ordered on stream s
int comm_allreduce(carmel_hip_comm* c, double* dev, size_t n, bool op_max, hipStream_t s);
// buf holds world * count doubles: afterwards this rank's piece buf[rank * count ..) is the sum over the ranks of that piece
int comm_reduce_scatter(carmel_hip_comm* c, double* buf, size_t count, hipStream_t s);
// ... every rank's piece is copied to all ranks
int comm_all_gather(carmel_hip_comm* c, double* buf, size_t count, hipStream_t s);
