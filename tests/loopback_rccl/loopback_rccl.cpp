// loopback_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl that serves several "ranks" on ONE device.
//
// A one-GPU test box cannot run the device group's RCCL branch with more than one rank (RCCL refuses a device that is
// listed twice), so the receive offsets and the stream ordering of `ncclGroupStart; ncclAllGather x G; ncclGroupEnd`
// (nano-vectordb_amd/csrc/nvdb_group.cpp) would meet an 8-GPU node unexecuted.  This library exports the six RCCL entry
// points the group binds with dlopen -- same names, same signatures (rccl/rccl.h) -- and implements the all-gather with
// the collective's ordering semantics on plain HIP streams:
//
//   at ncclGroupEnd, for the G queued calls (rank j: send_j, recv_j, stream_j):
//     every stream_j records "my send buffer is ready";
//     stream_i waits for ALL ready events, copies send_j -> recv_i + j * bytes for every j (hipMemcpyAsync), records "done";
//     every stream_j waits for all done events (a send buffer may be reused only when every rank has read it).
//
// The product never links or ships this file; the group loads it only when NVDB_GROUP_RCCL_LIB names it
// (tests/test_gpu_group.py).  Counters let the test see that the RCCL branch ran and that communicators were reused.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <mutex>
#include <vector>

namespace {
struct Clique {
  int n = 0;
  std::atomic<int> alive{0};
};
}  // namespace

struct ncclComm {
  Clique* clique;
  int rank, device;
  hipEvent_t ready, done;
};

namespace {
struct Op { const void* send; void* recv; size_t bytes; ncclComm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
std::atomic<int> g_inits{0}, g_allgathers{0}, g_groups{0}, g_destroys{0};

size_t dtype_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

ncclResult_t flush() {
  std::vector<Op> ops;
  ops.swap(g_ops);
  if (ops.empty()) return ncclSuccess;
  Clique* cl = ops[0].comm->clique;
  // one call per rank of ONE clique, equal sizes: what the group issues; anything else is a usage error here
  if (static_cast<int>(ops.size()) != cl->n) return ncclInvalidUsage;
  std::vector<const Op*> by_rank(cl->n, nullptr);
  for (const Op& o : ops) {
    if (o.comm->clique != cl || o.bytes != ops[0].bytes || by_rank[o.comm->rank]) return ncclInvalidUsage;
    by_rank[o.comm->rank] = &o;
  }
#define LB_HIP(call) do { if ((call) != hipSuccess) return ncclUnhandledCudaError; } while (0)
  for (const Op* o : by_rank) { LB_HIP(hipSetDevice(o->comm->device)); LB_HIP(hipEventRecord(o->comm->ready, o->stream)); }
  for (const Op* r : by_rank) {
    LB_HIP(hipSetDevice(r->comm->device));
    for (const Op* s : by_rank) if (s != r) LB_HIP(hipStreamWaitEvent(r->stream, s->comm->ready, 0));
    for (const Op* s : by_rank)
      LB_HIP(hipMemcpyAsync(static_cast<char*>(r->recv) + static_cast<size_t>(s->comm->rank) * s->bytes, s->send, s->bytes, hipMemcpyDefault, r->stream));
    LB_HIP(hipEventRecord(r->comm->done, r->stream));
  }
  for (const Op* s : by_rank) {
    LB_HIP(hipSetDevice(s->comm->device));
    for (const Op* r : by_rank) if (s != r) LB_HIP(hipStreamWaitEvent(s->stream, r->comm->done, 0));
  }
#undef LB_HIP
  return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist) {
  if (!comm || ndev <= 0) return ncclInvalidArgument;
  Clique* cl = new Clique();
  cl->n = ndev;
  cl->alive = ndev;
  for (int i = 0; i < ndev; ++i) {
    ncclComm* c = new ncclComm{cl, i, devlist ? devlist[i] : i, nullptr, nullptr};
    if (hipSetDevice(c->device) != hipSuccess || hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess)
      return ncclUnhandledCudaError;
    comm[i] = c;
  }
  ++g_inits;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclInvalidArgument;
  (void)hipSetDevice(c->device);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->done) (void)hipEventDestroy(c->done);
  if (--c->clique->alive == 0) delete c->clique;
  delete c;
  ++g_destroys;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  ++g_groups;
  return flush();
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  const size_t eb = dtype_bytes(datatype);
  if (!sendbuff || !recvbuff || !comm || eb == 0) return ncclInvalidArgument;
  g_ops.push_back(Op{sendbuff, recvbuff, sendcount * eb, comm, stream});
  ++g_allgathers;
  return g_depth > 0 ? ncclSuccess : flush();     // outside a group: only a clique of one rank can complete
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "loopback stand-in: HIP call failed";
    case ncclInvalidArgument: return "loopback stand-in: invalid argument";
    case ncclInvalidUsage: return "loopback stand-in: invalid usage (one equal-sized ncclAllGather per rank inside a group)";
    default: return "loopback stand-in: error";
  }
}

// test hook: [0] ncclCommInitAll calls, [1] ncclAllGather calls, [2] outermost ncclGroupEnd calls, [3] ncclCommDestroy calls
void nvdb_loopback_rccl_counters(int* out4) {
  out4[0] = g_inits; out4[1] = g_allgathers; out4[2] = g_groups; out4[3] = g_destroys;
}

}  // extern "C"
