// nvdb_group.cpp -- one process, several GPUs: the row-sharded flat scan with an RCCL all-gather of the per-shard
// top-k lists (include/nvdb_hip.h, "device group").  Layered ENTIRELY on the public per-device C ABI
// (nvdb_hip_create / _upload_corpus / _search_batch_dev / _search_check / _merge_topk_strided_dev): a group is G
// contexts, G streams, G packed result buffers and one communicator per device.
//
//   per batch (<= 1024 queries):   H2D queries -> every device                      (G streams, one host thread each)
//                                  nvdb_hip_search_batch_dev on every shard          -> packed [ids | scores] per device
//                                  ncclGroupStart; ncclAllGather x G; ncclGroupEnd   (B*k*12 bytes per rank over xGMI)
//                                  merge_topk_kernel on device 0 out of its gathered buffer -> D2H
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy already in the process if there is one -- PyTorch bundles
// its own -- else /opt/rocm's), so the library has no link-time dependency on it and single-GPU users never load it.
// Where RCCL cannot serve the device list (the same device listed twice -- the one-GPU rehearsal -- or no RCCL at all)
// the exchange is G peer copies into device 0's gathered buffer; the merge is the same kernel either way.  The
// reference has no multi-GPU path; this replaces the host-side merge of nvdb::FlatIndexHIPSharded (round 2).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/nvdb_hip.h"
#pragma GCC visibility pop

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
  bool override_lib = false;                    // loaded from NVDB_GROUP_RCCL_LIB (a stand-in that also serves repeated devices)
  bool ok() const { return handle != nullptr; }
};

// One table per library path.  Default: the librccl already in the process (PyTorch bundles one), else /opt/rocm's.
// NVDB_GROUP_RCCL_LIB=<path>, read when a group is created, names another library with the same six entry points; the tests
// use it to drive THIS file's RCCL branch with several ranks on one device through a loopback stand-in
// (tests/loopback_rccl: it serves a device listed more than once, which RCCL itself refuses).
Rccl& rccl(const std::string& override_path) {
  static std::mutex mu;
  static std::map<std::string, Rccl> tables;
  std::lock_guard<std::mutex> lk(mu);
  auto it = tables.find(override_path);
  if (it != tables.end()) return it->second;
  Rccl x;
  x.override_lib = !override_path.empty();
  const std::vector<std::string> names = x.override_lib ? std::vector<std::string>{override_path}
                                                        : std::vector<std::string>{"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  for (const std::string& name : names) {
    x.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (x.handle) break;
  }
  auto done = [&](Rccl& r) -> Rccl& { return tables.emplace(override_path, r).first->second; };
  if (!x.handle) { const char* why = dlerror(); x.why = std::string("librccl not loadable: ") + (why ? why : "?"); return done(x); }   // (dlerror() clears itself: read it once)
#define NVDB_SYM(field, sym)                                                         \
  x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.handle, sym));               \
  if (!x.field) { x.why = std::string("librccl lacks ") + sym; x.handle = nullptr; return done(x); }
  NVDB_SYM(CommInitAll, "ncclCommInitAll")
  NVDB_SYM(CommDestroy, "ncclCommDestroy")
  NVDB_SYM(GroupStart, "ncclGroupStart")
  NVDB_SYM(GroupEnd, "ncclGroupEnd")
  NVDB_SYM(AllGather, "ncclAllGather")
  NVDB_SYM(GetErrorString, "ncclGetErrorString")
#undef NVDB_SYM
  return done(x);
}

std::string g_group_create_err;

}  // namespace

struct nvdb_hip_group {
  std::vector<int> devices;
  std::vector<nvdb_hip_ctx*> ctx;
  std::vector<hipStream_t> stream;
  std::vector<hipEvent_t> done;                 // per device: its search (+ its copy in peer-copy mode) has been enqueued up to here
  std::vector<ncclComm_t> comm;                 // empty in peer-copy mode
  Rccl* R = nullptr;                            // the library the communicators came from
  std::vector<void*> dq, packed, gathered;      // per device: queries, [ids | scores] of its shard, all shards' packed blocks
  void* merged = nullptr;                       // device 0: [ids | scores] of the merged lists
  void* pinned = nullptr;                       // host staging of the merged lists
  size_t q_bytes = 0, pack_bytes = 0, pinned_bytes = 0;
  uint64_t n = 0;
  uint32_t dim = 0, dtype = 0;
  bool use_rccl = false;
  std::string why_not_rccl;
  std::string err;
  uint64_t searches = 0, fallbacks = 0;
};

namespace {

nvdb_status gfail(nvdb_hip_group* g, nvdb_status s, const std::string& m) { g->err = m; return s; }

#define GHIP(g, call)                                                                                    \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) return gfail(g, NVDB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

nvdb_status grow(nvdb_hip_group* g, std::vector<void*>& bufs, size_t idx, size_t want) {
  if (bufs[idx]) GHIP(g, hipFree(bufs[idx]));
  bufs[idx] = nullptr;
  GHIP(g, hipMalloc(&bufs[idx], want));
  return NVDB_OK;
}

// (re)size the per-device buffers for batches of nq queries and lists of k entries
nvdb_status ensure_buffers(nvdb_hip_group* g, uint32_t nq, uint32_t k) {
  const size_t G = g->ctx.size();
  const size_t qb = static_cast<size_t>(nq) * g->dim * 4, pb = static_cast<size_t>(nq) * k * 12;
  if (qb > g->q_bytes) {
    for (size_t i = 0; i < G; ++i) {
      GHIP(g, hipSetDevice(g->devices[i]));
      // + 8 zero query rows: the host API pads its own staging the same way (the exact kernel reads query groups of 8)
      const size_t padq = 8 * static_cast<size_t>(g->dim) * 4;
      nvdb_status st = grow(g, g->dq, i, qb + padq);
      if (st) return st;
      GHIP(g, hipMemset(g->dq[i], 0, qb + padq));
    }
    g->q_bytes = qb;
  }
  if (pb > g->pack_bytes) {
    for (size_t i = 0; i < G; ++i) {
      GHIP(g, hipSetDevice(g->devices[i]));
      nvdb_status st = grow(g, g->packed, i, pb);
      if (st) return st;
      // RCCL: every rank receives all blocks; peer-copy mode: only device 0 does
      if (g->use_rccl || i == 0) { if ((st = grow(g, g->gathered, i, pb * G))) return st; }
    }
    GHIP(g, hipSetDevice(g->devices[0]));
    if (g->merged) GHIP(g, hipFree(g->merged));
    g->merged = nullptr;
    GHIP(g, hipMalloc(&g->merged, pb));
    g->pack_bytes = pb;
  }
  if (pb > g->pinned_bytes) {
    if (g->pinned) (void)hipHostFree(g->pinned);
    g->pinned = nullptr; g->pinned_bytes = 0;
    GHIP(g, hipHostMalloc(&g->pinned, pb, hipHostMallocDefault));
    g->pinned_bytes = pb;
  }
  return NVDB_OK;
}

void shard_range(uint64_t n, size_t i, size_t G, uint64_t* lo, uint64_t* hi) { *lo = n * i / G; *hi = n * (i + 1) / G; }

// one batch of nq <= 1024 queries through the device-side exchange; NVDB_ERR_INTERNAL = a shard's self-check tripped
nvdb_status search_sub_batch(nvdb_hip_group* g, const float* queries, uint32_t nq, uint32_t k, uint64_t* out_ids, float* out_scores) {
  const size_t G = g->ctx.size();
  const size_t qb = static_cast<size_t>(nq) * g->dim * 4, ib = static_cast<size_t>(nq) * k * 8, sb = static_cast<size_t>(nq) * k * 4, pb = ib + sb;
  // 1. queries up + local search on every shard: one host thread per device, so that the G enqueue sequences overlap
  std::vector<nvdb_status> sts(G, NVDB_OK);
  std::vector<std::string> errs(G);
  auto shard = [&](size_t i) {
    hipError_t e = hipSetDevice(g->devices[i]);
    // the 8 rows after the batch are zeros on every call (a smaller batch after a larger one would otherwise leave stale queries there)
    if (e == hipSuccess) e = hipMemsetAsync(static_cast<char*>(g->dq[i]) + qb, 0, 8 * static_cast<size_t>(g->dim) * 4, g->stream[i]);
    if (e == hipSuccess) e = hipMemcpyAsync(g->dq[i], queries, qb, hipMemcpyHostToDevice, g->stream[i]);
    if (e != hipSuccess) { sts[i] = NVDB_ERR_HIP; errs[i] = hipGetErrorString(e); return; }
    char* p = static_cast<char*>(g->packed[i]);
    sts[i] = nvdb_hip_search_batch_dev(g->ctx[i], static_cast<const float*>(g->dq[i]), nq, k, reinterpret_cast<uint64_t*>(p),
                                       reinterpret_cast<float*>(p + ib), g->stream[i]);
    if (sts[i]) { errs[i] = nvdb_hip_last_error(g->ctx[i]); return; }
    if (!g->use_rccl) {
      // peer-copy exchange: my block into device 0's gathered buffer, on my stream
      e = hipMemcpyPeerAsync(static_cast<char*>(g->gathered[0]) + i * pb, g->devices[0], p, g->devices[i], pb, g->stream[i]);
      if (e == hipSuccess) e = hipEventRecord(g->done[i], g->stream[i]);
      if (e != hipSuccess) { sts[i] = NVDB_ERR_HIP; errs[i] = hipGetErrorString(e); }
    }
  };
  if (G == 1) shard(0);
  else {
    std::vector<std::thread> th;
    for (size_t i = 0; i < G; ++i) th.emplace_back(shard, i);
    for (auto& t : th) t.join();
  }
  for (size_t i = 0; i < G; ++i) if (sts[i]) return gfail(g, sts[i], "shard " + std::to_string(i) + ": " + errs[i]);
  // 2. the exchange
  if (g->use_rccl) {
    Rccl& R = *g->R;
    ncclResult_t r = R.GroupStart();
    for (size_t i = 0; i < G && r == ncclSuccess; ++i)
      r = R.AllGather(g->packed[i], g->gathered[i], pb, ncclUint8, g->comm[i], g->stream[i]);
    const ncclResult_t r2 = R.GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) return gfail(g, NVDB_ERR_HIP, std::string("ncclAllGather: ") + R.GetErrorString(r != ncclSuccess ? r : r2));
  } else {
    GHIP(g, hipSetDevice(g->devices[0]));
    for (size_t i = 1; i < G; ++i) GHIP(g, hipStreamWaitEvent(g->stream[0], g->done[i], 0));
  }
  // 3. merge on device 0 out of the gathered buffer (block i = [ids | scores] of shard i), results down, one synchronisation
  GHIP(g, hipSetDevice(g->devices[0]));
  char* gat = static_cast<char*>(g->gathered[0]);
  char* mer = static_cast<char*>(g->merged);
  nvdb_status st = nvdb_hip_merge_topk_strided_dev(g->ctx[0], reinterpret_cast<const uint64_t*>(gat), reinterpret_cast<const float*>(gat + ib), pb, pb,
                                                   static_cast<uint32_t>(G), nq, k, reinterpret_cast<uint64_t*>(mer), reinterpret_cast<float*>(mer + ib), g->stream[0]);
  if (st) return gfail(g, st, std::string("merge: ") + nvdb_hip_last_error(g->ctx[0]));
  GHIP(g, hipMemcpyAsync(g->pinned, mer, pb, hipMemcpyDeviceToHost, g->stream[0]));
  for (size_t i = 0; i < G; ++i) { GHIP(g, hipSetDevice(g->devices[i])); GHIP(g, hipStreamSynchronize(g->stream[i])); }
  // 4. every shard's self-check (list overflow / bound violation): the caller falls back to the per-shard host API on a trip
  bool tripped = false;
  for (size_t i = 0; i < G; ++i) {
    const nvdb_status chk = nvdb_hip_search_check(g->ctx[i], nullptr);
    if (chk == NVDB_ERR_INTERNAL) tripped = true;
    else if (chk) return gfail(g, chk, "shard " + std::to_string(i) + ": " + nvdb_hip_last_error(g->ctx[i]));
  }
  if (tripped) return NVDB_ERR_INTERNAL;
  std::memcpy(out_ids, g->pinned, ib);
  std::memcpy(out_scores, static_cast<char*>(g->pinned) + ib, sb);
  return NVDB_OK;
}

// the round-2 flow, kept for what the device exchange does not take (a tripped self-check): every shard through the host API (which retries and falls back by itself), lists merged on the host
nvdb_status search_host_merge(nvdb_hip_group* g, const float* queries, uint32_t nq, uint32_t k, uint64_t* out_ids, float* out_scores) {
  const size_t G = g->ctx.size();
  const size_t per = static_cast<size_t>(nq) * k;
  std::vector<uint64_t> ids(G * per, ~0ull);
  std::vector<float> sc(G * per, -__builtin_huge_valf());
  std::vector<nvdb_status> sts(G, NVDB_OK);
  std::vector<std::thread> th;
  for (size_t i = 0; i < G; ++i)
    th.emplace_back([&, i] { sts[i] = nvdb_hip_search_batch(g->ctx[i], queries, nq, k, ids.data() + i * per, sc.data() + i * per, nullptr, nullptr); });
  for (auto& t : th) t.join();
  for (size_t i = 0; i < G; ++i) if (sts[i]) return gfail(g, sts[i], "shard " + std::to_string(i) + ": " + nvdb_hip_last_error(g->ctx[i]));
  g->fallbacks++;
  return nvdb_merge_topk_host(ids.data(), sc.data(), static_cast<uint32_t>(G), nq, k, out_ids, out_scores);
}

}  // namespace

extern "C" {

nvdb_status nvdb_hip_group_create(const int* devices, uint32_t n_devices, nvdb_hip_group** out) {
  if (!out) return NVDB_ERR_INVALID;
  *out = nullptr;
  if (!devices || n_devices == 0) { g_group_create_err = "group_create: empty device list"; return NVDB_ERR_INVALID; }
  auto* g = new nvdb_hip_group();
  g->devices.assign(devices, devices + n_devices);
  const size_t G = n_devices;
  g->dq.assign(G, nullptr); g->packed.assign(G, nullptr); g->gathered.assign(G, nullptr);
  auto bail = [&](nvdb_status s, const std::string& m) { g_group_create_err = m; nvdb_hip_group_destroy(g); return s; };
  for (size_t i = 0; i < G; ++i) {
    nvdb_hip_ctx* c = nullptr;
    const nvdb_status st = nvdb_hip_create(devices[i], &c);
    if (st) return bail(st, nvdb_hip_last_error(nullptr));
    g->ctx.push_back(c);
    hipStream_t s = nullptr;
    hipEvent_t e = nullptr;
    if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
      return bail(NVDB_ERR_HIP, "group_create: stream / event creation failed");
    g->stream.push_back(s);
    g->done.push_back(e);
  }
  // RCCL wants distinct devices; NVDB_GROUP_NO_RCCL=1 forces the peer-copy exchange (tests)
  const bool distinct = std::set<int>(g->devices.begin(), g->devices.end()).size() == G;
  const char* off = std::getenv("NVDB_GROUP_NO_RCCL");
  const char* lib = std::getenv("NVDB_GROUP_RCCL_LIB");
  const bool no_rccl = off && off[0] == '1';
  Rccl* R = (no_rccl || (!distinct && !(lib && lib[0]))) ? nullptr : &rccl(lib ? lib : "");
  if (no_rccl) g->why_not_rccl = "NVDB_GROUP_NO_RCCL=1";
  else if (!R) g->why_not_rccl = "a device is listed more than once";
  else if (!R->ok()) g->why_not_rccl = R->why;
  else {
    g->comm.assign(G, nullptr);
    const ncclResult_t r = R->CommInitAll(g->comm.data(), static_cast<int>(G), g->devices.data());
    if (r != ncclSuccess) { g->why_not_rccl = std::string("ncclCommInitAll: ") + R->GetErrorString(r); g->comm.clear(); }
    else { g->use_rccl = true; g->R = R; }
  }
  *out = g;
  return NVDB_OK;
}

void nvdb_hip_group_destroy(nvdb_hip_group* g) {
  if (!g) return;
  for (size_t i = 0; i < g->stream.size(); ++i) { (void)hipSetDevice(g->devices[i]); (void)hipStreamSynchronize(g->stream[i]); }
  for (ncclComm_t c : g->comm) if (c && g->R) (void)g->R->CommDestroy(c);
  for (size_t i = 0; i < g->ctx.size(); ++i) {       // only devices a context was created on (a bad ordinal never reaches HIP)
    (void)hipSetDevice(g->devices[i]);
    for (auto* v : {&g->dq, &g->packed, &g->gathered}) if (i < v->size() && (*v)[i]) (void)hipFree((*v)[i]);
    if (i < g->done.size()) (void)hipEventDestroy(g->done[i]);
    if (i < g->stream.size()) (void)hipStreamDestroy(g->stream[i]);
  }
  if (g->merged && !g->ctx.empty()) { (void)hipSetDevice(g->devices[0]); (void)hipFree(g->merged); }
  if (g->pinned) (void)hipHostFree(g->pinned);
  for (nvdb_hip_ctx* c : g->ctx) nvdb_hip_destroy(c);
  delete g;
}

const char* nvdb_hip_group_last_error(const nvdb_hip_group* g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }

uint32_t nvdb_hip_group_size(const nvdb_hip_group* g) { return g ? static_cast<uint32_t>(g->ctx.size()) : 0; }

nvdb_hip_ctx* nvdb_hip_group_ctx(nvdb_hip_group* g, uint32_t shard) { return (g && shard < g->ctx.size()) ? g->ctx[shard] : nullptr; }

int nvdb_hip_group_exchange(const nvdb_hip_group* g, const char** why) {
  if (!g) return -1;
  if (why) *why = g->use_rccl ? (g->R->override_lib ? "rccl all-gather (library named by NVDB_GROUP_RCCL_LIB)" : "rccl all-gather") : g->why_not_rccl.c_str();
  return g->use_rccl ? 1 : 0;
}

nvdb_status nvdb_hip_group_upload_corpus(nvdb_hip_group* g, const void* rows, const float* scales, uint64_t n, uint32_t dim, uint32_t dtype) {
  if (!g) return NVDB_ERR_INVALID;
  const size_t bpe = dtype == NVDB_DTYPE_F32 ? 4 : dtype == NVDB_DTYPE_F16 ? 2 : dtype == NVDB_DTYPE_I8 ? 1 : 0;
  if (!rows || n == 0 || dim == 0 || bpe == 0) return gfail(g, NVDB_ERR_INVALID, bpe ? "group upload: empty corpus" : "Unsupported base dtype (Float32/Float16/Int8 only)");
  if (n < g->ctx.size()) return gfail(g, NVDB_ERR_INVALID, "group upload: fewer rows than shards");
  const size_t G = g->ctx.size();
  std::vector<nvdb_status> sts(G, NVDB_OK);
  std::vector<std::thread> th;
  for (size_t i = 0; i < G; ++i)
    th.emplace_back([&, i] {
      uint64_t lo, hi;
      shard_range(n, i, G, &lo, &hi);
      sts[i] = nvdb_hip_upload_corpus(g->ctx[i], static_cast<const char*>(rows) + lo * dim * bpe, scales ? scales + lo : nullptr, hi - lo, dim, dtype, lo);
    });
  for (auto& t : th) t.join();
  for (size_t i = 0; i < G; ++i) if (sts[i]) return gfail(g, sts[i], "shard " + std::to_string(i) + ": " + nvdb_hip_last_error(g->ctx[i]));
  g->n = n; g->dim = dim; g->dtype = dtype;
  return NVDB_OK;
}

nvdb_status nvdb_hip_group_generate_corpus(nvdb_hip_group* g, uint64_t seed, uint64_t n, uint32_t dim, uint32_t dtype) {
  if (!g) return NVDB_ERR_INVALID;
  if (n < g->ctx.size() || dim == 0) return gfail(g, NVDB_ERR_INVALID, "group generate: fewer rows than shards");
  const size_t G = g->ctx.size();
  for (size_t i = 0; i < G; ++i) {
    uint64_t lo, hi;
    shard_range(n, i, G, &lo, &hi);
    const nvdb_status st = nvdb_hip_generate_corpus(g->ctx[i], seed, hi - lo, dim, dtype, lo);
    if (st) return gfail(g, st, "shard " + std::to_string(i) + ": " + nvdb_hip_last_error(g->ctx[i]));
  }
  g->n = n; g->dim = dim; g->dtype = dtype;
  return NVDB_OK;
}

nvdb_status nvdb_hip_group_set_option(nvdb_hip_group* g, const char* key, int64_t value) {
  if (!g) return NVDB_ERR_INVALID;
  for (size_t i = 0; i < g->ctx.size(); ++i) {
    const nvdb_status st = nvdb_hip_set_option(g->ctx[i], key, value);
    if (st) return gfail(g, st, nvdb_hip_last_error(g->ctx[i]));
  }
  return NVDB_OK;
}

nvdb_status nvdb_hip_group_search_batch(nvdb_hip_group* g, const float* queries, uint32_t nq, uint32_t k, uint64_t* out_ids,
                                        float* out_scores, uint32_t* out_k_eff, nvdb_hip_group_stats* stats) {
  if (!g) return NVDB_ERR_INVALID;
  if (g->n == 0) return gfail(g, NVDB_ERR_NO_CORPUS, "Empty base");
  if (nq > 0 && k > 0 && (!queries || !out_ids || !out_scores)) return gfail(g, NVDB_ERR_INVALID, queries ? "null output" : "Null query");
  if (out_k_eff) *out_k_eff = static_cast<uint32_t>(std::min<uint64_t>(k, g->n));
  const uint64_t fb0 = g->fallbacks;
  if (nq && k) {
    for (uint32_t q0 = 0; q0 < nq; q0 += 1024) {
      const uint32_t b = std::min<uint32_t>(1024, nq - q0);
      const float* q = queries + static_cast<size_t>(q0) * g->dim;
      uint64_t* oi = out_ids + static_cast<size_t>(q0) * k;
      float* os = out_scores + static_cast<size_t>(q0) * k;
      nvdb_status st;
      if ((st = ensure_buffers(g, b, k))) return st;
      st = search_sub_batch(g, q, b, k, oi, os);
      if (st == NVDB_ERR_INTERNAL) st = search_host_merge(g, q, b, k, oi, os);     // a shard's self-check tripped
      if (st) return st;
      g->searches++;
    }
  }
  if (stats) {
    stats->shards = static_cast<uint32_t>(g->ctx.size());
    stats->exchange = g->use_rccl ? 1u : 0u;
    stats->host_merge_fallbacks = static_cast<uint32_t>(g->fallbacks - fb0);
    stats->bytes_per_rank = static_cast<uint64_t>(std::min<uint32_t>(nq, 1024)) * k * 12;
  }
  return NVDB_OK;
}

}  // extern "C"
