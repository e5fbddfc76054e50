// nvdb_corpus.cpp -- the context and its resident corpus: create / destroy, upload / adopt / generate (fp16 / int8 shadow copies,
// row-norm pass), options, statistics, host-side helpers of the C ABI (include/nvdb_hip.h).
#include "nvdb_ctx.h"
#include "kernels_corpus.h"

namespace {

std::string g_create_err;

void free_corpus(nvdb_hip_ctx* c) {
  if (c->owned) {
    if (c->rows) (void)hipFree(c->rows);
    if (c->scales) (void)hipFree(c->scales);
  }
  if (c->shadow16) { (void)hipFree(c->shadow16); c->shadow16 = nullptr; }
  if (c->shadow8) { (void)hipFree(c->shadow8); c->shadow8 = nullptr; }
  if (c->shadow8_scales) { (void)hipFree(c->shadow8_scales); c->shadow8_scales = nullptr; }
  c->rows = nullptr; c->scales = nullptr; c->owned = false; c->n = 0; c->dim = 0; c->fdim = 0; c->dtype = 0; c->max_norm = 0.f;
  c->cap_hint = 0;
  c->q8shadow = false; c->resid_max = 0.f; c->filter_max_norm = 0.f;
}

nvdb_status compute_max_norm(nvdb_hip_ctx* c) {
  nvdb_status st = ensure(c, c->misc, 64);
  if (st) return st;
  HIPCHK(c, hipMemsetAsync(c->misc.p, 0, 64, c->stream));
  uint32_t* bits = static_cast<uint32_t*>(c->misc.p) + 8;
  const unsigned grid = static_cast<unsigned>(std::min<uint64_t>((c->n + 3) / 4, 4096));
  if (c->dtype == NVDB_DTYPE_F32) row_norm_max_kernel<DT_F32><<<grid, 256, 0, c->stream>>>(c->rows, c->scales, c->n, c->dim, bits);
  else if (c->dtype == NVDB_DTYPE_F16) row_norm_max_kernel<DT_F16><<<grid, 256, 0, c->stream>>>(c->rows, c->scales, c->n, c->dim, bits);
  else row_norm_max_kernel<DT_I8><<<grid, 256, 0, c->stream>>>(c->rows, c->scales, c->n, c->dim, bits);
  HIPCHK(c, hipGetLastError());
  uint32_t h = 0, hb[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(hb, bits, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  h = hb[0];
  std::memcpy(&c->max_norm, &h, 4);
  c->i8_scales_signed = hb[1] != 0;
  // Which dim do the MFMA kernels run at?  fp16 corpus with an instantiated dim: the corpus itself, no copy.
  // fp32 corpus, or fp16 with another dim <= 1536: an fp16 shadow copy, rows zero-padded to the next instantiated
  // dim (skipped when values would overflow a half).  int8: its own instantiations, no shadow.
  c->fdim = c->dim;
  c->filter_max_norm = c->max_norm;
  c->q8shadow = false; c->resid_max = 0.f;
  // Option q8_shadow (off by default): an fp16 / fp32 corpus is FILTERED through an int8 copy of itself -- half (a quarter) of the bytes per
  // row and integer MFMAs at twice the fp16 rate; the rows' quantisation residual joins the filter's error bound and every survivor is
  // re-scored from the ORIGINAL rows, so ids and scores stay bit-exact.  Costs n x dim bytes of HBM.
  if (c->opt_q8_shadow && c->dtype != NVDB_DTYPE_I8 && i8_filter_dim(c->dim)) {
    const uint32_t sdim = c->dim;
    const size_t count = static_cast<size_t>(c->n) * sdim, pad = static_cast<size_t>(PAD_ROWS) * sdim + 4096;
    const size_t n_pad = (static_cast<size_t>(c->n) + PAD_ROWS - 1) / PAD_ROWS * PAD_ROWS + PAD_ROWS;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shadow8), count + pad));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shadow8_scales), n_pad * 4));
    HIPCHK(c, hipMemsetAsync(c->shadow8 + count, 0, pad, c->stream));
    HIPCHK(c, hipMemsetAsync(c->shadow8_scales, 0, n_pad * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(bits, 0, 8, c->stream));
    if (c->dtype == NVDB_DTYPE_F32) shadow_q8_kernel<float><<<grid, 256, 0, c->stream>>>(static_cast<const float*>(c->rows), c->shadow8, c->shadow8_scales, c->n, c->dim, sdim, bits);
    else shadow_q8_kernel<_Float16><<<grid, 256, 0, c->stream>>>(static_cast<const _Float16*>(c->rows), c->shadow8, c->shadow8_scales, c->n, c->dim, sdim, bits);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hb, bits, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (hb[1] == 0) {
      std::memcpy(&c->resid_max, &hb[0], 4);
      // row norms of the shadow (what the int8 kernels stream): the query side of the bound and the lo-plane margin are priced with them
      HIPCHK(c, hipMemsetAsync(bits, 0, 8, c->stream));
      row_norm_max_kernel<DT_I8><<<grid, 256, 0, c->stream>>>(c->shadow8, c->shadow8_scales, c->n, sdim, bits);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpyAsync(hb, bits, 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      std::memcpy(&c->filter_max_norm, &hb[0], 4);
      c->q8shadow = true;
      c->i8_scales_signed = false;
      return NVDB_OK;                                          // (no fp16 shadow beside it)
    }
    (void)hipFree(c->shadow8); c->shadow8 = nullptr;           // a non-finite value in the corpus: no shadow, the usual routing
    (void)hipFree(c->shadow8_scales); c->shadow8_scales = nullptr;
  }
  if (c->dtype != NVDB_DTYPE_I8 && c->dim <= F16_FILTER_MAX_DIM && !(c->dtype == NVDB_DTYPE_F16 && f16_filter_dim(c->dim)) && c->opt_f32_shadow) {
    uint32_t sdim = 128;
    while (!f16_filter_dim(sdim) || sdim < c->dim) sdim += 128;
    const size_t count = static_cast<size_t>(c->n) * sdim;
    const size_t pad = static_cast<size_t>(PAD_ROWS) * sdim * 2 + 4096;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shadow16), count * 2 + pad));
    HIPCHK(c, hipMemsetAsync(reinterpret_cast<char*>(c->shadow16) + count * 2, 0, pad, c->stream));
    HIPCHK(c, hipMemsetAsync(bits, 0, 4, c->stream));
    if (c->dtype == NVDB_DTYPE_F32) shadow_f16_kernel<float><<<4096, 256, 0, c->stream>>>(static_cast<const float*>(c->rows), c->shadow16, c->n, c->dim, sdim, bits);
    else shadow_f16_kernel<_Float16><<<4096, 256, 0, c->stream>>>(static_cast<const _Float16*>(c->rows), c->shadow16, c->n, c->dim, sdim, bits);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&h, bits, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float maxabs; std::memcpy(&maxabs, &h, 4);
    if (!(maxabs < 60000.f)) { (void)hipFree(c->shadow16); c->shadow16 = nullptr; }
    else c->fdim = sdim;
  }
  if (c->dtype == NVDB_DTYPE_I8 && c->dim <= I8_FILTER_MAX_DIM && !i8_filter_dim(c->dim) && c->opt_f32_shadow) {
    uint32_t sdim = 256;
    while (!i8_filter_dim(sdim) || sdim < c->dim) sdim += 128;
    const size_t count = static_cast<size_t>(c->n) * sdim, pad = static_cast<size_t>(PAD_ROWS) * sdim + 4096;
    const size_t n_pad = (static_cast<size_t>(c->n) + PAD_ROWS - 1) / PAD_ROWS * PAD_ROWS + PAD_ROWS;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shadow8), count + pad));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->shadow8_scales), n_pad * 4));
    HIPCHK(c, hipMemsetAsync(c->shadow8 + count, 0, pad, c->stream));
    HIPCHK(c, hipMemsetAsync(c->shadow8_scales, 0, n_pad * 4, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->shadow8_scales, c->scales, static_cast<size_t>(c->n) * 4, hipMemcpyDeviceToDevice, c->stream));
    shadow_i8_kernel<<<4096, 256, 0, c->stream>>>(static_cast<const signed char*>(c->rows), c->shadow8, c->n, c->dim, sdim);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->fdim = sdim;
  }
  return NVDB_OK;
}

// Corpus rows host -> HBM.  The caller's rows are pageable (an mmap of the vecbin: reference VectorDataset, src/vector_dataset.cpp:24-157;
// its CUDA track uploads the base with one cudaMemcpy from that mmap, src/cuda_refine.cu:179-204).  One hipMemcpy from pageable memory
// stages through the runtime's single bounce buffer: 21-25 GB/s on the bench host, 0.74 s of the CLI's wall for 15 GB.  Here T host
// threads each own two pinned 8 MB buffers and a stream: copy a chunk out of the mmap (this is also where its pages are faulted in,
// T at a time), enqueue its H2D DMA, go on with the next chunk while the DMA runs; a buffer is reused when its event has fired.
// NVDB_UPLOAD_THREADS overrides T (1 = the plain chunked hipMemcpy).
nvdb_status upload_rows(nvdb_hip_ctx* c, void* dst, const void* src, size_t bytes) {
  unsigned T = std::min(6u, std::max(2u, std::thread::hardware_concurrency() / 2));   // 6: the plateau on the bench host (profiles/r04_upload_bench.txt)
  if (const char* e = std::getenv("NVDB_UPLOAD_THREADS")) T = static_cast<unsigned>(std::max(1, std::atoi(e)));
  constexpr size_t CH = size_t(8) << 20;
  const size_t nch = (bytes + CH - 1) / CH;
  if (T <= 1 || nch < 4 * static_cast<size_t>(T)) {
    const size_t chunk = size_t(256) << 20;
    for (size_t off = 0; off < bytes; off += chunk) {
      const size_t take = std::min(chunk, bytes - off);
      HIPCHK(c, hipMemcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, take, hipMemcpyHostToDevice));
    }
    return NVDB_OK;
  }
  std::vector<std::string> errs(T);
  std::atomic<size_t> next{0};
  auto worker = [&](unsigned t) {
    hipStream_t st = nullptr;
    void* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess && errs[t].empty()) errs[t] = std::string(what) + ": " + hipGetErrorString(e); return e == hipSuccess; };
    bool ok = chk(hipSetDevice(c->device), "hipSetDevice") && chk(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    const auto tp0 = std::chrono::steady_clock::now();
    for (int b = 0; b < 2 && ok; ++b)
      ok = chk(hipHostMalloc(&pin[b], CH, hipHostMallocDefault), "hipHostMalloc") && chk(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming), "hipEventCreate");
    if (t == 0 && std::getenv("NVDB_UPLOAD_DEBUG")) std::fprintf(stderr, "[nvdb upload] thread 0: stream + 2 pinned buffers of %zu MB in %.1f ms\n", CH >> 20, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
    // chunks are handed out in order (an atomic counter): the threads walk the file front to back together
    for (int b = 0; ok; b ^= 1) {
      const size_t i = next.fetch_add(1);
      if (i >= nch) break;
      const size_t off = i * CH, take = std::min(CH, bytes - off);
      if (used[b]) ok = chk(hipEventSynchronize(ev[b]), "hipEventSynchronize");
      if (!ok) break;
      std::memcpy(pin[b], static_cast<const char*>(src) + off, take);
      ok = chk(hipMemcpyAsync(static_cast<char*>(dst) + off, pin[b], take, hipMemcpyHostToDevice, st), "hipMemcpyAsync") && chk(hipEventRecord(ev[b], st), "hipEventRecord");
      used[b] = true;
    }
    if (st) (void)hipStreamSynchronize(st);
    for (int b = 0; b < 2; ++b) { if (ev[b]) (void)hipEventDestroy(ev[b]); if (pin[b]) (void)hipHostFree(pin[b]); }
    if (st) (void)hipStreamDestroy(st);
  };
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t) th.emplace_back(worker, t);
  for (auto& x : th) x.join();
  for (const auto& e : errs) if (!e.empty()) return fail(c, NVDB_ERR_HIP, "upload: " + e);
  return NVDB_OK;
}

nvdb_status check_corpus_args(nvdb_hip_ctx* c, uint64_t n, uint32_t dim, uint32_t dtype) {
  if (!c) return NVDB_ERR_INVALID;
  if (n == 0 || dim == 0) return fail(c, NVDB_ERR_INVALID, "corpus: count and dim must be > 0");
  if (bpe_of(dtype) == 0) return fail(c, NVDB_ERR_INVALID, "Unsupported base dtype (Float32/Float16/Int8 only)");
  if (n >= 0xFFFFFFF0ull) return fail(c, NVDB_ERR_UNSUPPORTED, "corpus shard must hold fewer than 2^32-16 rows (shard it)");
  return NVDB_OK;
}

}  // namespace

extern "C" {

int nvdb_hip_abi_version(void) { return NVDB_HIP_ABI_VERSION; }

int nvdb_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

nvdb_status nvdb_hip_create(int device_ordinal, nvdb_hip_ctx** out_ctx) {
  if (!out_ctx) return NVDB_ERR_INVALID;
  *out_ctx = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_err = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return NVDB_ERR_HIP;
  }
  if (device_ordinal < 0 || device_ordinal >= ndev) { g_create_err = "device ordinal out of range"; return NVDB_ERR_INVALID; }
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess) { g_create_err = hipGetErrorString(e); return NVDB_ERR_HIP; }
  auto* c = new nvdb_hip_ctx();
  c->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
    g_create_err = hipGetErrorString(e);
    delete c;
    return NVDB_ERR_HIP;
  }
  {
    float w0[24];
    for (int i = 0; i < 24; ++i) w0[i] = i < 8 ? 1.f : 0.f;
    if ((e = hipMalloc(&c->xcdw.p, sizeof(w0))) != hipSuccess || (e = hipMemcpy(c->xcdw.p, w0, sizeof(w0), hipMemcpyHostToDevice)) != hipSuccess) {
      g_create_err = hipGetErrorString(e);
      if (c->xcdw.p) (void)hipFree(c->xcdw.p);
      (void)hipStreamDestroy(c->stream);
      delete c;
      return NVDB_ERR_HIP;
    }
    c->xcdw.bytes = sizeof(w0);
  }
  *out_ctx = c;
  return NVDB_OK;
}

void nvdb_hip_destroy(nvdb_hip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  free_corpus(c);
  if (c->hostblock.p) c->misc.p = nullptr;         // (misc lives inside the host API's result block)
  for (DevBuf* b : {&c->q32, &c->q16, &c->qscale, &c->qinv, &c->ebound, &c->slack, &c->thr, &c->cnt, &c->overflow, &c->cand,
                    &c->out_ids, &c->out_scores, &c->misc, &c->hostblock, &c->hitlog, &c->prog, &c->qdelta, &c->rq, &c->rcand, &c->rout_ids, &c->rout_dist, &c->lk_scores, &c->lk_sel, &c->lk_hist, &c->lk_state, &c->xcdw, &c->tickets})
    if (b->p) (void)hipFree(b->p);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (auto& k : c->klaunch) { (void)hipEventDestroy(k.e0); (void)hipEventDestroy(k.e1); }
  for (hipEvent_t e : c->kl_pool) (void)hipEventDestroy(e);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->rpinned) (void)hipHostFree(c->rpinned);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* nvdb_hip_last_error(const nvdb_hip_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

nvdb_status nvdb_hip_upload_corpus(nvdb_hip_ctx* c, const void* rows, const float* scales, uint64_t n, uint32_t dim,
                                   uint32_t dtype, uint64_t global_row_base) {
  nvdb_status st = check_corpus_args(c, n, dim, dtype);
  if (st) return st;
  if (!rows) return fail(c, NVDB_ERR_INVALID, "upload_corpus: null rows");
  if (dtype == NVDB_DTYPE_I8 && !scales) return fail(c, NVDB_ERR_INVALID, "upload_corpus: int8 corpus needs per-row scales");
  HIPCHK(c, hipSetDevice(c->device));
  free_corpus(c);
  const size_t bytes = static_cast<size_t>(n) * dim * bpe_of(dtype);
  const size_t pad = static_cast<size_t>(PAD_ROWS) * dim * bpe_of(dtype) + 4096;   // zero rows up to a whole tile (+ slack for vector loads)
  const bool dbg = std::getenv("NVDB_UPLOAD_DEBUG") != nullptr;      // stderr: where an upload's time goes
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  HIPCHK(c, hipMalloc(&c->rows, bytes + pad));
  HIPCHK(c, hipMemset(static_cast<char*>(c->rows) + bytes, 0, pad));
  c->owned = true;
  const double t_alloc = since();
  if ((st = upload_rows(c, c->rows, rows, bytes))) return st;
  const double t_rows = since();
  if (dtype == NVDB_DTYPE_I8) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->scales), (n + PAD_ROWS) * sizeof(float)));
    HIPCHK(c, hipMemset(c->scales + n, 0, PAD_ROWS * sizeof(float)));
    HIPCHK(c, hipMemcpy(c->scales, scales, n * sizeof(float), hipMemcpyHostToDevice));
  }
  c->n = n; c->dim = dim; c->dtype = dtype; c->row_base = global_row_base;
  st = compute_max_norm(c);
  if (dbg) std::fprintf(stderr, "[nvdb upload] %.2f GB: device allocation %.1f ms, rows %.1f ms (%.1f GB/s), scales + row-norm pass (+ shadow copy) %.1f ms\n",
                        bytes / 1e9, t_alloc, t_rows - t_alloc, bytes / 1e6 / (t_rows - t_alloc), since() - t_rows);
  return st;
}

nvdb_status nvdb_hip_adopt_corpus(nvdb_hip_ctx* c, void* dev_rows, float* dev_scales, uint64_t n, uint32_t dim, uint32_t dtype,
                                  uint64_t global_row_base) {
  nvdb_status st = check_corpus_args(c, n, dim, dtype);
  if (st) return st;
  if (!dev_rows) return fail(c, NVDB_ERR_INVALID, "adopt_corpus: null rows");
  if (dtype == NVDB_DTYPE_I8 && !dev_scales) return fail(c, NVDB_ERR_INVALID, "adopt_corpus: int8 corpus needs per-row scales");
  HIPCHK(c, hipSetDevice(c->device));
  free_corpus(c);
  c->rows = dev_rows; c->scales = dev_scales; c->owned = false;
  c->n = n; c->dim = dim; c->dtype = dtype; c->row_base = global_row_base;
  return compute_max_norm(c);
}

nvdb_status nvdb_hip_generate_corpus(nvdb_hip_ctx* c, uint64_t seed, uint64_t n, uint32_t dim, uint32_t dtype,
                                     uint64_t global_row_base) {
  nvdb_status st = check_corpus_args(c, n, dim, dtype);
  if (st) return st;
  HIPCHK(c, hipSetDevice(c->device));
  free_corpus(c);
  const size_t bytes = static_cast<size_t>(n) * dim * bpe_of(dtype);
  const size_t pad = static_cast<size_t>(PAD_ROWS) * dim * bpe_of(dtype) + 4096;
  HIPCHK(c, hipMalloc(&c->rows, bytes + pad));
  HIPCHK(c, hipMemset(static_cast<char*>(c->rows) + bytes, 0, pad));
  c->owned = true;
  if (dtype == NVDB_DTYPE_I8) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->scales), (n + PAD_ROWS) * sizeof(float)));
    HIPCHK(c, hipMemset(c->scales + n, 0, PAD_ROWS * sizeof(float)));
  }
  c->n = n; c->dim = dim; c->dtype = dtype; c->row_base = global_row_base;
  // launch in slabs so that a single launch stays well inside the grid-size limit
  const uint64_t slab = 1ull << 24;
  for (uint64_t r0 = 0; r0 < n; r0 += slab) {
    const uint64_t cntr = std::min(slab, n - r0);
    const unsigned grid = static_cast<unsigned>((cntr + 3) / 4);
    char* rp = static_cast<char*>(c->rows) + r0 * dim * bpe_of(dtype);
    if (dtype == NVDB_DTYPE_F32) gen_rows_kernel<DT_F32><<<grid, 256, 0, c->stream>>>(seed, global_row_base + r0, cntr, dim, rp, nullptr);
    else if (dtype == NVDB_DTYPE_F16) gen_rows_kernel<DT_F16><<<grid, 256, 0, c->stream>>>(seed, global_row_base + r0, cntr, dim, rp, nullptr);
    else gen_rows_kernel<DT_I8><<<grid, 256, 0, c->stream>>>(seed, global_row_base + r0, cntr, dim, rp, c->scales + r0);
    HIPCHK(c, hipGetLastError());
  }
  return compute_max_norm(c);
}

nvdb_status nvdb_hip_corpus_info(const nvdb_hip_ctx* c, uint64_t* n, uint32_t* dim, uint32_t* dtype, uint64_t* base, float* mx) {
  if (!c) return NVDB_ERR_INVALID;
  if (n) *n = c->n;
  if (dim) *dim = c->dim;
  if (dtype) *dtype = c->dtype;
  if (base) *base = c->row_base;
  if (mx) *mx = c->max_norm;
  return c->rows ? NVDB_OK : NVDB_ERR_NO_CORPUS;
}

nvdb_status nvdb_hip_download_rows(nvdb_hip_ctx* c, uint64_t row0, uint64_t nrows, void* rows_out, float* scales_out) {
  if (!c) return NVDB_ERR_INVALID;
  if (!c->rows) return fail(c, NVDB_ERR_NO_CORPUS, "Empty base");
  if (row0 + nrows > c->n || !rows_out) return fail(c, NVDB_ERR_INVALID, "download_rows: range out of bounds");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t rb = static_cast<size_t>(c->dim) * bpe_of(c->dtype);
  HIPCHK(c, hipMemcpy(rows_out, static_cast<const char*>(c->rows) + row0 * rb, nrows * rb, hipMemcpyDeviceToHost));
  if (scales_out && c->scales) HIPCHK(c, hipMemcpy(scales_out, c->scales + row0, nrows * sizeof(float), hipMemcpyDeviceToHost));
  return NVDB_OK;
}

nvdb_status nvdb_hip_set_option(nvdb_hip_ctx* c, const char* key, int64_t value) {
  if (!c || !key) return NVDB_ERR_INVALID;
  const std::string k(key);
  if (k == "path") { if (value < 0 || value > 2) return fail(c, NVDB_ERR_INVALID, "path must be 0,1,2"); c->opt_path = value; }
  else if (k == "chunk0_rows") { if (value < 256) return fail(c, NVDB_ERR_INVALID, "chunk0_rows must be >= 256"); c->opt_chunk0 = value; }
  else if (k == "cand_cap") { if (value < 0 || value > SELECT_MAX_CAP) return fail(c, NVDB_ERR_INVALID, "cand_cap out of range"); c->opt_cap = value; }
  else if (k == "time_kernels") { c->opt_time_kernels = value ? 1 : 0; }
  else if (k == "sync_every") { if (value < 1 || (value & (value - 1))) return fail(c, NVDB_ERR_INVALID, "sync_every must be a power of two"); c->opt_sync_every = value; }
  else if (k == "sync_lead") { if (value < 1) return fail(c, NVDB_ERR_INVALID, "sync_lead must be >= 1"); c->opt_sync_lead = value; }
  else if (k == "sibling_sync") { c->opt_sibling_sync = value ? 1 : 0; }
  else if (k == "f32_shadow") { c->opt_f32_shadow = value ? 1 : 0; }
  else if (k == "q8_shadow") { c->opt_q8_shadow = value ? 1 : 0; }
#ifdef NVDB_HIP_DEV
  else if (k == "mfma16") { c->opt_mfma16 = value ? 1 : 0; }
  else if (k == "i8_wide") { c->opt_i8_wide = value ? 1 : 0; }
  else if (k == "i8_pipe") { c->opt_i8_pipe = value ? 1 : 0; }
  else if (k == "i8_waves8") { c->opt_i8_waves8 = value ? 1 : 0; }
  else if (k == "i8_mfma16") { c->opt_i8_mfma16 = value ? 1 : 0; }
  else if (k == "i8_small8") { c->opt_i8_small8 = value ? 1 : 0; }
#else
  // kernel variants that lost their A/B (32x32x16 fp16 build for batches > 128, two-plane int8 kernel, filter_i8w_kernel at 64 queries
  // per wave, 8-wave int8 build, 32x32x32 int8 build at d >= 384) live in libnvdb_hip_dev.so only; the product accepts their default values
  else if (k == "mfma16" || k == "i8_wide" || k == "i8_pipe" || k == "i8_mfma16" || k == "i8_small8") { if (!value) return fail(c, NVDB_ERR_UNSUPPORTED, k + " = 0 selects a developer-build kernel variant (libnvdb_hip_dev.so)"); }
  else if (k == "i8_waves8") { if (value) return fail(c, NVDB_ERR_UNSUPPORTED, "i8_waves8 = 1 selects a developer-build kernel variant (libnvdb_hip_dev.so)"); }
#endif

  else if (k == "i8_defer") { c->opt_i8_defer = value ? 1 : 0; }
  else if (k == "xcd_balance") { c->opt_xcd_balance = value ? 1 : 0; }
  else if (k == "i8_lo_bits") { if (value < 2 || value > 7) return fail(c, NVDB_ERR_INVALID, "i8_lo_bits must be in [2,7]"); c->opt_i8_lo_bits = value; }
  else if (k == "boot_tiles") { if (value < 0 || value > SELECT_MAX_CAP) return fail(c, NVDB_ERR_INVALID, "boot_tiles out of range"); c->opt_boot_tiles = value; }
#ifdef NVDB_HIP_DEV
  else if (k == "debug_rows") { c->dbg_rows = value < 0 ? 0 : value; }
#endif
  else if (k == "waves8") { c->opt_waves8 = value ? 1 : 0; }
  else if (k == "fuse") { c->opt_fuse = value ? 1 : 0; }
  else if (k == "zero_copy") { c->opt_zero_copy = value ? 1 : 0; }
  else if (k == "exact_mfma") { c->opt_exact_mfma = value ? 1 : 0; }
  else if (k == "exact_img") { c->opt_exact_img = value ? 1 : 0; }
  else if (k == "exact_prescan") { c->opt_exact_prescan = value ? 1 : 0; }
  else if (k == "exact_wgs") { if (value < 1 || value > 16) return fail(c, NVDB_ERR_INVALID, "exact_wgs must be in [1,16]"); c->opt_exact_wgs = value; }
  else if (k == "exact_lds") { if (value < 0 || value > 2) return fail(c, NVDB_ERR_INVALID, "exact_lds must be 0, 1 or 2"); c->opt_exact_lds = value; }
  else if (k == "time_launches") { c->opt_time_launches = value ? 1 : 0; }
  else if (k == "tile_permute") { c->opt_tile_permute = value ? 1 : 0; }
  else if (k == "rescore8") { c->opt_rescore8 = value < 0 ? 0 : (value > 2 ? 2 : value); }
  else if (k == "mfma_boot") { c->opt_mfma_boot = value ? 1 : 0; }
  else if (k == "refine_v2") { c->opt_refine_v2 = value < 0 ? 0 : (value > 2 ? 2 : value); }
  else if (k == "refine_pinned") { c->opt_refine_pinned = value ? 1 : 0; }
  else if (k == "largek_budget_mb") { if (value < 1) return fail(c, NVDB_ERR_INVALID, "largek_budget_mb must be >= 1"); c->opt_largek_budget_mb = value; }
  else if (k == "chunk_growth") { if (value != 0 && (value < 2 || value > 64)) return fail(c, NVDB_ERR_INVALID, "chunk_growth must be 0 (automatic) or in [2,64]"); c->opt_growth = value; }
  else if (k == "min_filter_batch") { if (value < 1) return fail(c, NVDB_ERR_INVALID, "min_filter_batch must be >= 1"); c->opt_min_filter_batch = value; }
  else return fail(c, NVDB_ERR_INVALID, "unknown option: " + k);
  return NVDB_OK;
}

nvdb_status nvdb_hip_get_stats(nvdb_hip_ctx* c, nvdb_hip_scan_stats* stats) {
  if (!c || !stats) return NVDB_ERR_INVALID;
  if (c->stats_lazy) {                             // the small-call path skips this read-back; do it now
    c->stats_lazy = false;
    unsigned long long tot = 0;
    if (c->last_nq) {
      HIPCHK(c, hipSetDevice(c->device));
      std::vector<uint32_t> cn(c->last_nq);
      HIPCHK(c, hipMemcpy(cn.data(), c->cnt.p, c->last_nq * 4, hipMemcpyDeviceToHost));
      for (uint32_t v : cn) tot += std::min(v, c->last_cap);
    }
    c->stats.candidates = tot;
  }
  *stats = c->stats;
  return NVDB_OK;
}

// ---- host helpers -----------------------------------------------------------------------------------
#ifdef NVDB_HIP_DEV
uint32_t nvdb_permuted_tile(uint32_t g, uint32_t n_tiles) {
  uint32_t mul, mask;
  perm_params(n_tiles, mul, mask);
  return perm_tile_raw(g, mul, mask, n_tiles);
}
#endif

void nvdb_synth_rows_f32(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, float* out) {
  std::vector<int32_t> raw(dim);
  for (uint64_t r = 0; r < nrows; ++r) {
    const uint32_t key = synth_row_key(seed, row0 + r);
    uint64_t ss = 0;
    for (uint32_t c = 0; c < dim; ++c) { raw[c] = synth_raw(key, c); ss += static_cast<uint64_t>(static_cast<int64_t>(raw[c]) * raw[c]); }
    const double inv = synth_inv_norm(ss);
    for (uint32_t c = 0; c < dim; ++c) out[r * dim + c] = synth_elem(raw[c], inv);
  }
}

void nvdb_f32_to_f16(const float* src, uint16_t* dst, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) dst[i] = f32_to_f16_rne(src[i]);
}

void nvdb_quantize_i8_rows(const float* rows, uint64_t nrows, uint32_t dim, int8_t* out, float* scales) {
  for (uint64_t r = 0; r < nrows; ++r) scales[r] = quantize_i8_row(rows + r * dim, dim, out + r * dim);
}

}  // extern "C"
