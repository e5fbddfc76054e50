// nvdb_refine.cpp -- exact-L2 refine rerank (reference src/cuda_refine.cu:839-1173; kernels: kernels_refine.h).
#include "nvdb_ctx.h"
#include "kernels_refine.h"

extern "C" {

// ---- refine -----------------------------------------------------------------------------------------
static nvdb_status refine_args(nvdb_hip_ctx* c, const void* q, const void* cand, uint32_t K, const void* out_ids) {
  if (!c) return NVDB_ERR_INVALID;
  if (!c->rows || c->n == 0) return fail(c, NVDB_ERR_NO_CORPUS, "Empty base");
  if (c->dtype != NVDB_DTYPE_F16 && c->dtype != NVDB_DTYPE_F32)
    return fail(c, NVDB_ERR_UNSUPPORTED, "refine supports base dtype fp16/fp32 only");
  if (K > NVDB_HIP_REFINE_KMAX) return fail(c, NVDB_ERR_INVALID, "refine: K not supported (K<=64)");
  if (!q || !cand || !out_ids) return fail(c, NVDB_ERR_INVALID, "refine: null pointer");
  return NVDB_OK;
}

static nvdb_status launch_refine(nvdb_hip_ctx* c, hipStream_t s, const float* dq, const uint32_t* dc, uint32_t Q, uint32_t R,
                                 uint32_t K, uint32_t* doi, float* dod) {
  const bool al = aligned_rows(c->dtype, c->dim);
  // v3 (whole rows per request, four lanes per row): fp16 rows of 512 / 1024 / 1536 bytes
  if (c->opt_refine_v2 >= 2 && c->dtype == NVDB_DTYPE_F16 && refine3_dim(c->dim)) {
#define NVDB_REFINE3(D)                                                                                                        \
    {                                                                                                                          \
      constexpr size_t lds = static_cast<size_t>(REFINE3_WAVES) * refine3_slot_bytes<D>();                                      \
      const void* fn = reinterpret_cast<const void*>(refine_l2_rows_kernel<D>);                                                 \
      if (!c->lds_attr_set.count(fn)) {                                                                                        \
        HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                 \
        c->lds_attr_set.insert(fn);                                                                                            \
      }                                                                                                                        \
      refine_l2_rows_kernel<D><<<Q, 64 * REFINE3_WAVES, lds, s>>>(c->rows, c->n, dq, dc, R, K, doi, dod);                       \
    }
    if (c->dim == 768) NVDB_REFINE3(768) else if (c->dim == 512) NVDB_REFINE3(512) else if (c->dim == 384) NVDB_REFINE3(384) else NVDB_REFINE3(256)
#undef NVDB_REFINE3
    HIPCHK(c, hipGetLastError());
    return NVDB_OK;
  }
  // v2 (coalesced gather through LDS): whole 16-byte steps only (f16: dim % 8 == 0, f32: dim % 4 == 0)
  if (al && c->opt_refine_v2 && static_cast<uint64_t>(c->dim) * bpe_of(c->dtype) >= 256) {
    constexpr size_t lds = 4 * 2 * 64 * 256;
    const void* fn = c->dtype == NVDB_DTYPE_F16 ? reinterpret_cast<const void*>(refine_l2_lds_kernel<DT_F16>)
                                                : reinterpret_cast<const void*>(refine_l2_lds_kernel<DT_F32>);
    if (!c->lds_attr_set.count(fn)) {
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      c->lds_attr_set.insert(fn);
    }
    if (c->dtype == NVDB_DTYPE_F16) refine_l2_lds_kernel<DT_F16><<<Q, 256, lds, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
    else refine_l2_lds_kernel<DT_F32><<<Q, 256, lds, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
    HIPCHK(c, hipGetLastError());
    return NVDB_OK;
  }
  if (c->dtype == NVDB_DTYPE_F16) {
    if (al) refine_l2_kernel<DT_F16, true><<<Q, 256, 0, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
    else refine_l2_kernel<DT_F16, false><<<Q, 256, 0, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
  } else {
    if (al) refine_l2_kernel<DT_F32, true><<<Q, 256, 0, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
    else refine_l2_kernel<DT_F32, false><<<Q, 256, 0, s>>>(c->rows, c->n, c->dim, dq, dc, R, K, doi, dod);
  }
  HIPCHK(c, hipGetLastError());
  return NVDB_OK;
}

nvdb_status nvdb_hip_refine_l2_topk_dev(nvdb_hip_ctx* c, const float* dq, const uint32_t* dc, uint32_t Q, uint32_t R, uint32_t K,
                                        uint32_t* doi, float* dod, void* hip_stream) {
  if (c && (K == 0 || Q == 0 || R == 0)) return NVDB_OK;
  nvdb_status st = refine_args(c, dq, dc, K, doi);
  if (st) return st;
  HIPCHK(c, hipSetDevice(c->device));
  return launch_refine(c, hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream, dq, dc, Q, R, K, doi, dod);
}

nvdb_status nvdb_hip_refine_l2_topk(nvdb_hip_ctx* c, const float* queries, const uint32_t* cand_ids, uint32_t Q, uint32_t R,
                                    uint32_t K, uint32_t* out_ids, float* out_dist, nvdb_hip_timing* timing) {
  if (timing) std::memset(timing, 0, sizeof(*timing));
  if (c && (K == 0 || Q == 0 || R == 0)) return NVDB_OK;       // cuda_refine.cu:853-857
  nvdb_status st = refine_args(c, queries, cand_ids, K, out_ids);
  if (st) return st;
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const size_t qb = static_cast<size_t>(Q) * c->dim * 4, cb = static_cast<size_t>(Q) * R * 4, ob = static_cast<size_t>(Q) * K * 4;
  if ((st = ensure(c, c->rq, qb))) return st;
  if ((st = ensure(c, c->rcand, cb))) return st;
  if ((st = ensure(c, c->rout_ids, ob))) return st;
  if ((st = ensure(c, c->rout_dist, ob))) return st;
  hipEvent_t e0 = get_event(c, 56), e1 = get_event(c, 57), e2 = get_event(c, 58), e3 = get_event(c, 59);
  // optional pinned staging (reference: CUDA_PINNED, src/cuda_refine.cu:875, 902-914): inputs are packed into pinned host
  // buffers BEFORE the timed region, the asynchronous copies then run at the link's rate instead of through the runtime's
  // pageable bounce buffers; results come back into pinned memory and are copied out after the synchronisation.
  const void* h_q = queries;
  const void* h_c = cand_ids;
  void* h_oi = out_ids;
  void* h_od = out_dist;
  if (c->opt_refine_pinned) {
    const size_t need = qb + cb + 2 * ob;
    if (c->rpinned_bytes < need) {
      if (c->rpinned) (void)hipHostFree(c->rpinned);
      c->rpinned = nullptr; c->rpinned_bytes = 0;
      HIPCHK(c, hipHostMalloc(&c->rpinned, need, hipHostMallocDefault));
      c->rpinned_bytes = need;
    }
    char* pin = static_cast<char*>(c->rpinned);
    std::memcpy(pin, queries, qb);
    std::memcpy(pin + qb, cand_ids, cb);
    h_q = pin; h_c = pin + qb; h_oi = pin + qb + cb; h_od = pin + qb + cb + ob;
  }
  HIPCHK(c, hipEventRecord(e0, s));
  HIPCHK(c, hipMemcpyAsync(c->rq.p, h_q, qb, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipMemcpyAsync(c->rcand.p, h_c, cb, hipMemcpyHostToDevice, s));
  HIPCHK(c, hipEventRecord(e1, s));
  if ((st = launch_refine(c, s, static_cast<const float*>(c->rq.p), static_cast<const uint32_t*>(c->rcand.p), Q, R, K,
                          static_cast<uint32_t*>(c->rout_ids.p), out_dist ? static_cast<float*>(c->rout_dist.p) : nullptr)))
    return st;
  HIPCHK(c, hipEventRecord(e2, s));
  HIPCHK(c, hipMemcpyAsync(h_oi, c->rout_ids.p, ob, hipMemcpyDeviceToHost, s));
  if (out_dist) HIPCHK(c, hipMemcpyAsync(h_od, c->rout_dist.p, ob, hipMemcpyDeviceToHost, s));
  HIPCHK(c, hipEventRecord(e3, s));
  HIPCHK(c, hipStreamSynchronize(s));
  if (c->opt_refine_pinned) {
    std::memcpy(out_ids, h_oi, ob);
    if (out_dist) std::memcpy(out_dist, h_od, ob);
  }
  if (timing) {
    (void)hipEventElapsedTime(&timing->h2d_ms, e0, e1);
    (void)hipEventElapsedTime(&timing->kernel_ms, e1, e2);
    (void)hipEventElapsedTime(&timing->d2h_ms, e2, e3);
    timing->total_ms = timing->h2d_ms + timing->kernel_ms + timing->d2h_ms;
    const bool rows_kernel = c->opt_refine_v2 >= 2 && c->dtype == NVDB_DTYPE_F16 && refine3_dim(c->dim);
    timing->threads = rows_kernel ? 64 * REFINE3_WAVES : 256; timing->nwarps = rows_kernel ? REFINE3_WAVES : 4; timing->K = K; timing->R = R;
    timing->shmem_bytes = rows_kernel ? static_cast<size_t>(REFINE3_WAVES) * (c->dim == 768 ? refine3_slot_bytes<768>() : c->dim == 512 ? refine3_slot_bytes<512>() : c->dim == 384 ? refine3_slot_bytes<384>() : refine3_slot_bytes<256>())
                                      : (c->opt_refine_v2 ? 4 * 2 * 64 * 256 : 4 * 64 * 8 + 16);
  }
  return NVDB_OK;
}

}  // extern "C"
