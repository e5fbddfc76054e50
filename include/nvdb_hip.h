/*
 * nvdb_hip.h -- C ABI of libnvdb_hip.so: the MI355X (gfx950) replacement for nano-vectordb's
 * flat-scan hot path and its exact-L2 refine stage.
 *
 * Plain C, plain pointers and sizes, no C++/torch types, no exceptions.  Every call returns an
 * nvdb_status (0 = OK); nvdb_hip_last_error(ctx) holds the message of the last failure.  The host
 * C++ wrappers (nano-vectordb_amd/host/include/nvdb/) translate status != 0 into
 * std::runtime_error, the reference's flat-path convention (src/flat_index.cpp:17).
 *
 * Each entry point names the reference interface it stands in for (file:line into the reference
 * tree).  INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Threading: a context is single-owner (one host thread at a time), like the reference's
 * process-global CUDA state (src/cuda_refine.cu:26-90); different contexts are independent, which
 * is how the multi-GPU path runs one context per device / per process.
 */
#ifndef NVDB_HIP_H
#define NVDB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVDB_HIP_ABI_VERSION 3

/* dtype codes == VecbinHeader::dtype (include/nvdb/vecbin_format.h:10-14) */
enum { NVDB_DTYPE_F32 = 1, NVDB_DTYPE_F16 = 2, NVDB_DTYPE_I8 = 3 };

typedef enum nvdb_status {
  NVDB_OK = 0,
  NVDB_ERR_INVALID = 1,      /* bad argument (null pointer, dim/dtype mismatch, k too large ...) */
  NVDB_ERR_HIP = 2,          /* a HIP runtime call failed (no device, out of memory ...)         */
  NVDB_ERR_UNSUPPORTED = 3,  /* shape outside what the kernels implement                          */
  NVDB_ERR_NO_CORPUS = 4,    /* search/refine before a corpus is resident ("Empty base")         */
  NVDB_ERR_INTERNAL = 5      /* a self-check failed (filter error bound violated, overflow ...)   */
} nvdb_status;

/* The flat path takes ANY k (clamped to the row count like the reference, src/flat_index.cpp:24): k <= 64 runs on
 * wavefront-resident lists / the MFMA filter, 65..1024 on the filter path where it is eligible (MFMA bootstrap, a dtype / dim
 * with a bootstrap build, >= 512*k rows), every other k > 64 on the any-k path (scores -> radix select -> sort).
 * Largest K of the refine path = the reference's NVDB_CUDA_KMAX = 64 (src/cuda_refine.cu:12-14, 858-862). */
#define NVDB_HIP_REFINE_KMAX 64

typedef struct nvdb_hip_ctx nvdb_hip_ctx;

/* Field-for-field mirror of nvdb::CudaRefineTiming (include/nvdb/cuda_refine.h:7-22); the flat
 * path fills h2d/kernel/d2h/total and leaves the rest zero.  dbg_* stay zero: the in-kernel
 * clock64 sampling (src/cuda_refine.cu:416-418, 495-500) is replaced by rocprofv3. */
typedef struct nvdb_hip_timing {
  float h2d_ms, kernel_ms, d2h_ms, total_ms;
  uint32_t threads;
  uint32_t nwarps;          /* number of 64-lane wavefronts per workgroup */
  uint32_t K;
  uint32_t R;
  size_t shmem_bytes;       /* LDS bytes per workgroup of the dominant kernel */
  uint32_t dbg_q;
  double dbg_dist_cycles_avg, dbg_write_cycles_avg, dbg_merge_cycles_avg;
  double dbg_dist_pct, dbg_write_pct, dbg_merge_pct;
} nvdb_hip_timing;

/* What the last flat search did (for tests, bench.py and the roofline arithmetic). */
typedef struct nvdb_hip_scan_stats {
  uint32_t path;               /* 1 = exact fp32 scan, 2 = MFMA filter + exact rescore (k <= 1024), 3 = any-k (k > 64 off the filter path) */
  uint32_t chunks;             /* corpus chunks (kernel launches of the dominant kernel)          */
  uint64_t rows_scanned;       /* rows x query-tiles streamed by the dominant kernel              */
  uint64_t candidates;         /* (query,row) pairs that reached the exact rescore                */
  uint32_t overflow_queries;   /* queries whose candidate list overflowed (re-run on path 1)      */
  uint32_t bound_violations;   /* |filter - exact| > bound seen by the rescore (must be 0)        */
  float    filter_kernel_ms;   /* sum of hipEvent times of the dominant kernel's launches (host API with a timing struct and option "time_launches" = 1; else 0) */
  float    other_kernel_ms;    /* prep + select + rescore + merge                                  */
  uint32_t i8_stage1_tiles;    /* int8 two-stage kernel: (wave, tile) pairs that went past the hi-plane quick test */
  uint32_t i8_stage2_blocks;   /* ... 32-query blocks for which the lo plane was multiplied after all           */
  uint32_t sticky_overflow;    /* nvdb_hip_search_check: a list / log overflow in ANY device-API search since the last check */
  uint32_t sticky_violations;  /* ... bound violations summed over those searches (both cleared by the check)    */
} nvdb_hip_scan_stats;

/* ---------------------------------------------------------------------------------------------
 * library / device
 * ------------------------------------------------------------------------------------------- */
int nvdb_hip_abi_version(void);
/* Number of visible HIP devices, or -1 when the runtime cannot initialise (no GPU). */
int nvdb_hip_device_count(void);

/* One context per GPU.  Replaces the reference's implicit device 0 + file-static caches
 * (src/cuda_refine.cu:26-90, :951).  Owns a HIP stream, the resident corpus and a grow-only
 * workspace (reference: ensure_workspace, src/cuda_refine.cu:144-176). */
nvdb_status nvdb_hip_create(int device_ordinal, nvdb_hip_ctx** out_ctx);
void nvdb_hip_destroy(nvdb_hip_ctx* ctx);
const char* nvdb_hip_last_error(const nvdb_hip_ctx* ctx);   /* ctx may be NULL: last create() error */

/* ---------------------------------------------------------------------------------------------
 * corpus residency -- replaces ensure_base_on_gpu (src/cuda_refine.cu:179-204) and the
 * FlatIndex(const VectorDataset*) constructor's borrowed pointer (include/nvdb/flat_index.h:11-16)
 * ------------------------------------------------------------------------------------------- */

/* Copy a row-major corpus into HBM (staged through pinned chunks; `rows` may be an mmap).
 * rows: n*dim elements of dtype; scales: n floats, required iff dtype == I8 (the vecbin layout
 * stores them after the payload, src/vector_dataset.cpp:86-87).  global_row_base is added to
 * every returned id (row-sharding across GPUs: shard g holds rows [base, base+n)). */
nvdb_status nvdb_hip_upload_corpus(nvdb_hip_ctx* ctx, const void* rows, const float* scales,
                                   uint64_t n, uint32_t dim, uint32_t dtype, uint64_t global_row_base);

/* Use a corpus that is already in HBM (e.g. a torch tensor); not copied, not freed. */
nvdb_status nvdb_hip_adopt_corpus(nvdb_hip_ctx* ctx, void* dev_rows, float* dev_scales,
                                  uint64_t n, uint32_t dim, uint32_t dtype, uint64_t global_row_base);

/* Generate the synthetic corpus rows [global_row_base, global_row_base+n) directly in HBM
 * (counter-based rows keyed (seed,row,col), L2-normalised, then f16 = RNE /
 * int8 = the reference quantiser's rule).  Bit-identical to nvdb_synth_rows() on the CPU. */
nvdb_status nvdb_hip_generate_corpus(nvdb_hip_ctx* ctx, uint64_t seed, uint64_t n, uint32_t dim,
                                     uint32_t dtype, uint64_t global_row_base);

nvdb_status nvdb_hip_corpus_info(const nvdb_hip_ctx* ctx, uint64_t* n, uint32_t* dim, uint32_t* dtype,
                                 uint64_t* global_row_base, float* max_row_norm);

/* Copy rows [row0,row0+nrows) (local indices) back to the host; scales_out may be NULL. */
nvdb_status nvdb_hip_download_rows(nvdb_hip_ctx* ctx, uint64_t row0, uint64_t nrows, void* rows_out,
                                   float* scales_out);

/* ---------------------------------------------------------------------------------------------
 * flat scan: batched query x corpus dot products + fused top-k
 * ------------------------------------------------------------------------------------------- */

/* Exact top-k by dot product for nq fp32 queries [nq][dim] (host memory).
 * Stands in for FlatIndex::search_topk_dot / FlatIndexOMP::search_topk_dot called nq times
 * (src/flat_index.cpp:16-48, src/flat_index_omp.cpp:16-85) and for the bench-side batched loop
 * batched_scan_omp_or_st (apps/nvdb_bench.cpp:47-159).
 *   out_ids[nq][k], out_scores[nq][k]: best first; ids are global (base + local row).
 *   k is clamped to the resident row count like the reference (flat_index.cpp:24); *out_k_eff
 *   (optional) receives min(k, n); slots >= k_eff hold id UINT64_MAX / score -inf.
 *   Scores are bit-identical to the reference's AVX2 kernels (simd_dot.cpp:26-49, 102-124,
 *   160-199); ties are ordered (score desc, id asc).
 *   k == 0 -> NVDB_OK, nothing written (flat_index.cpp:18).  Any k > 0 is accepted. */
nvdb_status nvdb_hip_search_batch(nvdb_hip_ctx* ctx, const float* queries, uint32_t nq, uint32_t k,
                                  uint64_t* out_ids, float* out_scores, uint32_t* out_k_eff,
                                  nvdb_hip_timing* timing);

/* Same, with queries and outputs already in HBM and all work enqueued on `hip_stream`
 * (a hipStream_t; NULL = the context's own NON-BLOCKING stream -- note that the legacy default stream also has the
 * handle NULL: a caller whose other work is on the default stream must pass an explicit stream or synchronise the
 * device, the context's stream is not ordered with it).  Returns after enqueueing; no host sync, so
 * overflow / bound self-checks are reported by nvdb_hip_search_check() after the caller has
 * synchronised the stream; the check covers EVERY search enqueued since the previous check (sticky flags), its
 * per-query detail and statistics describe the last one.  This is the form the multi-GPU path uses before its all-gather. */
nvdb_status nvdb_hip_search_batch_dev(nvdb_hip_ctx* ctx, const float* dev_queries, uint32_t nq, uint32_t k,
                                      uint64_t* dev_out_ids, float* dev_out_scores, void* hip_stream);
nvdb_status nvdb_hip_search_check(nvdb_hip_ctx* ctx, nvdb_hip_scan_stats* stats);
/* Statistics of the last nvdb_hip_search_batch() (summed over its sub-batches). */
nvdb_status nvdb_hip_get_stats(nvdb_hip_ctx* ctx, nvdb_hip_scan_stats* stats);

/* Merge per-shard top-k lists (e.g. after an RCCL all-gather): in[s][nq][k] -> out[nq][k] with
 * the same (score desc, id asc) order.  Device buffers, enqueued on hip_stream.  Any k the flat path accepts (the reference
 * bounds k by N only, src/flat_index.cpp:24): up to nshards*k = 4096 the lists of one query are ranked in LDS; longer ones are
 * merged by binary searches and must arrive sorted best-first with their padding (id ~0, -inf) last, as every search emits them. */
nvdb_status nvdb_hip_merge_topk_dev(nvdb_hip_ctx* ctx, const uint64_t* dev_ids, const float* dev_scores,
                                    uint32_t nshards, uint32_t nq, uint32_t k, uint64_t* dev_out_ids,
                                    float* dev_out_scores, void* hip_stream);
/* Same with explicit byte strides between consecutive shards' blocks: lets each rank all-gather ONE packed buffer
 * [ids (nq*k*8 bytes) | scores (nq*k*4 bytes)] and merge it in place (stride = nq*k*12 for both). */
nvdb_status nvdb_hip_merge_topk_strided_dev(nvdb_hip_ctx* ctx, const uint64_t* dev_ids, const float* dev_scores,
                                            size_t stride_ids_bytes, size_t stride_scores_bytes, uint32_t nshards,
                                            uint32_t nq, uint32_t k, uint64_t* dev_out_ids, float* dev_out_scores,
                                            void* hip_stream);
/* Host version of the same merge (no GPU needed). */
nvdb_status nvdb_merge_topk_host(const uint64_t* ids, const float* scores, uint32_t nshards, uint32_t nq,
                                 uint32_t k, uint64_t* out_ids, float* out_scores);

/* ---------------------------------------------------------------------------------------------
 * device group: ONE process driving several GPUs of a node (the reference has no multi-GPU path; the north star's
 * "corpus row-sharded across the GPUs, RCCL all-gather of per-shard partial top-k over xGMI").  Shard g holds the
 * contiguous rows [g*n/G, (g+1)*n/G) on devices[g] with global ids.  A batch is searched on every shard
 * (nvdb_hip_search_batch_dev, one host thread per device for the enqueue), each device's [ids | scores] block
 * (nq*k*12 bytes) is all-gathered with ONE ncclAllGather per device inside ncclGroupStart/End on the devices' streams,
 * and the k-way merge runs on devices[0] (same (score desc, id asc) order: the result equals the unsharded search bit
 * for bit).  RCCL is bound with dlopen at group creation; when it cannot serve the list (a device named twice, RCCL
 * absent, NVDB_GROUP_NO_RCCL=1) the exchange is G peer copies into devices[0] and the same merge kernel.
 * A shard whose self-check trips (list overflow, non-finite query) sends the sub-batch through the per-shard host API
 * (which retries / falls back by itself) and a host-side merge.
 * NVDB_GROUP_RCCL_LIB=<path> (read at group creation) binds the six RCCL entry points from another library instead: the tests
 * drive the RCCL branch with several ranks on one device through a loopback stand-in (tests/loopback_rccl).
 * One process per GPU (bench.py, torch.distributed) uses nvdb_hip_search_batch_dev + the caller's all-gather +
 * nvdb_hip_merge_topk_strided_dev instead -- same kernels, same packed layout.
 * ------------------------------------------------------------------------------------------- */
typedef struct nvdb_hip_group nvdb_hip_group;
typedef struct nvdb_hip_group_stats {
  uint32_t shards;                /* G */
  uint32_t exchange;              /* 1 = RCCL all-gather, 0 = peer copies into devices[0] */
  uint32_t host_merge_fallbacks;  /* sub-batches of the last call that went through the host merge */
  uint64_t bytes_per_rank;        /* packed block one rank contributes per sub-batch: nq*k*12 */
} nvdb_hip_group_stats;

nvdb_status nvdb_hip_group_create(const int* devices, uint32_t n_devices, nvdb_hip_group** out_group);
void nvdb_hip_group_destroy(nvdb_hip_group* group);
const char* nvdb_hip_group_last_error(const nvdb_hip_group* group);          /* NULL: last create() error */
uint32_t nvdb_hip_group_size(const nvdb_hip_group* group);
/* the shard's own context (options, statistics, corpus_info); owned by the group */
nvdb_hip_ctx* nvdb_hip_group_ctx(nvdb_hip_group* group, uint32_t shard);
/* 1 = RCCL, 0 = peer copies; *why (optional) names the reason */
int nvdb_hip_group_exchange(const nvdb_hip_group* group, const char** why);
/* row-shard a host corpus / the synthetic corpus over the group's devices (same arguments as the per-device calls) */
nvdb_status nvdb_hip_group_upload_corpus(nvdb_hip_group* group, const void* rows, const float* scales, uint64_t n,
                                         uint32_t dim, uint32_t dtype);
nvdb_status nvdb_hip_group_generate_corpus(nvdb_hip_group* group, uint64_t seed, uint64_t n, uint32_t dim, uint32_t dtype);
nvdb_status nvdb_hip_group_set_option(nvdb_hip_group* group, const char* key, int64_t value);   /* every shard */
/* Same contract as nvdb_hip_search_batch (host queries in, [nq][k] global ids + scores out, any nq, k clamped). */
nvdb_status nvdb_hip_group_search_batch(nvdb_hip_group* group, const float* queries, uint32_t nq, uint32_t k,
                                        uint64_t* out_ids, float* out_scores, uint32_t* out_k_eff,
                                        nvdb_hip_group_stats* stats);

/* Tunables (defaults are what bench.py measures; the table with meanings is in INTEGRATION.md section 4b):
 * "path" (0 auto, 1 exact, 2 mfma-filter), "chunk0_rows", "chunk_growth", "cand_cap", "min_filter_batch", "mfma_boot", "waves8",
 * "sibling_sync", "sync_every", "sync_lead", "tile_permute", "f32_shadow" (set before the upload), "exact_mfma" (exact scores on the fp32
 * matrix cores), "exact_lds", "i8_defer", "i8_lo_bits", "boot_tiles", "xcd_balance", "rescore8", "refine_v2", "refine_pinned" (reference CUDA_PINNED:
 * pinned host staging in nvdb_hip_refine_l2_topk), "largek_budget_mb" (HBM for the any-k path's score matrix), "time_kernels" (1: start /
 * stop events attached to every launch of the dominant kernel, read by nvdb_hip_collect_kernel_times), "time_launches" (the same for one host-API
 * call with a timing struct -> stats.filter_kernel_ms).  "mfma16", "i8_wide", "i8_pipe", "i8_waves8", "i8_mfma16", "i8_small8" select kernel
 * variants that exist in libnvdb_hip_dev.so only: the product accepts their default values (1, 1, 1, 0, 1) and returns
 * NVDB_ERR_UNSUPPORTED for the others.  Unknown key -> NVDB_ERR_INVALID. */
nvdb_status nvdb_hip_set_option(nvdb_hip_ctx* ctx, const char* key, int64_t value);

/* With "time_kernels" = 1: sum of the hipEvent durations of the dominant (filter) kernel's launches
 * since the last call, with their algorithmic flops (2 * queries * rows * dim) and corpus bytes
 * (rows * dim * bytes/elem).  Synchronises on the recorded events.  bench.py's roofline uses it. */
nvdb_status nvdb_hip_collect_kernel_times(nvdb_hip_ctx* ctx, uint32_t* launches, double* total_ms,
                                          double* total_flops, double* total_bytes);

/* ---------------------------------------------------------------------------------------------
 * exact-L2 refine (rerank of R candidates per query) -- replaces nvdb::cuda_l2_topk_batch
 * (include/nvdb/cuda_refine.h:25-38, src/cuda_refine.cu:839-1173) on the resident corpus
 * ------------------------------------------------------------------------------------------- */

/* queries [Q][dim] f32, cand_ids [Q][R] u32 local row ids (0xFFFFFFFF or >= n are skipped,
 * cuda_refine.cu:437).  out_ids [Q][K] ascending by squared L2 distance, padded with 0xFFFFFFFF;
 * out_dist [Q][K] padded with 1e30f, or NULL for ids only (CUDA_RETURN_DIST=0, :876).
 * Distances use the reference kernel's fp32 order (cuda_refine.cu:326-382 for f16 rows,
 * :383-392 for f32 rows -- the latter is implemented properly; the reference never launches it,
 * :1055-1085).  Corpus dtype must be F16 or F32 (apps/nvdb_ivf_eval.cpp:519-525).
 * K == 0 || Q == 0 || R == 0 -> NVDB_OK with zeroed timing (:853-857); K > 64 -> INVALID (:858-862). */
nvdb_status nvdb_hip_refine_l2_topk(nvdb_hip_ctx* ctx, const float* queries, const uint32_t* cand_ids,
                                    uint32_t Q, uint32_t R, uint32_t K, uint32_t* out_ids, float* out_dist,
                                    nvdb_hip_timing* timing);
nvdb_status nvdb_hip_refine_l2_topk_dev(nvdb_hip_ctx* ctx, const float* dev_queries, const uint32_t* dev_cand_ids,
                                        uint32_t Q, uint32_t R, uint32_t K, uint32_t* dev_out_ids,
                                        float* dev_out_dist, void* hip_stream);

/* ---------------------------------------------------------------------------------------------
 * host-side helpers that define the corpus bits (no GPU needed)
 * ------------------------------------------------------------------------------------------- */

/* Synthetic rows [row0,row0+nrows) as fp32 (the generator nvdb_hip_generate_corpus runs on device). */
void nvdb_synth_rows_f32(uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t dim, float* out);
/* fp32 -> IEEE half, round-to-nearest-even (tools/nvdb_convert_f16.cpp:99-107, the F16C path). */
void nvdb_f32_to_f16(const float* src, uint16_t* dst, uint64_t n);
/* per-row int8 quantisation (apps/nvdb_quantize_i8.cpp:12-16, 71-80). */
void nvdb_quantize_i8_rows(const float* rows, uint64_t nrows, uint32_t dim, int8_t* out, float* scales);

#ifdef __cplusplus
}
#endif
#endif /* NVDB_HIP_H */
