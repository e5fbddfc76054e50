// kernels_filter_i8s.h -- the int8 two-stage filter (batches > 128) on the 16x16x64 integer MFMA.
//
// Same algorithm, same LDS image, same loader and same logged survivors as filter_i8p_kernel<.., DEFER = false> (kernels_filter.h:
// hi plane of 64 queries per wave stationary in AGPRs, 64-row tiles in three stages, first-stage test in the shadow of the
// other row block's MFMAs on biased accumulators, survivors logged and finished exactly after the stream by
// verify_and_scatter_i8).  What changes is the matrix instruction: v_mfma_i32_16x16x64_i8 instead of v_mfma_i32_32x32x32_i8.
// Same operations per clock on paper (MI355X_MICROARCH.md, Matrix cores: "I8 32x32x32 / 16x16x64: the cycles of the BF16 form
// of the same MxN at 2x the K"), but the chip holds a higher clock on the 16x16 shape under load (same guide, DVFS give-back
// item 7; the fp16 kernel gained 3 % from its 16x16x32 build, profiles/r01b_mfma_shape_ab.txt), and this kernel sits at the
// power wall.
//
// Mapping per wave (lane = (x15 = lane % 16, g4 = lane / 16)):
//   B fragment (nb, s): queries 16 nb + x15 of the wave's 64, bytes [64 s + 16 g4, +16) of their hi plane   -> 4 x DIM/64 fragments
//   A fragment (rb, s): row 16 rb + x15 of the tile, chunk 4 s + g4, stored at position chunk ^ x15 of the row image
//                       (the image filter_i8p_kernel's loader writes: conflict-free for this read pattern too)
//   D (rb, nb): lane holds rows 16 rb + 4 g4 + j (j = 0..3) of query 16 nb + x15: 4 registers; a 32-row block = 2 x 4 tiles
//   => a lane tests 32 values per 32-row block as before, but against 4 thresholds (its 4 queries) and 8 row scales.
#pragma once
#include "kernels_filter.h"

namespace nvdbhip {

typedef int intx4_t __attribute__((ext_vector_type(4)));

#define NVDB_MFMA_I8S_FROM(acc, a, b, c0) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c0))
#define NVDB_MFMA_I8S_ACC(acc, a, b) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))

// VAR (STAMP builds only; wrong results): 1 = no test, no logging (structure alone), 2 = test, nothing logged,
// 3 = structure alone without the in-loop LDS-DMA issue (the stages keep the first tiles), 4 = structure alone without the A-fragment LDS reads.
// WPB = 8: the same 256 queries per workgroup on 8 waves of 32 (two waves per SIMD, 128 + 128 registers each): a wave's LDS-DMA issue,
// ring priming and barrier waits run beside its SIMD partner's MFMAs.
template <int DIM, bool SYNC = false, bool STAMP = false, int RING = 6, int VAR = 0, int WPB = 4>
__global__ __launch_bounds__(64 * WPB, 1) void filter_i8s_kernel(
    const signed char* __restrict__ rows, const float* __restrict__ scales, uint32_t row_lo, uint32_t row_hi,
    const signed char* __restrict__ qhi, const signed char* __restrict__ qlo, uint32_t nq, uint32_t QT,
    const float* __restrict__ thr, const float* __restrict__ qscale, const float* __restrict__ qinv,
    const float* __restrict__ qdelta, Hit* __restrict__ hitlog, ScatterArgs sa, uint32_t* __restrict__ prog,
    uint32_t sync_mask, uint32_t sync_lead, uint32_t* __restrict__ stage_counts) {
  static_assert(WPB == 4 || WPB == 8, "4 waves x 64 queries or 8 waves x 32 queries");
  constexpr bool NOTEST = VAR == 1 || VAR == 3 || VAR == 4;
  constexpr int NB = 16 / WPB, MB = 2;                       // blocks of 16 queries per wave; 2 blocks of 32 rows per tile
  constexpr int NV = 8 * NB;                                 // values a lane tests per 32-row block
  constexpr int KS = DIM / 64;                               // k-steps of 64 bytes
  constexpr int ROW_BYTES = DIM;
  constexpr int TROWS = FILTER_ROWS * MB;
  constexpr int NSTAGE = 3;
  constexpr int DATA_BYTES = TROWS * ROW_BYTES;
  constexpr int STAGE_BYTES = DATA_BYTES + WPB * 256;        // the tile + per-wave 256-byte copies of its 64 row scales
  constexpr int PIECES = DATA_BYTES / 1024, PPW = PIECES / WPB;
  constexpr int CHUNKS_PER_ROW = ROW_BYTES / 16;
  constexpr int NSLOT = 2 * NB * KS;                         // MFMAs (16 cycles each) per 32-row block
  constexpr int NFRAG = 2 * KS;                              // A fragments per 32-row block
  static_assert(DIM % 128 == 0 && DIM <= 768, "row stride multiple of 128 bytes (swz_chunk); 64 queries x DIM bytes = 192 AGPRs at most");
  static_assert(PIECES % WPB == 0 && PPW % 2 == 0 && NSTAGE * STAGE_BYTES <= 160 * 1024 && PPW + 1 < 64 && NFRAG >= RING, "shape / LDS / vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x15 = lane & 15, g4 = lane >> 4;
  const uint32_t wave_gid = blockIdx.x * WPB + wave;

  const uint32_t nwg = gridDim.x, b = blockIdx.x;
  const uint32_t S = nwg / QT;
  uint32_t stream, qt;
  if ((nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0) { const uint32_t xcd = b & 7u, i = b >> 3; qt = i % QT; stream = (i / QT) * 8u + xcd; }
  else { qt = b % QT; stream = b / QT; }
  const uint32_t tiles_total = (row_hi - row_lo) / TROWS;
  const bool xcd_map = (nwg & 7u) == 0 && ((nwg >> 3) % QT) == 0;
  uint32_t t_lo, t_hi;
  stream_tile_range(tiles_total, S, stream, xcd_map, sa.xcdw, t_lo, t_hi);
  const uint32_t NT = t_hi - t_lo;
  if (NT == 0) return;

  // stationary operand: hi plane of this wave's 4 blocks of 16 queries, all of K, in AGPRs
  const uint32_t qbase = qt * 256u + wave * (16u * NB);
  float4_t bq[NB * KS];
#pragma unroll
  for (int f = 0; f < NB * KS; ++f) {
    const int nb = f / KS, s = f % KS;
    bq[f] = *reinterpret_cast<const float4_t*>(qhi + static_cast<uint64_t>(qbase + nb * 16 + x15) * DIM + 64 * s + 16 * g4);
  }
#pragma unroll
  for (int f = 0; f < NB * KS; ++f) asm volatile("" ::"a"(bq[f]));
  // biased accumulators (see filter_i8p_kernel): every tile starts at the bits of 2^23, |H| < 2^23, so the int32 sum read AS A FLOAT is
  // 2^23 + H (H >= 0) or 2^23 + H / 2 (H < 0, only ever over-estimated): the test is one v_fma per value
  intx4_t bias0;
#pragma unroll
  for (int r = 0; r < 4; ++r) bias0[r] = I8_ACC_BIAS;
  asm volatile("" : "+v"(bias0));
  uint32_t qid[NB];
  float t1q[NB];
  const float lo_unit = __builtin_bit_cast(float, (127u - sa.lo_bits) << 23);   // 2^-lo_bits: the first stage compares H alone
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    qid[nb] = qbase + nb * 16 + x15;
    const bool real = qid[nb] < nq;
    const float T = real ? thr[qid[nb]] * qscale[qid[nb]] : __builtin_huge_valf();
    t1q[nb] = real ? (T - (1.001f * qdelta[qid[nb]] + 2e-6f * fabsf(T) + 1e-5f)) * lo_unit : __builtin_huge_valf();   // as filter_i8w_kernel
    asm volatile("" ::"v"(t1q[nb]));
  }
  const bool wave_has_queries = qbase < nq;

  uint32_t src_off[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const uint32_t P = static_cast<uint32_t>((wave * PPW + i) * 64 + lane);
    const uint32_t r = P / CHUNKS_PER_ROW, cpos = P % CHUNKS_PER_ROW;
    src_off[i] = r * ROW_BYTES + (swz_chunk<ROW_BYTES>(cpos, r) << 4);
  }
  const uint32_t sc_off = (lane & (8 * MB - 1)) * 16;
  // A fragment (rb, s): chunk 4 s + g4 of row 16 rb + x15 at position swz_chunk(4 s + g4, x15):  a16 ^ ((s & 3) << 6)  +  256 (s >> 2)  +  16 ROW_BYTES rb
  // (row strides that are odd multiples of 128 bytes, d = 384: the swizzle group is 8 chunks, a16 ^ ((s & 1) << 6)  +  128 (s >> 1))
  const uint32_t a16 = static_cast<uint32_t>(x15) * ROW_BYTES + (swz_chunk<ROW_BYTES>(static_cast<uint32_t>(g4), static_cast<uint32_t>(x15)) << 4);

  const char* gbase = reinterpret_cast<const char*>(rows);
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(NVDB_LPTR(smem)));
  const uint32_t g_lo = row_lo / TROWS + t_lo;
  auto tile_row0 = [&](uint32_t t_rel) -> uint32_t { return perm_tile(g_lo + (t_rel < NT ? t_rel : NT - 1), sa) * TROWS; };
  auto issue_piece = [&](uint32_t row0, uint32_t buf, int i) {
    glds16(src_off[i], gbase + static_cast<uint64_t>(row0) * ROW_BYTES, lds_base + buf * STAGE_BYTES + (wave * PPW + i) * 1024);
  };
  auto issue_scales = [&](uint32_t row0, uint32_t buf) {
    if (lane < 8 * MB) glds16(sc_off, reinterpret_cast<const char*>(scales + row0), lds_base + buf * STAGE_BYTES + DATA_BYTES + wave * 256);
  };
  auto loop_piece = [&](uint32_t row0, uint32_t buf, int i) { if constexpr (VAR != 3) issue_piece(row0, buf, i); };
  auto loop_scales = [&](uint32_t row0, uint32_t buf) { if constexpr (VAR != 3) issue_scales(row0, buf); };
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) issue_piece(tile_row0(st), st, i);
    issue_scales(tile_row0(st), st);
  }

  uint32_t wcnt = 0, n_stage1 = 0;
  Hit* mylog = hitlog + static_cast<uint64_t>(wave_gid) * FILTER_LOGCAP;

  intx4_t acc0[2][NB], acc1[2][NB];                // row block 0 / 1 of the tile in flight: [16-row half][query block]
  // row scales of a block for this lane's 8 rows (16 h + 4 g4 + j) and the constants -2^23 * scale of the biased test;
  // set 0 = block 0 of the current tile (filled during the first half, used in the second), set 1 = block 1 (filled during the
  // second half, used in the NEXT tile's first half -- which then touches nothing of this tile's stage: one barrier per tile)
  float scv[2][8], scc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { scv[0][i] = scv[1][i] = 0.f; scc[0][i] = scc[1][i] = 0.f; }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc1[h][nb] = bias0;        // tile 0's first half "tests" block 1 of a tile that does not exist (ignored)
  float mx[NB][2];                                 // running max of H * scale per query block and 16-row half
  float4_t ar[RING];

  auto read_a = [&](const char* stage, int u, int mb) -> float4_t {         // u = 2 s + h: k-step s, 16-row half h of row block mb
    const int s = u >> 1, h = u & 1;
    if constexpr (VAR == 4) { float4_t z = {0.f, 1.f, 2.f, 3.f}; asm volatile("" : "+v"(z)); return z; }
    const uint32_t off = swz16<ROW_BYTES>() ? (a16 ^ ((s & 3) << 6)) + (s >> 2) * 256 : (a16 ^ ((s & 1) << 6)) + (s >> 1) * 128;
    return *reinterpret_cast<const float4_t*>(stage + off + (2 * mb + h) * 16 * ROW_BYTES);
  };
  // in the shadow of the MFMAs: slot 0 reads the 8 scales of block mb of `stage` into set `set`, slots SW0.. multiply one each by -2^23
  constexpr int SW0 = NSLOT >= 64 ? 8 : 4;         // (d = 384: 48 slots per half; the 32 test units need 34 of them)
  auto scale_step = [&](const char* stage, int mb, int set, int w) {
    if constexpr (NOTEST) return;
    if (w == 0) {
      const float* sc_lds = reinterpret_cast<const float*>(stage + DATA_BYTES + wave * 256);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 v = *reinterpret_cast<const float4*>(sc_lds + 32 * mb + 16 * h + 4 * g4);
        scv[set][4 * h] = v.x; scv[set][4 * h + 1] = v.y; scv[set][4 * h + 2] = v.z; scv[set][4 * h + 3] = v.w;
      }
    }
    const int i = w - SW0;
    if (i >= 0 && i < 8) scc[set][i] = scv[set][i] * -8388608.f;
  };
  // Stage-1 test of a block, one value per TWO MFMA slots where they suffice (32 values over the 96 slots of a half at d = 768, 4 waves; an MFMA holds the
  // issue port for 8 of its 16 cycles): unit u = value v: (2^23 + H) * scale - 2^23 * scale = H * scale rounded once; odd v folds
  // the pair into the running maximum of its (query block, 16-row half).  Value v = 8 nb + 4 h + j.
  float tm[NV];
  constexpr int W0 = SW0 + 10;                     // the constants are ready (slots SW0 .. SW0 + 7)
  constexpr int SPU = (NSLOT - W0 - 4) / NV >= 2 ? 2 : 1;   // slots per unit
  static_assert(W0 + NV * SPU + 2 <= NSLOT, "the test ends before the half's last MFMA");
  auto test_step = [&](const intx4_t (&a)[2][NB], const float (&sc)[8], const float (&sccs)[8], int w) {
    if constexpr (NOTEST) return;
    const int rel = w - W0;
    if (rel < 0 || rel % SPU != 0) return;
    const int v = rel / SPU;
    if (v < NV) {
      const int nb = v >> 3, h = (v >> 2) & 1, j = v & 3;
      const int b0 = a[h][nb][j];
      tm[v] = __builtin_fmaf(__builtin_bit_cast(float, b0), sc[4 * h + j], sccs[4 * h + j]);
    }
    const int p = v - 1;                             // fold the pair completed one unit ago
    if (p >= 0 && p < NV && (p & 1)) mx[p >> 3][(p >> 2) & 1] = vmax3(mx[p >> 3][(p >> 2) & 1], tm[p - 1], tm[p]);
  };
  auto reset_max = [&]() {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { mx[nb][0] = -__builtin_huge_valf(); mx[nb][1] = -__builtin_huge_valf(); }
  };
  float flagv = -1.f;                              // >= 0 iff some value of the tested block reaches its first-stage threshold
  auto combine_flags = [&]() {
    if constexpr (NOTEST) return;
    float m[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) m[nb] = vmax3(mx[nb][0], mx[nb][1], mx[nb][1]) - t1q[nb];
    flagv = NB == 4 ? vmax3(vmax3(m[0], m[1], m[NB / 2]), m[NB - 1], m[NB - 1]) : vmax3(m[0], m[NB - 1], m[NB - 1]);
  };
  auto any_flag = [&]() -> bool { return __builtin_amdgcn_ballot_w64(flagv >= 0.f) != 0; };
  // log the values of a tested block that pass the first stage -- row scale, row, query, H -- for the exact finish after the stream
  auto rare_log = [&](const intx4_t (&a)[2][NB], const float (&sc)[8], uint32_t row0, int mb) {
    ++n_stage1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (!__builtin_amdgcn_ballot_w64(mx[nb][h] >= t1q[nb])) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int H = a[h][nb][j] - I8_ACC_BIAS;
          const bool hit = static_cast<float>(H) * sc[4 * h + j] >= t1q[nb];
          const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
          if (m) {
            const uint32_t idx = wcnt + static_cast<uint32_t>(__builtin_popcountll(m & ((1ull << lane) - 1ull)));
            if (hit && idx < FILTER_LOGCAP)
              mylog[idx] = Hit{sc[4 * h + j], row0 + 32u * mb + 16u * h + 4u * g4 + j, qid[nb], static_cast<uint32_t>(H)};
            wcnt += static_cast<uint32_t>(__builtin_popcountll(m));
          }
        }
      }
  };

  uint32_t sync_strikes = 0;
  uint64_t stamp_c = 0, stamp_r = 0;
  if constexpr (STAMP) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  const uint32_t xw_t0 = static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime());
  // LDS-DMA issue of tile t + 2: half of a wave's pieces in each half of the tile, the first three in the bubble while the
  // A-fragment ring fills after the barrier
  constexpr int HP = PPW / 2, HP0 = HP < 3 ? HP : 3;
  constexpr int D1 = (HP - HP0) > 0 ? (HP - HP0) : 1;
  constexpr int EV1 = NFRAG / D1, EV2 = NFRAG / (PPW - HP);
  static_assert((HP == HP0 || NFRAG % (HP - HP0) == 0) && NFRAG % (PPW - HP) == 0, "pieces spread evenly over the A fragments of both halves");
  for (uint32_t t = 0; t < NT; ++t) {
    if constexpr (SYNC) {
      if (wave == 0 && (t & sync_mask) == 0) sibling_rendezvous(prog + static_cast<uint64_t>(stream) * 8, qt, t, sync_lead, sync_strikes, lane);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 1) : "memory");         // tile t has landed (tile t + 1 may still be in flight)
    __builtin_amdgcn_s_barrier();
    const uint32_t next_row0 = tile_row0(t + 2), next_buf = (t + 2) % NSTAGE;
    const char* stage = smem + (t % NSTAGE) * STAGE_BYTES;
    if (!wave_has_queries) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) issue_piece(next_row0, next_buf, i);
      issue_scales(next_row0, next_buf);
      continue;
    }
    // ---- first half: block 0 of tile t  ||  test of block 1 of tile t - 1 (t == 0: garbage, tested and ignored) ----------
    reset_max();
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) ar[u] = read_a(stage, u, 0);
#pragma unroll
    for (int i = 0; i < HP0; ++i) loop_piece(next_row0, next_buf, i);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NFRAG; ++u) {
      if (u + RING - 1 < NFRAG) ar[(u + RING - 1) % RING] = read_a(stage, u + RING - 1, 0);
      else ar[(u + RING - 1) % RING] = read_a(stage, u + RING - 1 - NFRAG, 1);          // the ring runs through both halves
      const float4_t av = ar[u % RING];
      const int s = u >> 1, h = u & 1;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (s == 0) NVDB_MFMA_I8S_FROM(acc0[h][nb], av, bq[nb * KS], bias0);
        else NVDB_MFMA_I8S_ACC(acc0[h][nb], av, bq[nb * KS + s]);
        const int w = NB * u + nb;
        scale_step(stage, 0, 0, w);
        test_step(acc1, scv[1], scc[1], w);
        if (w == NSLOT - 1) combine_flags();
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (HP > HP0) {
        if (u % EV1 == EV1 - 1) loop_piece(next_row0, next_buf, HP0 + u / EV1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (VAR == 0 && t > 0 && any_flag()) rare_log(acc1, scv[1], tile_row0(t - 1), 1);
    if constexpr (VAR != 0) { asm volatile("" ::"v"(flagv)); }
    // ---- second half: block 1 of tile t  ||  rest of the loads of tile t + 2, test of block 0 of tile t ---------------------
    reset_max();
    constexpr int R0 = NFRAG % RING;                                       // ring slot of block 1's first fragment
#pragma unroll
    for (int u = 0; u < NFRAG; ++u) {
      if (u + RING - 1 < NFRAG) ar[(R0 + u + RING - 1) % RING] = read_a(stage, u + RING - 1, 1);
      const float4_t av = ar[(R0 + u) % RING];
      const int s = u >> 1, h = u & 1;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (s == 0) NVDB_MFMA_I8S_FROM(acc1[h][nb], av, bq[nb * KS], bias0);
        else NVDB_MFMA_I8S_ACC(acc1[h][nb], av, bq[nb * KS + s]);
        const int w = NB * u + nb;
        scale_step(stage, 1, 1, w);
        test_step(acc0, scv[0], scc[0], w);
        if (w == NSLOT - 1) combine_flags();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (u % EV2 == EV2 - 1) loop_piece(next_row0, next_buf, HP + u / EV2);
      if (u == 0) loop_scales(next_row0, next_buf);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (VAR == 0 && any_flag()) rare_log(acc0, scv[0], tile_row0(t), 0);
    if constexpr (VAR != 0) { asm volatile("" ::"v"(flagv)); }
  }
  if (wave_has_queries) {                          // block 1 of the last tile
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    reset_max();
#pragma unroll
    for (int w = W0; w < W0 + (NV + 1) * SPU; ++w) test_step(acc1, scv[1], scc[1], w);
    combine_flags();
    if (VAR == 0 && any_flag()) rare_log(acc1, scv[1], tile_row0(NT - 1), 1);
  }
  if constexpr (STAMP) {
    const uint64_t dc = __builtin_amdgcn_s_memtime() - stamp_c, dr = __builtin_amdgcn_s_memrealtime() - stamp_r;
    if (wave == 0 && lane == 0) {
      uint64_t* out = reinterpret_cast<uint64_t*>(prog + static_cast<uint64_t>(gridDim.x) * 8) + static_cast<uint64_t>(blockIdx.x) * 2;
      out[0] = dc; out[1] = dr;
    }
  }
  if constexpr (SYNC) { if (wave == 0 && lane == 0) __hip_atomic_store(prog + static_cast<uint64_t>(stream) * 8 + qt, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  if (stage_counts && lane == 0 && n_stage1) atomicAdd(stage_counts, n_stage1);
  if (wave == 0 && lane == 0) record_xcd_speed(sa.xcdw, xcd_map, stream & 7u, NT, static_cast<uint32_t>(__builtin_amdgcn_s_memrealtime()) - xw_t0);
  verify_and_scatter_i8<DIM>(mylog, wcnt, sa, lane, rows, qlo, thr, qscale, qinv);
}

}  // namespace nvdbhip
