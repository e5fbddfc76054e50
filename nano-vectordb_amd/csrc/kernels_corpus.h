// kernels_corpus.h -- one-time kernels over the resident corpus: the synthetic generator, the fp16 / int8 shadow copies the
// MFMA filter streams when the corpus itself cannot be streamed, and the row-norm pass behind the filter's error bound.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_exact.h"
#include "nvdb_common.h"

namespace nvdbhip {

// ------------------------------------------------------------------------------------------------
// synthetic corpus generator: one wave per row (bit-identical to nvdb_synth_rows_f32 on the host
// followed by the RNE half conversion / the reference's int8 quantiser).
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void gen_rows_kernel(uint64_t seed, uint64_t row_base, uint64_t n, uint32_t dim,
                                                       void* __restrict__ rows, float* __restrict__ scales) {
  const int lane = threadIdx.x & 63;
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const uint32_t key = synth_row_key(seed, row_base + r);
  unsigned long long ss = 0;
  for (uint32_t c = lane; c < dim; c += 64) { const long long v = synth_raw(key, c); ss += static_cast<unsigned long long>(v * v); }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const double inv = synth_inv_norm(ss);
  if constexpr (DT == DT_F32) {
    float* out = static_cast<float*>(rows) + r * dim;
    for (uint32_t c = lane; c < dim; c += 64) out[c] = synth_elem(synth_raw(key, c), inv);
  } else if constexpr (DT == DT_F16) {
    _Float16* out = static_cast<_Float16*>(rows) + r * dim;
    for (uint32_t c = lane; c < dim; c += 64) out[c] = static_cast<_Float16>(synth_elem(synth_raw(key, c), inv));
  } else {
    float mx = 0.f;
    for (uint32_t c = lane; c < dim; c += 64) mx = fmaxf(mx, fabsf(synth_elem(synth_raw(key, c), inv)));
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float scale = mx > 0.f ? mx / 127.f : 1.f;
    const float is = 1.0f / scale;
    signed char* out = static_cast<signed char*>(rows) + r * dim;
    for (uint32_t c = lane; c < dim; c += 64) {
      float qv = rintf(synth_elem(synth_raw(key, c), inv) * is);
      qv = fminf(fmaxf(qv, -127.f), 127.f);
      out[c] = static_cast<signed char>(static_cast<int>(qv));
    }
    if (lane == 0) scales[r] = scale;
  }
}

// "shadow" copy streamed by the MFMA filter when the corpus itself cannot be: an fp32 corpus (rounded to fp16,
// round-to-nearest-even) and/or a dim the kernels are not instantiated for (rows zero-padded to sdim).  Also
// returns max |x| (float bits via atomicMax).  The original rows stay the arbiter: every survivor is re-scored
// from them in the reference's order.
template <typename SrcT>
__global__ __launch_bounds__(256) void shadow_f16_kernel(const SrcT* __restrict__ src, _Float16* __restrict__ dst, size_t n,
                                                         uint32_t dim, uint32_t sdim, uint32_t* __restrict__ maxabs_bits) {
  float mx = 0.f;
  const size_t count = n * sdim;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < count; i += static_cast<size_t>(gridDim.x) * 256) {
    const size_t r = i / sdim;
    const uint32_t c = static_cast<uint32_t>(i - r * sdim);
    const float v = c < dim ? static_cast<float>(src[r * dim + c]) : 0.f;
    dst[i] = static_cast<_Float16>(v);
    mx = fmaxf(mx, fabsf(v));
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(maxabs_bits, __builtin_bit_cast(uint32_t, mx));
}

// int8 corpus whose dim the kernels are not instantiated for: copy with rows zero-padded to sdim bytes (the swizzled LDS image
// needs a row stride that is a multiple of 128 bytes, swz_chunk); 16 source bytes per thread where alignment allows
static __global__ __launch_bounds__(256) void shadow_i8_kernel(const signed char* __restrict__ src, signed char* __restrict__ dst, size_t n,
                                                        uint32_t dim, uint32_t sdim) {
  const size_t count = n * sdim;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < count; i += static_cast<size_t>(gridDim.x) * 256) {
    const size_t r = i / sdim;
    const uint32_t c = static_cast<uint32_t>(i - r * sdim);
    dst[i] = c < dim ? src[r * dim + c] : static_cast<signed char>(0);
  }
}

// int8 FILTER shadow of an fp16 / fp32 corpus (option q8_shadow): the rows quantised per row by the reference's own rule
// (apps/nvdb_quantize_i8.cpp:71-80: scale = max|x| / 127, rint(x / scale), clamp +-127), rows zero-padded to sdim, plus -- what turns
// a lossy copy into a usable FILTER -- the largest quantisation residual ||x - scale * x_q||_2 over all rows (float bits via atomicMax):
// |<q, x> - <q, scale * x_q>| <= ||q|| * that, which the query prep adds to the filter's error bound.  One wave per row.
// out_bits[0] = max residual norm (slightly inflated), out_bits[1] = 1 if a row held a non-finite value (no shadow then).
template <typename SrcT>
__global__ __launch_bounds__(256) void shadow_q8_kernel(const SrcT* __restrict__ src, signed char* __restrict__ dst, float* __restrict__ dst_scales,
                                                        uint64_t n, uint32_t dim, uint32_t sdim, uint32_t* __restrict__ out_bits) {
  const int lane = threadIdx.x & 63;
  float wres = 0.f;
  bool bad = false;
  for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); r < n; r += static_cast<uint64_t>(gridDim.x) * 4) {
    const SrcT* row = src + r * dim;
    float mx = 0.f;
    for (uint32_t c = lane; c < dim; c += 64) mx = fmaxf(mx, fabsf(static_cast<float>(row[c])));
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (!(mx < 3.0e38f)) bad = true;
    const float scale = mx > 0.f ? mx / 127.f : 1.f;
    const float inv = 1.0f / scale;
    float ss = 0.f;
    signed char* out = dst + r * sdim;
    for (uint32_t c = lane; c < sdim; c += 64) {
      float qv = 0.f, x = 0.f;
      if (c < dim) { x = static_cast<float>(row[c]); qv = fminf(fmaxf(rintf(x * inv), -127.f), 127.f); }
      out[c] = static_cast<signed char>(static_cast<int>(qv));
      const float d = x - qv * scale;
      ss = __builtin_fmaf(d, d, ss);
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) dst_scales[r] = scale;
    wres = fmaxf(wres, sqrtf(ss) * 1.0001f);
  }
  if (lane == 0 && wres > 0.f) atomicMax(out_bits, __builtin_bit_cast(uint32_t, wres));
  if (lane == 0 && (bad || !(wres < 3.0e38f))) out_bits[1] = 1u;
}

// max over rows of the (dequantised) L2 norm, slightly inflated; result as float bits via atomicMax
template <int DT>
__global__ __launch_bounds__(256) void row_norm_max_kernel(const void* __restrict__ rows, const float* __restrict__ scales,
                                                           uint64_t n, uint32_t dim, uint32_t* __restrict__ out_bits) {
  const int lane = threadIdx.x & 63;
  float wmax = 0.f;
  for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); r < n; r += static_cast<uint64_t>(gridDim.x) * 4) {
    const void* rp = row_ptr<DT>(rows, r, dim);
    float ss = 0.f;
    for (uint32_t c = lane; c < dim; c += 64) { const float v = load1<DT>(rp, c); ss = __builtin_fmaf(v, v, ss); }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    float nrm = sqrtf(ss) * 1.0001f;
    if constexpr (DT == DT_I8) {
      const float sc = scales[r];
      nrm *= fabsf(sc);
      // a negative / NaN row scale, or one so large that 2^23 * scale overflows (inf - inf = NaN would lose the flag silently):
      // the biased-accumulator test assumes neither -> such corpora take the in-loop (unbiased) build
      if (!(sc >= 0.f && sc < 1e30f) && lane == 0) out_bits[1] = 1u;
    }
    wmax = fmaxf(wmax, nrm);
  }
  if (lane == 0 && wmax > 0.f) atomicMax(out_bits, __builtin_bit_cast(uint32_t, wmax));
}

}  // namespace nvdbhip
