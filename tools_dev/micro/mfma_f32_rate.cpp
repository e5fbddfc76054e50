// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 in the patterns the exact kernels use.  Developer tool (GPU box):
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_f32_rate tools_dev/micro/mfma_f32_rate.cpp && /tmp/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));

// VARIANT 0: 8 chains, A/B VGPR constants, builtin.  1: + one v_cvt_f32_f16 feeding each MFMA's A (like the fp16 kernels).
// 2: asm MFMA with B in AGPR, acc in VGPR.  3: asm, B in AGPR, A produced one step ahead by cvt.  4: builtin, 16 chains (two tiles).
template <int VARIANT>
__global__ __launch_bounds__(256, 1) void k(float* out, const unsigned* in, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  float b[24][8];
#pragma unroll
  for (int t = 0; t < 24; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) b[t][j] = __uint_as_float(in[(t * 8 + j) * 64 + lane]) ;
  unsigned raw[24][4];
#pragma unroll
  for (int t = 0; t < 24; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[t][j] = in[4096 + (t * 4 + j) * 64 + lane];
  floatx4 acc[VARIANT == 4 ? 16 : 8];
#pragma unroll
  for (int j = 0; j < (VARIANT == 4 ? 16 : 8); ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
  if constexpr (VARIANT == 2 || VARIANT == 3) {
#pragma unroll
    for (int t = 0; t < 24; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"a"(b[t][j]));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 24; ++t) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (VARIANT == 1 || VARIANT == 3) {
          const unsigned w = raw[t][j >> 1];
          x[j] = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<unsigned short>((j & 1) ? (w >> 16) : (w & 0xFFFFu))));
        } else x[j] = __uint_as_float(raw[t][j >> 1] + j);
      }
      if constexpr (VARIANT == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));        // all conversions of the step before its first MFMA
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (VARIANT == 2 || VARIANT == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(x[j]), "a"(b[t][j]));
        else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], b[t][j], acc[j], 0, 0, 0);
        if constexpr (VARIANT == 4) acc[8 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], b[t][(j + 1) & 7], acc[8 + j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (VARIANT == 2 || VARIANT == 3) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < (VARIANT == 4 ? 16 : 8); ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V> void run(const char* name, float* out, unsigned* in, unsigned long long* cyc, int mf_per_iter) {
  const int iters = 2000;
  k<V><<<256, 256>>>(out, in, iters, cyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<V><<<256, 256>>>(out, in, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double c = 0; for (auto v : h) c += double(v); c /= 256;
  printf("%-70s %8.1f cycles per MFMA, %7.3f ms, %6.1f TFLOP/s\n", name, c / (double(iters) * mf_per_iter), ms, 2048.0 * mf_per_iter * iters * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
  float* out; unsigned* in; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&in, 65536 * 4); hipMalloc(&cyc, 256 * 8);
  std::vector<unsigned> h(65536); for (int i = 0; i < 65536; ++i) h[i] = 0x3C003800u + (i * 2654435761u) % 0x3000;   // plausible halves / floats
  hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  run<0>("builtin, 8 chains, A/B in VGPRs (compiler's choice of files)", out, in, cyc, 192);
  run<1>("builtin, 8 chains, one v_cvt_f32_f16 per MFMA", out, in, cyc, 192);
  run<2>("asm, B in AGPRs, acc in VGPRs", out, in, cyc, 192);
  run<3>("asm, B in AGPRs, acc in VGPRs, a step's 8 cvt before its MFMAs", out, in, cyc, 192);
  run<4>("builtin, 16 chains (two tiles per step)", out, in, cyc, 384);
  return 0;
}
