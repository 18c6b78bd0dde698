// Epilogue glue for the PyTorch-ROCm policy/value net: bias + residual + ReLU of a conv output in ONE pass over the fp16
// channels_last activation (HBM-bound: 2 B read + 2 B written per element, + 2 B for the residual).  The reference runs
// conv -> BatchNorm -> ReLU / (+x) -> ReLU as separate PyTorch kernels (src_py/elfgames/go/df_model3.py:62-110); with
// eval BatchNorm folded into the conv weights what remains per conv is exactly this epilogue.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/elf_amd.h"
#include "engine_host.h"

namespace {

struct alignas(16) H8 { unsigned int v[4]; };   // 8 x 16-bit floats

// unpack / pack two 16-bit floats of one dword; BF = bfloat16 (upper half of an fp32), else IEEE half
template <bool BF>
__device__ __forceinline__ float2 unpack2(unsigned int w) {
  if (BF) return make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u));
  return __half22float2(*reinterpret_cast<const __half2*>(&w));
}
template <bool BF>
__device__ __forceinline__ unsigned int pack2(float2 f) {
  if (BF) {   // round to nearest even, NaN kept quiet
    auto r = [](float x) -> unsigned int {
      unsigned int u = __float_as_uint(x);
      if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
      return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    };
    return r(f.x) | (r(f.y) << 16);
  }
  __half2 h = __float22half2_rn(f);
  return *reinterpret_cast<unsigned int*>(&h);
}

// lanes own 16 B (8 halfs); a wave covers 1 KiB per load/store instruction; grid-stride so that ~8 waves per SIMD cover any size
template <bool RES, bool BIAS, bool BF>
__global__ __launch_bounds__(256) void k_bias_act(H8* __restrict__ x, const H8* __restrict__ bias, const H8* __restrict__ res,
                                                  int64_t n8, int c8, int relu) {
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += step) {
    H8 a = x[i];
    H8 b, r;
    if (BIAS) b = bias[(int)(i % c8)];
    if (RES) r = res[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float2 f = unpack2<BF>(a.v[k]);
      if (BIAS) { float2 g = unpack2<BF>(b.v[k]); f.x += g.x; f.y += g.y; }
      if (RES) { float2 g = unpack2<BF>(r.v[k]); f.x += g.x; f.y += g.y; }
      if (relu) { f.x = fmaxf(f.x, 0.0f); f.y = fmaxf(f.y, 0.0f); }
      a.v[k] = pack2<BF>(f);
    }
    x[i] = a;
  }
}

}  // namespace

template <bool BF>
static int launch_bias_act(void* x, const void* bias, const void* res, int64_t rows, int channels, int relu, void* stream) {
  if (!x || rows < 0 || channels <= 0 || (channels & 7) != 0) return ELFGO_E_BADARG;
  if ((((uintptr_t)x | (uintptr_t)bias | (uintptr_t)res) & 15) != 0) return ELFGO_E_BADARG;
  const int64_t n8 = rows * (int64_t)(channels / 8);
  if (n8 == 0) return 0;
  const int c8 = channels / 8;
  // no handle here: the pass runs on the device that owns x (the caller's stream must belong to it)
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, x) != hipSuccess) { (void)hipGetLastError(); return ELFGO_E_BADARG; }
  DevGuard _dg(at.device);
  int64_t blocks = (n8 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;   // 256 CUs x 32 resident waves / 4 waves per block, x4 oversubscription
  dim3 g((unsigned)blocks), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (res && bias) hipLaunchKernelGGL((k_bias_act<true, true, BF>), g, b, 0, st, (H8*)x, (const H8*)bias, (const H8*)res, n8, c8, relu);
  else if (res) hipLaunchKernelGGL((k_bias_act<true, false, BF>), g, b, 0, st, (H8*)x, (const H8*)nullptr, (const H8*)res, n8, c8, relu);
  else if (bias) hipLaunchKernelGGL((k_bias_act<false, true, BF>), g, b, 0, st, (H8*)x, (const H8*)bias, (const H8*)nullptr, n8, c8, relu);
  else hipLaunchKernelGGL((k_bias_act<false, false, BF>), g, b, 0, st, (H8*)x, (const H8*)nullptr, (const H8*)nullptr, n8, c8, relu);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int elfnet_bias_act_f16(void* x, const void* bias, const void* res, int64_t rows, int channels, int relu, void* stream) {
  return launch_bias_act<false>(x, bias, res, rows, channels, relu, stream);
}
extern "C" int elfnet_bias_act_bf16(void* x, const void* bias, const void* res, int64_t rows, int channels, int relu, void* stream) {
  return launch_bias_act<true>(x, bias, res, rows, channels, relu, stream);
}
