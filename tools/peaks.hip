// Achievable peaks on the box the bench runs on ("also measure an achievable peak with a plain MFMA
// microbenchmark and report the fraction against both"):
//   * bf16 MFMA: v_mfma_f32_16x16x32_bf16 on register operands only, 8 independent accumulators per wave, 2 or 4 waves
//     per SIMD, for a short (~0.1 s, the length of one _sample launch) and a long (~1 s, DVFS settled) run;
//   * HBM: float4 copy (read + write) and read-only sum over buffers far larger than L2 + MALL.
// Build:  hipcc --offload-arch=gfx950 -O3 tools/peaks.hip -o tools/peaks      Run: tools/peaks  (prints one JSON line)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef __bf16 bf16v8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

__global__ void __launch_bounds__(256) k_mfma(float* out, int iters) {
  bf16v8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)(0.001f * (threadIdx.x & 7) + 0.01f * j);
    b[j] = (__bf16)(0.002f * (threadIdx.x & 3) - 0.01f * j);
  }
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int t = 1; t < 8; ++t) s += acc[t];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];      // keeps the loop alive, never true
}

__global__ void __launch_bounds__(256) k_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_read(const f32x4* __restrict__ src, float* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += src[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[0] = s[0];
}

template <class F>
static int timed(F launch, int reps, std::vector<float>& ms) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(a, 0));
    launch();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float t;
    CHECK(hipEventElapsedTime(&t, a, b));
    ms.push_back(t);
  }
  return 0;
}

static float median(std::vector<float> v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out;
  CHECK(hipMalloc(&out, 4096));
  printf("{\"device\": \"%s\", \"cus\": %d", prop.gcnArchName, cus);

  // ---- MFMA -----------------------------------------------------------------------------------------------------
  for (int wgs_per_cu = 2; wgs_per_cu <= 4; wgs_per_cu += 2) {          // 256-thread WGs: 2 or 4 waves per SIMD
    const int grid = cus * wgs_per_cu;
    for (int pass = 0; pass < 2; ++pass) {
      // one MFMA = 16 cycles at full rate; iters x 32 MFMAs per wave; aim for ~0.1 s and ~1 s at ~2 GHz
      const int iters = (pass == 0 ? 400000 : 4000000) / wgs_per_cu / 2;
      std::vector<float> ms;
      if (timed([&] { hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, 1000); }, 1, ms)) return 1;   // warm-up
      ms.clear();
      if (timed([&] { hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, iters); }, pass == 0 ? 5 : 2, ms)) return 1;
      const double flop = (double)grid * 4 * iters * 32 * 16384.0;
      const float t = median(ms);
      printf(", \"mfma_bf16_TFLOPs_%dwaves_%s\": %.1f, \"mfma_ms_%dwaves_%s\": %.1f", wgs_per_cu, pass ? "1s" : "0.1s",
             flop / t / 1e9, wgs_per_cu, pass ? "1s" : "0.1s", t);
    }
  }

  // ---- HBM ------------------------------------------------------------------------------------------------------
  const size_t bytes = (size_t)4 << 30;
  f32x4 *src, *dst;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMalloc(&dst, bytes));
  CHECK(hipMemset(src, 1, bytes));
  CHECK(hipMemset(dst, 0, bytes));
  const size_t n = bytes / sizeof(f32x4);
  for (int mult = 8; mult <= 32; mult *= 2) {
    std::vector<float> ms;
    if (timed([&] { hipLaunchKernelGGL(k_copy, dim3(cus * mult), dim3(256), 0, 0, src, dst, n); }, 7, ms)) return 1;
    printf(", \"hbm_copy_GBs_grid%dxCU\": %.0f", mult, 2.0 * bytes / median(ms) / 1e6);
    ms.clear();
    if (timed([&] { hipLaunchKernelGGL(k_read, dim3(cus * mult), dim3(256), 0, 0, src, out, n); }, 7, ms)) return 1;
    printf(", \"hbm_read_GBs_grid%dxCU\": %.0f", mult, 1.0 * bytes / median(ms) / 1e6);
  }
  printf("}\n");
  CHECK(hipFree(src));
  CHECK(hipFree(dst));
  CHECK(hipFree(out));
  return 0;
}
