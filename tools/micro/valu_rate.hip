// Issue rate of the vector ALU for the instructions the traversal kernels are made
// of: independent chains of one instruction per wavefront, enough wavefronts to
// fill every SIMD, time per instruction per SIMD in cycles at the measured clock.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rate tools/micro/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kChains = 8, kIters = 65536;

template <int OP>
__global__ void __launch_bounds__(256) k_rate(double* out, double seed) {
  double a[kChains];
  float f[kChains];
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p[kChains];
  for (int c = 0; c < kChains; c++) {
    a[c] = seed + c + threadIdx.x;
    f[c] = (float)a[c];
    p[c] = float2v{f[c], f[c] + 1.0f};
  }
  const double m = 1.0000001, b = 1e-9;
  const float mf = 1.0000001f, bf = 1e-9f;
  const float2v mp = {mf, mf}, bp = {bf, bf};
  for (int i = 0; i < kIters; i++) {
#pragma unroll
    for (int c = 0; c < kChains; c++) {
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(m));
      if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
      if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(mf), "v"(bf));
      if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[c]) : "v"(mp), "v"(bp));
      if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(mf), "v"(bf));
    }
  }
  double s = 0.0;
  for (int c = 0; c < kChains; c++) s += a[c] + f[c] + p[c].x + p[c].y;
  if (s == 12345.678) out[0] = s;
}

template <int OP>
void run(const char* name, double* out, int cus, double mhz) {
  const int blocks = cus * 8;   // 8 blocks of 4 wavefronts per CU: 8 wavefronts per SIMD
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  k_rate<OP><<<blocks, 256>>>(out, 1.0);
  CHECK(hipEventRecord(e0));
  k_rate<OP><<<blocks, 256>>>(out, 1.0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double instrPerSimd = 8.0 * kChains * kIters;   // 8 wavefronts on each SIMD
  const double cycles = ms * 1e-3 * mhz * 1e6 / instrPerSimd;
  printf("| %s | %.3f | %.2f |\n", name, ms, cycles);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  double* out;
  CHECK(hipMalloc(&out, 8));
  const double mhz = prop.clockRate / 1000.0;
  printf("# %s, %d CUs, %.0f MHz: cycles per wavefront instruction per SIMD (8 wavefronts per SIMD, 8 independent chains each)\n\n",
         prop.name, prop.multiProcessorCount, mhz);
  printf("| instruction | ms | cycles / instruction |\n|---|---|---|\n");
  run<0>("v_fma_f64", out, prop.multiProcessorCount, mhz);
  run<1>("v_mul_f64", out, prop.multiProcessorCount, mhz);
  run<2>("v_add_f64", out, prop.multiProcessorCount, mhz);
  run<3>("v_fma_f32", out, prop.multiProcessorCount, mhz);
  run<4>("v_pk_fma_f32 (two results per lane)", out, prop.multiProcessorCount, mhz);
  run<5>("v_max3_f32", out, prop.multiProcessorCount, mhz);
  CHECK(hipFree(out));
  return 0;
}
