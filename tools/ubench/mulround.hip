// mulround.hip -- what one Q15 "multiply by a constant, round half away from zero, >> 15" step
// (FixedPoint::operator*=, tmc3/FixedPoint.h:115-123) costs on gfx950 as 64-bit integer code
// (32-bit multipliers, quarter rate) and as exact double arithmetic (|a * c| < 2^52: fma + trunc),
// as a dependent chain of one wavefront (latency) and with the chip full (throughput).
//   hipcc --offload-arch=gfx950 -O3 -o mulround mulround.hip && ./mulround
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ int64_t mul_i64_u32(int64_t a, int64_t c)
{
  const uint32_t cl = (uint32_t)c;
  const uint64_t lo = (uint64_t)(uint32_t)a * cl;
  const uint32_t hi = (uint32_t)((uint64_t)a >> 32) * cl + (uint32_t)(lo >> 32);
  return (int64_t)(((uint64_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ int64_t fp_mul_c(int64_t a, int64_t c)
{
  const int64_t p = mul_i64_u32(a, c);
  return (p + (1 << 14) + (p >> 63)) >> 15;
}
// c15 = c * 2^-15 (exact); round half away from zero
__device__ __forceinline__ double fp_mul_d(double a, double c15)
{
  return __builtin_trunc(__builtin_fma(a, c15, __builtin_copysign(0.5, a)));
}

template<int MODE>
__global__ __launch_bounds__(256) void
chain(int64_t* out, int iters, int64_t x0, int64_t ca, int64_t cb, unsigned long long* cyc)
{
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  int64_t xi = x0 + tid, yi = x0 - tid;
  double xd = (double)xi, yd = (double)yi;
  const double da = (double)ca * (1.0 / 32768), db = (double)cb * (1.0 / 32768);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (MODE == 0) {  // butterfly step, integer
      const int64_t l = fp_mul_c(xi, ca) + fp_mul_c(yi, cb);
      const int64_t h = fp_mul_c(yi, ca) - fp_mul_c(xi, cb);
      xi = l; yi = h;
    } else if (MODE == 1) {  // butterfly step, double
      const double l = fp_mul_d(xd, da) + fp_mul_d(yd, db);
      const double h = fp_mul_d(yd, da) - fp_mul_d(xd, db);
      xd = l; yd = h;
    } else if (MODE == 2) {  // single dependent multiply, integer
      xi = fp_mul_c(xi, ca) + 977;
    } else {                 // single dependent multiply, double
      xd = fp_mul_d(xd, da) + 977.0;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (MODE & 1) { xi = (int64_t)xd; yi = (int64_t)yd; }
  out[tid * 2] = xi; out[tid * 2 + 1] = yi;
  if (tid == 0) *cyc = t1 - t0;
}

int main()
{
  const int iters = 4096;
  int64_t* d; unsigned long long* dc;
  const int maxthreads = 256 * 256 * 16;
  hipMalloc(&d, sizeof(int64_t) * 2 * maxthreads); hipMalloc(&dc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<int64_t> ref(2), got(2 * 64);
  // a = 23170 (~1/sqrt 2), b = 23170: values stay bounded
  const int64_t x0 = 123456789012ll, ca = 23170, cb = 23169;
  int64_t want[4][2];
  for (int mode = 0; mode < 4; mode++) {
    for (int cfg = 0; cfg < 2; cfg++) {
      const int blocks = cfg == 0 ? 1 : 256 * 16, threads = cfg == 0 ? 64 : 256;
      float ms = 0;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        switch (mode) {
        case 0: chain<0><<<blocks, threads>>>(d, iters, x0, ca, cb, dc); break;
        case 1: chain<1><<<blocks, threads>>>(d, iters, x0, ca, cb, dc); break;
        case 2: chain<2><<<blocks, threads>>>(d, iters, x0, ca, cb, dc); break;
        default: chain<3><<<blocks, threads>>>(d, iters, x0, ca, cb, dc); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      unsigned long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
      hipMemcpy(got.data(), d, 16 * 64, hipMemcpyDeviceToHost);
      if (cfg == 0) { want[mode][0] = got[2]; want[mode][1] = got[3]; }
      const double steps = (double)iters * blocks * threads;
      printf("{\"mode\": \"%s\", \"cfg\": \"%s\", \"ms\": %.4f, \"memtime_ticks_per_iter\": %.2f, \"ns_per_iter_one_wave\": %.2f, \"G_iters_per_s\": %.2f, \"x1\": %lld, \"y1\": %lld}\n",
             mode == 0 ? "butterfly_i64" : mode == 1 ? "butterfly_f64" : mode == 2 ? "mul_i64" : "mul_f64",
             cfg == 0 ? "one_wave" : "chip_full_16_waves_per_cu_x4", ms, (double)cyc / iters, ms * 1e6 / iters,
             steps / (ms * 1e6), (long long)got[2], (long long)got[3]);
    }
  }
  printf("{\"exact\": %s}\n", (want[0][0] == want[1][0] && want[0][1] == want[1][1] && want[2][0] == want[3][0]) ? "true" : "false");
  return 0;
}
