// tools/ubench/handoff_pollers.hip -- round 6 micro-benchmark: does polling DELAY the write it waits for?  The sc1 / sc1 ping-pong of
// tools/ubench/handoff.hip between two wavefronts on different XCDs, with P more wavefronts (other compute units, all XCDs) polling the SAME
// two granule lines meanwhile, at two poll rates (back to back; with s_sleep 16 between polls) and with 1 or 64 lanes per polling load.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/handoff_pollers.hip -o /tmp/hp && /tmp/hp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template<int SLEEP, int LANES>
__global__ void pingpong(uint32_t* box, int peer, int iters, int pollers, unsigned long long* out, int* stop)
{
  const int b = blockIdx.x;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(box, 0, 4096, 0x00020000);
  if (b != 0 && b != peer) {
    // a bystander: polls both granules until told to stop (bounded)
    if (b >= 2 + pollers + (peer > 1 ? 0 : 0) || (int)threadIdx.x >= LANES)
      return;
    uint32_t acc = 0;
    for (int spin = 0; spin < 4000000; spin++) {
      asm volatile("" ::: "memory");
      const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 16);
      const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, 64, 0, 16);
      acc += r0.z + r1.z;
      if (SLEEP)
        __builtin_amdgcn_s_sleep(SLEEP);
      if ((spin & 63) == 0 && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        break;
    }
    if (acc == 0x12345678u)
      out[2] = acc;
    return;
  }
  const int me = b == 0 ? 0 : 1;
  if (threadIdx.x != 0)
    return;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  bool dead = false;
  for (int i = 1; i <= iters && !dead; i++) {
    const u32x4 g = {(uint32_t)i, 0u, (uint32_t)i, 0u};
    if (me == 0) {
      __builtin_amdgcn_raw_buffer_store_b128(g, rs, 0, 0, 16);
      for (int spin = 0;; spin++) {
        asm volatile("" ::: "memory");
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, 64, 0, 16);
        if (r.z == (uint32_t)i)
          break;
        if (spin > 400000) {
          dead = true;
          break;
        }
      }
    } else {
      for (int spin = 0;; spin++) {
        asm volatile("" ::: "memory");
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, 16);
        if (r.z == (uint32_t)i)
          break;
        if (spin > 400000) {
          dead = true;
          break;
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(g, rs, 64, 0, 16);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[me] = dead ? 0ull : t1 - t0;
  if (me == 0)
    __hip_atomic_store(stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template<int SLEEP, int LANES>
void run(const char* name, int pollers)
{
  uint32_t* box;
  unsigned long long* out;
  int* stop;
  (void)hipMalloc(&box, 4096);
  (void)hipMalloc(&out, 32);
  (void)hipMalloc(&stop, 4);
  const int iters = 2000, peer = 1;
  for (int rep = 0; rep < 2; rep++) {
    (void)hipMemset(box, 0, 4096);
    (void)hipMemset(stop, 0, 4);
    (void)hipMemset(out, 0, 32);
    (void)hipDeviceSynchronize();
    pingpong<SLEEP, LANES><<<2 + pollers, 64>>>(box, peer, iters, pollers, out, stop);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    if (rep == 1)
      printf("%-34s %4d bystanders: %.0f cycles per one-way hop%s, err %d\n", name, pollers, (double)h[0] / iters / 2,
             h[0] && h[1] ? "" : "  NEVER ARRIVED", (int)e);
  }
  (void)hipFree(box);
  (void)hipFree(out);
  (void)hipFree(stop);
}

int main()
{
  for (int p : {0, 1, 4, 16, 64, 254}) {
    run<0, 1>("back to back, 1 lane", p);
    run<0, 64>("back to back, 64 lanes", p);
    run<16, 1>("s_sleep 16 between, 1 lane", p);
    run<16, 64>("s_sleep 16 between, 64 lanes", p);
  }
  return 0;
}
