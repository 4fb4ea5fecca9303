// tools/ubench/handoff.hip -- round 6 micro-benchmark: latency of a 16-byte {value, tag} granule hand-off between two
// wavefronts on different CUs of the SAME XCD and of DIFFERENT XCDs, by store / load flavour (MI355X).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/handoff.hip -o /tmp/handoff && /tmp/handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template<int ST, int LD>  // aux bits: 0 plain, 1 sc0, 16 sc1, 17 sc0 sc1; ST = 100: a plain store, then an sc1 store of the same granule
__global__ void pingpong(uint32_t* box, int peer, int iters, unsigned long long* out, int* xcc)
{
  const int b = blockIdx.x;
  if (b != 0 && b != peer)
    return;
  const int me = b == 0 ? 0 : 1;
  if (threadIdx.x == 0) {
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[me] = id & 0xf;
  }
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(box, 0, 4096, 0x00020000);
  // two granules: [0] written by ping, [1] written by pong; lane 0 only
  if (threadIdx.x != 0)
    return;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  bool dead = false;  // a flavour whose data never arrives must not hang the GPU: bounded polls
  for (int i = 1; i <= iters && !dead; i++) {
    if (me == 0) {
      const u32x4 g = {(uint32_t)i, 0u, (uint32_t)i, 0u};
      if (ST == 100) {
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, 0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, 0, 0, 16);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, 0, 0, ST);
      }
      for (int spin = 0;; spin++) {
        asm volatile("" ::: "memory");  // (the poll is a fresh load every time)
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, 64, 0, LD);
        if (r.z == (uint32_t)i)
          break;
        if (spin > 200000) {
          dead = true;
          break;
        }
      }
    } else {
      for (int spin = 0;; spin++) {
        asm volatile("" ::: "memory");  // (the poll is a fresh load every time)
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 0, LD);
        if (r.z == (uint32_t)i)
          break;
        if (spin > 200000) {
          dead = true;
          break;
        }
      }
      const u32x4 g = {(uint32_t)i, 0u, (uint32_t)i, 0u};
      if (ST == 100) {
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, 64, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, 64, 0, 16);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, 64, 0, ST);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[me] = dead ? 0ull : t1 - t0;
}

template<int ST, int LD>
void run(const char* name, int peer)
{
  uint32_t* box;
  unsigned long long* out;
  int* xcc;
  hipMalloc(&box, 4096);
  hipMalloc(&out, 16);
  hipMalloc(&xcc, 8);
  hipMemset(box, 0, 4096);
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 2; rep++) {
    hipMemset(box, 0, 4096);
    hipDeviceSynchronize();
    hipEventRecord(a);
    pingpong<ST, LD><<<64, 64>>>(box, peer, iters, out, xcc);
    hipEventRecord(b);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2];
    int hx[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
    if (rep == 1)
      printf("%-28s peer block %2d  xcc %d -> %d : %.3f us per one-way hop (%.0f ticks)%s, err %d\n", name, peer, hx[0], hx[1],
             ms * 1000.0 / iters / 2, (double)h[0] / iters / 2, h[0] && h[1] ? "" : "  NEVER ARRIVED (bounded polls ran out)", (int)e);
  }
  hipFree(box);
  hipFree(out);
  hipFree(xcc);
}

int main()
{
  for (int peer : {8, 1, 16, 3}) {
    run<16, 16>("store sc1 / load sc1", peer);
    run<0, 16>("store plain / load sc1", peer);
    run<17, 17>("store sc0sc1 / load sc0sc1", peer);
    run<0, 17>("store plain / load sc0sc1", peer);
    run<1, 16>("store sc0 / load sc1", peer);
    run<100, 16>("store plain+sc1 / load sc1", peer);
  }
  return 0;
}
