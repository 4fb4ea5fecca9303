// tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE.
//
// A lock-step wavefront emulator: the device headers under
// mpeg-pcc-tmc13_amd/csrc compile against THIS file with plain g++ (it shadows
// <hip/hip_runtime.h> on the include path of the emulator build only) and the
// kernels then run on the CPU, one workgroup at a time, every thread a fiber
// (ucontext).  Wave collectives (__ballot, __shfl, ds_bpermute, DPP moves),
// __syncthreads and s_sleep are rendezvous points handled by a scheduler
// (emu_core.cpp).  A collective that the live lanes of a wavefront do not all
// reach together is reported as an error -- which is exactly the convergence
// discipline the gfx950 kernels need -- and a workgroup that can make no
// progress is reported as a deadlock.
//
// What this is for: the index arithmetic, list construction and hand-off
// protocols of the kernels can be checked against the oracle in the CPU test
// tier (-m "not gpu"), before a GPU is involved.  It is NOT a product path:
// nothing under mpeg-pcc-tmc13_amd/ includes or links it, and the library built for
// gfx950 never sees this header.
#pragma once
#define GPCC_EMU 1

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define address_space(n) unused
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

typedef void* hipStream_t;
typedef void* hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }

namespace emu {

enum OpKind { kOpBallot = 1, kOpShfl, kOpBarrier, kOpSleep };

struct Uint3 { unsigned x, y, z; };
extern Uint3 g_block_idx, g_block_dim, g_grid_dim;
Uint3 cur_thread_idx();
Uint3 cur_block_idx();
int cur_lane();
// how many workgroups of a launch run together (default 1 = one after another); more than one
// only for kernels without __shared__ data (see emu_core.cpp)
void set_concurrent_blocks(unsigned n);

// Rendezvous of the live lanes of the calling fiber's wavefront: every lane
// contributes `v`; on return snap[0..63] holds all contributions (0 for lanes
// that have exited) and *active the mask of lanes that took part.
void wave_exchange(int kind, uint64_t v, const uint64_t** snap, uint64_t* active);
void block_barrier();
void sleep_yield();

typedef void (*KernelThunk)(void* closure);
void launch(dim3 grid, dim3 block, KernelThunk fn, void* closure);

template<class F>
void launch_fn(dim3 grid, dim3 block, F f)
{
  launch(grid, block, [](void* c) { (*static_cast<F*>(c))(); }, &f);
}

template<class T>
inline uint64_t to_bits(T v)
{
  static_assert(sizeof(T) <= 8, "value wider than 64 bits");
  uint64_t b = 0;
  memcpy(&b, &v, sizeof(T));
  return b;
}
template<class T>
inline T from_bits(uint64_t b)
{
  T v;
  memcpy(&v, &b, sizeof(T));
  return v;
}

}  // namespace emu

#define threadIdx (emu::cur_thread_idx())
#define blockIdx (emu::cur_block_idx())
#define blockDim (emu::g_block_dim)
#define gridDim (emu::g_grid_dim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch_fn(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

// ---- wave collectives ------------------------------------------------------------
inline unsigned long long emu_ballot(int site, int pred)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpBallot | (site << 8), pred ? 1 : 0, &s, &act);
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++)
    if (((act >> i) & 1) && s[i])
      m |= 1ull << i;
  return m;
}
inline int emu_any(int site, int pred) { return emu_ballot(site, pred) != 0; }
inline int emu_all(int site, int pred)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpBallot | (site << 8), pred ? 1 : 0, &s, &act);
  for (int i = 0; i < 64; i++)
    if (((act >> i) & 1) && !s[i])
      return 0;
  return 1;
}
template<class T>
inline T emu_shfl(int site, T v, int src, int width = 64)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), emu::to_bits(v), &s, &act);
  const int lane = emu::cur_lane();
  const int from = (lane & ~(width - 1)) | (src & (width - 1));
  return emu::from_bits<T>(s[from & 63]);
}
template<class T>
inline T emu_shfl_xor(int site, T v, int mask, int width = 64)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), emu::to_bits(v), &s, &act);
  const int lane = emu::cur_lane();
  const int from = lane ^ mask;
  if ((from & ~(width - 1)) != (lane & ~(width - 1)))
    return v;
  return emu::from_bits<T>(s[from & 63]);
}
template<class T>
inline T emu_shfl_up(int site, T v, unsigned d, int width = 64)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), emu::to_bits(v), &s, &act);
  const int lane = emu::cur_lane();
  const int from = lane - (int)d;
  if (from < (lane & ~(width - 1)))
    return v;
  return emu::from_bits<T>(s[from]);
}
template<class T>
inline T emu_shfl_down(int site, T v, unsigned d, int width = 64)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), emu::to_bits(v), &s, &act);
  const int lane = emu::cur_lane();
  const int from = lane + (int)d;
  if (from >= (lane & ~(width - 1)) + width)
    return v;
  return emu::from_bits<T>(s[from]);
}
inline int emu_builtin_amdgcn_ds_bpermute(int site, int addr, int v)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), (uint32_t)v, &s, &act);
  return (int)(uint32_t)s[((unsigned)addr >> 2) & 63];
}
// forward permute: lane i sends v to lane addr/4 (an unwritten lane reads 0;
// two senders to one lane: the higher lane wins here, unspecified on hardware)
inline int emu_builtin_amdgcn_ds_permute(int site, int addr, int v)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(
    emu::kOpShfl | (site << 8), ((uint64_t)(((unsigned)addr >> 2) & 63) << 32) | (uint32_t)v, &s, &act);
  const int lane = emu::cur_lane();
  int r = 0;
  for (int i = 0; i < 64; i++)
    if (((act >> i) & 1) && (int)(s[i] >> 32) == lane)
      r = (int)(uint32_t)s[i];
  return r;
}
inline int emu_builtin_amdgcn_readlane(int site, int v, int src)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), (uint32_t)v, &s, &act);
  return (int)(uint32_t)s[src & 63];
}
inline int emu_builtin_amdgcn_readfirstlane(int site, int v)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), (uint32_t)v, &s, &act);
  for (int i = 0; i < 64; i++)
    if ((act >> i) & 1)
      return (int)(uint32_t)s[i];
  return v;
}
// the DPP controls the kernels use: quad_perm, row_half_mirror, row_mirror
inline int emu_builtin_amdgcn_update_dpp(int site, int old, int src, int ctrl, int row_mask, int bank_mask, bool)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), (uint32_t)src, &s, &act);
  const int lane = emu::cur_lane();
  int from;
  if (ctrl >= 0 && ctrl <= 0xff)
    from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl == 0x141)
    from = (lane & ~7) | (7 - (lane & 7));
  else if (ctrl == 0x140)
    from = (lane & ~15) | (15 - (lane & 15));
  else {
    fprintf(stderr, "emu: unsupported dpp_ctrl 0x%x\n", ctrl);
    abort();
  }
  (void)old;
  (void)row_mask;
  (void)bank_mask;
  return (int)(uint32_t)s[from];
}
#define __ballot(...) emu_ballot(__LINE__, __VA_ARGS__)
#define __any(...) emu_any(__LINE__, __VA_ARGS__)
#define __all(...) emu_all(__LINE__, __VA_ARGS__)
#define __shfl(...) emu_shfl(__LINE__, __VA_ARGS__)
#define __shfl_xor(...) emu_shfl_xor(__LINE__, __VA_ARGS__)
#define __shfl_up(...) emu_shfl_up(__LINE__, __VA_ARGS__)
#define __shfl_down(...) emu_shfl_down(__LINE__, __VA_ARGS__)
#define __builtin_amdgcn_ds_bpermute(...) emu_builtin_amdgcn_ds_bpermute(__LINE__, __VA_ARGS__)
#define __builtin_amdgcn_ds_permute(...) emu_builtin_amdgcn_ds_permute(__LINE__, __VA_ARGS__)
#define __builtin_amdgcn_readlane(...) emu_builtin_amdgcn_readlane(__LINE__, __VA_ARGS__)
#define __builtin_amdgcn_readfirstlane(...) emu_builtin_amdgcn_readfirstlane(__LINE__, __VA_ARGS__)
#define __builtin_amdgcn_update_dpp(...) emu_builtin_amdgcn_update_dpp(__LINE__, __VA_ARGS__)
// wave_barrier: a scheduling fence on hardware (the LDS executes a wavefront's
// instructions in order); here the lanes really have to meet, because lanes run
// one after the other between rendezvous points
inline void emu_wave_barrier(int site)
{
  const uint64_t* s;
  uint64_t act;
  emu::wave_exchange(emu::kOpShfl | (site << 8), 0, &s, &act);
}
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier(__LINE__)
inline void __syncthreads() { emu::block_barrier(); }
inline void __builtin_amdgcn_s_sleep(int) { emu::sleep_yield(); }
inline void __builtin_amdgcn_s_barrier() { emu::block_barrier(); }
inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }
inline void __threadfence() {}

// ---- atomics (one fiber runs at a time) -------------------------------------------
template<class T, class U>
inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template<class T, class U>
inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template<class T, class U>
inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template<class T, class U>
inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template<class T, class U>
inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template<class T, class U>
inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template<class T, class U, class V>
inline T atomicCAS(T* p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define __hip_atomic_fetch_or(p, v, order, scope) atomicOr((p), (v))
#define __hip_atomic_fetch_max(p, v, order, scope) atomicMax((p), (v))
#define __hip_atomic_exchange(p, v, order, scope) atomicExch((p), (v))

// ---- bit helpers -------------------------------------------------------------------
// ---- buffer instructions (the 16-byte mail-box granules of the dependency-ordered kernels) ----
// A resource is base + size; a load outside it returns zeros and a store there is dropped,
// as the hardware does.  One 16-byte access is one memcpy: whole, like the granule on the device.
struct emu_u32x4 {
  uint32_t x, y, z, w;
};
struct emu_rsrc {
  char* base;
  uint32_t bytes;
};
inline emu_rsrc emu_make_buffer_rsrc(const void* p, int, int num_bytes, int)
{
  return emu_rsrc{(char*)const_cast<void*>(p), (uint32_t)num_bytes};
}
inline emu_u32x4 emu_raw_buffer_load_b128(emu_rsrc r, int voffset, int soffset, int)
{
  emu_u32x4 v{0, 0, 0, 0};
  const uint32_t off = (uint32_t)voffset + (uint32_t)soffset;
  if (off + 16 <= r.bytes)
    memcpy(&v, r.base + off, 16);
  return v;
}
inline void emu_raw_buffer_store_b128(emu_u32x4 v, emu_rsrc r, int voffset, int soffset, int)
{
  const uint32_t off = (uint32_t)voffset + (uint32_t)soffset;
  if (off + 16 <= r.bytes)
    memcpy(r.base + off, &v, 16);
}
#define __builtin_amdgcn_make_buffer_rsrc(...) emu_make_buffer_rsrc(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_load_b128(...) emu_raw_buffer_load_b128(__VA_ARGS__)
#define __builtin_amdgcn_raw_buffer_store_b128(...) emu_raw_buffer_store_b128(__VA_ARGS__)

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
template<class A, class B>
inline auto min(A a, B b) -> decltype(a + b) { return a < b ? a : b; }
template<class A, class B>
inline auto max(A a, B b) -> decltype(a + b) { return a > b ? a : b; }
