// LDS atomic throughput vs bank-conflict degree on gfx950: hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic
// every wave issues ITER x 4 ds_or_rtn_b32 (or variants) on its own 8 KB table; addresses: lane l -> bank (l % 32 / k) * ... so that
// exactly k lanes of each 32-lane group share a bank (different words).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t __attribute__((address_space(3))) *lds32;
template <int MODE>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, int kdeg, int randomise)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x;
  for (uint32_t o = lane * 16; o < 8192; o += 1024) *(uint4 *)(smem + o) = make_uint4(0, 0, 0, 0);
  // lane -> (bank, word): k lanes per bank inside a group of 32
  const uint32_t g = lane & 31;
  uint32_t bank = g / kdeg, wsel = g % kdeg;
  uint32_t acc = 0;
  uint32_t rnd = lane * 2654435761u + blockIdx.x * 40503u + 12345u;
  // addresses are fixed per lane (8 of them, computed once): the loop is LDS instructions + one VALU each
  uint32_t ad[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (randomise == 1) { rnd = rnd * 1664525u + 1013904223u; ad[j] = (rnd >> 8) & 0x1ffcu; }
    else if (randomise == 2) {
      // like a sorted sub-row of ~220 postings over 65536 targets, lane l holding postings 4l..4l+3: target ~ (4l + j) * 298 +- noise
      rnd = rnd * 1664525u + 1013904223u;
      const uint32_t t = ((lane * 4u + (uint32_t)(j & 3)) * 298u + ((rnd >> 10) % 600u) + (uint32_t)(j >> 2) * 13u) & 0xffffu;
      ad[j] = (t >> 3) & 0x1ffcu;
    }
    else ad[j] = (((wsel + (uint32_t)j * 7u) & 63u) * 32u + bank) * 4u;
  }
  uint32_t bit = 1u << (lane & 31);
  for (int it = 0; it < iters; ++it) {
    uint32_t old[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) old[j] = __hip_atomic_fetch_or((lds32)(uintptr_t)ad[j], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 1) { old[j] = *(volatile lds32)(uintptr_t)ad[j]; }
      else if (MODE == 2) { __hip_atomic_fetch_or((lds32)(uintptr_t)ad[j], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); old[j] = 0; }
      else { *(volatile lds32)(uintptr_t)ad[j] = bit; old[j] = 0; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += old[j];
    bit = (bit << 1) | (bit >> 31);
  }
  out[blockIdx.x * 64 + lane] = acc;
}
int main(int argc, char **argv)
{
  int waves_per_cu = argc > 1 ? atoi(argv[1]) : 12;
  int iters = 10000;
  uint32_t *d; hipMalloc(&d, 256 * 32 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[] = {"ds_or_rtn_b32", "ds_read_b32", "ds_or_b32 (no rtn)", "ds_write_b32"};
  for (int mode = 0; mode < 4; ++mode)
    for (int cfg = 0; cfg < 7; ++cfg) {
      if (mode == 1) continue;
      int kdeg = cfg < 5 ? (cfg == 0 ? 1 : cfg == 1 ? 2 : cfg == 2 ? 3 : cfg == 3 ? 4 : 8) : 1, rnd = cfg == 5 ? 1 : (cfg == 6 ? 2 : 0);
      const int grid = 256 * waves_per_cu;
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 12 * 1024, 0, d, iters, kdeg, rnd);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 12 * 1024, 0, d, iters, kdeg, rnd);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 12 * 1024, 0, d, iters, kdeg, rnd);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 12 * 1024, 0, d, iters, kdeg, rnd);
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_cu = (double)waves_per_cu * iters * 8;
      printf("%-20s %-10s waves/CU %2d: %.3f ms, %.1f ns per wave-instruction per CU (= %.1f cycles at 2.4 GHz)\n", names[mode],
             rnd == 2 ? "sorted" : rnd ? "random" : (kdeg == 1 ? "1-way" : kdeg == 2 ? "2-way" : kdeg == 3 ? "3-way" : kdeg == 4 ? "4-way" : "8-way"), waves_per_cu, ms,
             ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    }
  return 0;
}
