// tools/gather_bench.hip -- calibration microbenchmark:
// random aligned B-byte block reads (B = 16..128) over a table of T bytes, one block per lane
// per iteration, as k_search issues them (16-byte vector loads, uncoalesced across lanes).
// Prints achieved GB/s and reads/s; run under rocprofv3 --pmc FETCH_SIZE to calibrate the
// counter for this access pattern (known byte count = reads * B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int B>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ tab, uint64_t nblocks,
                                                uint32_t iters, uint32_t* out) {
  uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;           // xorshift64
    const uint64_t b = (uint64_t)(((unsigned __int128)s * nblocks) >> 64);
    const uint4* p = tab + b * (B / 16);
#pragma unroll
    for (int j = 0; j < B / 16; j++) { uint4 v = p[j]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x12345678u) out[0] = acc;   // keep the loads alive
}
// dependent-chain variant: next address depends on the loaded data (latency bound per lane,
// like a DFS step), several independent chains per lane are NOT used: one chain per lane.
template <int B>
__global__ __launch_bounds__(256) void k_chase(const uint4* __restrict__ tab, uint64_t nblocks,
                                               uint32_t iters, uint32_t* out) {
  uint64_t s = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 999;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint64_t b = (uint64_t)(((unsigned __int128)(s + acc) * nblocks) >> 64);
    const uint4* p = tab + b * (B / 16);
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < B / 16; j++) { uint4 v = p[j]; x += v.x ^ v.y ^ v.z ^ v.w; }
    acc = x & 1u;   // tables are zero-filled: acc stays 0 but the dependency is real
  }
  if (acc == 0x12345678u) out[0] = acc;
}


// quad-cooperative variant: 4 adjacent lanes read ONE random 64-byte block (16 B each), and every
// lane takes part in 4 different blocks per iteration: same number of load instructions per lane
// as k_gather<64>, but each wave-instruction touches 16 lines instead of 64.
__global__ __launch_bounds__(256) void k_gather_quad(const uint4* __restrict__ tab, uint64_t nblocks,
                                                     uint32_t iters, uint32_t* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t s = (uint64_t)(tid >> 2) * 0x9E3779B97F4A7C15ull + 777;
  const uint32_t piece = tid & 3;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const uint64_t b = (uint64_t)(((unsigned __int128)s * nblocks) >> 64);
      uint4 v = tab[b * 4 + piece];
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
static void run_quad(const uint4* tab, uint64_t bytes, int grid, uint32_t iters, uint32_t* out) {
  const uint64_t nblocks = bytes / 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_gather_quad, dim3(grid), dim3(256), 0, 0, tab, nblocks, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double reads = (double)grid * 256 * iters;   // blocks = lanes*iters*4/4
    if (rep == 1)
      printf("quad   B= 64 table=%6.0f MiB grid=%5d iters=%4u : %8.3f ms  %7.2f Gblocks/s  %8.1f GB/s\n",
             bytes / 1048576.0, grid, iters, ms, reads / ms / 1e6, reads * 64 / ms / 1e6);
  }
}

// group-cooperative variant, general: G adjacent lanes read ONE random block of G x W bytes (W bytes each) with one
// load instruction, and every lane takes part in 4 blocks per iteration.  (4, 16) is the quad pattern above;
// (8, 16) one 128-byte block per instruction as eight 16-byte pieces; (16, 8) a 128-byte block of sixteen 8-byte
// entries - k_search's read of a PAM-pair block; (16, 4) a 64-byte block of sixteen 4-byte words.  The question
// DESIGN 9.2 turns on: does a 128-byte block cost the memory system one random request or two?
template <int G, int W>
__global__ __launch_bounds__(256) void k_gather_group(const uint8_t* __restrict__ tab, uint64_t nblocks, uint32_t iters,
                                                      uint32_t* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t s = (uint64_t)(tid / G) * 0x9E3779B97F4A7C15ull + 4242;
  const uint32_t piece = tid % G;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const uint64_t b = (uint64_t)(((unsigned __int128)s * nblocks) >> 64);
      const uint8_t* p = tab + b * (uint64_t)(G * W) + piece * W;
      if (W == 16) { const uint4 v = *(const uint4*)p; acc += v.x ^ v.y ^ v.z ^ v.w; }
      else if (W == 8) { const uint2 v = *(const uint2*)p; acc += v.x ^ v.y; }
      else { acc += *(const uint32_t*)p; }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}
template <int G, int W>
static void run_group(const uint4* tab, uint64_t bytes, int grid, uint32_t iters, uint32_t* out) {
  const uint64_t nblocks = bytes / (G * W);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather_group<G, W>), dim3(grid), dim3(256), 0, 0, (const uint8_t*)tab, nblocks, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double blocks = (double)grid * 256 * iters * 4 / G;
    const double lines = blocks * ((G * W + 63) / 64);
    if (rep == 1)
      printf("group  %2d lanes x %2d B = %3d-byte blocks table=%6.0f MiB : %8.3f ms  %7.2f Gblocks/s  %7.2f G 64-byte lines/s  %8.1f GB/s\n",
             G, W, G * W, bytes / 1048576.0, ms, blocks / ms / 1e6, lines / ms / 1e6, blocks * G * W / ms / 1e6);
  }
}

template <int B>
static void run(const uint4* tab, uint64_t bytes, int grid, uint32_t iters, uint32_t* out, bool chase) {
  const uint64_t nblocks = bytes / B;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    if (chase) hipLaunchKernelGGL(k_chase<B>, dim3(grid), dim3(256), 0, 0, tab, nblocks, iters, out);
    else hipLaunchKernelGGL(k_gather<B>, dim3(grid), dim3(256), 0, 0, tab, nblocks, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double reads = (double)grid * 256 * iters;
    if (rep == 1)
      printf("%s B=%3d table=%6.0f MiB grid=%5d iters=%4u : %8.3f ms  %7.2f Greads/s  %8.1f GB/s\n",
             chase ? "chase " : "gather", B, bytes / 1048576.0, grid, iters, ms, reads / ms / 1e6,
             reads * B / ms / 1e6);
  }
}

int main(int argc, char** argv) {
  const uint64_t max_bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 4096ull) << 20;
  uint4* tab; uint32_t* out;
  if (hipMalloc(&tab, max_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 64);
  hipMemset(tab, 0, max_bytes);
  hipDeviceSynchronize();
  const uint64_t sizes[] = {3200ull << 20, 12000ull << 20, max_bytes};
  for (uint64_t sz : sizes) {
    if (sz > max_bytes) continue;
    for (int grid : {5120}) {
      run<16>(tab, sz, grid, 256, out, false);
      run<32>(tab, sz, grid, 256, out, false);
      run<64>(tab, sz, grid, 256, out, false);
      run<128>(tab, sz, grid, 128, out, false);
      run_quad(tab, sz, grid, 256, out);
      run_group<8, 16>(tab, sz, grid, 128, out);
      run_group<16, 8>(tab, sz, grid, 128, out);
      run_group<16, 4>(tab, sz, grid, 128, out);
      run_group<8, 8>(tab, sz, grid, 128, out);
    }

  }
  return 0;
}
