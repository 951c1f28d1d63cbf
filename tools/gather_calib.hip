// tools/gather_calib.hip -- what the memory system's random-request ceiling IS.
//
// k_search asks for ~4.8e10 random blocks per second and tools/gather_bench showed that rate to be the same for 16-, 64-
// and 128-byte blocks over 12-40 GB tables.  This tool varies the two things a request touches on its way - the cache
// level that holds the table and the page its address translates through - with everything else fixed:
//
//   table     2 MB (inside one XCD's 4 MB L2), 128 MB (inside the 256 MB Infinity Cache), 40 GB (HBM)
//   region    every wave-instruction draws its 64 lanes' blocks from ONE region of the table, chosen at random per
//             instruction: 4 KB (one page whatever the page size), 64 KB, 2 MB (one large page), 64 MB, 1 GB, the table.
//             The blocks stay random inside the region, so the DRAM side sees the same scatter; what changes is how
//             many translations (and, below 2 MB, DRAM rows/channels) one instruction needs.
//   pattern   G adjacent lanes read one aligned block of G x W bytes: (16, 8) = a 128-byte block of sixteen 8-byte
//             entries (k_search's PAM-pair table read), (4, 16) = a 64-byte block (an Occ block, a deep-table line),
//             (1, 16) = one 16-byte word per lane, 64 blocks per instruction (a context-word gather).
//
// One kernel instantiation per pattern and one dispatch per (table, region): `rocprofv3 --pmc ... -- tools/gather_calib`
// gives the counters per row (FETCH_SIZE: bytes per request by block size; TCC_HIT/TCC_MISS/TCC_EA0_RDREQ: where the
// requests are served).  Build: hipcc -O3 --offload-arch=gfx950 -o tools/gather_calib tools/gather_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

template <int G, int W>
__global__ __launch_bounds__(256) void k_calib(const uint8_t* __restrict__ tab, uint64_t table_bytes, uint64_t region_bytes, uint32_t iters,
                                               uint32_t* out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t wave = tid >> 6, gid = tid / G, piece = tid % G;
  const uint64_t n_regions = table_bytes / region_bytes, blocks_per_region = region_bytes / (uint64_t)(G * W);
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint64_t key = (uint64_t)it * 4u + j;
      const uint64_t r = (uint64_t)(((unsigned __int128)mix(((uint64_t)wave << 32) ^ key ^ 0x1234567ull) * n_regions) >> 64);
      const uint64_t b = (uint64_t)(((unsigned __int128)mix(((uint64_t)gid << 32) ^ key ^ 0xABCDEF01ull) * blocks_per_region) >> 64);
      const uint8_t* p = tab + r * region_bytes + b * (uint64_t)(G * W) + piece * W;
      if (W == 16) {
        const uint4 v = *(const uint4*)p;
        acc += v.x ^ v.y ^ v.z ^ v.w;
      } else if (W == 8) {
        const uint2 v = *(const uint2*)p;
        acc += v.x ^ v.y;
      } else {
        acc += *(const uint32_t*)p;
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc; /* keep the loads alive */
}

template <int G, int W>
static void run(const uint8_t* tab, uint64_t table_bytes, uint64_t region_bytes, uint32_t* out) {
  const int grid = 5120;
  const uint32_t iters = 128;
  if (region_bytes > table_bytes) region_bytes = table_bytes;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_calib<G, W>), dim3(grid), dim3(256), 0, 0, tab, table_bytes, region_bytes, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double blocks = (double)grid * 256 * iters * 4 / G;
  printf("{\"pattern\": \"%d lanes x %d B\", \"block_bytes\": %d, \"table_mib\": %.0f, \"region_kib\": %.0f, \"ms\": %.3f, \"gblocks_per_s\": %.2f, "
         "\"gb_per_s\": %.1f, \"blocks_per_dispatch\": %.0f}\n",
         G, W, G * W, table_bytes / 1048576.0, region_bytes / 1024.0, ms, blocks / ms / 1e6, blocks * G * W / ms / 1e6, blocks);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const uint64_t max_bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 40960ull) << 20;
  uint8_t* tab;
  uint32_t* out;
  if (hipMalloc(&tab, max_bytes) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  hipMalloc(&out, 64);
  hipMemset(tab, 0, max_bytes);
  hipDeviceSynchronize();
  const uint64_t KB = 1024, MB = 1024 * KB, GB = 1024 * MB;
  const uint64_t tables[] = {2 * MB, 128 * MB, max_bytes};
  for (uint64_t t : tables) {
    if (t > max_bytes) continue;
    run<16, 8>(tab, t, t, out);
    run<4, 16>(tab, t, t, out);
    run<1, 16>(tab, t, t, out);
  }
  const uint64_t regions[] = {4 * KB, 64 * KB, 2 * MB, 64 * MB, 1 * GB};
  for (uint64_t r : regions) {
    run<16, 8>(tab, max_bytes, r, out);
    run<4, 16>(tab, max_bytes, r, out);
    run<1, 16>(tab, max_bytes, r, out);
  }
  return 0;
}
