// fetch_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE on gfx950 in the access patterns
// of mrk::scan_bm_kernel (tools/traffic.sh runs it under `rocprofv3 --pmc FETCH_SIZE`):
//   calib_stream4   each wave reads 256-B rows (64 lanes x 4 B, coalesced) of two arrays in bursts of 4 rows -- the
//                   bitmap windows;
//   calib_stream16  16 B per lane (1 KiB per wave instruction) -- the pattern MI355X_MICROARCH.md calibrated (x 0.5);
//   calib_gather4_gG  every lane reads ONE 4-B word out of each cell of G consecutive words, cells in ascending order
//                   (word = cell * G + hash(cell) % G) -- the tf / field gathers by rank at match density 1/G.  The host
//                   counts the distinct 64-B and 128-B lines the launch touches with the same hash.
// Prints one JSON line: per kernel the exact bytes requested / lines touched per launch.  Buffers are 4 GiB each
// (>> the 256 MiB Infinity Cache) and every launch sweeps them once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__host__ __device__ inline uint32_t mix(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return (uint32_t)x;
}

constexpr int WG = 256;

// rows of 64 words; wave w of the grid owns rows [w * rows_per_wave, +rows_per_wave)
__global__ __launch_bounds__(WG) void calib_stream4(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t rows_per_wave,
                                                    uint64_t n_rows, uint32_t* __restrict__ out) {
  const uint64_t wave = (uint64_t)blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t r0 = wave * rows_per_wave, r1 = r0 + rows_per_wave;
  if (r1 > n_rows) r1 = n_rows;
  uint32_t acc = 0;
  for (uint64_t r = r0; r < r1; r += 4) {
    uint32_t av[4], bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint64_t rr = r + i < r1 ? r + i : r1 - 1;
      av[i] = a[rr * 64 + lane];
      bv[i] = b[rr * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc += __popc(av[i] & bv[i]);
  }
  if (acc == 0x7fffffffu) out[wave] = acc; // never true for the test data; keeps the loads alive
}

__global__ __launch_bounds__(WG) void calib_stream16(const uint4* __restrict__ a, uint64_t n_vec, uint32_t* __restrict__ out) {
  const uint64_t stride = (uint64_t)gridDim.x * WG;
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < n_vec; i += stride) {
    const uint4 v = a[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x7fffffffu) out[0] = acc;
}

template <int G>
__global__ __launch_bounds__(WG) void calib_gather4(const uint32_t* __restrict__ a, uint64_t cells_per_wave, uint64_t n_cells,
                                                    uint32_t* __restrict__ out) {
  const uint64_t wave = (uint64_t)blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t c0 = wave * cells_per_wave, c1 = c0 + cells_per_wave;
  if (c1 > n_cells) c1 = n_cells;
  uint32_t acc = 0;
  for (uint64_t c = c0 + lane; c < c1; c += 64) acc += a[c * G + mix(c) % G];
  if (acc == 0x7fffffffu) out[wave] = acc;
}

template <int G>
static void run_gather(const uint32_t* d, uint64_t n_words, uint32_t* d_out, int reps, bool last) {
  const uint64_t n_cells = n_words / G;
  const uint64_t waves = 256ull * 4 * 16; // 16 waves per SIMD-set: plenty in flight
  const uint64_t cpw = (n_cells + waves - 1) / waves;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((calib_gather4<G>), dim3((unsigned)(waves / (WG / 64))), dim3(WG), 0, 0, d, cpw, n_cells, d_out);
  CK(hipDeviceSynchronize());
  uint64_t l64 = 0, l128 = 0, p64 = ~0ull, p128 = ~0ull;
  for (uint64_t c = 0; c < n_cells; ++c) { // words ascend with the cell: distinct lines = changes of the line number
    const uint64_t w = c * G + mix(c) % G;
    if ((w >> 4) != p64) ++l64, p64 = w >> 4;
    if ((w >> 5) != p128) ++l128, p128 = w >> 5;
  }
  printf("\"calib_gather4<%d>\": {\"words_read\": %llu, \"lines64\": %llu, \"lines128\": %llu, \"launches\": %d}%s", G,
         (unsigned long long)n_cells, (unsigned long long)l64, (unsigned long long)l128, reps, last ? "" : ", ");
}

int main(int argc, char** argv) {
  const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4;
  const int reps = argc > 2 ? atoi(argv[2]) : 3;
  const uint64_t bytes = gib << 30, n_words = bytes / 4;
  uint32_t *a = nullptr, *b = nullptr, *out = nullptr;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&out, 1 << 22));
  CK(hipMemset(a, 0x5a, bytes));
  CK(hipMemset(b, 0x33, bytes));
  CK(hipDeviceSynchronize());
  printf("{\"buffer_bytes\": %llu, ", (unsigned long long)bytes);
  {
    const uint64_t n_rows = n_words / 64, waves = 256ull * 4 * 6; // 6144 work items like the bench's bitmap launches
    const uint64_t rpw = ((n_rows + waves - 1) / waves + 3) / 4 * 4;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(calib_stream4, dim3((unsigned)(waves / (WG / 64))), dim3(WG), 0, 0, a, b, rpw, n_rows, out);
    CK(hipDeviceSynchronize());
    printf("\"calib_stream4\": {\"bytes\": %llu, \"launches\": %d}, ", (unsigned long long)(2 * bytes), reps);
  }
  {
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(calib_stream16, dim3(256 * 8), dim3(WG), 0, 0, (const uint4*)a, bytes / 16, out);
    CK(hipDeviceSynchronize());
    printf("\"calib_stream16\": {\"bytes\": %llu, \"launches\": %d}, ", (unsigned long long)bytes, reps);
  }
  run_gather<2>(a, n_words, out, reps, false);
  run_gather<4>(a, n_words, out, reps, false);
  run_gather<8>(a, n_words, out, reps, false);
  run_gather<16>(a, n_words, out, reps, false);
  run_gather<32>(a, n_words, out, reps, false);
  run_gather<64>(a, n_words, out, reps, true);
  printf("}\n");
  CK(hipFree(a));
  CK(hipFree(b));
  CK(hipFree(out));
  return 0;
}
