// Microbenchmark (diagnostic, not part of the library): read bandwidth of the access pattern the
// full-sequence kernels use -- a workgroup walks ROWS x PIECE-byte pieces of a (rows, T) fp32
// matrix (row pitch T*4 bytes) -- against plain linear streaming of the same bytes.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/rowpiece_bw.hip -o gpurun_out/rowpiece_bw && gpurun_out/rowpiece_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));

// grid.x = (T / cols_per_wg) * (rows / ROWS); each workgroup reads ROWS rows x cols_per_wg columns,
// in steps of `piece` columns (piece * 4 bytes contiguous per row per step), 256 threads, 16 B per lane.
template <int ROWS>
__global__ __launch_bounds__(256) void rowpiece(const float *__restrict__ x, float *__restrict__ sink, int T,
                                                int cols_per_wg, int piece, int n_colblk) {
  const int cb = blockIdx.x % n_colblk, rb = blockIdx.x / n_colblk;
  const int lanes_per_row = piece / 4;            // lanes covering one row piece
  const int rows_per_pass = 256 / lanes_per_row;  // rows covered by one load instruction of the block
  const int r_in = threadIdx.x / lanes_per_row, c_in = 4 * (threadIdx.x % lanes_per_row);
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < cols_per_wg; c0 += piece) {
#pragma unroll 4
    for (int r = r_in; r < ROWS; r += rows_per_pass) {
      const float *p = x + (size_t)(rb * ROWS + r) * T + (size_t)cb * cols_per_wg + c0 + c_in;
      acc += *(const v4 *)p;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = acc.x;
}
__global__ __launch_bounds__(256) void linear(const float *__restrict__ x, float *__restrict__ sink, size_t n4) {
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += ((const v4 *)x)[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = acc.x;
}
int main() {
  const int T = 16000, rows = 16 * 64 * 24;  // 24 activation tensors of (16, 64, 16000): 1.57 GB > Infinity Cache
  const size_t n = (size_t)rows * T;
  float *x, *sink;
  hipMalloc(&x, n * 4); hipMalloc(&sink, 4);
  hipMemset(x, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch, const char *name) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-58s %8.3f ms  %6.2f TB/s\n", name, ms, n * 4.0 / ms / 1e9);
  };
  time([&] { hipLaunchKernelGGL(linear, dim3(256 * 8), dim3(256), 0, 0, x, sink, n / 4); }, "linear, 2048 workgroups grid-stride");
  for (int piece : {64, 128, 256, 512}) {
    for (int cols : {512, 2000}) {
      if (cols % piece) continue;
      const int ncb = T / cols;
      char name[128]; snprintf(name, sizeof name, "64 rows x %4d B pieces, %4d columns per workgroup", piece * 4, cols);
      time([&] { hipLaunchKernelGGL(rowpiece<64>, dim3(ncb * (rows / 64)), dim3(256), 0, 0, x, sink, T, cols, piece, ncb); }, name);
    }
  }
  for (int piece : {64, 256}) {
    const int cols = 2000 / piece * piece == 2000 ? 2000 : 512, ncb = T / cols;
    char name[128]; snprintf(name, sizeof name, "256 rows x %4d B pieces, %4d columns per workgroup", piece * 4, cols);
    time([&] { hipLaunchKernelGGL(rowpiece<256>, dim3(ncb * (rows / 256)), dim3(256), 0, 0, x, sink, T, cols, piece, ncb); }, name);
  }
  return 0;
}
