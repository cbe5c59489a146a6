// Probe: ds_read_b64_tr_b16 through __builtin_amdgcn_ds_read_tr16_b64_v4i16 as the B operand fetch of a 32x32x16
// MFMA from a [row = k][col = t] tile of 16-bit values: lane l (li = l & 31, h = l >> 5) must receive
// tile[k0 + 8 h + j][c0 + li] in element j.  hipcc --offload-arch=gfx950 -O2 scripts/probes/tr_read.hip -o tr_read && ./tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int PITCH = 72;  // ushorts per row (144 bytes)
__global__ void k(const unsigned short *src, unsigned short *out, int k0, int c0) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[128 * PITCH];
  for (int i = threadIdx.x; i < 128 * PITCH; i += 64) tile[i] = src[i];
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  const int R0 = k0 + 8 * (g >> 1), C0 = c0 + 16 * (g & 1);
  typedef __attribute__((address_space(3))) v4s lds_v4s;
  lds_v4s *a0 = (lds_v4s *)(__attribute__((address_space(3))) unsigned short *)&tile[(R0 + q) * PITCH + C0 + 4 * p];
  lds_v4s *a1 = (lds_v4s *)(__attribute__((address_space(3))) unsigned short *)&tile[(R0 + 4 + q) * PITCH + C0 + 4 * p];
  const v4s x = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a0), y = __builtin_amdgcn_ds_read_tr16_b64_v4i16(a1);
  for (int e = 0; e < 4; ++e) {
    out[l * 8 + e] = (unsigned short)x[e];
    out[l * 8 + 4 + e] = (unsigned short)y[e];
  }
}
int main() {
  unsigned short h[128 * PITCH], *d, *o, r[64 * 8];
  for (int row = 0; row < 128; ++row)
    for (int c = 0; c < PITCH; ++c) h[row * PITCH + c] = (unsigned short)(row * 100 + c);
  hipMalloc(&d, sizeof h);
  hipMalloc(&o, sizeof r);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  int bad = 0;
  for (int k0 = 0; k0 < 128; k0 += 16)
    for (int c0 = 0; c0 < 64; c0 += 32) {
      k<<<1, 64>>>(d, o, k0, c0);
      hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int want = (k0 + 8 * (l >> 5) + j) * 100 + c0 + (l & 31);
          if (r[l * 8 + j] != want) {
            if (bad < 8) printf("k0 %d c0 %d lane %d elem %d: got %d want %d\n", k0, c0, l, j, r[l * 8 + j], want);
            ++bad;
          }
        }
    }
  printf("tr_read probe: %d mismatches\n", bad);
  return bad != 0;
}
