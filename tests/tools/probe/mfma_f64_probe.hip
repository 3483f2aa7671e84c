// Development probe: lane maps of v_mfma_f64_4x4x4_4b_f64 and v_mfma_f64_16x16x4_f64 on gfx950, found with exact
// integer data (a_l = l + 1, b = indicator of one lane).  hipcc --offload-arch=gfx950 mfma_f64_probe.hip -o mfma_f64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe4(double *out) {   // out[p*64 + lane]
  const int l = threadIdx.x;
  for (int p = 0; p < 64; ++p) {
    double a = l + 1, b = (l == p) ? 1.0 : 0.0, c = 0.0;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
    out[p * 64 + l] = d;
  }
}
int main() {
  double *d; hipMalloc(&d, 64 * 64 * 8);
  hipLaunchKernelGGL(probe4, dim3(1), dim3(64), 0, 0, d);
  static double h[64 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // for B lane p: which output lanes are non-zero and which A lane they carry
  for (int p = 0; p < 64; ++p) {
    printf("b-lane %2d:", p);
    for (int o = 0; o < 64; ++o) if (h[p * 64 + o] != 0.0) printf(" out%d<-a%d", o, (int)h[p * 64 + o] - 1);
    printf("\n");
  }
  return 0;
}
