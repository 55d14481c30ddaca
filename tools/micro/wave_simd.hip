// Dev micro-test: where do the waves of a workgroup land?  Every wave of 256- and 512-thread workgroups records its
// HW_ID (SIMD_ID = bits 5:4, CU_ID = bits 11:8, SE_ID = 15:13, WAVE_ID 3:0) - the premise of a wave-specialised kernel
// (one gather wave + one MFMA wave per SIMD) is that wave w of an 8-wave workgroup runs on SIMD w % 4.
// hipcc --offload-arch=gfx950 -O3 -o wave_simd wave_simd.hip && ./wave_simd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(unsigned* out, int spin) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  volatile float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;    // keep the workgroups resident together for a while
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
  for (int nt : {256, 512}) {
    const int nw = nt / 64, blocks = 1024;
    unsigned* d;
    hipMalloc(&d, blocks * nw * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(nt), 0, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * nw);
    hipMemcpy(h.data(), d, blocks * nw * 4, hipMemcpyDeviceToHost);
    int rr = 0, distinct4 = 0, same_cu = 0;
    int hist[8][4] = {};
    for (int b = 0; b < blocks; ++b) {
      bool ok = true, cu_ok = true;
      unsigned seen = 0;
      for (int w = 0; w < nw; ++w) {
        const unsigned id = h[b * nw + w], simd = (id >> 4) & 3;
        hist[w][simd]++;
        if (simd != ((h[b * nw] >> 4) + w) % 4) ok = false;
        if (((id >> 8) & 0xff) != ((h[b * nw] >> 8) & 0xff)) cu_ok = false;
        if (w < 4) seen |= 1u << simd;
      }
      rr += ok; same_cu += cu_ok; distinct4 += seen == 0xf;
    }
    printf("%d threads: %d workgroups; waves on consecutive SIMDs (w0+w)%%4: %d; first 4 waves on 4 distinct SIMDs: %d; all waves one CU: %d\n",
           nt, blocks, rr, distinct4, same_cu);
    for (int w = 0; w < nw; ++w) printf("  wave %d -> SIMD histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 4; ++b) { printf("  wg %d:", b); for (int w = 0; w < nw; ++w) printf(" %08x", h[b * nw + w]); printf("\n"); }
    hipFree(d);
  }
  return 0;
}
