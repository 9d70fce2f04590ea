// cumask_probe.hip -- which physical CUs does bit i of a stream's CU mask (hipExtStreamCreateWithCUMask) enable on MI355X?
// Launches many workgroups on a masked stream; each records (XCC_ID, SE, CU) from the hardware id registers.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <vector>

__global__ void probe(uint32_t* out) {
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay resident a little so that the grid spreads over every enabled CU
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) {}
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
  const int NB = 4096;
  uint32_t* d;
  hipMalloc(&d, sizeof(uint32_t) * 2 * NB);
  hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, s, d);
  hipStreamSynchronize(s);
  std::vector<uint32_t> h(2 * NB);
  hipMemcpy(h.data(), d, sizeof(uint32_t) * 2 * NB, hipMemcpyDeviceToHost);
  std::set<uint32_t> cus;
  int per_xcc[16] = {0};
  for (int i = 0; i < NB; ++i) {
    const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
    cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
  }
  for (uint32_t c : cus) per_xcc[c >> 16]++;
  printf("%s: %zu distinct CUs; per XCC:", name, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("\n");
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  std::vector<uint32_t> all(8, 0xffffffffu), first32(8, 0), last32(8, 0), every8(8, 0), first224(8, 0xffffffffu);
  first32[0] = 0xffffffffu;
  last32[7] = 0xffffffffu;
  first224[7] = 0;
  for (int i = 0; i < 256; i += 8) every8[i / 32] |= 1u << (i % 32);
  run("all 256 bits", all);
  run("bits 0..31", first32);
  run("bits 224..255", last32);
  run("bits 0..223", first224);
  run("every 8th bit", every8);
  return 0;
}
