// Probe: how many workgroups of a given shape (threads, static + dynamic LDS) does a gfx950 CU actually hold at once?
// Every workgroup records (xcc, se, cu, t_start, t_end); the host sweeps over time and reports max / mean concurrency.
// build: hipcc -O2 --offload-arch=gfx950 tools/occupancy_probe.hip -o gpurun_out/occupancy_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

struct rec {
  unsigned           hwid, xcc;
  unsigned long long t0, t1;
};

template <int STATIC_BYTES>
__global__ void __launch_bounds__(192) probe(rec* out, int spin)
{
  __shared__ int8_t    fixed[STATIC_BYTES];
  extern __shared__ int dyn[];
  unsigned long long   t0 = (unsigned long long)wall_clock64();
  fixed[threadIdx.x] = (int8_t)threadIdx.x;
  dyn[threadIdx.x]   = threadIdx.x;
  __syncthreads();
  int acc = 0;
  for (int i = 0; i < spin; ++i) {
    acc += fixed[(threadIdx.x + i) % 192] + dyn[(threadIdx.x * 7 + i) % 192];
    __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = rec{hwid, xcc & 0xfu, t0, (unsigned long long)wall_clock64()};
    if (acc == 0x7fffffff)
      out[blockIdx.x].hwid = 0;
  }
}

int main(int argc, char** argv)
{
  const int n = 8192, spin = 2000;
  rec*      d;
  hipMalloc(&d, n * sizeof(rec));
  std::vector<rec> h(n);
  const int dyns[] = {4608, 5632, 6656, 8704, 10752, 11776, 12800, 13120, 13824, 14848};
  hipFuncSetAttribute((const void*)probe<26112>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 - 26112);
  hipFuncSetAttribute((const void*)probe<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 - 256);
  for (int variant = 0; variant < 2; ++variant)
    for (int dyn : dyns) {
      if (variant == 0)
        hipLaunchKernelGGL(probe<26112>, dim3(n), dim3(192), dyn, 0, d, spin);
      else
        hipLaunchKernelGGL(probe<256>, dim3(n), dim3(192), dyn + 25856, 0, d, spin);
      if (hipDeviceSynchronize() != hipSuccess) {
        printf("launch failed\n");
        return 1;
      }
      hipMemcpy(h.data(), d, n * sizeof(rec), hipMemcpyDeviceToHost);
      std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
      for (auto& r : h) {
        // HW_ID: [11:8] cu_id, [12] sh_id, [14:13] se_id (gfx9 layout) -> key on xcc + bits 8..15
        unsigned key = (r.xcc << 16) | ((r.hwid >> 8) & 0xff);
        ev[key].push_back({r.t0, +1});
        ev[key].push_back({r.t1, -1});
      }
      int    mx = 0;
      double mean = 0;
      for (auto& kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int                cur = 0, m = 0;
        unsigned long long last = kv.second.front().first, busy = 0, wsum = 0;
        for (auto& e : kv.second) {
          wsum += (e.first - last) * cur;
          busy += cur ? (e.first - last) : 0;
          last = e.first;
          cur += e.second;
          m = std::max(m, cur);
        }
        mx = std::max(mx, m);
        mean += busy ? (double)wsum / busy : 0;
      }
      printf("%s lds static+dyn = %6d B : CU keys %zu, max concurrent WGs/CU %d, mean %.2f\n", variant ? "all-dynamic" : "static26112",
             26112 + dyn, ev.size(), mx, mean / ev.size());
    }
  return 0;
}
