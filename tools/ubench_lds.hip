// ubench_lds.hip -- what do the LDS instructions of the event loops cost on gfx950, in the units of ubench_valu_issue.hip?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_lds.hip -o /tmp/ubench_lds && /tmp/ubench_lds
// One workgroup of 1024 threads per CU (4 waves per SIMD, 16 per CU: the event kernels' shape), a 64 KiB region of LDS, every lane
// its own address: RANDOM (uniform over the region), NEAR (lane i within a few cells of lane i - 1: what pixel-sorted events
// produce) or LINEAR (lane i -> cell i).  Loop of 8 independent instructions + s_waitcnt; s_memtime inside the kernel.
// Printed: cycles per wave-instruction as seen by ONE wave, and per CU (/ 16 waves) = what the LDS pipe needs per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

enum Op { READ2 = 0, READ_B32, ADD_F64, ADD_U64, ADD_F32, ADD_U32, ADD_F64_SPARSE, NOPS };
static const char* kNames[] = {"ds_read2_b32 (offsets 0,1)", "ds_read_b32", "ds_add_f64", "ds_add_u64", "ds_add_f32", "ds_add_u32",
                               "ds_add_f64, 6 of 64 lanes"};
enum Pat { RANDOM = 0, NEAR, LINEAR };
static const char* kPat[] = {"random", "near", "linear"};

template <int OP>
__global__ void __launch_bounds__(1024) bench(unsigned long long* out, const unsigned* addr, int iters) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = 0.0;
  unsigned a[8];
  for (int k = 0; k < 8; ++k) a[k] = addr[k * 1024 + threadIdx.x];  // byte addresses, 8-byte aligned for the 64-bit ops
  __syncthreads();
  double dv = 1.0;
  unsigned long long uv = 1;
  float fv = 1.0f;
  unsigned u = 1;
  float2 r[8];
  const bool sparse_on = (threadIdx.x & 63) % 11 == 0;  // 6 lanes of 64
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; ++it) {
    if (OP == READ2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(r[k]) : "v"(a[k]));
    } else if (OP == READ_B32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_read_b32 %0, %1" : "=v"(r[k].x) : "v"(a[k]));
    } else if (OP == ADD_F64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_add_f64 %0, %1" : : "v"(a[k]), "v"(dv) : "memory");
    } else if (OP == ADD_U64) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_add_u64 %0, %1" : : "v"(a[k]), "v"(uv) : "memory");
    } else if (OP == ADD_F32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_add_f32 %0, %1" : : "v"(a[k]), "v"(fv) : "memory");
    } else if (OP == ADD_U32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("ds_add_u32 %0, %1" : : "v"(a[k]), "v"(u) : "memory");
    } else if (OP == ADD_F64_SPARSE) {
      if (sparse_on) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("ds_add_f64 %0, %1" : : "v"(a[k]), "v"(dv) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float s = 0;
  for (int k = 0; k < 8; ++k) s += r[k].x + r[k].y;
  if (s == 1.2345f) out[4096] = 1;
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 16 + threadIdx.x / 64) % 4096] = t1 - t0;
}

template <int OP>
void run(unsigned long long* d_out, unsigned long long* h_out, unsigned* d_addr) {
  const int iters = 500;
  unsigned* h = (unsigned*)malloc(8 * 1024 * 4);
  for (int pat = 0; pat < 3; ++pat) {
    srand(7);
    for (int k = 0; k < 8; ++k) {
      unsigned cur = rand() % 8000;
      for (int t = 0; t < 1024; ++t) {
        unsigned cell;  // 8-byte cell index in [0, 8190)
        if (pat == RANDOM) cell = rand() % 8190;
        else if (pat == NEAR) { if ((t & 63) == 0) cur = rand() % 8000; cur = (cur + rand() % 3) % 8190; cell = cur; }
        else cell = (t + 64 * k) % 8190;
        h[k * 1024 + t] = cell * 8;
      }
    }
    hipMemcpy(d_addr, h, 8 * 1024 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
      bench<OP><<<256, 1024, 65536>>>(d_out, d_addr, iters);
      hipDeviceSynchronize();
    }
    hipMemcpy(h_out, d_out, 4096 * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < 4096; ++i) sum += (double)h_out[i];
    const double cyc_wave = sum / 4096 / (iters * 8.0);
    printf("%-28s %-7s %7.2f cycles per instruction per wave, %6.2f per CU\n", kNames[OP], kPat[pat], cyc_wave, cyc_wave / 16.0);
  }
  free(h);
}

int main() {
  unsigned long long *d_out, *h_out = (unsigned long long*)malloc(4097 * 8);
  unsigned* d_addr;
  hipMalloc(&d_out, 4097 * 8);
  hipMalloc(&d_addr, 8 * 1024 * 4);
  hipFuncSetAttribute((const void*)bench<READ2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  run<READ2>(d_out, h_out, d_addr); run<READ_B32>(d_out, h_out, d_addr); run<ADD_F64>(d_out, h_out, d_addr);
  run<ADD_U64>(d_out, h_out, d_addr); run<ADD_F32>(d_out, h_out, d_addr); run<ADD_U32>(d_out, h_out, d_addr);
  run<ADD_F64_SPARSE>(d_out, h_out, d_addr);
  return 0;
}
