// gemm_lab_h.hip -- measuring bench of the bf16-operand GEMM tile kernel (audiogan_amd/csrc/gemm_bf16_tile.h).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab_h.hip -o tools/gemm_lab_h ;  ./tools/gemm_lab_h [M N K]
// Prints per (layout, tile shape): microseconds, TFLOP/s, and the largest deviation from a float64 host reference on a sample
// of the output (operands are exactly representable bf16 values, so only the fp32 summation order differs).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

#define AG_GEMMH_TILE_CASES(F) F(1, 256, 128, 2, 2, 1, 4) F(2, 256, 256, 2, 4, 2, 2) F(3, 256, 256, 4, 4, 2, 1) F(4, 256, 256, 2, 4, 1, 2)
#include "../audiogan_amd/csrc/gemm_bf16_tile.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned short f2bf(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 16384, N = argc > 3 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
  std::vector<unsigned short> a((size_t)M * K), b((size_t)N * K), at((size_t)M * K), bt((size_t)N * K);
  uint32_t s = 777;
  auto nextf = [&]() {
    s = s * 1664525u + 1013904223u; const float u1 = ((s >> 8) + 1) / 16777217.f;
    s = s * 1664525u + 1013904223u; const float u2 = (s >> 8) / 16777216.f;
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
  };
  for (auto& v : a) v = f2bf(nextf());
  for (auto& v : b) v = f2bf(nextf());
  for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) at[(size_t)k * M + m] = a[(size_t)m * K + k];
  for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) bt[(size_t)k * N + n] = b[(size_t)n * K + k];
  unsigned short *dA, *dB, *dAt, *dBt, *dC16;
  float* dC;
  CK(hipMalloc(&dA, a.size() * 2)); CK(hipMalloc(&dB, b.size() * 2)); CK(hipMalloc(&dAt, a.size() * 2)); CK(hipMalloc(&dBt, b.size() * 2));
  CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dC16, (size_t)M * N * 2));
  CK(hipMemcpy(dA, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, b.data(), b.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dAt, at.data(), a.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dBt, bt.data(), b.size() * 2, hipMemcpyHostToDevice));
  // float64 reference on a sample
  const int NS = 256;
  std::vector<int> sm(NS), sn(NS);
  std::vector<double> ref(NS);
  for (int i = 0; i < NS; ++i) {
    s = s * 1664525u + 1013904223u; sm[i] = (s >> 8) % M;
    s = s * 1664525u + 1013904223u; sn[i] = (s >> 8) % N;
    double acc = 0;
    for (int k = 0; k < K; ++k) acc += (double)bf2f(a[(size_t)sm[i] * K + k]) * bf2f(b[(size_t)sn[i] * K + k]);
    ref[i] = acc;
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("M %d N %d K %d (bf16 operands)\n", M, N, K);
  std::vector<float> hc((size_t)M * N);
  for (int c16 = 0; c16 < 2; ++c16)
    for (int shape = 1; shape <= 4; ++shape)
      for (int lay = 0; lay < 4; ++lay) {
        const int ta = lay >> 1, tb = lay & 1;
        GemmH q;
        q.A = ta ? dAt : dA; q.B = tb ? dB : dBt; q.C = c16 ? nullptr : dC; q.C16 = c16 ? dC16 : nullptr; q.bias = nullptr; q.res = nullptr;
        q.res16 = nullptr; q.gate16 = nullptr; q.part = nullptr;
        q.lda = ta ? M : K; q.ldb = tb ? K : N; q.ldc = N; q.ldc16 = N; q.ldres = q.ldres16 = q.ldgate16 = 0;
        q.M = M; q.N = N; q.K = K; q.ksplit = 1; q.kchunk = K; q.act = 0; q.alpha = 1.f; q.beta = 0.f; q.slope = 0.f;
        CK(hipMemset(dC, 0, (size_t)M * N * 4));
        for (int i = 0; i < 5; ++i) gemm_bf16t_launch(q, ta, tb, shape, 0);
        CK(hipDeviceSynchronize());
        const int reps = 40;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) gemm_bf16t_launch(q, ta, tb, shape, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
        double dev = -1;
        if (!c16) {
          CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
          dev = 0;
          for (int i = 0; i < NS; ++i) { const double d = fabs(hc[(size_t)sm[i] * N + sn[i]] - ref[i]); if (d > dev) dev = d; }
        }
        int bm, bn;
        gemm_bf16t_dims(shape, bm, bn);
        printf("%s ta=%d tb=%d shape %d (%dx%d)  %8.1f us %7.1f TF   dev %g\n", c16 ? "C16" : "C32", ta, tb, shape, bm, bn, us, tf, dev);
        fflush(stdout);
      }
  return 0;
}
