// SPDX-License-Identifier: GPL-3.0-or-later
// valu_probe.hip -- dev probe (round 5): what the instructions a cheaper streaming filter would be made of cost on gfx950,
// and what the packed SAD instructions compute exactly.
//
//   * throughput (wave instructions per cycle and SIMD, from 256 CUs x 4 SIMDs with 8 waves each issuing an unrolled
//     stream of independent instructions): v_add_u32 (reference: full rate), v_alignbit_b32, v_sub_u32_sdwa on byte
//     lanes, v_pk_sub_u16, v_pk_min_u16, v_qsad_pk_u16_u8, v_mqsad_pk_u16_u8, v_sad_u8, v_msad_u8
//   * semantics of v_qsad_pk_u16_u8 / v_mqsad_pk_u16_u8 / v_msad_u8 against a CPU model (which operand is the window,
//     which the reference, which bytes the masked forms leave out)
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe.bin && ./tools/valu_probe.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
   do {                                                                                           \
      hipError_t e_ = (x);                                                                        \
      if (e_ != hipSuccess) {                                                                     \
         fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                  \
         exit(1);                                                                                 \
      }                                                                                           \
   } while (0)

constexpr int kIters = 4096;
constexpr int kUnroll = 16;

// 16 independent chains of one instruction, kIters times; the results are folded so that nothing is dead
template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t *out, uint32_t seed)
{
   uint32_t a[kUnroll];
   unsigned long long q[kUnroll];
#pragma unroll
   for (int k = 0; k < kUnroll; k++) {
      a[k] = seed * (k + 1) + threadIdx.x;
      q[k] = ((unsigned long long)a[k] << 32) | (a[k] * 2654435761u);
   }
   const uint32_t c = seed ^ 0x01020304u;
   for (int i = 0; i < kIters; i++) {
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
         if constexpr (OP == 0) {
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 1) {
            asm volatile("v_alignbit_b32 %0, %0, %1, 24" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 2) {
            asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_0"
                         : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 3) {
            asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 4) {
            asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 5) {
            asm volatile("v_qsad_pk_u16_u8 %0, %0, %1, 0" : "+v"(q[k]) : "v"(c));
         }
         else if constexpr (OP == 6) {
            asm volatile("v_mqsad_pk_u16_u8 %0, %0, %1, 0" : "+v"(q[k]) : "v"(c));
         }
         else if constexpr (OP == 7) {
            asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 8) {
            asm volatile("v_msad_u8 %0, %0, %1, 0" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 9) {
            asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 10) {
            asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 11) {
            asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 13) {
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 14) {
            asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 15) {
            asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 16) {
            asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 17) {
            asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[k]));
         }
         else if constexpr (OP == 18) {
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "s"(c));
         }
         else if constexpr (OP == 19) {
            asm volatile("v_alignbit_b32 %0, %0, %1, 24" : "+v"(a[k]) : "s"(c));
         }
         else if constexpr (OP == 20) {
            asm volatile("v_and_b32 %0, 0x7f7f7f7f, %0" : "+v"(a[k]));
         }
         else if constexpr (OP == 21) {
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(c));
         }
         else if constexpr (OP == 12) {
            asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(*reinterpret_cast<__uint128_t *>(&q[k & ~1])) : "v"(q[k | 1]), "v"(c));
         }
      }
   }
   uint32_t r = 0;
#pragma unroll
   for (int k = 0; k < kUnroll; k++) {
      r ^= a[k] ^ (uint32_t)q[k] ^ (uint32_t)(q[k] >> 32);
   }
   if (r == 0x12345u) {
      out[threadIdx.x] = r;
   }
}

__global__ void semantics(const unsigned long long *win, const uint32_t *ref, const unsigned long long *acc, unsigned long long *q,
                          unsigned long long *mq, uint32_t *msad, int n)
{
   const int i = blockIdx.x * blockDim.x + threadIdx.x;
   if (i < n) {
      q[i] = __builtin_amdgcn_qsad_pk_u16_u8(win[i], ref[i], acc[i]);
      mq[i] = __builtin_amdgcn_mqsad_pk_u16_u8(win[i], ref[i], acc[i]);
      msad[i] = __builtin_amdgcn_msad_u8((uint32_t)win[i], ref[i], (uint32_t)acc[i]);
   }
}

template <int OP>
double run(const char *name, uint32_t *d_out, int cus, double mhz)
{
   const int blocks = cus * 8;                      // 8 workgroups of 4 waves per CU = 8 waves per SIMD
   hipEvent_t e0, e1;
   CHECK(hipEventCreate(&e0));
   CHECK(hipEventCreate(&e1));
   for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, nullptr, d_out, 12345u + rep);
   }
   CHECK(hipEventRecord(e0, nullptr));
   const int reps = 5;
   for (int rep = 0; rep < reps; rep++) {
      hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, nullptr, d_out, 777u + rep);
   }
   CHECK(hipEventRecord(e1, nullptr));
   CHECK(hipEventSynchronize(e1));
   float ms = 0;
   CHECK(hipEventElapsedTime(&ms, e0, e1));
   const double wave_insts = (double)reps * blocks * 4.0 * kIters * kUnroll;
   const double per_simd_cycle = wave_insts / (cus * 4.0) / (ms * 1e-3 * mhz * 1e6);
   printf("%-22s %8.3f ms  %.3f wave instructions per cycle and SIMD (1 wave64 instruction = 4 cycles at full rate: 0.25)  => %.2f x the cost of v_add_u32's slot\n",
          name, ms / reps, per_simd_cycle, 0.25 / per_simd_cycle);
   return per_simd_cycle;
}

int main()
{
   int dev = 0, cus = 0, khz = 0;
   CHECK(hipGetDevice(&dev));
   CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
   CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev));
   const double mhz = khz / 1000.0;
   printf("# valu_probe: %d CUs, %.0f MHz (attribute; the achieved clock under load may be lower)\n", cus, mhz);
   uint32_t *d_out = nullptr;
   CHECK(hipMalloc(&d_out, 4096));
   run<0>("v_add_u32", d_out, cus, mhz);
   run<0>("v_add_u32 (again)", d_out, cus, mhz);
   run<1>("v_alignbit_b32", d_out, cus, mhz);
   run<2>("v_sub_u32_sdwa (byte)", d_out, cus, mhz);
   run<3>("v_pk_sub_u16", d_out, cus, mhz);
   run<4>("v_pk_min_u16", d_out, cus, mhz);
   run<5>("v_qsad_pk_u16_u8", d_out, cus, mhz);
   run<6>("v_mqsad_pk_u16_u8", d_out, cus, mhz);
   run<7>("v_sad_u8", d_out, cus, mhz);
   run<8>("v_msad_u8", d_out, cus, mhz);
   run<9>("v_perm_b32", d_out, cus, mhz);
   run<10>("v_bfi_b32", d_out, cus, mhz);
   run<11>("v_and_or_b32", d_out, cus, mhz);
   run<12>("v_mqsad_u32_u8", d_out, cus, mhz);
   run<13>("v_xor_b32 (VOP2)", d_out, cus, mhz);
   run<14>("v_sub_u32 (VOP2)", d_out, cus, mhz);
   run<15>("v_add_u32_e64 (VOP3)", d_out, cus, mhz);
   run<16>("v_add3_u32", d_out, cus, mhz);
   run<17>("v_lshlrev_b32 imm", d_out, cus, mhz);
   run<18>("v_xor_b32 v, s", d_out, cus, mhz);
   run<19>("v_alignbit_b32 v, s", d_out, cus, mhz);
   run<20>("v_and_b32 literal", d_out, cus, mhz);
   run<21>("v_mov_b32_dpp wave_shr", d_out, cus, mhz);

   // ---- semantics ---------------------------------------------------------------------------------------------------
   const int n = 1 << 16;
   std::vector<unsigned long long> win(n), acc(n), q(n), mq(n);
   std::vector<uint32_t> ref(n), ms(n);
   uint64_t x = 88172645463325252ull;
   auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
   for (int i = 0; i < n; i++) {
      win[i] = rnd();
      ref[i] = (uint32_t)rnd();
      acc[i] = (i & 1) ? 0 : (rnd() & 0x00ff00ff00ff00ffull);
      if (i % 3 == 0) {
         ref[i] &= (uint32_t)rnd() | 0xff00ff00u;      // some zero bytes in the reference
      }
      if (i % 5 == 0) {
         win[i] &= rnd() | 0x00ff00ff00ff00ffull;      // ... and in the window
      }
   }
   unsigned long long *d_win, *d_acc, *d_q, *d_mq;
   uint32_t *d_ref, *d_ms;
   CHECK(hipMalloc(&d_win, n * 8)); CHECK(hipMalloc(&d_acc, n * 8)); CHECK(hipMalloc(&d_q, n * 8)); CHECK(hipMalloc(&d_mq, n * 8));
   CHECK(hipMalloc(&d_ref, n * 4)); CHECK(hipMalloc(&d_ms, n * 4));
   CHECK(hipMemcpy(d_win, win.data(), n * 8, hipMemcpyHostToDevice));
   CHECK(hipMemcpy(d_acc, acc.data(), n * 8, hipMemcpyHostToDevice));
   CHECK(hipMemcpy(d_ref, ref.data(), n * 4, hipMemcpyHostToDevice));
   hipLaunchKernelGGL(semantics, dim3(n / 256), dim3(256), 0, nullptr, d_win, d_ref, d_acc, d_q, d_mq, d_ms, n);
   CHECK(hipMemcpy(q.data(), d_q, n * 8, hipMemcpyDeviceToHost));
   CHECK(hipMemcpy(mq.data(), d_mq, n * 8, hipMemcpyDeviceToHost));
   CHECK(hipMemcpy(ms.data(), d_ms, n * 4, hipMemcpyDeviceToHost));
   // model: lane j of the result = acc lane j + sum over k of |byte (j + k) of the 64-bit window - byte k of the reference|;
   // masked forms, two hypotheses: bytes are left out where the REFERENCE byte is 0 / where the WINDOW byte is 0
   long bad_q = 0, bad_mq_ref = 0, bad_mq_win = 0, bad_ms_ref = 0, bad_ms_win = 0;
   for (int i = 0; i < n; i++) {
      unsigned long long eq = 0, em_ref = 0, em_win = 0;
      for (int j = 0; j < 4; j++) {
         uint32_t s = 0, sr = 0, sw = 0;
         for (int k = 0; k < 4; k++) {
            const int wb = (int)((win[i] >> (8 * (j + k))) & 0xff), rb = (int)((ref[i] >> (8 * k)) & 0xff);
            const uint32_t d = (uint32_t)abs(wb - rb);
            s += d;
            sr += rb ? d : 0;
            sw += wb ? d : 0;
         }
         const uint32_t a = (uint32_t)((acc[i] >> (16 * j)) & 0xffff);
         eq |= (unsigned long long)((s + a) & 0xffff) << (16 * j);
         em_ref |= (unsigned long long)((sr + a) & 0xffff) << (16 * j);
         em_win |= (unsigned long long)((sw + a) & 0xffff) << (16 * j);
      }
      bad_q += q[i] != eq;
      bad_mq_ref += mq[i] != em_ref;
      bad_mq_win += mq[i] != em_win;
      uint32_t sr = 0, sw = 0;
      for (int k = 0; k < 4; k++) {
         const int wb = (int)((win[i] >> (8 * k)) & 0xff), rb = (int)((ref[i] >> (8 * k)) & 0xff);
         sr += rb ? (uint32_t)abs(wb - rb) : 0;
         sw += wb ? (uint32_t)abs(wb - rb) : 0;
      }
      bad_ms_ref += ms[i] != sr + (uint32_t)acc[i];
      bad_ms_win += ms[i] != sw + (uint32_t)acc[i];
   }
   printf("semantics over %d random cases: v_qsad_pk_u16_u8 = sliding 4-byte SAD of src0[63:0] bytes j..j+3 against src1, + src2 lane j: %ld mismatches\n", n, bad_q);
   printf("   v_mqsad_pk_u16_u8 masked where the REFERENCE (src1) byte is 0: %ld mismatches; masked where the WINDOW (src0) byte is 0: %ld mismatches\n",
          bad_mq_ref, bad_mq_win);
   printf("   v_msad_u8 masked where src1 byte is 0: %ld mismatches; where src0 byte is 0: %ld mismatches\n", bad_ms_ref, bad_ms_win);
   return 0;
}
