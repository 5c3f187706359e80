// valubench.hip - issue-rate microbenchmark for the integer VALU ops the kernels lean on (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valubench scripts/valubench.hip && /tmp/valubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 64
#define ITER 256

#define KERNEL(name, body)                                                                  \
  __global__ __launch_bounds__ (256) void name (uint32_t *out, uint32_t seed)               \
  {                                                                                         \
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;                \
    uint32_t a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19;                        \
    uint32_t b = seed | 1, c = seed + 77;                                                   \
    for (int i = 0; i < ITER; i++) {                                                        \
      _Pragma ("unroll") for (int r = 0; r < REP / 8; r++) {                                \
        body (a0) body (a1) body (a2) body (a3) body (a4) body (a5) body (a6) body (a7)     \
      }                                                                                     \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
  }

#define OP_ADD(x) asm volatile ("v_add_u32 %0, %0, %1" : "+v" (x) : "v" (b));
#define OP_MUL24(x) asm volatile ("v_mul_u32_u24 %0, %0, %1" : "+v" (x) : "v" (b));
#define OP_MAD24(x) asm volatile ("v_mad_u32_u24 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_MULLO(x) asm volatile ("v_mul_lo_u32 %0, %0, %1" : "+v" (x) : "v" (b));
#define OP_MULHI24(x) asm volatile ("v_mul_hi_i32_i24 %0, %0, %1" : "+v" (x) : "v" (b));
#define OP_ADD3(x) asm volatile ("v_add3_u32 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_PKSHR(x) asm volatile ("v_pk_lshrrev_b16 %0, 1, %0" : "+v" (x));
#define OP_PKMUL(x) asm volatile ("v_pk_mul_lo_u16 %0, %0, %1" : "+v" (x) : "v" (b));
#define OP_PKMAD(x) asm volatile ("v_pk_mad_u16 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_PKADD(x) asm volatile ("v_pk_add_u16 %0, %0, %1" : "+v" (x) : "v" (b));
#define OP_PERM(x) asm volatile ("v_perm_b32 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_AND(x) asm volatile ("v_and_b32 %0, %0, %1" : "+v" (x) : "v" (b));
#define OP_BFE(x) asm volatile ("v_bfe_u32 %0, %0, 8, 8" : "+v" (x));
#define OP_LSHLOR(x) asm volatile ("v_lshl_or_b32 %0, %0, 1, %1" : "+v" (x) : "v" (b));
#define OP_MED3(x) asm volatile ("v_med3_i32 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_CNDMASK(x) asm volatile ("v_cndmask_b32 %0, %0, %1, vcc" : "+v" (x) : "v" (b));
#define OP_SDWA(x) asm volatile ("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v" (x) : "v" (b));
#define OP_DOT4(x) asm volatile ("v_dot4_u32_u8 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_SAD(x) asm volatile ("v_sad_u8 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_FMA(x) asm volatile ("v_fma_f32 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_ASHRPK(x) asm volatile ("v_lshrrev_b32 %0, 3, %0" : "+v" (x));
#define OP_LERP(x) asm volatile ("v_lerp_u8 %0, %0, %1, %2" : "+v" (x) : "v" (b), "v" (c));
#define OP_DOT4I(x) asm volatile ("v_dot4_i32_i8 %0, %1, %2, %0" : "+v" (x) : "v" (b), "v" (c));
#define OP_SATPK(x) asm volatile ("v_sat_pk_u8_i16 %0, %0" : "+v" (x));
#define OP_ASHR16S(x) asm volatile ("v_ashrrev_i16_sdwa %0, %1, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:WORD_0" : "+v" (x) : "v" (b));
#define OP_XOR(x) asm volatile ("v_xor_b32 %0, %0, %1" : "+v" (x) : "v" (b));
#define OP_MOV(x) asm volatile ("v_mov_b32 %0, %1" : "+v" (x) : "v" (b));

KERNEL (k_add, OP_ADD) KERNEL (k_mul24, OP_MUL24) KERNEL (k_mad24, OP_MAD24) KERNEL (k_mullo, OP_MULLO)
KERNEL (k_mulhi24, OP_MULHI24) KERNEL (k_add3, OP_ADD3) KERNEL (k_pkshr, OP_PKSHR) KERNEL (k_pkmul, OP_PKMUL)
KERNEL (k_pkmad, OP_PKMAD) KERNEL (k_pkadd, OP_PKADD) KERNEL (k_perm, OP_PERM) KERNEL (k_and, OP_AND) KERNEL (k_bfe, OP_BFE)
KERNEL (k_lshlor, OP_LSHLOR) KERNEL (k_med3, OP_MED3) KERNEL (k_cndmask, OP_CNDMASK) KERNEL (k_sdwa, OP_SDWA)
KERNEL (k_dot4, OP_DOT4) KERNEL (k_sad, OP_SAD) KERNEL (k_fma, OP_FMA) KERNEL (k_shr, OP_ASHRPK)
KERNEL (k_lerp, OP_LERP) KERNEL (k_dot4i, OP_DOT4I) KERNEL (k_satpk, OP_SATPK) KERNEL (k_ashr16s, OP_ASHR16S) KERNEL (k_xor, OP_XOR) KERNEL (k_mov, OP_MOV)

typedef void (*kern_t) (uint32_t *, uint32_t);

static void run (const char *name, kern_t k, uint32_t *out)
{
  const int blocks = 256 * 8 * 4;
  hipEvent_t e0, e1;
  hipEventCreate (&e0);
  hipEventCreate (&e1);
  hipLaunchKernelGGL (k, dim3 (blocks), dim3 (256), 0, 0, out, 1u);
  hipDeviceSynchronize ();
  hipEventRecord (e0);
  hipLaunchKernelGGL (k, dim3 (blocks), dim3 (256), 0, 0, out, 2u);
  hipEventRecord (e1);
  hipEventSynchronize (e1);
  float ms;
  hipEventElapsedTime (&ms, e0, e1);
  const double ops = (double) blocks * 256 * ITER * REP;
  printf ("%-10s %8.3f ms  %7.2f T lane-ops/s  (%.1f lanes/clk/CU at 2.4 GHz)\n", name, ms, ops / ms / 1e9,
      ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main ()
{
  uint32_t *out;
  hipMalloc (&out, 256 * 8 * 4 * 256 * 4);
#define R(n) run (#n, n, out);
  R (k_add) R (k_mul24) R (k_mad24) R (k_mullo) R (k_mulhi24) R (k_add3) R (k_pkshr) R (k_pkmul) R (k_pkmad) R (k_pkadd)
  R (k_perm) R (k_and) R (k_bfe) R (k_lshlor) R (k_med3) R (k_cndmask) R (k_sdwa) R (k_dot4) R (k_sad) R (k_fma) R (k_shr)
  R (k_lerp) R (k_dot4i) R (k_satpk) R (k_ashr16s) R (k_xor) R (k_mov)
  return 0;
}
