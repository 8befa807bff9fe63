// valu_pairs.hip -- what does a wave64 VALU instruction cost a gfx950 SIMD, as a function of the instructions AROUND it?
// Hand-ordered inline-asm bodies (the compiler cannot reorder inside one asm block), 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_pairs.hip -o /tmp/valu_pairs && /tmp/valu_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REGS "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), \
             "+v"(r8), "+v"(r9), "+v"(r10), "+v"(r11), "+v"(r12), "+v"(r13), "+v"(r14), "+v"(r15)

// every body is 16 VALU instructions, repeated 8 times per loop iteration (128 VALU per taken branch)
#define BODY_DEP1 /* one dependent chain */ \
  "v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n" \
  "v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n" \
  "v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n" \
  "v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n v_fma_f32 %0, %0, %14, %1\n"
#define BODY_ILP2 /* two chains interleaved A B A B */ \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n"
#define BODY_ILP2_AABB /* two chains, A A B B */ \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %2, %2, %14, %15\n"
#define BODY_ILP3 \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %0, %0, %14, %15\n" \
  "v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %0, %0, %14, %15\n"
#define BODY_ILP4 \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n"
#define BODY_ILP8 \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n" \
  "v_fma_f32 %5, %5, %14, %15\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n v_fma_f32 %8, %8, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n" \
  "v_fma_f32 %5, %5, %14, %15\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n v_fma_f32 %8, %8, %14, %15\n"
// VOP2 forms (4-byte encodings)
#define BODY_VOP2_DEP1 \
  "v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" \
  "v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" \
  "v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" \
  "v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
#define BODY_VOP2_ILP4 \
  "v_mul_f32 %0, %0, %15\n v_mul_f32 %2, %2, %15\n v_mul_f32 %3, %3, %15\n v_mul_f32 %4, %4, %15\n" \
  "v_add_f32 %0, %0, %15\n v_add_f32 %2, %2, %15\n v_add_f32 %3, %3, %15\n v_add_f32 %4, %4, %15\n" \
  "v_mul_f32 %0, %0, %15\n v_mul_f32 %2, %2, %15\n v_mul_f32 %3, %3, %15\n v_mul_f32 %4, %4, %15\n" \
  "v_add_f32 %0, %0, %15\n v_add_f32 %2, %2, %15\n v_add_f32 %3, %3, %15\n v_add_f32 %4, %4, %15\n"
// packed
#define BODY_PK_DEP1 \
  "v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n" \
  "v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n" \
  "v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n" \
  "v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %0, %0, %0, %1\n"
#define BODY_PK_ILP4 \
  "v_pk_fma_f32 %0, %0, %0, %4\n v_pk_fma_f32 %1, %1, %1, %4\n v_pk_fma_f32 %2, %2, %2, %4\n v_pk_fma_f32 %3, %3, %3, %4\n" \
  "v_pk_fma_f32 %0, %0, %0, %4\n v_pk_fma_f32 %1, %1, %1, %4\n v_pk_fma_f32 %2, %2, %2, %4\n v_pk_fma_f32 %3, %3, %3, %4\n" \
  "v_pk_fma_f32 %0, %0, %0, %4\n v_pk_fma_f32 %1, %1, %1, %4\n v_pk_fma_f32 %2, %2, %2, %4\n v_pk_fma_f32 %3, %3, %3, %4\n" \
  "v_pk_fma_f32 %0, %0, %0, %4\n v_pk_fma_f32 %1, %1, %1, %4\n v_pk_fma_f32 %2, %2, %2, %4\n v_pk_fma_f32 %3, %3, %3, %4\n"
// a packed op next to an independent plain op
#define BODY_PK_PLAIN_MIX \
  "v_pk_fma_f32 %0, %0, %0, %4\n v_fma_f32 %10, %10, %14, %15\n v_pk_fma_f32 %1, %1, %1, %4\n v_fma_f32 %11, %11, %14, %15\n" \
  "v_pk_fma_f32 %2, %2, %2, %4\n v_fma_f32 %12, %12, %14, %15\n v_pk_fma_f32 %3, %3, %3, %4\n v_fma_f32 %13, %13, %14, %15\n" \
  "v_pk_fma_f32 %0, %0, %0, %4\n v_fma_f32 %10, %10, %14, %15\n v_pk_fma_f32 %1, %1, %1, %4\n v_fma_f32 %11, %11, %14, %15\n" \
  "v_pk_fma_f32 %2, %2, %2, %4\n v_fma_f32 %12, %12, %14, %15\n v_pk_fma_f32 %3, %3, %3, %4\n v_fma_f32 %13, %13, %14, %15\n"
// transcendental: rcp dependent / independent
#define BODY_RCP_ILP4 \
  "v_rcp_f32 %0, %0\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n" \
  "v_rcp_f32 %0, %0\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n" \
  "v_rcp_f32 %0, %0\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n" \
  "v_rcp_f32 %0, %0\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n"
// one rcp per 4 plain ops, all independent (does the transcendental unit run beside the main pipe?)
#define BODY_RCP_MIX \
  "v_rcp_f32 %0, %0\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n" \
  "v_rcp_f32 %5, %5\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n v_fma_f32 %8, %8, %14, %15\n" \
  "v_rcp_f32 %0, %0\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n" \
  "v_rcp_f32 %5, %5\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %14, %15\n v_fma_f32 %8, %8, %14, %15\n"
// f64
#define BODY_F64_ILP4 \
  "v_fma_f64 %0, %0, %0, %4\n v_fma_f64 %1, %1, %1, %4\n v_fma_f64 %2, %2, %2, %4\n v_fma_f64 %3, %3, %3, %4\n" \
  "v_fma_f64 %0, %0, %0, %4\n v_fma_f64 %1, %1, %1, %4\n v_fma_f64 %2, %2, %2, %4\n v_fma_f64 %3, %3, %3, %4\n" \
  "v_fma_f64 %0, %0, %0, %4\n v_fma_f64 %1, %1, %1, %4\n v_fma_f64 %2, %2, %2, %4\n v_fma_f64 %3, %3, %3, %4\n" \
  "v_fma_f64 %0, %0, %0, %4\n v_fma_f64 %1, %1, %1, %4\n v_fma_f64 %2, %2, %2, %4\n v_fma_f64 %3, %3, %3, %4\n"
#define BODY_ADDF64_ILP4 \
  "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n" \
  "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n" \
  "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n" \
  "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
// plain ops with an s_nop 0 between pairs / cndmask / integer
#define BODY_INT_ILP4 \
  "v_add_u32 %0, %0, %15\n v_add_u32 %2, %2, %15\n v_add_u32 %3, %3, %15\n v_add_u32 %4, %4, %15\n" \
  "v_xor_b32 %0, %0, %15\n v_xor_b32 %2, %2, %15\n v_xor_b32 %3, %3, %15\n v_xor_b32 %4, %4, %15\n" \
  "v_add_u32 %0, %0, %15\n v_add_u32 %2, %2, %15\n v_add_u32 %3, %3, %15\n v_add_u32 %4, %4, %15\n" \
  "v_xor_b32 %0, %0, %15\n v_xor_b32 %2, %2, %15\n v_xor_b32 %3, %3, %15\n v_xor_b32 %4, %4, %15\n"
// dependent chain of length 2 then switch: A1 A2 B1 B2 C1 C2 D1 D2 (distance 8 between reuse)
#define BODY_ILP4_PAIRS \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n v_fma_f32 %4, %4, %14, %15\n" \
  "v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %2, %2, %14, %15\n v_fma_f32 %2, %2, %14, %15\n" \
  "v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %3, %3, %14, %15\n v_fma_f32 %4, %4, %14, %15\n v_fma_f32 %4, %4, %14, %15\n"

enum { T_DEP1, T_ILP2, T_ILP2_AABB, T_ILP3, T_ILP4, T_ILP8, T_VOP2_DEP1, T_VOP2_ILP4, T_INT_ILP4, T_ILP4_PAIRS, T_RCP_ILP4, T_RCP_MIX, T_COUNT32 };

template<int T>
__global__ void k32( float * out, int iters )
	{
	float r0 = threadIdx.x * 1e-3f + 1.0f, r1 = 0.5f, r2 = r0 + 1, r3 = r0 + 2, r4 = r0 + 3, r5 = r0 + 4, r6 = r0 + 5, r7 = r0 + 6, r8 = r0 + 7;
	float r9 = 1, r10 = 2, r11 = 3, r12 = 4, r13 = 5, r14 = 6, r15 = 0.25f;
	for( int it = 0; it < iters; ++it )
		{
		if constexpr( T == T_DEP1 ) asm volatile( BODY_DEP1 BODY_DEP1 BODY_DEP1 BODY_DEP1 BODY_DEP1 BODY_DEP1 BODY_DEP1 BODY_DEP1 : REGS );
		if constexpr( T == T_ILP2 ) asm volatile( BODY_ILP2 BODY_ILP2 BODY_ILP2 BODY_ILP2 BODY_ILP2 BODY_ILP2 BODY_ILP2 BODY_ILP2 : REGS );
		if constexpr( T == T_ILP2_AABB ) asm volatile( BODY_ILP2_AABB BODY_ILP2_AABB BODY_ILP2_AABB BODY_ILP2_AABB BODY_ILP2_AABB BODY_ILP2_AABB BODY_ILP2_AABB BODY_ILP2_AABB : REGS );
		if constexpr( T == T_ILP3 ) asm volatile( BODY_ILP3 BODY_ILP3 BODY_ILP3 BODY_ILP3 BODY_ILP3 BODY_ILP3 BODY_ILP3 BODY_ILP3 : REGS );
		if constexpr( T == T_ILP4 ) asm volatile( BODY_ILP4 BODY_ILP4 BODY_ILP4 BODY_ILP4 BODY_ILP4 BODY_ILP4 BODY_ILP4 BODY_ILP4 : REGS );
		if constexpr( T == T_ILP8 ) asm volatile( BODY_ILP8 BODY_ILP8 BODY_ILP8 BODY_ILP8 BODY_ILP8 BODY_ILP8 BODY_ILP8 BODY_ILP8 : REGS );
		if constexpr( T == T_VOP2_DEP1 ) asm volatile( BODY_VOP2_DEP1 BODY_VOP2_DEP1 BODY_VOP2_DEP1 BODY_VOP2_DEP1 BODY_VOP2_DEP1 BODY_VOP2_DEP1 BODY_VOP2_DEP1 BODY_VOP2_DEP1 : REGS );
		if constexpr( T == T_VOP2_ILP4 ) asm volatile( BODY_VOP2_ILP4 BODY_VOP2_ILP4 BODY_VOP2_ILP4 BODY_VOP2_ILP4 BODY_VOP2_ILP4 BODY_VOP2_ILP4 BODY_VOP2_ILP4 BODY_VOP2_ILP4 : REGS );
		if constexpr( T == T_INT_ILP4 ) asm volatile( BODY_INT_ILP4 BODY_INT_ILP4 BODY_INT_ILP4 BODY_INT_ILP4 BODY_INT_ILP4 BODY_INT_ILP4 BODY_INT_ILP4 BODY_INT_ILP4 : REGS );
		if constexpr( T == T_ILP4_PAIRS ) asm volatile( BODY_ILP4_PAIRS BODY_ILP4_PAIRS BODY_ILP4_PAIRS BODY_ILP4_PAIRS BODY_ILP4_PAIRS BODY_ILP4_PAIRS BODY_ILP4_PAIRS BODY_ILP4_PAIRS : REGS );
		if constexpr( T == T_RCP_ILP4 ) asm volatile( BODY_RCP_ILP4 BODY_RCP_ILP4 BODY_RCP_ILP4 BODY_RCP_ILP4 BODY_RCP_ILP4 BODY_RCP_ILP4 BODY_RCP_ILP4 BODY_RCP_ILP4 : REGS );
		if constexpr( T == T_RCP_MIX ) asm volatile( BODY_RCP_MIX BODY_RCP_MIX BODY_RCP_MIX BODY_RCP_MIX BODY_RCP_MIX BODY_RCP_MIX BODY_RCP_MIX BODY_RCP_MIX : REGS );
		}
	out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9 + r10 + r11 + r12 + r13 + r14 + r15;
	}

typedef float v2f __attribute__(( ext_vector_type( 2 ) ));
enum { P_DEP1, P_ILP4, P_MIX, P_COUNT };
template<int T>
__global__ void kpk( float * out, int iters )
	{
	v2f r0 = { threadIdx.x * 1e-3f + 1.0f, 2.0f }, r1 = r0 + 1.0f, r2 = r0 + 2.0f, r3 = r0 + 3.0f, r4 = { 0.25f, 0.5f };
	float r10 = 2, r11 = 3, r12 = 4, r13 = 5, r15 = 0.25f;
	for( int it = 0; it < iters; ++it )
		{
		if constexpr( T == P_DEP1 ) asm volatile( BODY_PK_DEP1 BODY_PK_DEP1 BODY_PK_DEP1 BODY_PK_DEP1 BODY_PK_DEP1 BODY_PK_DEP1 BODY_PK_DEP1 BODY_PK_DEP1 : "+v"( r0 ), "+v"( r1 ), "+v"( r2 ), "+v"( r3 ), "+v"( r4 ) );
		if constexpr( T == P_ILP4 ) asm volatile( BODY_PK_ILP4 BODY_PK_ILP4 BODY_PK_ILP4 BODY_PK_ILP4 BODY_PK_ILP4 BODY_PK_ILP4 BODY_PK_ILP4 BODY_PK_ILP4 : "+v"( r0 ), "+v"( r1 ), "+v"( r2 ), "+v"( r3 ), "+v"( r4 ) );
		if constexpr( T == P_MIX ) asm volatile( BODY_PK_PLAIN_MIX BODY_PK_PLAIN_MIX BODY_PK_PLAIN_MIX BODY_PK_PLAIN_MIX BODY_PK_PLAIN_MIX BODY_PK_PLAIN_MIX BODY_PK_PLAIN_MIX BODY_PK_PLAIN_MIX : "+v"( r0 ), "+v"( r1 ), "+v"( r2 ), "+v"( r3 ), "+v"( r4 ), "+v"( r10 ), "+v"( r10 ), "+v"( r10 ), "+v"( r10 ), "+v"( r10 ),
			"+v"( r10 ), "+v"( r11 ), "+v"( r12 ), "+v"( r13 ), "+v"( r15 ), "+v"( r15 ) );
		}
	out[blockIdx.x * blockDim.x + threadIdx.x] = r0.x + r1.x + r2.y + r3.x + r4.x + r10 + r11 + r12 + r13 + r15;
	}

enum { D_FMA4, D_ADD4, D_COUNT };
template<int T>
__global__ void kf64( float * out, int iters )
	{
	double r0 = threadIdx.x * 1e-3 + 1.0, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = 0.25;
	for( int it = 0; it < iters; ++it )
		{
		if constexpr( T == D_FMA4 ) asm volatile( BODY_F64_ILP4 BODY_F64_ILP4 BODY_F64_ILP4 BODY_F64_ILP4 BODY_F64_ILP4 BODY_F64_ILP4 BODY_F64_ILP4 BODY_F64_ILP4 : "+v"( r0 ), "+v"( r1 ), "+v"( r2 ), "+v"( r3 ), "+v"( r4 ) );
		if constexpr( T == D_ADD4 ) asm volatile( BODY_ADDF64_ILP4 BODY_ADDF64_ILP4 BODY_ADDF64_ILP4 BODY_ADDF64_ILP4 BODY_ADDF64_ILP4 BODY_ADDF64_ILP4 BODY_ADDF64_ILP4 BODY_ADDF64_ILP4 : "+v"( r0 ), "+v"( r1 ), "+v"( r2 ), "+v"( r3 ), "+v"( r4 ) );
		}
	out[blockIdx.x * blockDim.x + threadIdx.x] = float( r0 + r1 + r2 + r3 + r4 );
	}

static float * g_out;
template<class K> static double run( K kern, int waves_per_simd )
	{
	// clocks: an idle MI355X takes tens of milliseconds of load to settle, so every configuration is run back to back for ~40 ms
	// before the timed launches, and the timed region is 8 launches of ~4 ms
	const int threads = 64 * 4 * waves_per_simd, blocks = 256, iters = 8000;
	hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
	for( int i = 0; i < 10; ++i ) kern<<<blocks, threads>>>( g_out, iters );
	hipEventRecord( e0 );
	for( int i = 0; i < 8; ++i ) kern<<<blocks, threads>>>( g_out, iters );
	hipEventRecord( e1 ); hipEventSynchronize( e1 );
	float ms; hipEventElapsedTime( &ms, e0, e1 );
	hipEventDestroy( e0 ); hipEventDestroy( e1 );
	return double( ms ) * 1e6 / ( 8.0 * double( iters ) * 128 * waves_per_simd );   // ns per wave-instruction per SIMD
	}

template<int T> static void row32( const char * name )
	{
	printf( "%-34s", name );
	for( int w : { 1, 2, 3, 4 } ) printf( "  %dw %5.2f ns", w, run( k32<T>, w ) );
	printf( "\n" );
	}
template<int T> static void rowpk( const char * name )
	{
	printf( "%-34s", name );
	for( int w : { 1, 2, 3, 4 } ) printf( "  %dw %5.2f ns", w, run( kpk<T>, w ) );
	printf( "\n" );
	}
template<int T> static void rowf64( const char * name )
	{
	printf( "%-34s", name );
	for( int w : { 1, 2, 3, 4 } ) printf( "  %dw %5.2f ns", w, run( kf64<T>, w ) );
	printf( "\n" );
	}

int main()
	{
	hipMalloc( &g_out, sizeof( float ) * 256 * 1024 );
	for( int i = 0; i < 80; ++i ) k32<T_ILP4><<<256, 1024>>>( g_out, 4000 );      // ~150 ms: let the clocks settle
	hipDeviceSynchronize();
	printf( "ns per wave-instruction per SIMD (1 / 2 / 3 / 4 waves per SIMD); 0.95 ns = 2 cycles at 2.1 GHz\n" );
	row32<T_DEP1>( "fma dep chain (ILP 1)" );
	row32<T_ILP2>( "fma ILP 2 (A B A B)" );
	row32<T_ILP2_AABB>( "fma ILP 2 (A A B B)" );
	row32<T_ILP3>( "fma ILP 3" );
	row32<T_ILP4>( "fma ILP 4" );
	row32<T_ILP8>( "fma ILP 8" );
	row32<T_ILP4_PAIRS>( "fma 4 chains, A A B B C C D D" );
	row32<T_VOP2_DEP1>( "mul/add VOP2 dep chain" );
	row32<T_VOP2_ILP4>( "mul/add VOP2 ILP 4" );
	row32<T_INT_ILP4>( "add_u32/xor ILP 4" );
	row32<T_RCP_ILP4>( "rcp ILP 4" );
	row32<T_RCP_MIX>( "1 rcp + 3 fma, independent" );
	rowpk<P_DEP1>( "pk_fma dep chain" );
	rowpk<P_ILP4>( "pk_fma ILP 4" );
	rowpk<P_MIX>( "pk_fma + fma alternating, indep" );
	rowf64<D_FMA4>( "fma_f64 ILP 4" );
	rowf64<D_ADD4>( "add_f64 ILP 4" );
	return 0;
	}
