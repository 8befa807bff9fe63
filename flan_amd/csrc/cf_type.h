// cf_type.h -- the complex / pair type of the kernels: a native 2-float vector, so that elementwise arithmetic on it
// compiles to gfx950's packed fp32 instructions (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32: two results per issue slot).
#pragma once
namespace flanhip {
typedef float cf __attribute__(( ext_vector_type( 2 ) ));
}
