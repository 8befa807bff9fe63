// cf_type.h -- the complex / pair type of the kernels: a native 2-float vector (an aligned register pair: 8-byte LDS and global accesses
// take it whole).  conversions.hip is compiled WITHOUT packed fp32 (flan_amd/build.py): on gfx950 a v_pk_*_f32 occupies the SIMD for
// two plain instructions' time (profiles/r03_a_issue_model.txt), so element-wise arithmetic on a cf is two plain instructions there.
#pragma once
namespace flanhip {
typedef float cf __attribute__(( ext_vector_type( 2 ) ));
typedef float v4f_t __attribute__(( ext_vector_type( 4 ) ));   // four floats as one register tuple (16-byte LDS / global accesses: per-bin constant tables)
struct d2 { double x, y; };   // a complex number in double (tables evaluated on the host: the unit circle of the direct sums, Bluestein's chirp)
}
