#!/usr/bin/env python3
"""Generates tools/ubench/issue_model.hip: what does each kind of instruction of the dft 2048 kernels' frame loop cost a gfx950 SIMD?

Every test is a 64-instruction inline-asm body on explicitly named registers (the compiler cannot reorder or re-allocate it), looped;
run with 1, 2, 3 and 4 wavefronts per SIMD on all 256 CUs; the wavefront reads s_memtime / s_memrealtime around the loop, so the
result is in shader CYCLES per wave-instruction per SIMD (and the clock the chip held is printed beside it).

    python tools/ubench/gen_issue_model.py && hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_model.hip -o tools/ubench/issue_model
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = []


def rep(lines, n=64):
    out = []
    while len(out) < n:
        out.extend(lines)
    return out[:n]


def add(name, body, per=None):
    """per: instructions counted per body (default: all 64)"""
    TESTS.append((name, body, per if per is not None else len(body)))


# registers: v8..v47 data (initialised to small floats), v1..v7 constants, s[4:5] a mask
def chains(fmt, n, base=8, step=1):
    return rep([fmt.format(r=base + step * i, r1=base + step * i + 1) for i in range(n)])


add("v_fma_f32      8 chains, banks 0/1/2", chains("v_fma_f32 v{r}, v{r}, v1, v2", 8))
add("v_fma_f32      8 chains, all operands bank 0 (r % 4 == 0)", chains("v_fma_f32 v{r}, v{r}, v4, v48", 8, 8, 4))
add("v_fma_f32      8 chains, src0 = src1 bank", rep(["v_fma_f32 v{r}, v{r}, v{s}, v2".format(r=8 + 4 * i, s=4) for i in range(8)]))
add("v_mul_f32_e32  8 chains", chains("v_mul_f32_e32 v{r}, v{r}, v1", 8))
add("v_add_f32_e32  8 chains", chains("v_add_f32_e32 v{r}, v{r}, v3", 8))
add("v_fmac_f32_e32 8 chains", chains("v_fmac_f32_e32 v{r}, v1, v2", 8))
add("v_fmaak_f32    8 chains (literal)", chains("v_fmaak_f32 v{r}, v{r}, v1, 0x3e000000", 8))
add("v_fma_f32      8 chains, inline constant + sgpr", chains("v_fma_f32 v{r}, v{r}, 0.5, s6", 8))
add("v_fma_f32      1 chain (dependent)", rep(["v_fma_f32 v8, v8, v1, v2"]))
add("v_fmac_f32_e32 1 chain (dependent)", rep(["v_fmac_f32_e32 v8, v1, v2"]))
add("v_mul_f32_e32  1 chain (dependent)", rep(["v_mul_f32_e32 v8, v8, v1"]))
add("v_add_f32_e32  1 chain (dependent)", rep(["v_add_f32_e32 v8, v8, v3"]))
add("v_fma_f32      2 chains A B A B", chains("v_fma_f32 v{r}, v{r}, v1, v2", 2))
add("v_fma_f32      2 chains A A B B", rep(["v_fma_f32 v8, v8, v1, v2", "v_fma_f32 v8, v8, v1, v2", "v_fma_f32 v9, v9, v1, v2", "v_fma_f32 v9, v9, v1, v2"]))
add("v_fma_f32      3 chains", chains("v_fma_f32 v{r}, v{r}, v1, v2", 3))
add("v_fma_f32      4 chains", chains("v_fma_f32 v{r}, v{r}, v1, v2", 4))
add("mul -> add dependent pairs, 4 chains", rep(sum([["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + i) for i in range(4)], ["v_add_f32_e32 v{r}, v{r}, v3".format(r=8 + i) for i in range(4)]], [])))
add("v_cndmask_b32_e32 (vcc) 8 chains", chains("v_cndmask_b32_e32 v{r}, v{r}, v1, vcc", 8))
add("v_cndmask_b32_e64 (s[4:5]) 8 chains", chains("v_cndmask_b32_e64 v{r}, v{r}, v1, s[4:5]", 8))
add("v_cmp_gt_f32 vcc + dependent v_cndmask, 8 regs", rep(sum([["v_cmp_gt_f32_e32 vcc, v{r}, v1".format(r=8 + i), "v_cndmask_b32_e32 v{r}, v{r}, v2, vcc".format(r=16 + i)] for i in range(8)], [])))
add("v_cmp_gt_f32_e64 s[8:9] .. 4 apart + v_cndmask_e64", rep(sum([["v_cmp_gt_f32_e64 s[{s}:{s1}], v{r}, v1".format(s=8 + 2 * i, s1=9 + 2 * i, r=8 + i) for i in range(4)], ["v_cndmask_b32_e64 v{r}, v{r}, v2, s[{s}:{s1}]".format(s=8 + 2 * i, s1=9 + 2 * i, r=16 + i) for i in range(4)]], [])))
add("v_max_f32_e64 |a|,|b| 8 chains", chains("v_max_f32_e64 v{r}, |v{r}|, |v1|", 8))
add("v_bfi_b32 8 chains", chains("v_bfi_b32 v{r}, v5, v{r}, v1", 8))
add("v_trunc_f32 8 chains", chains("v_trunc_f32_e32 v{r}, v{r}", 8))
add("v_min3_f32 8 chains", chains("v_min3_f32 v{r}, v{r}, v1, v2", 8))
add("v_ldexp_f32 8 chains", chains("v_ldexp_f32 v{r}, v{r}, v7", 8))
add("v_mov_b32 8 regs", chains("v_mov_b32_e32 v{r}, v1", 8))
add("v_mov_b64 4 pairs", chains("v_mov_b64_e32 v[{r}:{r1}], v[2:3]", 4, 8, 2))
add("v_lshl_add_u64 4 pairs", chains("v_lshl_add_u64 v[{r}:{r1}], v[{r}:{r1}], 0, v[2:3]", 4, 8, 2))
add("v_mul_lo_u32 8 chains", chains("v_mul_lo_u32 v{r}, v{r}, v7", 8))
add("v_add_u32 8 chains", chains("v_add_u32_e32 v{r}, v{r}, v7", 8))
add("v_rcp_f32 8 chains", chains("v_rcp_f32_e32 v{r}, v{r}", 8))
add("v_sqrt_f32 8 chains", chains("v_sqrt_f32_e32 v{r}, v{r}", 8))
add("1 v_rcp_f32 : 7 v_fma_f32, independent", rep(["v_rcp_f32_e32 v16, v17"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]))
add("1 v_rcp_f32 : 15 v_fma_f32, independent", rep(["v_rcp_f32_e32 v16, v17"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + (i % 8)) for i in range(15)]))
add("v_cvt_f64_f32 4 pairs", chains("v_cvt_f64_f32_e32 v[{r}:{r1}], v1", 4, 8, 2))
add("v_add_f64 4 chains", chains("v_add_f64 v[{r}:{r1}], v[{r}:{r1}], v[2:3]", 4, 8, 2))
add("v_fma_f64 4 chains", chains("v_fma_f64 v[{r}:{r1}], v[{r}:{r1}], v[2:3], v[2:3]", 4, 8, 2))
add("1 v_add_f64 : 7 v_fma_f32, independent", rep(["v_add_f64 v[16:17], v[16:17], v[2:3]"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]))
add("v_div_scale_f32 8 regs", chains("v_div_scale_f32 v{r}, vcc, v{r}, v1, v{r}", 8))
add("v_div_fixup_f32 8 chains", chains("v_div_fixup_f32 v{r}, v{r}, v1, v2", 8))
add("v_pk_fma_f32 4 chains", chains("v_pk_fma_f32 v[{r}:{r1}], v[{r}:{r1}], v[2:3], v[2:3]", 4, 8, 2))
add("v_readlane_b32 + 7 fma", rep(["v_readlane_b32 s10, v16, 3"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]))
add("1 s_add_u32 : 3 v_fma_f32 (counted: the 48 fma)", rep(["s_add_u32 s11, s11, 1"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(3)]), 48)
add("1 s_nop 0 : 3 v_fma_f32 (counted: the 48 fma)", rep(["s_nop 0"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(3)]), 48)

add("v_fma_f32 with neg modifier, 8 chains", chains("v_fma_f32 v{r}, v{r}, v1, -v2", 8))
add("v_mul_f32_e64 with neg modifier, 8 chains", chains("v_mul_f32_e64 v{r}, v{r}, -v1", 8))
add("v_sub_f32_e32 8 chains", chains("v_sub_f32_e32 v{r}, v{r}, v3", 8))
add("v_max_f32_e32 8 chains", chains("v_max_f32_e32 v{r}, v{r}, v1", 8))
add("v_and_b32_e32 8 chains", chains("v_and_b32_e32 v{r}, v{r}, v5", 8))
add("v_xor_b32_e32 8 chains", chains("v_xor_b32_e32 v{r}, v{r}, v7", 8))
add("v_ashrrev_i32_e32 8 chains", chains("v_ashrrev_i32_e32 v{r}, 31, v{r}", 8))
add("v_rndne_f32 8 chains", chains("v_rndne_f32_e32 v{r}, v{r}", 8))
add("v_floor_f32 8 chains", chains("v_floor_f32_e32 v{r}, v{r}", 8))
add("v_cvt_i32_f32 8 chains", chains("v_cvt_i32_f32_e32 v{r}, v{r}", 8))
add("v_med3_f32 8 chains", chains("v_med3_f32 v{r}, v{r}, v1, v2", 8))
add("v_mul_f32_e32 with sgpr src0, 8 chains", chains("v_mul_f32_e32 v{r}, s6, v{r}", 8))
add("v_mul_f32_e32 with inline constant, 8 chains", chains("v_mul_f32_e32 v{r}, 0.5, v{r}", 8))
add("v_add_f32_e64 with |abs|, 8 chains", chains("v_add_f32_e64 v{r}, |v{r}|, v3", 8))
add("v_fma_f32 srcs in banks 1,2,3 dst bank 0", rep(["v_fma_f32 v{d}, v{a}, v{b}, v{c}".format(d=8 + 4 * i, a=9 + 4 * i, b=10 + 4 * i, c=11 + 4 * i) for i in range(8)]))
add("v_fma_f32 src0/src1 same bank (r, r+4), 8 chains", rep(["v_fma_f32 v{d}, v{a}, v{b}, v2".format(d=8 + i, a=8 + i, b=16 + 4 * (i // 4) + (i % 4)) for i in range(8)]))
add("1 v_bfi_b32 : 7 v_mul_f32_e32", rep(["v_bfi_b32 v16, v5, v17, v1"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + i) for i in range(7)]))
add("1 v_bfi_b32 : 3 v_mul_f32_e32", rep(["v_bfi_b32 v16, v5, v17, v1"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + i) for i in range(3)]))
add("1 v_rcp_f32 : 7 v_mul_f32_e32", rep(["v_rcp_f32_e32 v16, v17"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + i) for i in range(7)]))
add("1 v_rcp_f32 : 15 v_mul_f32_e32", rep(["v_rcp_f32_e32 v16, v17"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]))
add("1 v_rcp_f32 : 31 v_mul_f32_e32", rep(["v_rcp_f32_e32 v16, v17"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(31)]))
add("cmp_e32 + cndmask_e32 : 6 v_mul_f32_e32", rep(["v_cmp_gt_f32_e32 vcc, v17, v1"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + i) for i in range(3)] + ["v_cndmask_b32_e32 v16, v18, v2, vcc"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=11 + i) for i in range(3)]))
add("1 ds_read_b64 : 15 v_mul_f32_e32 (counted: the 60 mul)", rep(["ds_read_b64 v[32:33], v40 offset:512"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]), 60)
add("1 ds_write_b64 : 15 v_mul_f32_e32 (counted: the 60 mul)", rep(["ds_write_b64 v40, v[34:35] offset:1024"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]), 60)
add("1 ds_read2_b64 : 15 v_mul_f32_e32 (counted: the 60 mul)", rep(["ds_read2_b64 v[32:35], v40 offset0:64 offset1:128"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]), 60)
add("1 ds_write2_b64 : 15 v_mul_f32_e32 (counted: the 60 mul)", rep(["ds_write2_b64 v40, v[32:33], v[34:35] offset0:64 offset1:128"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]), 60)
add("1 ds_read_b128 : 15 v_mul_f32_e32 (counted: the 60 mul)", rep(["ds_read_b128 v[32:35], v41 offset:2048"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]), 60)
add("1 ds_write_b128 : 15 v_mul_f32_e32 (counted: the 60 mul)", rep(["ds_write_b128 v41, v[32:35] offset:4096"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]), 60)
add("1 ds_write_b32 : 15 v_mul_f32_e32 (counted: the 60 mul)", rep(["ds_write_b32 v42, v32 offset:4096"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]), 60)
add("1 ds_read_b32 : 15 v_mul_f32_e32 (counted: the 60 mul)", rep(["ds_read_b32 v32, v42 offset:4096"] + ["v_mul_f32_e32 v{r}, v{r}, v1".format(r=8 + (i % 8)) for i in range(15)]), 60)
# LDS: v40 = lane * 8 (conflict-free b64), v41 = lane * 16
add("ds_read_b64 only (16 in flight, then wait)", rep(["ds_read_b64 v[{r}:{r1}], v40 offset:{o}".format(r=8 + 2 * i, r1=9 + 2 * i, o=512 * i) for i in range(15)] + ["s_waitcnt lgkmcnt(0)"], 64), 60)
add("ds_write_b64 only (16, then wait)", rep(["ds_write_b64 v40, v[{r}:{r1}] offset:{o}".format(r=8 + 2 * i, r1=9 + 2 * i, o=512 * i) for i in range(15)] + ["s_waitcnt lgkmcnt(0)"], 64), 60)
add("1 ds_read_b64 : 7 v_fma_f32 (counted: the 56 fma)", rep(["ds_read_b64 v[32:33], v40 offset:512"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]), 56)
add("1 ds_write_b64 : 7 v_fma_f32 (counted: the 56 fma)", rep(["ds_write_b64 v40, v[34:35] offset:1024"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]), 56)
add("1 ds_read_b64 : 3 v_fma_f32 (counted: the 48 fma)", rep(["ds_read_b64 v[32:33], v40 offset:512"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(3)]), 48)
add("1 ds_write_b64 : 3 v_fma_f32 (counted: the 48 fma)", rep(["ds_write_b64 v40, v[34:35] offset:1024"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(3)]), 48)
add("1 ds_read2_b64 : 7 v_fma_f32 (counted: the 56 fma)", rep(["ds_read2_b64 v[32:35], v40 offset0:64 offset1:128"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]), 56)
add("1 ds_write2_b64 : 7 v_fma_f32 (counted: the 56 fma)", rep(["ds_write2_b64 v40, v[32:33], v[34:35] offset0:64 offset1:128"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]), 56)
add("1 ds_read_b128 : 7 v_fma_f32 (counted: the 56 fma)", rep(["ds_read_b128 v[32:35], v41 offset:2048"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]), 56)
add("1 ds_bpermute_b32 : 7 v_fma_f32 (counted: the 56 fma)", rep(["ds_bpermute_b32 v32, v42, v33"] + ["v_fma_f32 v{r}, v{r}, v1, v2".format(r=8 + i) for i in range(7)]), 56)
# the frame loop's mix in miniature: 16 LDS, 2 trans, 2 f64, rest plain fp32 in 8 chains
mix = []
for i in range(64):
    if i % 16 == 5:
        mix.append("ds_read_b64 v[32:33], v40 offset:512")
    elif i % 16 == 11:
        mix.append("ds_write_b64 v40, v[34:35] offset:1024")
    elif i == 20:
        mix.append("v_rcp_f32_e32 v16, v17")
    elif i == 40:
        mix.append("v_sqrt_f32_e32 v18, v19")
    elif i == 30:
        mix.append("v_add_f64 v[20:21], v[20:21], v[2:3]")
    elif i == 50:
        mix.append("v_cvt_f64_f32_e32 v[22:23], v1")
    else:
        mix.append(["v_fma_f32 v{r}, v{r}, v1, v2", "v_mul_f32_e32 v{r}, v{r}, v1", "v_add_f32_e32 v{r}, v{r}, v3", "v_fmac_f32_e32 v{r}, v1, v2"][i % 4].format(r=8 + (i % 8)))
add("frame-loop mix: 8 LDS, 2 trans, 2 f64, 52 plain fp32 (all 64 counted)", mix)

CLOBBER = ", ".join('"v%d"' % i for i in range(1, 52)) + ', "s4", "s5", "s6", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "vcc", "memory"'
INIT = ["v_mov_b32 v1, 0x3f7fff00", "v_mov_b32 v2, 0x33000000", "v_mov_b32 v3, 0x2f000000", "v_mov_b32 v4, 0x3f7ffe00", "v_mov_b32 v48, 0x33000000",
        "v_mov_b32 v5, 0x7fffffff", "v_mov_b32 v6, 0x3f000000", "v_mov_b32 v7, 0", "s_mov_b64 s[4:5], 0x5555", "s_mov_b32 s6, 0x33000000", "s_mov_b32 s11, 0"]
INIT += ["v_cvt_f32_u32 v%d, v0" % r for r in range(8, 48)]
INIT += ["v_mov_b32 v50, 0x3a000000"] + ["v_fma_f32 v%d, v%d, v50, 1.0" % (r, r) for r in range(8, 48)]
INIT += ["v_and_b32 v40, 63, v0", "v_lshlrev_b32 v41, 4, v40", "v_lshlrev_b32 v42, 2, v40", "v_lshlrev_b32 v40, 3, v40", "v_mov_b32 v3, 0", "v_mov_b32 v2, 0x33000000", "v_mov_b32 v3, 0x2f000000"]


def main():
    out = ["// GENERATED by tools/ubench/gen_issue_model.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <cstdio>", "#include <vector>", "#include <algorithm>", ""]
    out.append("#define CLOBBER " + CLOBBER)
    for i, (name, body, per) in enumerate(TESTS):
        out.append("__global__ void k%d( unsigned long long * stamps, int iters )" % i)
        out.append("\t{")
        out.append("\textern __shared__ unsigned char smem[];")
        out.append("\tasm volatile( \"%s\" ::: CLOBBER );" % "\\n\\t".join(INIT))
        out.append("\tunsigned long long t0, t1, r0, r1;")
        out.append("\tasm volatile( \"s_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"( t0 ), \"=s\"( r0 ) :: \"memory\" );")
        out.append("\tfor( int it = 0; it < iters; ++it )")
        out.append("\t\tasm volatile( \"%s\" ::: CLOBBER );" % "\\n\\t".join(body))
        out.append("\tasm volatile( \"s_waitcnt lgkmcnt(0)\\n\\ts_memtime %0\\n\\ts_memrealtime %1\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"( t1 ), \"=s\"( r1 ) :: \"memory\" );")
        out.append("\tif( ( threadIdx.x & 63 ) == 0 ) { const size_t w = ( size_t( blockIdx.x ) * ( blockDim.x >> 6 ) + ( threadIdx.x >> 6 ) ) * 2; stamps[w] = t1 - t0; stamps[w + 1] = r1 - r0; }")
        out.append("\tif( iters < 0 ) smem[threadIdx.x] = 0;")
        out.append("\t}")
        out.append("")
    out.append("struct Test { const char * name; void ( *kern )( unsigned long long *, int ); int per; };")
    out.append("static const Test tests[] = {")
    for i, (name, body, per) in enumerate(TESTS):
        out.append("\t{ \"%s\", k%d, %d }," % (name, i, per))
    out.append("};")
    out.append(r"""
int main( int argc, char ** argv )
	{
	const int iters = 6000, blocks = 256;
	unsigned long long * d; hipMalloc( &d, sizeof( unsigned long long ) * 2 * blocks * 16 );
	std::vector<unsigned long long> h( 2 * blocks * 16 );
	for( int i = 0; i < 60; ++i ) hipLaunchKernelGGL( tests[0].kern, dim3( blocks ), dim3( 1024 ), 32768, 0, d, iters );   // ~100 ms: clocks settle
	hipDeviceSynchronize();
	printf( "cycles per counted wave-instruction per SIMD = ( life of the block's LAST wavefront ) / ( instructions per wavefront x wavefronts per SIMD ); [life of the first wavefront done / last]; (clock held, GHz); 1 / 2 / 3 / 4 wavefronts per SIMD, 256 blocks, one per CU\n" );
	for( const Test & t : tests )
		{
		printf( "%-72s", t.name );
		for( int w = 1; w <= 4; ++w )
			{
			const int threads = 256 * w;
			for( int i = 0; i < 3; ++i ) hipLaunchKernelGGL( t.kern, dim3( blocks ), dim3( threads ), 32768, 0, d, iters );
			hipDeviceSynchronize();
			hipMemcpy( h.data(), d, sizeof( unsigned long long ) * 2 * blocks * 4 * w, hipMemcpyDeviceToHost );
			// a block's time = its LAST wavefront's (the arbiter favours older wavefronts: the early finishers' lives say nothing about throughput)
			std::vector<double> cyc, ghz, cmin;
			for( int b = 0; b < blocks; ++b )
				{
				double mx = 0, mn = 1e30, g = 0;
				for( int i = b * 4 * w; i < ( b + 1 ) * 4 * w; ++i ) { mx = std::max( mx, double( h[2 * i] ) ); mn = std::min( mn, double( h[2 * i] ) ); g = std::max( g, double( h[2 * i] ) / double( h[2 * i + 1] ) * 0.1 ); }
				cyc.push_back( mx ); cmin.push_back( mn ); ghz.push_back( g );
				}
			std::sort( cyc.begin(), cyc.end() ); std::sort( ghz.begin(), ghz.end() ); std::sort( cmin.begin(), cmin.end() );
			const double c = cyc[cyc.size() / 2] / ( double( iters ) * t.per * w );
			printf( "  %dw %5.2f [first done %4.2f] (%.2f)", w, c, cmin[cmin.size() / 2] / cyc[cyc.size() / 2], ghz[ghz.size() / 2] );
			}
		printf( "\n" );
		fflush( stdout );
		}
	return 0;
	}
""")
    with open(os.path.join(HERE, "issue_model.hip"), "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.join(HERE, "issue_model.hip"), len(TESTS), "tests")


if __name__ == "__main__":
    main()
