// LDS-DMA helpers shared by the matrix kernels (csrc/conv.hip, csrc/wino4.hip): global / buffer loads that land in LDS
// without passing through VGPRs, issued from inline asm so that hipcc does not count them (with the builtin the
// compiler drains the DMA queue, s_waitcnt vmcnt(0), in front of every ds_read; here the counted waits of the
// kernels are the only ones).
//
// M0 carries the LDS base of a DMA.  It is written WITHOUT save / restore and without a clobber declaration (declaring it
// makes hipcc wrap every DMA in the s_mov pair this avoids: scalar instructions sit in the same in-order stream as the
// wave's MFMAs).  That is safe only while no compiler-generated use of m0 (movrel, sendmsg, GWS, addtid, its own
// LDS-DMA builtins) is live across these statements; tests/test_isa_m0.py disassembles the kernels and checks it.
#pragma once
#include "common.h"

namespace uoc {

// 64 lanes x 16 B land at LDS byte address `lds_dst` (wave-uniform) + 16*lane
__device__ __forceinline__ void glds16(const float *g, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The same through a raw buffer descriptor: address = descriptor base + soff (wave-uniform SGPR) + voff (per lane);
// a lane whose voff lies beyond the descriptor's num_records delivers ZEROS (raw-buffer range check).  Measured on gfx950
// (round 3, csrc/wino4.hip): the check covers voff + soff, so a descriptor must span everything soff can reach.  That is the whole per-lane address arithmetic of an implicit-GEMM chunk: the (tap, cin-slice) offset is
// one scalar, the pixel's offset a loop-invariant VGPR, an out-of-image tap the out-of-range constant.
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr unsigned kOobVoff = 0xFFFFFFF0u;
__device__ __forceinline__ void blds16(v4i srd, unsigned voff, unsigned soff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(srd), "s"(soff), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ v4i make_srd(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  v4i r;
  r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));   // stride 0, no swizzle
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

}  // namespace uoc
