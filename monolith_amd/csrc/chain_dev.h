// The hot-key chain: ONE wave adds the rows of one key in input order (the reference sums a key's
// gradients sequentially, embedding_hash_table_tf_bridge.cc:241-266, so a key with 30 000
// occurrences in a batch is a 30 000-long dependent fp32 add chain whatever runs it).
//
// A lone wave issues one instruction every ~4.5 cycles whether it depends on the previous one or
// not (scripts/exp_chain.hip: 4.66 cycles per dependent v_add_f32, 8.65 per pair of independent
// ones), so the chain costs its INSTRUCTION COUNT. The rows therefore arrive "packed"
// (k_long_pack): groups of four rows stored column-major, cell (g, e) = float4 of rows 4g .. 4g+3
// of column e at cell index g * D + e. The lane that owns column e then needs one ds_read_b128 per
// four rows and one v_add_f32 per row, and one LDS-DMA wave-instruction (64 lanes x 16 bytes,
// global -> LDS with no register in between) brings 1 KiB = 1024 / (4 D) rows:
// 1.6 instructions per row at D = 16.
//
// The wave owns a ring of R 1-KiB slots in LDS and fills it itself: no second wave, no barrier, no
// flag — the order is kept by the wave's own counted vmcnt. Loop body = two halves of eight groups
// (32 rows); while the adds of one half run from registers the reads of the next are in flight.
// Registers v64 .. v127 belong to the loop (the kernel is built for 128 VGPRs per lane).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mono {

typedef int chain_v4i __attribute__((ext_vector_type(4)));
// Buffer descriptor of the packed scratch (raw buffer: a byte offset at or past `bytes` is range
// checked by the hardware and the load dropped, so a wrong offset cannot fault). Wave-uniform.
__device__ __forceinline__ chain_v4i chain_rsrc(const void* base, uint32_t bytes) {
  const uint64_t p = (uint64_t)(uintptr_t)base;
  chain_v4i rs;
  rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)p);
  rs.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(p >> 32)) & 0xFFFF;
  rs.z = __builtin_amdgcn_readfirstlane((int)bytes);
  rs.w = 0x00020000;
  return rs;
}
// one LDS-DMA: 64 lanes x 16 bytes from rs + voff (per lane) to LDS lds_dst + lane * 16
__device__ __forceinline__ void chain_dma16(chain_v4i rs, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rs), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void chain_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#define MONO_CH_RD(reg, addr, o) "ds_read_b128 v[" #reg ":" #reg "+3], " addr " offset:%[" #o "]\n\t"
#define MONO_CH_RDA(addr)                                                                     \
  MONO_CH_RD(64, addr, o0) MONO_CH_RD(68, addr, o1) MONO_CH_RD(72, addr, o2)                  \
  MONO_CH_RD(76, addr, o3) MONO_CH_RD(80, addr, o4) MONO_CH_RD(84, addr, o5)                  \
  MONO_CH_RD(88, addr, o6) MONO_CH_RD(92, addr, o7)
#define MONO_CH_RDB(addr)                                                                     \
  MONO_CH_RD(96, addr, o8) MONO_CH_RD(100, addr, o9) MONO_CH_RD(104, addr, o10)               \
  MONO_CH_RD(108, addr, o11) MONO_CH_RD(112, addr, o12) MONO_CH_RD(116, addr, o13)            \
  MONO_CH_RD(120, addr, o14) MONO_CH_RD(124, addr, o15)
#define MONO_CH_AD(r) "v_add_f32 %[acc], %[acc], v" #r "\n\t"
#define MONO_CH_AD8(a, b, c, d, e, f, g, h)                                                   \
  MONO_CH_AD(a) MONO_CH_AD(b) MONO_CH_AD(c) MONO_CH_AD(d) MONO_CH_AD(e) MONO_CH_AD(f)         \
  MONO_CH_AD(g) MONO_CH_AD(h)
#define MONO_CH_ADDA                                                                          \
  MONO_CH_AD8(64, 65, 66, 67, 68, 69, 70, 71) MONO_CH_AD8(72, 73, 74, 75, 76, 77, 78, 79)     \
  MONO_CH_AD8(80, 81, 82, 83, 84, 85, 86, 87) MONO_CH_AD8(88, 89, 90, 91, 92, 93, 94, 95)
#define MONO_CH_ADDB                                                                          \
  MONO_CH_AD8(96, 97, 98, 99, 100, 101, 102, 103)                                             \
  MONO_CH_AD8(104, 105, 106, 107, 108, 109, 110, 111)                                         \
  MONO_CH_AD8(112, 113, 114, 115, 116, 117, 118, 119)                                         \
  MONO_CH_AD8(120, 121, 122, 123, 124, 125, 126, 127)
// The LDS-DMAs of one half: U slots of 1 KiB. m0 = LDS address of the slot (one wait state between
// a write to m0 and the DMA that reads it: the v_add that moves the lanes' offset on fills it).
// No immediate offsets: an instruction offset would move the LDS address as well.
#define MONO_CH_G "buffer_load_dwordx4 %[goff], %[rs], 0 offen lds\n\t"
#define MONO_CH_NX "s_add_u32 m0, m0, 0x400\n\tv_add_u32 %[goff], 0x400, %[goff]\n\t"
#define MONO_CH_DMA_U1 MONO_CH_G "v_add_u32 %[goff], 0x400, %[goff]\n\t"
#define MONO_CH_DMA_U2 MONO_CH_G MONO_CH_NX MONO_CH_DMA_U1
#define MONO_CH_DMA_U4 MONO_CH_G MONO_CH_NX MONO_CH_G MONO_CH_NX MONO_CH_DMA_U2
#define MONO_CH_DMA_U8 MONO_CH_G MONO_CH_NX MONO_CH_G MONO_CH_NX MONO_CH_G MONO_CH_NX MONO_CH_G MONO_CH_NX MONO_CH_DMA_U4
// Refill the slots whose rows are in registers now, then make sure the next half's slots have
// landed: R - U loads may stay pending behind them. Rows are left to fetch while more than
// R / 2U iterations remain; afterwards (out of line, .Lmono_ch_no*) everything pending is waited for.
#define MONO_CH_REFILL(DMAS, tag)                                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                                                                  \
  "s_add_u32 m0, %[rb], %[pos]\n\t"                                                           \
  "s_cmp_gt_u32 %[n], %[kleft]\n\t"                                                           \
  "s_cbranch_scc0 .Lmono_ch_no" tag "_%=\n\t" DMAS                                            \
  "s_waitcnt vmcnt(%[vmw])\n"                                                                 \
  ".Lmono_ch_j" tag "_%=:\n\t"
#define MONO_CHAIN_ASM(DMAS)                                                                  \
  asm volatile(                                                                               \
      "s_mov_b32 %[keep], m0\n\t"                                                             \
      MONO_CH_RDA("%[rd]")                                                                    \
      "s_branch .Lmono_ch_top_%=\n"                                                           \
      ".Lmono_ch_noa_%=:\n\t"                                                                 \
      "s_waitcnt vmcnt(0)\n\t"                                                                \
      "s_branch .Lmono_ch_ja_%=\n"                                                            \
      ".Lmono_ch_nob_%=:\n\t"                                                                 \
      "s_waitcnt vmcnt(0)\n\t"                                                                \
      "s_branch .Lmono_ch_jb_%=\n"                                                            \
      ".Lmono_ch_top_%=:\n\t"                                                                 \
      MONO_CH_REFILL(DMAS, "a")                                                               \
      MONO_CH_RDB("%[rd]")                                                                    \
      MONO_CH_ADDA                                                                            \
      "s_add_u32 %[pn], %[pos], %[adv]\n\t"                                                   \
      "s_cmp_eq_u32 %[pn], %[ringb]\n\t"                                                      \
      "s_cselect_b32 %[pn], 0, %[pn]\n\t"                                                     \
      "s_add_u32 %[pos], %[pos], %[half]\n\t"                                                 \
      "v_add_u32 %[rd], %[rdlane], %[pn]\n\t"                                                 \
      MONO_CH_REFILL(DMAS, "b")                                                               \
      MONO_CH_RDA("%[rd]")                                                                    \
      MONO_CH_ADDB                                                                            \
      "s_mov_b32 %[pos], %[pn]\n\t"                                                           \
      "s_sub_u32 %[n], %[n], 1\n\t"                                                           \
      "s_cmp_lg_u32 %[n], 0\n\t"                                                              \
      "s_cbranch_scc1 .Lmono_ch_top_%=\n\t"                                                   \
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                     \
      "s_mov_b32 m0, %[keep]\n\t"                                                             \
      : [acc] "+&v"(acc), [rd] "+&v"(rd), [goff] "+&v"(goff), [pos] "+&s"(pos),               \
        [pn] "=&s"(pn), [n] "+&s"(iters), [keep] "=&s"(keep)                                  \
      : [rdlane] "v"(rdlane), [rs] "s"(rs), [rb] "s"(ring_lds),                            \
        [o0] "n"(0 * GS), [o1] "n"(1 * GS), [o2] "n"(2 * GS), [o3] "n"(3 * GS),               \
        [o4] "n"(4 * GS), [o5] "n"(5 * GS), [o6] "n"(6 * GS), [o7] "n"(7 * GS),               \
        [o8] "n"(8 * GS), [o9] "n"(9 * GS), [o10] "n"(10 * GS), [o11] "n"(11 * GS),           \
        [o12] "n"(12 * GS), [o13] "n"(13 * GS), [o14] "n"(14 * GS), [o15] "n"(15 * GS),       \
        [adv] "n"(16 * GS), [half] "n"(8 * GS), [ringb] "n"(R * 1024),                        \
        [kleft] "n"(R / (2 * U)), [vmw] "n"(R - U)                                            \
      : "memory", "scc", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72",       \
        "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84",   \
        "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",   \
        "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",  \
        "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117",       \
        "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127")

// acc + row 0 + row 1 + ... in that order, one column per lane (lane e < D owns column e; the
// other lanes repeat a column and their result is to be ignored).
//   rs     : descriptor of the packed scratch (chain_rsrc)
//   off0   : byte offset of the key's packed rows in it, 1 KiB aligned, wave-uniform
//   nslots : 1-KiB slots of the run, a multiple of 2 U = D / 4 (the run is padded with rows of
//            -0.0f, which leave any sum unchanged)
//   ring_lds : LDS byte address of this wave's ring of R slots (R a multiple of 2 U, R - U <= 63)
template <int D, int R>
__device__ __forceinline__ float chain_sum(chain_v4i rs, uint32_t off0_, uint32_t nslots_,
                                           uint32_t ring_lds_, float acc) {
  static_assert(D == 8 || D == 16 || D == 32 || D == 64, "row widths the chain is built for");
  constexpr int GS = D * 16;  // bytes of one group of four rows
  constexpr int U = D / 8;    // slots per half (eight groups)
  static_assert(R % (2 * U) == 0 && R >= 2 * U && R - U <= 63, "ring size");
  // (wave-uniform by contract; made provably so for the "s" operands)
  const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)off0_);
  const uint32_t nslots = (uint32_t)__builtin_amdgcn_readfirstlane((int)nslots_);
  const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)ring_lds_);
  const uint32_t lane = __lane_id();
  uint32_t goff = off0 + lane * 16u;
  const uint32_t rdlane = ring_lds + (lane & (uint32_t)(D - 1)) * 16u;
  // prologue: fill the ring
  const uint32_t fill = nslots < (uint32_t)R ? nslots : (uint32_t)R;
  for (uint32_t k = 0; k < fill; ++k) chain_dma16(rs, goff + k * 1024u, ring_lds + k * 1024u);
  goff += fill * 1024u;
  if (fill == (uint32_t)R) chain_wait_vm<R - U>();
  else chain_wait_vm<0>();
  uint32_t rd = rdlane, pos = 0, pn, iters = nslots / (2 * U), keep;
  if constexpr (D == 8) MONO_CHAIN_ASM(MONO_CH_DMA_U1);
  else if constexpr (D == 16) MONO_CHAIN_ASM(MONO_CH_DMA_U2);
  else if constexpr (D == 32) MONO_CHAIN_ASM(MONO_CH_DMA_U4);
  else MONO_CHAIN_ASM(MONO_CH_DMA_U8);
  return acc;
}

}  // namespace mono
