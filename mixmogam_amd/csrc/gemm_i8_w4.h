// gemm_i8_w4.h -- LDS-DMA staging for the 4-wave kernels (k_scan_w4s.hip, k_scan_w4b.hip): one wave per SIMD,
// wave tile 128 x 128 over the 2-slot / 128-byte K-step LDS image of gemm_i8_core.h; each wave stages 64 rows of
// an operand tile as 8 pieces of 8 rows x 128 B.
#pragma once
#include "gemm_i8_core.h"

namespace mmg {

constexpr int W4_THREADS = 256;

struct StageOp4 {
  __amdgpu_buffer_rsrc_t rs;
  int v_even, v_odd, ld8;
};

__device__ __forceinline__ StageOp4 make_stage_op4(const int8_t* base, int64_t ld, int wave, int lane) {
  StageOp4 s;
  s.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  const int base_row = wave * 64 + (lane >> 3);
  const int c0 = (lane & 7) ^ ((base_row >> 1) & 7);
  s.v_even = base_row * (int)ld + c0 * 16;
  s.v_odd = base_row * (int)ld + (c0 ^ 4) * 16;
  s.ld8 = 8 * (int)ld;
  return s;
}

// piece i in 0..7: rows (wave*8 + i)*8 .. +8.  AUX: cache policy of the load (0 default, 2 = nt; experiments only)
template <int AUX = 0>
__device__ __forceinline__ void stage_piece4(const StageOp4& s, int k0, char* lds_tile, int wave, int i) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rs, (MMG_AS3 void*)(lds_tile + (wave * 8 + i) * 1024), 16,
                                           (i & 1) ? s.v_odd : s.v_even, k0 + i * s.ld8, 0, AUX);
}

}  // namespace mmg
