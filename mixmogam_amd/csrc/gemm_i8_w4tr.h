// gemm_i8_w4tr.h -- the 4-wave int8 MFMA job stream of gemm_i8_w4s.h for operands that are CONTRACTION-MAJOR in
// memory: both tiles are column windows of one row-major [k][col] image (the SNP-major genotype store itself:
// k = SNP, col = individual), so the kinship GEMMs K = S'S read the store as it lies and the individual-major
// transposed image (a 5 GB HBM round trip per call, 3.3 of 12.5 ms of the IBS kinship at N = 5000 x M = 1e6) is gone.
//
// What the MFMA wants -- 16 consecutive k per lane for the lane's row -- is produced by gfx950's transposed LDS read
// ds_read_b64_tr_b8 (tools/probe/ds_read_tr_b8.hip, measured: within a group of 16 consecutive lanes, lane 2q + p
// supplies the address of bytes 8p .. 8p+7 of row q of an 8 x 16 byte block, and lane i receives column i of the 8
// rows, row 0 in its lowest byte).  Two such reads (k rows +0..7, +8..15) fill one v4i operand of
// v_mfma_i32_32x32x32_i8: lanes 0-15 / 16-31 hold operand rows 0-15 / 16-31 with k 0..15, lanes 32-63 the same rows
// with k 16..31.
//
// LDS image of an operand tile per K step: [128 k rows][256 cols] bytes (32 KiB, the size of the row-major tile of
// gemm_i8_core.h), filled by the same lane-linear LDS-DMA (buffer_load ... lds, 16 B per lane, one instruction = 4 k
// rows x 256 B); the 16-byte chunks of row k are XOR-swizzled in PAIRS, position = chunk ^ ((k & 7) << 1), applied
// to the per-lane source offset of the DMA and again on the read side.  A transposed read of a 32-lane half covers 8
// consecutive k rows x 32 bytes (2 chunks): with the pair swizzle the 8 rows fall into 8 different bank octets --
// conflict free (bank = (addr / 4) % 64).
//
// Pipeline, barrier placement and the RAW / WAR argument: gemm_i8_w4s.h / k_scan_w4s.hip; per slice 16 MFMA and 16
// ds_read_b64_tr_b8 (two per fragment) instead of 8 ds_read_b128 -- the same LDS bytes.
#pragma once
#include "gemm_i8_w4s.h"

namespace mmg {

typedef int v2i __attribute__((ext_vector_type(2)));

struct StageOpTr {
  __amdgpu_buffer_rsrc_t rs;
  int v_even, v_odd;   // per-lane source offsets (bytes) of even / odd pieces: the pair swizzle differs by (k & 4)
  int ld4;             // 4 * ld
};

// base: byte 0 of the tile's column window at k row 0 of the stage; ld: row stride of the image
__device__ __forceinline__ StageOpTr make_stage_op_tr(const int8_t* base, int64_t ld, int lane) {
  StageOpTr s;
  s.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  const int rp = lane >> 4, cp = lane & 15;            // row within the 4-row piece, chunk POSITION in LDS
  s.v_even = rp * (int)ld + ((cp ^ (rp << 1)) << 4);
  s.v_odd = rp * (int)ld + ((cp ^ ((4 + rp) << 1)) << 4);
  s.ld4 = 4 * (int)ld;
  return s;
}

// piece i in 0..7 of this wave: k rows wave*32 + i*4 .. +4 (1 KiB of LDS)
__device__ __forceinline__ void stage_piece_tr(const StageOpTr& s, char* lds_tile, int wave, int i) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(s.rs, (MMG_AS3 void*)(lds_tile + (wave * 8 + i) * 1024), 16,
                                           (i & 1) ? s.v_odd : s.v_even, (wave * 8 + i) * s.ld4, 0, 0);
}

// Fragment of a 32-row operand block for K slice `slice` (32 k rows): two transposed reads.  INLINE ASM on purpose:
// the compiler models the ds_read_tr builtins as memory WRITES, and its waitcnt pass then puts an `s_waitcnt vmcnt(0)`
// in front of every one of them that follows an LDS-DMA load (32 stalls on the stage in flight per K step; plain
// ds_read_b128 get no such wait).  The price: the compiler does not know that the outputs arrive asynchronously, so
// every consumer is preceded by an explicit counted `s_waitcnt lgkmcnt(n)` tied to the fragment registers
// (frag_wait) -- the counts are derived in w4tr_slice.  addr: LDS byte address of the block for slice 0.
__device__ __forceinline__ void lds_frag_tr(v4i& f, uint32_t addr, int slice) {
  v2i lo, hi;
  switch (slice) {                                       // the offset is an instruction immediate
    case 0: asm volatile("ds_read_b64_tr_b8 %0, %2\n\tds_read_b64_tr_b8 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
    case 1: asm volatile("ds_read_b64_tr_b8 %0, %2 offset:8192\n\tds_read_b64_tr_b8 %1, %2 offset:10240" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
    case 2: asm volatile("ds_read_b64_tr_b8 %0, %2 offset:16384\n\tds_read_b64_tr_b8 %1, %2 offset:18432" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
    default: asm volatile("ds_read_b64_tr_b8 %0, %2 offset:24576\n\tds_read_b64_tr_b8 %1, %2 offset:26624" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
  }
  f = v4i{lo.x, lo.y, hi.x, hi.y};
}

// at most N LDS reads may still be outstanding when the instructions that consume f issue
template <int N>
__device__ __forceinline__ void frag_wait(v4i& f) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N));
}
template <int N>
__device__ __forceinline__ void frag_wait2(v4i& f, v4i& g) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f), "+v"(g) : "n"(N));
}

// End of a job: EVERY transposed read issued so far has landed, and until here the compiler must treat their destination
// registers as live.  The last slice of a job reads the fragments of a slice that does not exist (the stream has no "last
// slice" form); to the compiler those asm outputs are dead the moment they are written, so it handed their VGPRs to the
// epilogue's address arithmetic -- and a read that returned late overwrote a row / column index: a wild atomicAdd.  Alone
// on a CU the reads are back long before the epilogue starts; beside another kernel's LDS traffic (the upload stream of
// the chunk prefetcher) they sometimes were not: the GPU memory fault of round 4 (tools/stress_two_threads.py, 2 s to fault;
// DESIGN.md section 9).
__device__ __forceinline__ void frag_drain(Frag4& f, Frag4& g) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.a[2]), "+v"(f.a[3]), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3]),
                 "+v"(g.a[0]), "+v"(g.a[1]), "+v"(g.a[2]), "+v"(g.a[3]), "+v"(g.b[0]), "+v"(g.b[1]), "+v"(g.b[2]), "+v"(g.b[3]));
}

// lane-constant part of a fragment address: operand rows tile_row0 + (lane & 31) .. , k half (lane >> 5)
__device__ __forceinline__ int frag_base_tr(int tile_row0, int lane) {
  const int i16 = lane & 15, g = lane >> 4, h = g >> 1;
  const int chunk = (tile_row0 >> 4) + (g & 1);          // 16-column chunk of this lane group's rows
  return (h * 16 + (i16 >> 1)) * 256 + ((chunk ^ (i16 & 14)) << 4) + (i16 & 1) * 8;
}

// ---- operand formats of the stream --------------------------------------------------------------------------------
// FmtI8: int8 operands, v_mfma_i32_32x32x32_i8, K step = 128 k rows, tile image [128 k][256 B] (above).
struct FmtI8 {
  typedef v16i Acc;
  typedef StageOpTr Stage;
  static constexpr int KROWS = 128;
  static __device__ __forceinline__ Stage make(const int8_t* base, int64_t ld, int wave, int lane) { (void)wave; return make_stage_op_tr(base, ld, lane); }
  static __device__ __forceinline__ void piece(const Stage& s, char* tile, int wave, int i) { stage_piece_tr(s, tile, wave, i); }
  static __device__ __forceinline__ int frag_base(int tile_row0, int lane) { return frag_base_tr(tile_row0, lane); }
  static __device__ __forceinline__ void frag(v4i& f, uint32_t addr, int slice) { lds_frag_tr(f, addr, slice); }
  static __device__ __forceinline__ Acc mma(v4i a, v4i b, Acc c) { return mfma8(a, b, c); }
  static __device__ __forceinline__ Acc zero() { return v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
};

// FmtF4: FP4 (E2M1) operands -- 0/1 genotypes as the nibbles 0x0 / 0x2 (= 0.0 / 1.0) -- on
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales: twice the MACs per instruction and half the bytes per operand
// element of the int8 form; the fp32 accumulators hold exact integers while a job's partial counts stay below 2^24.
// Image: [k][column nibbles], low nibble = even column; K step = 256 k rows, so that the LDS tile is the [256 rows][128
// B] image of gemm_i8_core.h (32 KiB, 16-byte chunk swizzle c ^ ((row >> 1) & 7), staged by 8-row x 128-B pieces) with
// k as the ROW.  ds_read_b64_tr_b4 (tools/probe/ds_read_tr_b4.hip): within a group of 16 lanes, lane q supplies the
// address of the 8 bytes (16 nibble columns) of row q of a 16 x 16 nibble block and lane i receives column i, row 0 in
// its lowest nibble; two reads (k rows +0..15, +16..31) are one 16-byte operand (32 k).  Bank check: a 32-lane half
// reads 16 rows x 16 B; rows k, k+1 differ by 32 banks, row pairs by their swizzle: conflict free.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
struct FmtF4 {
  typedef v16f Acc;
  typedef StageOp4 Stage;
  static constexpr int KROWS = 256;
  static __device__ __forceinline__ Stage make(const int8_t* base, int64_t ld, int wave, int lane) { return make_stage_op4(base, ld, wave, lane); }
  static __device__ __forceinline__ void piece(const Stage& s, char* tile, int wave, int i) { stage_piece4(s, 0, tile, wave, i); }
  static __device__ __forceinline__ int frag_base(int tile_row0, int lane) {
    const int q = lane & 15, g = lane >> 4, h = g >> 1;
    return (h * 32 + q) * 128 + (((tile_row0 >> 5) ^ (q >> 1)) << 4) + (g & 1) * 8;
  }
  static __device__ __forceinline__ void frag(v4i& f, uint32_t addr, int slice) {
    v2i lo, hi;
    switch (slice) {
      case 0: asm volatile("ds_read_b64_tr_b4 %0, %2\n\tds_read_b64_tr_b4 %1, %2 offset:2048" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
      case 1: asm volatile("ds_read_b64_tr_b4 %0, %2 offset:8192\n\tds_read_b64_tr_b4 %1, %2 offset:10240" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
      case 2: asm volatile("ds_read_b64_tr_b4 %0, %2 offset:16384\n\tds_read_b64_tr_b4 %1, %2 offset:18432" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
      default: asm volatile("ds_read_b64_tr_b4 %0, %2 offset:24576\n\tds_read_b64_tr_b4 %1, %2 offset:26624" : "=&v"(lo), "=&v"(hi) : "v"(addr)); break;
    }
    f = v4i{lo.x, lo.y, hi.x, hi.y};
  }
  static __device__ __forceinline__ Acc mma(v4i a, v4i b, Acc c) {
    const v8i a8{a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8{b.x, b.y, b.z, b.w, 0, 0, 0, 0};   // FP4 uses the first four dwords
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  static __device__ __forceinline__ Acc zero() { return v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
};

struct W4JobTr {
  const int8_t* P;     // column 0 of the job's 256-column P window at k row 0 of the job
  const int8_t* Q;     // likewise for the Q window
  int nks;             // K steps of 128 rows (>= 1)
};

// One slice: 16 MFMA on `cur`; the 8 fragments of (slot `src`, slice) into `nxt` (one per MFMA pair, order a0 b0 a1 b1
// a2 b2 a3 b3, two reads each); DMA pieces [P0, P1) of the cursor's stage into slot `dst`.
// Counted waits.  LDS reads return in order.  When a slice starts, the 16 reads of `cur` (issued during the previous
// slice, or all complete after a barrier's lgkmcnt(0)) are the only LDS operations in flight; fragment position p
// (a_m: 2m, b_n: 2n + 1) is complete once at most 16 - 2(p + 1) + r reads are outstanding, r = reads this slice has
// issued so far (2 per finished MFMA pair, the pair's own two after its first MFMA).  The MFMAs run in ORD order, whose
// first use of positions 0..7 is at MFMA index 0, 0, 1, 2, 4, 6, 9, 12 with r = 0, 0, 2, 2, 4, 6, 10, 12:
// lgkmcnt(12), (12), (10), (10), (10), (12), (12).  No other lgkm operation may be issued inside the stream (scalar
// loads return out of order): job descriptors are held in registers.
// ABL (timing ablations, WRONG results, instantiated by `make EXPERIMENTS=1` builds only): 1 = no LDS-DMA inside the loop, 2 = no
// fragment reads inside the loop, 3 = neither (the MFMAs, the barrier and the waits alone).
template <class Fmt, int P0, int P1, bool ZERO, int ABL = 0>
__device__ __forceinline__ void w4tr_slice(typename Fmt::Acc (&acc)[4][4], Frag4& cur, Frag4& nxt, const char* src,
                                           const int (&ab)[4], const int (&bb)[4], int slice,
                                           const typename Fmt::Stage& sp, const typename Fmt::Stage& sq, char* dst,
                                           int wave) {
  static_assert(P1 - P0 <= 16, "at most two DMA pieces per MFMA pair");
  constexpr int PER = (P1 - P0 > 8) ? 2 : 1;             // DMA pieces per MFMA pair
  const uint32_t s32 = (uint32_t)(uintptr_t)src;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m0 = ORD_M[2 * i], n0 = ORD_N[2 * i], m1 = ORD_M[2 * i + 1], n1 = ORD_N[2 * i + 1];
    if (i == 0) frag_wait2<12>(cur.a[0], cur.b[0]);      // MFMA 0: a0 b0
    if (i == 1) frag_wait<10>(cur.b[1]);                 // MFMA 2: a0 b1
    if (i == 2) frag_wait<10>(cur.a[2]);                 // MFMA 4: a2 b0
    if (i == 3) frag_wait<10>(cur.b[2]);                 // MFMA 6: a0 b2
    if (i == 6) frag_wait<12>(cur.b[3]);                 // MFMA 12: a0 b3
    if (ZERO) acc[m0][n0] = Fmt::mma(cur.a[m0], cur.b[n0], Fmt::zero());
    else acc[m0][n0] = Fmt::mma(cur.a[m0], cur.b[n0], acc[m0][n0]);
    // the asm statements keep their order among themselves, but the MFMAs (no side effects) are free to sink below
    // them -- the scheduler bunched all 16 at the end of the slice: pin the source order MFMA / reads / MFMA / DMA
    __builtin_amdgcn_sched_barrier(0);
    if (ABL != 2 && ABL != 3) {
      if ((i & 1) == 0) Fmt::frag(nxt.a[i >> 1], s32 + (uint32_t)ab[i >> 1], slice);
      else Fmt::frag(nxt.b[i >> 1], s32 + (uint32_t)bb[i >> 1], slice);
    }
    if (i == 0) frag_wait<12>(cur.a[1]);                 // MFMA 1: a1 b0
    if (i == 4) frag_wait<12>(cur.a[3]);                 // MFMA 9: a3 b0
    if (ZERO) acc[m1][n1] = Fmt::mma(cur.a[m1], cur.b[n1], Fmt::zero());
    else acc[m1][n1] = Fmt::mma(cur.a[m1], cur.b[n1], acc[m1][n1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int pc = P0 + PER * i + u;
      if (pc < P1 && ABL != 1 && ABL != 3) {
        if (pc < 8) Fmt::piece(sp, dst, wave, pc);
        else Fmt::piece(sq, dst + TILE_BYTES, wave, pc - 8);
      }
    }
  }
}

// Runs jobs j0 .. j1-1 of this workgroup; job(j) -> W4JobTr (wave-uniform); ld: row stride of the image (bytes);
// pre / epi as in w4s_stream: acc[m][n] = the wave's 4 x 4 accumulator tiles, rows = P columns wm*128 + m*32..,
// columns = Q columns wn*128 + n*32.. (C layout of gemm_i8_core.h).
// N3: DMA pieces of stage t+2 issued right after the barrier of step t (slice 3), the other 16 - N3 in slice 0 of step
// t+1.  The operands of these kernels stream from HBM / the Infinity Cache (a 5 GB store, not an L2-resident digit
// image): the later a piece is issued the likelier the next barrier waits for it.
// PFD > 0: L2 prefetch.  The stream keeps ONE stage of LDS-DMA in flight (two LDS slots), so by Little's law a CU moves
// at most one stage (64 KiB) per memory round trip: an L2 hit returns well inside a K step (~1 us), a line that has to
// come from the Infinity Cache or HBM may not, and the barrier waits.  With PFD = d every wave touches the 128 lines
// of ITS rows of stage (cursor + d) -- two 4-byte-per-lane `buffer_load ... lds` into a 256-byte scratch behind the
// stage buffers (lds + LDS_BYTES + wave * 256: the caller allocates LDS_BYTES + 1024) -- one K step before the real
// DMA of that stage is issued, so the DMA finds its lines in L2.  Same-job stages only.
constexpr int W4TR_PF_LDS = 1024;

template <int N3 = 8, int PFD = 0, class Fmt = FmtI8, int ABL = 0, class JobFn, class PreFn, class EpiFn>
__device__ __forceinline__ void w4tr_stream(int j0, int j1, int64_t ld, char* lds, JobFn&& job, PreFn&& pre, EpiFn&& epi) {
  static_assert(N3 >= 8 && N3 <= 16, "N3");
  static_assert(PFD == 0 || Fmt::KROWS == 128 || Fmt::KROWS == 256, "the L2 prefetch knows the int8 and the FP4 stage shapes");
  if (j1 <= j0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int ab[4], bb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ab[i] = Fmt::frag_base(wm * 128 + i * 32, lane);
    bb[i] = Fmt::frag_base(wn * 128 + i * 32, lane) + TILE_BYTES;
  }
  const int64_t kstep_bytes = (int64_t)Fmt::KROWS * ld;

  // ---- issue cursor over the flattened stage stream (wave-uniform scalars); the descriptor base moves with the stage
  // (a K step is 128 rows = 128 * ld bytes: beyond 32-bit offsets for long contraction ranges)
  int cj = j0;
  W4JobTr cjb = job(cj);
  int cks = 0, cnks = cjb.nks;
  typename Fmt::Stage sp = Fmt::make(cjb.P, ld, wave, lane);
  typename Fmt::Stage sq = Fmt::make(cjb.Q, ld, wave, lane);
  auto rebase = [&]() {
    sp.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(cjb.P + cks * kstep_bytes), 0, 0x7fffffff, 0x00020000);
    sq.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(cjb.Q + cks * kstep_bytes), 0, 0x7fffffff, 0x00020000);
  };
  // one 128-byte line per lane: lane l -> k row wave*32 + (l >> 1) of the stage, line (l & 1) of the 256-byte window
  // (FP4 image: a stage is 256 k rows of one 128-byte line each: lane l -> k row wave * 64 + l)
  const int pf_off = Fmt::KROWS == 128 ? (wave * 32 + (lane >> 1)) * (int)ld + (lane & 1) * 128 : (wave * 64 + lane) * (int)ld;
  auto prefetch = [&]() {
    if (PFD > 0 && cks + PFD < cnks) {
      char* scratch = lds + LDS_BYTES + wave * 256;
      const __amdgpu_buffer_rsrc_t rp =
          __builtin_amdgcn_make_buffer_rsrc((void*)(cjb.P + (cks + PFD) * kstep_bytes), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rq =
          __builtin_amdgcn_make_buffer_rsrc((void*)(cjb.Q + (cks + PFD) * kstep_bytes), 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (MMG_AS3 void*)scratch, 4, pf_off, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (MMG_AS3 void*)scratch, 4, pf_off, 0, 0, 0);
    }
  };
  auto advance = [&]() {
    if (cks + 1 < cnks) { ++cks; rebase(); return; }
    if (cj + 1 < j1) {
      ++cj;
      cjb = job(cj);
      cks = 0;
      cnks = cjb.nks;
      rebase();
    }                                                    // else: stay on the last stage (harmless re-issue)
  };

  // ---- prologue: stage 0 complete, the first N3 pieces of stage 1 in flight, fragments of step 0 slice 0
#pragma unroll
  for (int i = 0; i < 8; ++i) Fmt::piece(sp, lds, wave, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) Fmt::piece(sq, lds + TILE_BYTES, wave, i);
  advance();                                             // -> stage 1
#pragma unroll
  for (int i = 0; i < N3; ++i) {
    if (i < 8) Fmt::piece(sp, lds + BUF_BYTES, wave, i);
    else Fmt::piece(sq, lds + BUF_BYTES + TILE_BYTES, wave, i - 8);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N3) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  prefetch();                                            // lines of stage 1+PFD (stage 2 for PFD = 1)

  Frag4 f0, f1;
  {
    const uint32_t l32 = (uint32_t)(uintptr_t)lds;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                        // the order the counted waits of the first slice assume
      Fmt::frag(f0.a[i], l32 + (uint32_t)ab[i], 0);
      Fmt::frag(f0.b[i], l32 + (uint32_t)bb[i], 0);
    }
  }

  typename Fmt::Acc acc[4][4];                           // written (not accumulated) by the first slice of every job
  int t = 0;
  auto step = [&](bool first) {
    char* cur = lds + (t & 1) * BUF_BYTES;
    char* oth = lds + ((t + 1) & 1) * BUF_BYTES;
    if (first) w4tr_slice<Fmt, N3, 16, true, ABL>(acc, f0, f1, cur, ab, bb, 1, sp, sq, oth, wave);
    else w4tr_slice<Fmt, N3, 16, false, ABL>(acc, f0, f1, cur, ab, bb, 1, sp, sq, oth, wave);
    w4tr_slice<Fmt, 16, 16, false, ABL>(acc, f1, f0, cur, ab, bb, 2, sp, sq, oth, wave);
    w4tr_slice<Fmt, 16, 16, false, ABL>(acc, f0, f1, cur, ab, bb, 3, sp, sq, oth, wave);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    advance();                                           // -> stage t+2
    prefetch();                                          // lines of stage t+2+PFD -> L2
    w4tr_slice<Fmt, 0, N3, false, ABL>(acc, f1, f0, oth, ab, bb, 0, sp, sq, cur, wave);
    ++t;
  };
  for (int jj = j0; jj < j1; ++jj) {
    const int nks = job(jj).nks;
    for (int ks = 0; ks < nks - 1; ++ks) step(ks == 0);
    pre(jj);
    step(nks == 1);
    frag_drain(f0, f1);                                  // the reads of the slice after the last one (see there)
    epi(jj, acc);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the re-issued tail stages must land before LDS is released
}

}  // namespace mmg
