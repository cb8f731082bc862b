// mlp_rc.hip -- the tiny-MLP chains (rgbnet + refnet, model/nerf.py:125-142,877,884,1009) with the ACTIVATIONS RESIDENT
// IN REGISTERS: forward chain and backward data-gradient chain, one persistent launch each.
//
// Formulation.  Every product is computed transposed, D^T[feature, sample] = W[feature, k] * X^T[k, sample], on
// v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand and the activations as the B operand.  A wave owns 32 samples
// (the MFMA's 32 columns) and ALL output features (8 row tiles of 32 for a 256-wide layer = 128 accumulator VGPRs).  The
// accumulator layout of a 32x32 tile -- lane (j = lane & 31, h = lane >> 5), register r holds row 8 (r >> 2) + 4 h + (r & 3),
// column j -- is, register by register, a legal B operand of the NEXT layer: B wants lanes 0..31 to hold one k and lanes
// 32..63 another, and register r of tile t holds k_a = 32 t + 8 (r >> 2) + (r & 3) in the lower half and k_a + 4 in the
// upper half.  So layer l + 1 walks its reduction in the order (k_a, k_a + 4) over (t, r) and feeds the accumulator
// registers of layer l straight back into the matrix core: the activations never leave the register file between layers
// -- no LDS round trip, no transpose, no barrier for them.  (The k order of a sum differs from the natural one; an fp32
// sum in any fixed order has the same error bound, and the order is the same in every launch: results are deterministic.)
//
// The matching A operand is plain: for k-group (t, r >> 2) lane (j, h) needs W[32 t' + j][32 t + 8 (r >> 2) + 4 h + 0..3],
// 16 contiguous bytes of a row-major weight row = one ds_read_b128 serving four MFMAs.  Weights stream through a ring of
// three 40 KB LDS slots, one 32-column chunk of all rows per slot, filled by LDS-DMA (global_load_lds_dwordx4: no register
// pass, no ds_write) from a chunk image that a small pack kernel writes each step (weights change every step): rows
// zero-padded to a multiple of 32, columns to a multiple of 32, each 128-byte row XOR-swizzled by ((row >> 1) & 7) in units
// of 16 bytes, which makes every ds_read_b128 lane group hit 16 distinct 4-bank groups.  One s_barrier per chunk
// (8192 MFMA cycles of work per wave); the DMA runs two chunks ahead, across layer and block boundaries.
//
// A 256-thread workgroup (one wave per SIMD, up to 512 VGPRs each) per CU walks blocks of 128 samples through all layers.
// Per 128 samples a layer's 256 KB of weights is read from L2 once (the LDS-resident form read it once per 64 samples).
//
// Forward extras: bias is the accumulators' initial value; ReLU in registers; each layer's output goes to HBM for the
// weight-gradient kernel (mlp_wgrad.hip); the SIGN of every ReLU input is kept as one bit per element in exactly the
// register layout (16 bytes per lane per layer; element e of a word at bit 31 - e), so that the backward chain applies the ReLU mask from 16 bytes instead of
// re-reading the 512-byte-per-lane activation.
// Backward: the same kernel on images of W^T (the pack kernel transposes), dY of the top layer loaded once from HBM, every
// layer's dY written out for the weight-gradient kernel.
#include "fgs_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// The A-operand reads of the chunk loop are issued as inline asm and waited for by hand.  Left to hipcc, the two reads in
// flight at any time (operands of the next two steps) were drained by an `s_waitcnt lgkmcnt(0)` every second step -- i.e. the
// matrix pipe waited for a ds_read_b128 issued 128 cycles earlier, ~50 idle cycles per step, 15 % of the chunk phase --
// where `lgkmcnt(1)` (the older of the two has landed; LDS returns in order) is all a step needs.
__device__ __forceinline__ unsigned rc_lds_addr(const float *p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)p;
}
template <int OFF_BYTES>
__device__ __forceinline__ void rc_read_a(floatx4 &dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF_BYTES) : "memory");
}

constexpr int RC_MAXL = 8;            // layers per chain
constexpr int RC_MAXCH = 80;          // chunks per pass over all layers
constexpr int RC_THREADS = 256;
constexpr int RC_SLOTS = 3;
constexpr int RC_SLOT_FLOATS = 256 * 32;   // 256 rows x 32 columns
constexpr int RC_SINK_FLOATS = 512;  // sink for the deferred stores of lanes without a sample: 256 floats + slack
constexpr int RC_BLOCK = 128;         // samples per workgroup pass (4 waves x 32)

struct RcLayer {
  int nt, nch;                 // output row tiles (rows_pad / 32, one of 4 / 6 / 8 / 10), reduction chunks of 32
  int n_rows;                  // valid output features (bias / store bound)
  int relu;
  const float *in;             // non-null: the carried input (<= 256 columns) is loaded from here instead of taken from the
  int64_t ld_in;               //           previous layer's registers
  int in_cols, in_valid;       // columns of the buffer (multiple of 4) / columns that carry data
  const float *ext;            // non-null: columns appended to the carried input (reduction chunks 8, 9)
  int64_t ld_ext;
  int ext_cols, ext_valid;
  const float *bias;
  int bias_slot;               // row of the LDS bias table (zeros for a layer without bias)
  uint4 *mask_w;               // forward: sign bits of the ReLU input, [ceil(M / 32)][64] uint4
  const uint4 *mask_r;         // backward: bits applied to this layer's output
  float *out;
  int64_t ldo;
  int n_store;                 // leading output columns stored (multiple of 4)
  int defer_store;             // the HBM copy of the output is issued from inside the next layer's chunks
};

struct RcArgs {
  int64_t M;
  const int64_t *m_dev;
  int n_layers, total_chunks;
  FgsStamps stamps;                        // fgs_dyn_t.stamps / fgs_mlp_rc_debug_stamps: per workgroup {s_memtime, s_memrealtime} x 2
  const float *img;
  int64_t sink_off;                        // floats from img to the sink (RC_SINK_FLOATS of them)
  // (chunk j of the stream starts at 1 KB piece j * 4 NTT of the image and has 4 NTT pieces: every layer of a launch has the
  // same padded row count, so no per-chunk table is needed -- a table lookup per chunk was two scalar loads and an exposed
  // s_waitcnt in front of every DMA issue)
  RcLayer L[RC_MAXL];
};

// ------------------------------------------------------------------------------------------------ weight image
struct PackLayer {
  const float *W;
  int64_t ldw;
  int n_out, n_in;      // W is [n_out][ldw] with n_in valid columns
  int rows_pad, nch;    // image geometry: rows (multiple of 32), chunks
  int64_t base;         // float offset of the layer's first chunk in the image
  int64_t f4_begin;     // first float4 of this layer in the flat work range
};
struct PackArgs {
  int n_layers, transpose;
  float *img;
  int64_t f4_total;
  PackLayer L[RC_MAXL];
};

// image element (row, k) = transpose ? W[k][row] : W[row][k]; zero outside.  One thread per float4 of the image.
__global__ __launch_bounds__(FGS_BLOCK) void k_rc_pack(PackArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.f4_total) return;
  int l = 0;
  while (l + 1 < a.n_layers && i >= a.L[l + 1].f4_begin) ++l;
  const PackLayer &L = a.L[l];
  const int64_t q = i - L.f4_begin;              // float4 index inside the layer: [chunk][row][c4']
  const int per_chunk = L.rows_pad * 8;
  const int c = (int)(q / per_chunk), rem = (int)(q - (int64_t)c * per_chunk);
  const int row = rem >> 3, c4s = rem & 7;
  const int c4 = c4s ^ ((row >> 1) & 7);         // the source column group stored at swizzled position c4s
  const int k0 = c * 32 + 4 * c4;
  const int n_rows = a.transpose ? L.n_in : L.n_out, n_k = a.transpose ? L.n_out : L.n_in;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    float x = 0.f;
    if (row < n_rows && k < n_k) x = a.transpose ? L.W[(int64_t)k * L.ldw + row] : L.W[(int64_t)row * L.ldw + k];
    v[j] = x;
  }
  *reinterpret_cast<float4 *>(a.img + L.base + (int64_t)c * per_chunk * 4 + (int64_t)rem * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// ------------------------------------------------------------------------------------------------ the chain
// Per-phase cycle counters (what scripts/rc_bench.py prints as "phases") cost eight scalar registers plus the stamp temporaries
// in a kernel that has none to spare (106 of 106: two more and hipcc spills scalars to scratch -- 576 bytes per lane, everything
// 1.85 x slower).  They are compiled in only with -DFGS_RC_PHASE_STAMPS (make rc-stamps); start / end stamps are always there.
#ifdef FGS_RC_PHASE_STAMPS
constexpr bool RC_PHASES = true;
#else
constexpr bool RC_PHASES = false;
#endif
struct RcState {
  const RcArgs *a;
  float *ring;
  float *sink;                // 1 KB of global memory nobody reads (behind the weight image)
  int issued, done, total_steps;     // chunk counters of this workgroup (uniform, 32-bit: compared in scalar registers)
  int issue_j, issue_slot, slot, total_chunks;
  int wave, lane;
  // the chunk whose DMA is being issued piece by piece between the MFMAs of the running chunk
  const float *dma_src;       // UNIFORM base of the chunk being fetched (scalar registers); the lane's 16 bytes are dma_lane
  unsigned dma_lane;          // lane * 16: with a uniform base the instruction takes the scalar-base form -- no vector
                              // address arithmetic per piece (every vector instruction between the MFMAs costs matrix-pipe time)
  float *dma_dst;
  int dma_p, dma_pieces;
  // output of the layer that just finished: its tiles (now the B operands in `prev`) are written to HBM tile by tile from
  // inside the NEXT layer's chunks (a 128 KB burst per layer and workgroup is store-issue bound: ~12k idle cycles)
  float *pend_row;            // deferred output: this lane's row + 4 h (out + row * ldo + 4 h) or the sink; the four float4 of
                              // a pending tile sit at compile-time offsets from it
  int pend_on;                // uniform: the layer that just finished left an output to be stored from inside these chunks
  unsigned long long t_init, t_chunks, t_epi, t_load;   // diagnostics (fgs_mlp_rc_debug_stamps): shader cycles per phase
  bool timed;
};

// start the DMA of the next chunk of the stream (nothing is issued yet: rc_dma_piece does that)
template <int NTT>
__device__ __forceinline__ void rc_dma_begin(RcState &s) {
  const RcArgs &a = *s.a;
  s.dma_pieces = 4 * NTT;
  s.dma_src = a.img + (int64_t)s.issue_j * (4 * NTT * 256);
  s.dma_dst = s.ring + s.issue_slot * RC_SLOT_FLOATS;
  s.dma_p = s.wave;
  ++s.issued;
  s.issue_j = (s.issue_j + 1 == s.total_chunks) ? 0 : s.issue_j + 1;
  s.issue_slot = (s.issue_slot + 1 == RC_SLOTS) ? 0 : s.issue_slot + 1;
}

// One 1 KB piece (this wave's next one) of the chunk being fetched.  BRANCH-FREE: when the wave has no piece left (a chunk
// with fewer rows than the issue slots provide for, or the end of the stream) the instruction still executes, re-fetching
// the chunk's first piece into a 1 KB dump area behind the ring -- straight-line code keeps hipcc's register allocation of
// the 500-register loop body stable (with a uniform branch per slot it spilled a whole B tile to scratch inside the loop).
__device__ __forceinline__ void rc_dma_piece(RcState &s) {
  const bool ok = s.dma_p < s.dma_pieces;
  const int p = ok ? s.dma_p : 0;
  float *dst = ok ? s.dma_dst + p * 256 : s.ring + RC_SLOTS * RC_SLOT_FLOATS + s.wave * 256;
  // (uniform base + this lane's 16-byte offset.  Issued by hand in the scalar-base form -- s_mov m0 / global_load_lds_dwordx4
  // v_off, s[base] -- the kernel lost its register allocation: 576 bytes of scratch per lane, everything 1.85 x slower; through
  // the builtin the address is a vector pair, one v_lshl_add_u64 per piece.)
  unsigned lane_off = s.dma_lane;
  asm volatile("" : "+v"(lane_off));      // (opaque: keeps hipcc from folding the lane offset into a 64-bit vector base)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(
                                       reinterpret_cast<const char *>(s.dma_src + p * 256) + lane_off),
                                   (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  s.dma_p += 4;
}

template <int NTT>
__device__ __forceinline__ void rc_dma(RcState &s) {     // a whole chunk at once (prologue)
  rc_dma_begin<NTT>(s);
  while (s.dma_p < s.dma_pieces) rc_dma_piece(s);
}

// One 32-column chunk: 16 k-steps x NTT row tiles = 4 NTT steps of four MFMAs each.  B = the 16 registers of one input tile.
// Tile-major order inside a k-group: the four MFMAs of a step form a dependent chain on one accumulator (the 32x32x2 form
// issues back to back on one accumulator: issue interval = dependent latency = 64 cycles).
//
// The A operands come through three registers in rotation (PHASE = rotation index of this chunk's step 0, a compile-time
// constant): the ds_read_b128 of step i + 2 is issued between the second and third MFMA of step i, and the rotation runs
// ACROSS chunks -- the last two steps read the first two operands of the NEXT chunk from its slot (S_next) -- so the matrix
// pipe does not drain at a chunk boundary.  That is legal because the synchronisation sits in the MIDDLE of a chunk:
// after step 2 NTT - 1 every wave waits for its own DMA pieces of the next chunk (issued a full chunk earlier) and meets
// the others at the one s_barrier of the chunk.  Past it (a) the next chunk is complete in LDS for everybody and (b)
// everybody has left the previous chunk, whose slot the chunk after next may overwrite: its DMA pieces are issued one at a
// time between the MFMAs of the second half (issued in a burst in front of a chunk, the 8-10 LDS-DMA instructions of a wave
// cost ~1000 cycles of idle matrix pipe per chunk).  sched_barrier(0) pins this order: left alone, hipcc sinks every LDS
// read to its use and the pipe then waits for LDS after every step.
template <int NTT, int PHASE, int NS, int STORE_TILE>
__device__ __forceinline__ void rc_chunk(RcState &s, const float *__restrict__ S, const float *__restrict__ S_next,
                                         const int (&rdoff)[4], const floatx16 &B, floatx16 (&acc)[NTT], floatx4 (&A)[3],
                                         int h) {
  constexpr int STEPS = 4 * NTT;
  constexpr int HALF = STEPS / 2;
  // LDS byte addresses of this lane's 16 bytes in row tile 0 of the chunk, per k-group q; row tile t adds 4096 (immediate)
  unsigned ra[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ra[q] = rc_lds_addr(S + rdoff[q]);
  const unsigned ra_next = rc_lds_addr((S_next ? S_next : S) + rdoff[0]);
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
    const int q = i / NTT, t = i % NTT;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");      // this step's operand has landed (the next step's may be in flight)
    const floatx4 a = A[(PHASE + i) % 3];
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, B[4 * q + 0], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, B[4 * q + 1], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (i + 2 < STEPS) {
      constexpr int dummy = 0; (void)dummy;
      const int q2 = (i + 2) / NTT;
      switch ((i + 2) % NTT) {        // (compile-time after unrolling: the row tile is the instruction's immediate offset)
        case 0: rc_read_a<0 * 4096>(A[(PHASE + i + 2) % 3], ra[q2]); break;
        case 1: rc_read_a<1 * 4096>(A[(PHASE + i + 2) % 3], ra[q2]); break;
        case 2: rc_read_a<2 * 4096>(A[(PHASE + i + 2) % 3], ra[q2]); break;
        case 3: rc_read_a<3 * 4096>(A[(PHASE + i + 2) % 3], ra[q2]); break;
        case 4: rc_read_a<4 * 4096>(A[(PHASE + i + 2) % 3], ra[q2]); break;
        case 5: rc_read_a<5 * 4096>(A[(PHASE + i + 2) % 3], ra[q2]); break;
        case 6: rc_read_a<6 * 4096>(A[(PHASE + i + 2) % 3], ra[q2]); break;
        default: rc_read_a<7 * 4096>(A[(PHASE + i + 2) % 3], ra[q2]); break;
      }
    } else if (S_next) {
      if (i + 2 - STEPS == 0) rc_read_a<0>(A[(PHASE + i + 2) % 3], ra_next);
      else rc_read_a<4096>(A[(PHASE + i + 2) % 3], ra_next);
    }
    // (the chunk's vector-memory instructions stay spread over its steps, one per step: in an isolated probe --
    // scripts/diag/rc_step_probe.hip -- one burst per chunk was cheaper, 230 vs 480 idle cycles for 8 LDS-DMA pieces, but in
    // this kernel, with the pieces' address arithmetic around them, the burst form measured 1.5 % slower)
    // the four float4 of the pending tile go out at steps HALF + 1, + 3, + 5, + 7 -- BEHIND the chunk's barrier.  On gfx9-class
    // hardware stores count in vmcnt like loads, and the barrier's s_waitcnt vmcnt(0) waits for their write acknowledgements:
    // issued in the first half (steps 1 .. 7) they had 2 - 3.6 K cycles to come back, from here a chunk and a half.
    if (STORE_TILE >= 0 && i >= HALF && i < HALF + 8 && (i & 1) == 1) {
      const int qs = (i - HALF) >> 1;
      constexpr int off = 32 * (STORE_TILE >= 0 ? STORE_TILE : 0);
      // UNCONDITIONAL: pend_row was chosen once per layer -- this lane's output row, or (sample beyond M) a 1 KB sink behind the
      // weight image.  A per-lane test here costs a vector compare whose result the scalar unit waits for behind whatever MFMA
      // is in the pipe: ~100 cycles per store, four stores per chunk.
      if (s.pend_on)
        *reinterpret_cast<float4 *>(s.pend_row + off + 8 * qs) = make_float4(B[4 * qs], B[4 * qs + 1], B[4 * qs + 2], B[4 * qs + 3]);
    }
    if (i >= HALF) {      // issue slot k of NS sits at step HALF + k * HALF / NS (compile-time)
#pragma unroll
      for (int k = 0; k < NS; ++k)
        if (HALF + (k * HALF) / NS == i) rc_dma_piece(s);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, B[4 * q + 2], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, B[4 * q + 3], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (i == STEPS / 2 - 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s.issued < s.total_steps) rc_dma_begin<NTT>(s);
      else s.dma_pieces = 0;                      // end of the stream: the issue slots only touch the dump area
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// 32 appended input columns (tile `c` of L.ext) of this lane's sample -> the 16 registers of a B tile
__device__ __forceinline__ void rc_load_ext(const RcLayer &L, int c, int64_t rowc, int h, floatx16 &B) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = 32 * c + 8 * q + 4 * h;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < L.ext_cols) v = *reinterpret_cast<const float4 *>(L.ext + rowc * L.ld_ext + col);
    if (col + 3 >= L.ext_valid) {      // padding columns of the buffer may hold anything: 0 * NaN would poison the sum
      if (col + 1 >= L.ext_valid) v.y = 0.f;
      if (col + 2 >= L.ext_valid) v.z = 0.f;
      v.w = 0.f;
      if (col >= L.ext_valid) v.x = 0.f;
    }
    B[4 * q] = v.x; B[4 * q + 1] = v.y; B[4 * q + 2] = v.z; B[4 * q + 3] = v.w;
  }
}

// chunks C, C + 1, ... of a layer (compile-time recursion: the rotation phase of a chunk is a template argument).
// Appended input columns (forward refnet layer 0: reduction chunks 8 and 9) need no registers of their own: input tile 0 is
// dead once chunk 0 has run, tile 1 after chunk 1, so the appended columns are loaded into THOSE registers right there --
// seven chunks (~25 us) before chunks 8 and 9 multiply by them.
template <int NTT, bool BWD, int C>
__device__ __forceinline__ void rc_chunks(RcState &s, const RcLayer &L, int nch, bool has_ext, const int (&rdoff)[4],
                                          floatx16 (&prev)[8], floatx16 (&acc)[NTT], floatx4 (&A)[3], int64_t rowc, int h) {
  // (nch / has_ext are read once per layer by the caller: behind the `memory` clobbers of the hand-issued instructions the
  // compiler re-read L.nch and L.ext from the kernel-argument segment at every chunk -- a scalar load and an exposed wait each)
  if constexpr (C < 10) {
    if (C < nch) {
      const float *S = s.ring + s.slot * RC_SLOT_FLOATS;
      s.slot = (s.slot + 1 == RC_SLOTS) ? 0 : s.slot + 1;
      ++s.done;
      const float *S_next = (s.done < s.total_steps) ? s.ring + s.slot * RC_SLOT_FLOATS : nullptr;
      constexpr int PH = (C * 4 * NTT) % 3;
      constexpr int NS = 2 * NTT / 2;        // DMA issue slots per chunk = the 1 KB pieces a wave fetches per chunk (rows / 32)
      // chunk C < 8 multiplies by input tile C = output tile C of the previous layer: its HBM copy goes out from here
      rc_chunk<NTT, PH, NS, (C < 8 ? C : -1)>(s, S, S_next, rdoff, prev[C < 8 ? C : C - 8], acc, A, h);
      if constexpr (!BWD && C < 2) {
        if (has_ext) rc_load_ext(L, C, rowc, h, prev[C]);
      }
      rc_chunks<NTT, BWD, C + 1>(s, L, nch, has_ext, rdoff, prev, acc, A, rowc, h);
    }
  }
}

template <int NTT, bool BWD>
__device__ __forceinline__ void rc_layer(RcState &s, const RcLayer &L, const int (&rdoff)[4], floatx16 (&prev)[8],
                                         floatx4 (&A)[3], int64_t row, int64_t rowc, bool row_ok, int64_t group, int h) {
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (RC_PHASES && s.timed) t0 = __builtin_amdgcn_s_memtime();
  floatx16 acc[NTT];
#pragma unroll
  for (int t = 0; t < NTT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  uint4 mbits = make_uint4(0u, 0u, 0u, 0u);
  if (BWD && L.mask_r) mbits = L.mask_r[group * 64 + s.lane];
  if (RC_PHASES && s.timed) t1 = __builtin_amdgcn_s_memtime();
  const int nch = __builtin_amdgcn_readfirstlane(L.nch);
  const bool has_ext = L.ext != nullptr;
  rc_chunks<NTT, BWD, 0>(s, L, nch, has_ext, rdoff, prev, acc, A, rowc, h);
  if (RC_PHASES && s.timed) t2 = __builtin_amdgcn_s_memtime();
  {   // the two operands prefetched for the next layer's first steps sit at rotation index (nch * 4 NTT) % 3: make that 0
    const int ph = (L.nch * 4 * NTT) % 3;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the hand-issued reads must have landed before they are copied)
    if (ph == 1) { const floatx4 t0 = A[1], t1 = A[2]; A[0] = t0; A[1] = t1; }
    else if (ph == 2) { const floatx4 t0 = A[2], t1 = A[0]; A[0] = t0; A[1] = t1; }
  }
  // ---- epilogue, one tile at a time: bias (from the LDS copy made at kernel start), activation / mask, sign bits, store;
  // the tile's 16 VGPRs then ARE the next layer's B operand
  // the HBM copy of this output is deferred into the next layer's chunks when that layer multiplies by all of its tiles
  // (next.nch >= NTT); the last layer of the chain stores here
  // Everything layer-uniform is read ONCE into scalars here, and the per-element work below is branch-free: written with the
  // conditions inside the unrolled loops (if (L.relu) ..., if (!defer && L.out && ...) ..., L.out + row * L.ldo per store)
  // hipcc re-read the layer record from the kernel-argument segment and rebuilt masks and 64-bit addresses per element --
  // behind the `memory` clobbers of the hand-issued instructions nothing is hoisted for it: 67 K cycles of epilogue per
  // 128-sample block where the arithmetic (read, bias, max, sign bit: five vector instructions per element) needs ~35 K.
  const bool defer = __builtin_amdgcn_readfirstlane(L.defer_store) != 0;
  const int n_store = __builtin_amdgcn_readfirstlane(L.n_store);
  float *const out_row = (L.out && row_ok) ? L.out + row * L.ldo : nullptr;      // this lane's output row
  s.pend_on = (defer && L.out) ? 1 : 0;
  s.pend_row = (out_row ? out_row : s.sink) + 4 * h;     // (whole tiles: the launcher defers only outputs of 32 NTT columns)
  float *const st_row = defer ? nullptr : out_row;
  const float lo = __builtin_amdgcn_readfirstlane(L.relu) ? 0.f : -INFINITY;      // max(v, -inf) = v: no ReLU
  unsigned bits[4] = {0u, 0u, 0u, 0u};
  // (backward, a layer without a mask: all ones keep every element)
  const bool has_mask = BWD && L.mask_r != nullptr;
  const unsigned mb[4] = {has_mask ? mbits.x : ~0u, has_mask ? mbits.y : ~0u, has_mask ? mbits.z : ~0u,
                          has_mask ? mbits.w : ~0u};
  const float *bias_l = s.ring + RC_SLOTS * RC_SLOT_FLOATS + 4 * 256 + L.bias_slot * 256 + 4 * h;
  // (the bias values of tile t + 1 are fetched while tile t is worked on: read where they are used, each of the 4 NTT
  // ds_read_b128 of a layer was waited for on the spot -- the compiler does not move an LDS read across the output stores)
  float4 bn[4];
  if (!BWD) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bn[q] = *reinterpret_cast<const float4 *>(bias_l + 8 * q);
  }
#pragma unroll
  for (int t = 0; t < NTT; ++t) {
    floatx16 v = acc[t];
    float4 bc[4];
    if (!BWD) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bc[q] = bn[q];
      if (t + 1 < NTT) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bn[q] = *reinterpret_cast<const float4 *>(bias_l + 32 * (t + 1) + 8 * q);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!BWD) {
        const float4 b = bc[q];
        v[4 * q] += b.x; v[4 * q + 1] += b.y; v[4 * q + 2] += b.z; v[4 * q + 3] += b.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * q + j;
        if (!BWD) {
          v[r] = fmaxf(v[r], lo);
          // "v > 0" of a ReLU output (only layers with ReLU store their bits): v >= +0 there, so v > 0 <=> its bit pattern p
          // is non-zero <=> bit 31 of p + 0x7fffffff.  One add and one v_alignbit that shifts the word left and takes that
          // bit in at the bottom: element e = 16 (t & 1) + r of a word ends up at bit 31 - e.  (Written as a compare and a
          // select -- also what hipcc makes of min(p, 1) -- every element cost a VCC round trip with its wait states.)
          if (t < 8) bits[t >> 1] = __builtin_amdgcn_alignbit(bits[t >> 1], __float_as_uint(v[r]) + 0x7fffffffu, 31);
        } else if (t < 8) {
          // keep / zero by the saved sign bit: the bit sign-extended to a 0 / ~0 word (one bit-field extract) and an AND --
          // a compare + select goes through an SGPR pair and its wait states
          const int keep = (int)(mb[t >> 1] << ((t & 1) * 16 + r)) >> 31;      // element e sits at bit 31 - e
          v[r] = __uint_as_float(__float_as_uint(v[r]) & (unsigned)keep);
        }
      }
      const int col = 32 * t + 8 * q + 4 * h;
      if (st_row && col < n_store)
        *reinterpret_cast<float4 *>(st_row + col) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
    if (t < 8) prev[t] = v;
    __builtin_amdgcn_sched_barrier(0);      // one tile at a time: 16 staging registers, not 128
  }
  if (!BWD && L.mask_w) L.mask_w[group * 64 + s.lane] = make_uint4(bits[0], bits[1], bits[2], bits[3]);
  if (RC_PHASES && s.timed) {
    s.t_init += t1 - t0; s.t_chunks += t2 - t1; s.t_epi += __builtin_amdgcn_s_memtime() - t2;
  }
}

// One instantiation per (direction, row tiles per layer): every layer of a launch has the same padded output width, so the
// kernel holds ONE copy of the 4 NTT-step loop body (with the layers' tile counts switched at run time, hipcc's register
// allocation over three inlined bodies spilled B tiles to scratch inside the loop).
template <bool BWD, int NTT>
__global__ __launch_bounds__(RC_THREADS, 1) void k_mlp_rc(RcArgs a) {
  // ring | 1 KB dump area per wave | bias table [RC_MAXL + 1][256] (row RC_MAXL: zeros)
  __shared__ __attribute__((aligned(16))) float ring[RC_SLOTS * RC_SLOT_FLOATS + 4 * 256 + (RC_MAXL + 1) * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int64_t M = fgs_rows(a.M, a.m_dev);
  const int64_t nb = (M + RC_BLOCK - 1) / RC_BLOCK;
  if ((int64_t)blockIdx.x >= nb) return;
  const int64_t my_blocks = (nb - blockIdx.x + gridDim.x - 1) / gridDim.x;
  unsigned long long *const stamps = fgs_stamp_wg(a.stamps);
  if (stamps && tid == 0) {
    stamps[0] = __builtin_amdgcn_s_memtime();
    stamps[1] = __builtin_amdgcn_s_memrealtime();
  }
  RcState s;
  s.a = &a; s.ring = ring; s.issued = 0; s.total_steps = __builtin_amdgcn_readfirstlane((int)(my_blocks * a.total_chunks));
  s.issue_j = 0; s.issue_slot = 0; s.slot = 0; s.wave = wave; s.lane = lane; s.done = 0; s.total_chunks = a.total_chunks;
  s.dma_p = 0; s.dma_pieces = 0; s.dma_src = a.img; s.dma_lane = (unsigned)lane * 16u; s.dma_dst = ring;
  s.pend_row = nullptr; s.pend_on = 0; s.sink = const_cast<float *>(a.img) + a.sink_off;
  s.timed = RC_PHASES && stamps != nullptr; s.t_init = s.t_chunks = s.t_epi = s.t_load = 0;
  rc_dma<NTT>(s);
  if (s.total_steps > 1) rc_dma<NTT>(s);
  {   // bias table -> LDS, once (a layer's epilogue then reads 16 bytes per output group instead of waiting on HBM)
    float *bt = ring + RC_SLOTS * RC_SLOT_FLOATS + 4 * 256;
    for (int i = tid; i < (RC_MAXL + 1) * 256; i += RC_THREADS) {
      const int l = i >> 8, c = i & 255;
      float v = 0.f;
      if (!BWD && l < a.n_layers && a.L[l].bias && c < a.L[l].n_rows) v = a.L[l].bias[c];
      bt[i] = v;
    }
  }
  int rdoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) rdoff[q] = l31 * 32 + 4 * ((2 * q + h) ^ ((l31 >> 1) & 7));
  // chunks 0 and 1 of the stream are in flight; the first mid-chunk barrier (inside chunk 0) covers chunk 1, this one chunk 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  floatx4 A[3];
  rc_read_a<0>(A[0], rc_lds_addr(ring + rdoff[0]));
  rc_read_a<4096>(A[1], rc_lds_addr(ring + rdoff[0]));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  A[2] = A[0];
  floatx16 prev[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[t][r] = 0.f;

  for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const int64_t group = b * 4 + wave;                  // 32-sample group of this wave
    const int64_t row = group * 32 + l31;
    const bool row_ok = row < M;
    const int64_t rowc = row_ok ? row : M - 1;           // loads of padding samples read a valid row; their results are dropped
    for (int l = 0; l < a.n_layers; ++l) {
      const RcLayer &L = a.L[l];
      unsigned long long tl = 0;
      if (RC_PHASES && s.timed) tl = __builtin_amdgcn_s_memtime();
      if (L.in) {
        // the carried input of this lane's sample: 32 float4 loads issued back to back (no per-load branch: a column group
        // beyond the buffer re-reads the row's last float4), THEN the padding columns are zeroed (they may hold anything:
        // 0 * NaN would poison the sum)
        const float *in_row = L.in + rowc * L.ld_in;
        const int in_cols = __builtin_amdgcn_readfirstlane(L.in_cols), in_valid = __builtin_amdgcn_readfirstlane(L.in_valid);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * c + 8 * q + 4 * h;
            const float4 v = *reinterpret_cast<const float4 *>(in_row + (col < in_cols ? col : in_cols - 4));
            prev[c][4 * q] = v.x; prev[c][4 * q + 1] = v.y; prev[c][4 * q + 2] = v.z; prev[c][4 * q + 3] = v.w;
          }
        if (in_valid < 256) {
#pragma unroll
          for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                prev[c][4 * q + i] = (32 * c + 8 * q + 4 * h + i < in_valid) ? prev[c][4 * q + i] : 0.f;
        }
      }
      if (RC_PHASES && s.timed) s.t_load += __builtin_amdgcn_s_memtime() - tl;
      rc_layer<NTT, BWD>(s, L, rdoff, prev, A, row, rowc, row_ok, group, h);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stamps && tid == 0) {
    stamps[2] = __builtin_amdgcn_s_memtime();
    stamps[3] = __builtin_amdgcn_s_memrealtime();
    stamps[4] = s.t_init; stamps[5] = s.t_chunks;
    stamps[6] = s.t_epi; stamps[7] = s.t_load;
  }
}

unsigned long long *g_rc_stamps = nullptr;

bool rc_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int rc_round_tiles(int rows) {   // output rows -> instantiated tile count
  const int t = (rows + 31) / 32;
  return t <= 4 ? 4 : t <= 6 ? 6 : t <= 8 ? 8 : 10;   // 10: rejected by the launcher
}

}  // namespace

// Diagnostics: while set, every workgroup of the chain kernels records {shader clock, 100 MHz wall clock} at its start and
// end into stamps[4 * workgroup ..] (>= 8 * 256 entries); in-kernel clock = d(shader) / d(wall) * 100 MHz.  NULL: off.
FGS_API int fgs_mlp_rc_debug_stamps(unsigned long long *stamps) {
  g_rc_stamps = stamps;
  return 0;
}

FGS_API int64_t fgs_mlp_rc_image_floats(int backward, int n_layers, const fgs_rc_layer_t *layers) {
  if (!layers || n_layers < 1 || n_layers > RC_MAXL) return -1;
  int64_t total = 0;
  for (int l = 0; l < n_layers; ++l) {
    const int rows = backward ? layers[l].n_in : layers[l].n_out, k = backward ? layers[l].n_out : layers[l].n_in;
    total += (int64_t)rc_round_tiles(rows) * 32 * ((k + 31) / 32) * 32;
  }
  return total + RC_SINK_FLOATS;
}

FGS_API int fgs_mlp_rc_chain(int backward, int64_t M, int n_layers, const fgs_rc_layer_t *layers, const float *in0,
                             int64_t ld_in0, int in0_cols, float *image_ws, int64_t image_ws_floats, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && n_layers >= 1 && n_layers <= RC_MAXL, FGS_E_RANGE,
              "fgs_mlp_rc_chain: M=%lld n_layers=%d (1..%d)", (long long)M, n_layers, RC_MAXL);
  if (M == 0) return 0;
  FGS_REQUIRE(layers && in0 && image_ws, FGS_E_INVALID, "fgs_mlp_rc_chain: null pointer");
  FGS_REQUIRE(in0_cols > 0 && in0_cols <= 256 && (in0_cols % 4) == 0 && (ld_in0 % 4) == 0 && ld_in0 >= in0_cols &&
                  rc_aligned16(in0) && rc_aligned16(image_ws),
              FGS_E_INVALID, "fgs_mlp_rc_chain: first input: 4..256 columns, multiple of 4, 16-byte aligned rows");
  const int64_t need = fgs_mlp_rc_image_floats(backward, n_layers, layers);
  FGS_REQUIRE(image_ws_floats >= need, FGS_E_INVALID, "fgs_mlp_rc_chain: image workspace %lld floats, need %lld",
              (long long)image_ws_floats, (long long)need);
  RcArgs a;
  PackArgs p;
  a.M = M; a.m_dev = fgs_dyn_rows(dyn); a.n_layers = n_layers; a.img = image_ws; a.stamps = fgs_dyn_stamps(dyn, g_rc_stamps);
  p.n_layers = n_layers; p.transpose = backward ? 1 : 0; p.img = image_ws;
  int carried = in0_cols;            // columns of the input carried in registers
  int64_t base = 0, f4 = 0;
  int chunk = 0;
  for (int l = 0; l < n_layers; ++l) {
    const fgs_rc_layer_t &U = layers[l];
    FGS_REQUIRE(U.W && U.n_out > 0 && U.n_in > 0 && U.ldw >= U.n_in, FGS_E_INVALID, "fgs_mlp_rc_chain: layer %d: bad weight", l);
    const int rows = backward ? U.n_in : U.n_out, k = backward ? U.n_out : U.n_in;
    FGS_REQUIRE(rows <= 256 && k <= 320, FGS_E_RANGE, "fgs_mlp_rc_chain: layer %d: %d x %d beyond 256 x 320", l, rows, k);
    const int ext_cols = backward ? 0 : U.ext_cols;
    // the input may be padded to a multiple of 4 columns (X0: 106 -> 108, refnet: 256 + 51 -> 256 + 52); the padding is
    // zeroed on load
    FGS_REQUIRE(k <= carried + ext_cols && carried + ext_cols < k + 4, FGS_E_INVALID,
                "fgs_mlp_rc_chain: layer %d reduces over %d columns but its input has %d (+%d appended)", l, k, carried, ext_cols);
    FGS_REQUIRE(carried <= 256 && ext_cols >= 0 && ext_cols <= 64 && (ext_cols % 4) == 0 &&
                    (ext_cols == 0 || (U.ext && carried == 256 && (U.ld_ext % 4) == 0 && rc_aligned16(U.ext))),
                FGS_E_INVALID, "fgs_mlp_rc_chain: layer %d: appended columns need a full 256-column carried input, <= 64 of them, "
                               "multiple of 4, aligned", l);
    FGS_REQUIRE(!U.out || ((U.ldo % 4) == 0 && rc_aligned16(U.out) && (U.n_store % 4) == 0 && U.n_store <= rc_round_tiles(rows) * 32 &&
                           U.ldo >= U.n_store), FGS_E_INVALID, "fgs_mlp_rc_chain: layer %d: bad output", l);
    FGS_REQUIRE((!U.bias || rc_aligned16(U.bias)) && (!U.mask_bits || rc_aligned16(U.mask_bits)), FGS_E_INVALID,
                "fgs_mlp_rc_chain: layer %d: bias / mask bits must be 16-byte aligned", l);
    FGS_REQUIRE(backward || (rows % 4) == 0, FGS_E_INVALID, "fgs_mlp_rc_chain: layer %d: n_out must be a multiple of 4", l);
    RcLayer &L = a.L[l];
    L.nt = rc_round_tiles(rows);
    FGS_REQUIRE(L.nt <= 8 && (l == 0 || L.nt == a.L[0].nt), FGS_E_INVALID,
                "fgs_mlp_rc_chain: layer %d produces %d columns: every layer of one chain must have the same output width "
                "rounded up to 128 / 192 / 256 (narrow or wider products go through fgs_gemm_f32)", l, rows);
    L.nch = (k + 31) / 32;
    L.n_rows = rows;
    L.relu = backward ? 0 : U.relu;
    L.in = (l == 0) ? in0 : nullptr; L.ld_in = ld_in0; L.in_cols = in0_cols;
    L.in_valid = (l == 0 && !ext_cols) ? k : in0_cols;
    L.ext = ext_cols ? U.ext : nullptr; L.ld_ext = U.ld_ext; L.ext_cols = ext_cols; L.ext_valid = k - carried;
    L.bias = backward ? nullptr : U.bias;
    L.bias_slot = (!backward && U.bias) ? l : RC_MAXL;
    L.mask_w = (!backward && U.relu) ? reinterpret_cast<uint4 *>(U.mask_bits) : nullptr;
    L.mask_r = backward ? reinterpret_cast<const uint4 *>(U.mask_bits) : nullptr;
    L.out = U.out; L.ldo = U.ldo; L.n_store = U.n_store;
    FGS_REQUIRE(chunk + L.nch <= RC_MAXCH, FGS_E_RANGE, "fgs_mlp_rc_chain: more than %d chunks", RC_MAXCH);
    PackLayer &P = p.L[l];
    P.W = U.W; P.ldw = U.ldw; P.n_out = U.n_out; P.n_in = U.n_in; P.rows_pad = L.nt * 32; P.nch = L.nch; P.base = base;
    P.f4_begin = f4;
    chunk += L.nch;
    base += (int64_t)P.rows_pad * 32 * L.nch;
    f4 += (int64_t)P.rows_pad * 8 * L.nch;
    carried = rows < 256 ? rows : 256;      // what the next layer finds in the registers
  }
  for (int l = 0; l < n_layers; ++l)     // deferred stores need the next layer to walk every tile of this output
    a.L[l].defer_store = (l + 1 < n_layers && a.L[l + 1].nch >= a.L[l].nt && !a.L[l + 1].in &&
                          (!a.L[l].out || a.L[l].n_store == 32 * a.L[l].nt)) ? 1 : 0;      // (deferred stores are whole tiles)
  a.total_chunks = chunk;
  a.sink_off = need - RC_SINK_FLOATS;
  p.f4_total = f4;
  hipStream_t st = fgs_s(stream);
  hipLaunchKernelGGL(k_rc_pack, dim3(fgs_blocks(f4)), dim3(FGS_BLOCK), 0, st, p);
  FGS_LAUNCH_OK("fgs_mlp_rc_chain (pack)");
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  const int64_t nb = (M + RC_BLOCK - 1) / RC_BLOCK;
  const unsigned grid = (unsigned)(nb < cus ? nb : cus);
  const int nt = a.L[0].nt;
#define RC_LAUNCH(B, N) hipLaunchKernelGGL((k_mlp_rc<B, N>), dim3(grid), dim3(RC_THREADS), 0, st, a)
  if (backward) { if (nt == 4) RC_LAUNCH(true, 4); else if (nt == 6) RC_LAUNCH(true, 6); else RC_LAUNCH(true, 8); }
  else { if (nt == 4) RC_LAUNCH(false, 4); else if (nt == 6) RC_LAUNCH(false, 6); else RC_LAUNCH(false, 8); }
#undef RC_LAUNCH
  FGS_LAUNCH_OK("fgs_mlp_rc_chain");
  return 0;
}
