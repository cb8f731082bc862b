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
// register layout (16 bytes per lane per layer), so that the backward chain applies the ReLU mask from 16 bytes instead of
// re-reading the 512-byte-per-lane activation.
// Backward: the same kernel on images of W^T (the pack kernel transposes), dY of the top layer loaded once from HBM, every
// layer's dY written out for the weight-gradient kernel.
#include "fgs_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int RC_MAXL = 8;            // layers per chain
constexpr int RC_MAXCH = 80;          // chunks per pass over all layers
constexpr int RC_THREADS = 256;
constexpr int RC_SLOTS = 3;
constexpr int RC_SLOT_FLOATS = 320 * 32;   // up to 320 rows x 32 columns
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
  uint4 *mask_w;               // forward: sign bits of the ReLU input, [ceil(M / 32)][64] uint4
  const uint4 *mask_r;         // backward: bits applied to this layer's output
  float *out;
  int64_t ldo;
  int n_store;                 // leading output columns stored (multiple of 4)
};

struct RcArgs {
  int64_t M;
  const int64_t *m_dev;
  int n_layers, total_chunks;
  const float *img;
  int chunk_piece0[RC_MAXCH];              // first 1 KB piece of chunk j in the image
  int chunk_pieces[RC_MAXCH];              // 1 KB pieces of chunk j (= rows_pad / 8).  (int, not a byte array: hipcc (ROCm 7.2)
                                           // folded the byte index of a uint8 table into the SGPR BASE of the neighbouring
                                           // s_load_dword -- base = kernarg + j, soffset = 3 j -- and the scalar unit drops
                                           // the base's two low bits before adding: chunk_piece0[1] read back chunk_piece0[0])
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
struct RcState {
  const RcArgs *a;
  float *ring;
  int64_t issued, total_steps;
  int issue_j, issue_slot, slot;
  int wave, lane;
};

__device__ __forceinline__ void rc_dma(RcState &s) {
  const RcArgs &a = *s.a;
  const int pieces = a.chunk_pieces[s.issue_j];
  const float *src = a.img + (int64_t)a.chunk_piece0[s.issue_j] * 256 + s.lane * 4;
  float *dst = s.ring + s.issue_slot * RC_SLOT_FLOATS;
  for (int p = s.wave; p < pieces; p += 4)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * 256),
                                     (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
  ++s.issued;
  s.issue_j = (s.issue_j + 1 == a.total_chunks) ? 0 : s.issue_j + 1;
  s.issue_slot = (s.issue_slot + 1 == RC_SLOTS) ? 0 : s.issue_slot + 1;
}

// one 32-column chunk: 16 k-steps x NTT row tiles.  B = the 16 registers of one input tile.
// Tile-major order inside a k-group: the four MFMAs of a tile form a dependent chain on its accumulator (the 32x32x2 form
// issues back to back on one accumulator: issue interval = dependent latency = 64 cycles), after which the tile's A
// registers are free and the ds_read_b128 of the NEXT k-group lands in them while the other tiles' MFMAs run -- a single
// A buffer of 4 NTT registers is enough to keep the LDS reads under the matrix work.
template <int NTT>
__device__ __forceinline__ void rc_chunk(const float *__restrict__ S, const int (&rdoff)[4], const floatx16 &B,
                                         floatx16 (&acc)[NTT]) {
  // software pipeline over the 4 NTT (k-group, tile) steps, three A registers in rotation: the ds_read_b128 of step i + 2
  // is issued between the second and third MFMA of step i.  sched_barrier(0) pins that order (left alone, the scheduler
  // sinks every read to its use and the matrix pipe waits for LDS four times per k-group).  hipcc drains lgkmcnt(0) in
  // front of every third step; by then the youngest read has two MFMAs (128 cycles) behind it.
  constexpr int STEPS = 4 * NTT;
  float4 A[3];
  A[0] = *reinterpret_cast<const float4 *>(S + rdoff[0]);
  A[1] = *reinterpret_cast<const float4 *>(S + 1024 + rdoff[0]);
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
    const int q = i / NTT, t = i % NTT;
    const float4 a = A[i % 3];
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, B[4 * q + 0], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, B[4 * q + 1], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (i + 2 < STEPS) {
      const int q2 = (i + 2) / NTT, t2 = (i + 2) % NTT;
      A[(i + 2) % 3] = *reinterpret_cast<const float4 *>(S + t2 * 1024 + rdoff[q2]);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, B[4 * q + 2], acc[t], 0, 0, 0);
    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, B[4 * q + 3], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NTT, bool BWD>
__device__ __forceinline__ void rc_layer(RcState &s, const RcLayer &L, const int (&rdoff)[4], floatx16 (&prev)[8],
                                         floatx16 (&ext)[2], int64_t row, int64_t rowc, bool row_ok, int64_t group, int h) {
  floatx16 acc[NTT];
  // accumulators start at the bias
#pragma unroll
  for (int t = 0; t < NTT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 32 * t + 8 * q + 4 * h;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!BWD && L.bias && col < L.n_rows) b = *reinterpret_cast<const float4 *>(L.bias + col);
      acc[t][4 * q] = b.x; acc[t][4 * q + 1] = b.y; acc[t][4 * q + 2] = b.z; acc[t][4 * q + 3] = b.w;
    }
  uint4 mbits = make_uint4(0u, 0u, 0u, 0u);
  if (BWD && L.mask_r) mbits = L.mask_r[group * 64 + s.lane];
#pragma unroll
  for (int c = 0; c < (BWD ? 8 : 10); ++c) {
    if (c < L.nch) {
      // chunk `c` of this layer has landed in every wave's share of the slot (each wave waits for its own DMA pieces, then
      // the barrier); all waves are also past the previous chunk, whose slot the next DMA overwrites
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s.issued < s.total_steps) rc_dma(s);
      const float *S = s.ring + s.slot * RC_SLOT_FLOATS;
      if (c < 8) rc_chunk<NTT>(S, rdoff, prev[c < 8 ? c : 0], acc);
      else if (!BWD) rc_chunk<NTT>(S, rdoff, ext[c >= 8 ? c - 8 : 0], acc);
      s.slot = (s.slot + 1 == RC_SLOTS) ? 0 : s.slot + 1;
    }
  }
  // ---- epilogue: activation / mask, sign bits, store, hand the tile registers to the next layer
  unsigned bits[4] = {0u, 0u, 0u, 0u};
  const unsigned mb[4] = {mbits.x, mbits.y, mbits.z, mbits.w};
#pragma unroll
  for (int t = 0; t < NTT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[t][r];
      if (!BWD) {
        if (L.relu) v = fmaxf(v, 0.f);
        if (t < 8 && v > 0.f) bits[t >> 1] |= 1u << ((t & 1) * 16 + r);
      } else if (t < 8) {
        if (L.mask_r && !((mb[t >> 1] >> ((t & 1) * 16 + r)) & 1u)) v = 0.f;
      }
      acc[t][r] = v;
    }
    if (L.out && row_ok) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 32 * t + 8 * q + 4 * h;
        if (col < L.n_store)
          *reinterpret_cast<float4 *>(L.out + row * L.ldo + col) =
              make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
      }
    }
    if (t < 8) prev[t] = acc[t];
  }
  if (!BWD && L.mask_w) L.mask_w[group * 64 + s.lane] = make_uint4(bits[0], bits[1], bits[2], bits[3]);
}

template <bool BWD>
__global__ __launch_bounds__(RC_THREADS, 1) void k_mlp_rc(RcArgs a) {
  __shared__ __attribute__((aligned(16))) float ring[RC_SLOTS * RC_SLOT_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int64_t M = fgs_rows(a.M, a.m_dev);
  const int64_t nb = (M + RC_BLOCK - 1) / RC_BLOCK;
  if ((int64_t)blockIdx.x >= nb) return;
  const int64_t my_blocks = (nb - blockIdx.x + gridDim.x - 1) / gridDim.x;
  RcState s;
  s.a = &a; s.ring = ring; s.issued = 0; s.total_steps = my_blocks * a.total_chunks;
  s.issue_j = 0; s.issue_slot = 0; s.slot = 0; s.wave = wave; s.lane = lane;
  rc_dma(s);
  if (s.total_steps > 1) rc_dma(s);
  int rdoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) rdoff[q] = l31 * 32 + 4 * ((2 * q + h) ^ ((l31 >> 1) & 7));
  floatx16 prev[8], ext[2];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[t][r] = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ext[t][r] = 0.f;

  for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
    const int64_t group = b * 4 + wave;                  // 32-sample group of this wave
    const int64_t row = group * 32 + l31;
    const bool row_ok = row < M;
    const int64_t rowc = row_ok ? row : M - 1;           // loads of padding samples read a valid row; their results are dropped
    for (int l = 0; l < a.n_layers; ++l) {
      const RcLayer &L = a.L[l];
      if (L.in) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * c + 8 * q + 4 * h;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < L.in_cols) v = *reinterpret_cast<const float4 *>(L.in + rowc * L.ld_in + col);
            if (col + 3 >= L.in_valid) {     // padding columns of the buffer may hold anything: 0 * NaN would poison the sum
              if (col + 1 >= L.in_valid) v.y = 0.f;
              if (col + 2 >= L.in_valid) v.z = 0.f;
              v.w = 0.f;
              if (col >= L.in_valid) v.x = 0.f;
            }
            prev[c][4 * q] = v.x; prev[c][4 * q + 1] = v.y; prev[c][4 * q + 2] = v.z; prev[c][4 * q + 3] = v.w;
          }
      }
      if (!BWD && L.ext) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * c + 8 * q + 4 * h;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < L.ext_cols) v = *reinterpret_cast<const float4 *>(L.ext + rowc * L.ld_ext + col);
            if (col + 3 >= L.ext_valid) {
              if (col + 1 >= L.ext_valid) v.y = 0.f;
              if (col + 2 >= L.ext_valid) v.z = 0.f;
              v.w = 0.f;
              if (col >= L.ext_valid) v.x = 0.f;
            }
            ext[c][4 * q] = v.x; ext[c][4 * q + 1] = v.y; ext[c][4 * q + 2] = v.z; ext[c][4 * q + 3] = v.w;
          }
      }
      switch (L.nt) {      // (a forward layer has at most 8 row tiles; only a backward layer can have 10: refnet's dZ)
        case 4: rc_layer<4, BWD>(s, L, rdoff, prev, ext, row, rowc, row_ok, group, h); break;
        case 6: rc_layer<6, BWD>(s, L, rdoff, prev, ext, row, rowc, row_ok, group, h); break;
        case 10: if (BWD) { rc_layer<BWD ? 10 : 8, BWD>(s, L, rdoff, prev, ext, row, rowc, row_ok, group, h); break; }
        default: rc_layer<8, BWD>(s, L, rdoff, prev, ext, row, rowc, row_ok, group, h); break;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

bool rc_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int rc_round_tiles(int rows) {   // output rows -> instantiated tile count
  const int t = (rows + 31) / 32;
  return t <= 4 ? 4 : t <= 6 ? 6 : t <= 8 ? 8 : 10;
}

}  // namespace

FGS_API int64_t fgs_mlp_rc_image_floats(int backward, int n_layers, const fgs_rc_layer_t *layers) {
  if (!layers || n_layers < 1 || n_layers > RC_MAXL) return -1;
  int64_t total = 0;
  for (int l = 0; l < n_layers; ++l) {
    const int rows = backward ? layers[l].n_in : layers[l].n_out, k = backward ? layers[l].n_out : layers[l].n_in;
    total += (int64_t)rc_round_tiles(rows) * 32 * ((k + 31) / 32) * 32;
  }
  return total;
}

FGS_API int fgs_mlp_rc_chain(int backward, int64_t M, int n_layers, const fgs_rc_layer_t *layers, const float *in0,
                             int64_t ld_in0, int in0_cols, float *image_ws, int64_t image_ws_floats, fgs_stream_t stream) {
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
  a.M = M; a.m_dev = fgs_row_ptr(); a.n_layers = n_layers; a.img = image_ws;
  p.n_layers = n_layers; p.transpose = backward ? 1 : 0; p.img = image_ws;
  int carried = in0_cols;            // columns of the input carried in registers
  int64_t base = 0, f4 = 0;
  int chunk = 0;
  for (int l = 0; l < n_layers; ++l) {
    const fgs_rc_layer_t &U = layers[l];
    FGS_REQUIRE(U.W && U.n_out > 0 && U.n_in > 0 && U.ldw >= U.n_in, FGS_E_INVALID, "fgs_mlp_rc_chain: layer %d: bad weight", l);
    const int rows = backward ? U.n_in : U.n_out, k = backward ? U.n_out : U.n_in;
    FGS_REQUIRE(rows <= 320 && k <= 320, FGS_E_RANGE, "fgs_mlp_rc_chain: layer %d: %d x %d beyond 320 x 320", l, rows, k);
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
    L.nch = (k + 31) / 32;
    L.n_rows = rows;
    L.relu = backward ? 0 : U.relu;
    L.in = (l == 0) ? in0 : nullptr; L.ld_in = ld_in0; L.in_cols = in0_cols;
    L.in_valid = (l == 0 && !ext_cols) ? k : in0_cols;
    L.ext = ext_cols ? U.ext : nullptr; L.ld_ext = U.ld_ext; L.ext_cols = ext_cols; L.ext_valid = k - carried;
    L.bias = backward ? nullptr : U.bias;
    L.mask_w = (!backward && U.relu) ? reinterpret_cast<uint4 *>(U.mask_bits) : nullptr;
    L.mask_r = backward ? reinterpret_cast<const uint4 *>(U.mask_bits) : nullptr;
    L.out = U.out; L.ldo = U.ldo; L.n_store = U.n_store;
    FGS_REQUIRE(chunk + L.nch <= RC_MAXCH, FGS_E_RANGE, "fgs_mlp_rc_chain: more than %d chunks", RC_MAXCH);
    PackLayer &P = p.L[l];
    P.W = U.W; P.ldw = U.ldw; P.n_out = U.n_out; P.n_in = U.n_in; P.rows_pad = L.nt * 32; P.nch = L.nch; P.base = base;
    P.f4_begin = f4;
    for (int c = 0; c < L.nch; ++c) {
      a.chunk_piece0[chunk] = (int)((base + (int64_t)c * P.rows_pad * 32) / 256);
      a.chunk_pieces[chunk] = P.rows_pad / 8;
      ++chunk;
    }
    base += (int64_t)P.rows_pad * 32 * L.nch;
    f4 += (int64_t)P.rows_pad * 8 * L.nch;
    carried = rows < 256 ? rows : 256;      // what the next layer finds in the registers
  }
  a.total_chunks = chunk;
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
  if (backward) hipLaunchKernelGGL(k_mlp_rc<true>, dim3(grid), dim3(RC_THREADS), 0, st, a);
  else hipLaunchKernelGGL(k_mlp_rc<false>, dim3(grid), dim3(RC_THREADS), 0, st, a);
  FGS_LAUNCH_OK("fgs_mlp_rc_chain");
  return 0;
}
