// mlp_wgrad.hip -- every weight (and bias) gradient of the tiny MLPs in ONE launch:  dW_l += dY_l^T X_l,  db_l += colsum(dY_l)
// for all layers l of rgbnet + refnet (model/nerf.py:125-142; the products autograd runs as seven separate addmm calls).
//
// Shape of the problem: outputs are tiny (256 x 108..308), the reduction runs over the M ~ 50-60 K surviving samples.  So the
// SAMPLES are split over the chip: a workgroup takes one (layer, block of <= 256 weight columns) and a contiguous range of
// samples, keeps the whole output block in its accumulators (16 MFMA tiles = 256 accumulator registers per wave) and streams
// its sample range through LDS into MFMA operands:
//   A operand (32x32x2: lane (j, h) holds A[i = j][k = h])  = dY[sample 2s + h][out-feature tile + j]
//   B operand                                               = X [sample 2s + h][in-feature tile + j]
// -- both are row segments of the row-major activations, so no transpose is needed anywhere.  Panels of 16 sample rows are
// gathered by LDS-DMA (global_load_lds_dwordx4: every lane names its own 16 source bytes, the data lands lane-linear) into
// COMPACT panels with compile-time pitch -- A [16][256], B [16][64 | 128 | 256] holding only the block's columns -- so every
// operand read is a ds_read_b32 with an immediate offset: no address arithmetic in the loop.  Four slots in rotation, one
// barrier per 16 samples, the DMA of a chunk issued three chunks ahead.  (Each LDS-DMA instruction blocks the wave's issue
// for ~60 cycles -- scripts/diag/rc_step_probe.hip -- but the alternative, global_load_dwordx4 into staging registers and
// ds_write_b128 a chunk later, measured 1.5 % SLOWER here: 9 123 vs 8 980 cycles per chunk.)
// Every k-step is 16 MFMAs per wave whatever the block's width -- the wave arrangement adapts:
//   mode A (129..256 columns): waves 2 x 2 over (rows, columns), 4 x 4 tiles each, all 8 k-steps of a chunk;
//   mode B ( 65..128 columns): waves 2 (rows) x 2 (k-steps: even / odd), 4 x 4 tiles each;
//   mode C (  1.. 64 columns): 4 waves on k-steps s = wave (mod 4), 8 x 2 tiles each
// (a narrow block computed 2 x 2 like a wide one issued 4 or 8 MFMAs per k-step and was bound by the per-sample streaming
// work: ~700 cycles per k-step instead of 256 / 512).  Everything that is not an MFMA -- the 8..10 operand reads of the next
// k-step, the DMA pieces, the bias-sum adds -- is placed BETWEEN the MFMAs of a k-step, one item per MFMA, so it executes in
// the 64-cycle shadow of the matrix pipe (issued in a burst between the MFMA groups it cost ~250 of 1274 cycles per k-step).
// At the end a workgroup adds its block to dW with fp32 atomics (two 128-byte row segments per instruction: the full-rate
// shape) -- one partial per workgroup.  All layers share the launch, so every CU is busy from start to end; workgroups are
// dealt to (layer, block) in proportion to the block's chunk time.
// Bias gradients ride along: the A fragments ARE dY, so a wave adds them up and adds the sums to db.
// Rows / columns of a tile that do not exist (n_out < 256, a block's last partial tile) are NOT masked: an MFMA output
// element depends only on its own row of A and column of B, so whatever the panel holds there only reaches accumulators that
// the flush skips.  Sample rows beyond M (the last chunk) do reach real outputs: they are zeroed in LDS once the chunk landed.
#include "fgs_common.h"
#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Cache policy of the panel loads: 2 = non-temporal.  The activations and their gradients (826 MB per launch) are read exactly
// once; streamed through the caches with the default policy they evict what the scatter kernels running beside this launch work
// on (sdf.grad's 16 MB of atomics, k0's touched voxels).  Same box, alternating builds: 1.840 / 1.844 ms per step against
// 1.849 / 1.849 with the default policy (0), the launch itself 463 us against 466.
#ifndef FGS_WG_DMA_CPOL
#define FGS_WG_DMA_CPOL 2
#endif
constexpr int WG_DMA_CPOL = FGS_WG_DMA_CPOL;
constexpr int WG_THREADS = 256;
constexpr int WG_MAXBLK = 24;       // (layer, column block) pairs per launch

struct WgBlock {
  const float *dY;
  const float *X;
  float *dW;
  float *dbias;          // null unless this is column block 0 of a layer with a bias gradient
  int64_t ld_dy, ld_x, ld_dw;
  int n_out, n_in;       // valid rows / columns of dW
  int col0, cols, mode;  // first column and width of the block; wave arrangement (WgMode)
  int wg0, n_wg;         // workgroups [wg0, wg0 + n_wg) split the samples of this block
  float stagger;         // a of the share function (wgrad_block)
  // store mode (fgs_mlp_wgrad_ws): every (workgroup, k-split wave group) writes its partial block with plain stores into its own
  // slice [n_out][cols] of `slab`, and k_wgrad_reduce adds the slices in order -- no float atomics (64 MB of them per launch at
  // the fine stage: ~50 us at the chip's atomic rate), and weight gradients that are bit-reproducible
  float *slab;           // null: the partials are added to dW with atomics
  int n_slices;          // n_wg x KS of the block's mode
};

struct WgArgs {
  int64_t M;
  const int64_t *m_dev;
  FgsStamps stamps;               // fgs_dyn_t.stamps / fgs_mlp_wgrad_debug_stamps: 8 words per workgroup
  int n_blocks;
  WgBlock B[WG_MAXBLK];
};

constexpr int WG_ROWS = 16;                       // samples per chunk = 8 k-steps
constexpr int WG_NS = 4;                          // ring slots
constexpr int WG_PA = WG_ROWS * 256;              // floats of the A panel
constexpr int WG_LDS_FLOATS = WG_NS * 2 * WG_PA;     // mode A slots (the largest): 128 KB -- with the 32 000 bytes of one
                                                     // k_feat_taps_bwd workgroup (it runs beside this kernel) exactly a CU's 160 KB

// Wave arrangements: RT x CT MFMA tiles per wave, KS = how many waves share a tile position and split a chunk's k-steps,
// PWB = floats per row of the compact B panel.  Waves = (4 / KS) positions x KS; positions are NWR row parts x NWC column
// parts (NWC = 2 only for KS = 1).  Rows covered = NWR * RT * 32, columns = NWC * CT * 32.
//   n_out <= 256:  0: 2x2 pos, 4x4 tiles (cols <= 256)   1: 2x1 pos x 2 k, 4x4 (cols <= 128)   2: 1 pos x 4 k, 8x2 (cols <= 64)
//   n_out <= 192:  3: 2x2 pos, 3x3 tiles (cols <= 192)   4: 2x1 pos x 2 k, 3x4 (cols <= 128)   5: 1 pos x 4 k, 6x2 (cols <= 64)
//   n_out <= 128:  6: 1 pos x 4 k, 4x4 tiles (cols <= 128)
// (the coarse stages' 192- and 128-wide layers ran as mode 0 / 1 before: 56 % / 25 % of their MFMAs on rows and columns
// that exist)
template <int MODE> struct WgMode;
template <> struct WgMode<0> { static constexpr int RT = 4, CT = 4, KS = 1, PWB = 256; };
template <> struct WgMode<1> { static constexpr int RT = 4, CT = 4, KS = 2, PWB = 128; };
template <> struct WgMode<2> { static constexpr int RT = 8, CT = 2, KS = 4, PWB = 64; };
template <> struct WgMode<3> { static constexpr int RT = 3, CT = 3, KS = 1, PWB = 192; };
template <> struct WgMode<4> { static constexpr int RT = 3, CT = 4, KS = 2, PWB = 128; };
template <> struct WgMode<5> { static constexpr int RT = 6, CT = 2, KS = 4, PWB = 64; };
template <> struct WgMode<6> { static constexpr int RT = 4, CT = 4, KS = 4, PWB = 128; };

// The chunk whose panels are being fetched: uniform base pointers + per-lane byte offsets, clamped per lane to the tensor's
// last 16 bytes (lanes of a row that straddles the end re-read the last float4 into places nobody uses or that get zeroed).
struct WgDma {
  const char *base_a, *base_b;    // dY / X at the chunk's first sample row
  unsigned lim_a, lim_b;          // largest byte offset from base that keeps a 16-byte load inside the tensor
  float *dst;                     // slot base (no chunk left: the pieces re-read the tensors' first rows into the slot the
                                  // chunk would have used -- free by the ring's invariant, read by nobody)
};

template <int MODE, int I>
__device__ __forceinline__ void wg_dma_piece(const WgDma &d, unsigned va0, unsigned col0_4, unsigned step_a, unsigned ld_b4,
                                             int wave, int lane) {
  using Md = WgMode<MODE>;
  constexpr int PWB = Md::PWB;
  constexpr int NB = PWB / 64;                // B pieces per wave (A: 4 -- one sample row of 256 floats each)
  static_assert(I < 4 + NB, "piece index");
  const char *src;
  float *dst;
  if (I < 4) {
    unsigned off = va0 + I * step_a;
    off = off < d.lim_a ? off : d.lim_a;
    src = d.base_a + off;
    dst = d.dst + (wave + 4 * I) * 256;
  } else {
    // B piece p = wave + 4 (I - 4): floats [256 p, 256 p + 256) of the compact [16][PWB] panel; this lane's float4 starts at
    // flat index f -> panel row f / PWB, panel column f % PWB (PWB is a compile-time constant: a multiply and a shift)
    const unsigned f = (unsigned)(wave + 4 * (I - 4)) * 256u + (unsigned)lane * 4u;
    const unsigned r = f / (unsigned)PWB, c = f - r * (unsigned)PWB;
    unsigned off = r * ld_b4 + col0_4 + c * 4u;
    off = off < d.lim_b ? off : d.lim_b;
    src = d.base_b + off;
    dst = d.dst + WG_PA + (wave + 4 * (I - 4)) * 256;
  }
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)dst, 16, 0, WG_DMA_CPOL);
}

template <int MODE, int I, int N>
struct WgPieces {       // pieces [I, N) of the pending chunk, back to back (prologue)
  static __device__ __forceinline__ void run(const WgDma &d, unsigned va0, unsigned col0_4, unsigned sa, unsigned ldb4, int wave,
                                             int lane) {
    wg_dma_piece<MODE, I>(d, va0, col0_4, sa, ldb4, wave, lane);
    WgPieces<MODE, I + 1, N>::run(d, va0, col0_4, sa, ldb4, wave, lane);
  }
};
template <int MODE, int N>
struct WgPieces<MODE, N, N> {
  static __device__ __forceinline__ void run(const WgDma &, unsigned, unsigned, unsigned, unsigned, int, int) {}
};

template <int MODE>
__device__ __forceinline__ void wgrad_block(const WgBlock &b, int64_t M, int j_in_block, float *lds,
                                            unsigned long long *stamps) {
  using Md = WgMode<MODE>;
  constexpr int RT = Md::RT, CT = Md::CT, KS = Md::KS, PWB = Md::PWB;
  constexpr int KPC = 8 / KS;                       // k-steps of a chunk that this wave computes
  constexpr int NP = 4 + PWB / 64;                  // DMA pieces per wave and chunk
  constexpr int PPK = (NP + KPC / 2 - 1) / (KPC / 2);   // pieces per k-step of a chunk's second half
  constexpr int SLOT = WG_PA + WG_ROWS * PWB;       // floats per slot
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  // position of the wave: row half wr, column half wc, k-step residue kg
  constexpr int NWC = KS == 1 ? 2 : 1, NPOS = 4 / KS, NM = RT * CT;      // column parts, tile positions, MFMAs per k-step
  const int pos = wave % NPOS, kg = wave / NPOS;
  const int wr = pos / NWC, wc = pos % NWC;
  const int row_base = wr * RT * 32, colp_base = wc * CT * 32;     // first row of the wave / first panel column
  // chunk range of this workgroup
  // (chunk indices and the row count as 32-bit SCALARS: M < 2^31 is checked by the launcher.  As 64-bit values -- the shares
  // below come out of floating-point arithmetic, i.e. out of vector registers -- every loop test was a vector compare whose
  // result the scalar branch had to wait for behind the MFMA in the pipe)
  const int Mi = __builtin_amdgcn_readfirstlane((int)M);
  const int NC = (Mi + WG_ROWS - 1) / WG_ROWS;
  // Staggered shares: workgroup j of n takes the chunks [NC F(j / n), NC F((j + 1) / n)), F(x) = (1 - a) x + a x^2 -- the
  // first a few per cent fewer than the last.  With equal shares all 256 workgroups reach the flush together and the memory-
  // side atomic units, idle until then, become the bottleneck for the last ~40 us (64 MB of fp32 atomics); staggered, the
  // early finishers' atomics run under the matrix work of the others.
  const double xa = (double)j_in_block / b.n_wg, xb = (double)(j_in_block + 1) / b.n_wg;
  const double sa = b.stagger;
  const int c0 = __builtin_amdgcn_readfirstlane(j_in_block == 0 ? 0 : (int)((double)NC * ((1.0 - sa) * xa + sa * xa * xa)));
  const int c1 = __builtin_amdgcn_readfirstlane(j_in_block + 1 == b.n_wg ? NC : (int)((double)NC * ((1.0 - sa) * xb + sa * xb * xb)));
  if (c0 >= c1) {
    if (b.slab) {        // (store mode: the slices of a workgroup without samples must read as zero)
      float *sl = b.slab + (int64_t)j_in_block * KS * b.n_out * b.cols;
      for (int64_t i = tid; i < (int64_t)KS * b.n_out * b.cols; i += WG_THREADS) sl[i] = 0.f;
    }
    return;
  }
  const unsigned ld_a4 = (unsigned)b.ld_dy * 4, ld_b4 = (unsigned)b.ld_x * 4;      // row pitch in bytes

  // ---- DMA: per-lane source offsets inside a chunk (bytes)
  //   A piece i: sample row wave + 4 i, 256 floats from its start:   lane * 16
  //   B piece i: 256 consecutive floats of the compact [16][PWB] panel (wg_dma_piece works out row and column per lane)
  const unsigned va0 = wave * ld_a4 + lane * 16;
  const unsigned col0_4 = (unsigned)b.col0 * 4u;
  const unsigned step_a = 4 * ld_a4;
  // (limits from 32-bit row counts: the scalar unit has no 64-bit signed compare, so `bytes - 16 - offset < 2^31` on uniform
  // 64-bit values became vector compares inside the loop.  A limit only has to cover the 16 rows of a chunk: rows are capped.)
  constexpr int ROW_CAP = 1 << 16;                 // x row pitch (<= 1280 bytes) stays below 2^31
  WgDma d;
  int issue_c = c0;
  int issue_slot = 0;
  auto dma_begin = [&]() __attribute__((always_inline)) {
    const int first = issue_c < c1 ? issue_c * WG_ROWS : 0;           // first sample row the pieces address
    int rows = Mi - first;                                           // >= 1: that row exists
    rows = rows < ROW_CAP ? rows : ROW_CAP;
    d.base_a = (const char *)b.dY + (int64_t)first * ld_a4;
    d.base_b = (const char *)b.X + (int64_t)first * ld_b4;
    d.lim_a = (unsigned)rows * ld_a4 - 16u;
    d.lim_b = (unsigned)rows * ld_b4 - 16u;
    d.dst = lds + issue_slot * SLOT;
    ++issue_c;
    issue_slot = issue_slot + 1 == WG_NS ? 0 : issue_slot + 1;
  };
  // rows of chunk c beyond M are zeroed in its slot (after it landed, before anybody reads it); uniform per workgroup
  auto zero_tail = [&](int c, int slot_of_c) __attribute__((always_inline)) {
    const int rows_left = Mi - c * WG_ROWS;
    if (rows_left >= WG_ROWS) return;
    float *S = lds + slot_of_c * SLOT;
    for (int i = rows_left * 256 + tid; i < WG_ROWS * 256; i += WG_THREADS) S[i] = 0.f;
    for (int i = rows_left * PWB + tid; i < WG_ROWS * PWB; i += WG_THREADS) S[WG_PA + i] = 0.f;
    __syncthreads();
  };

  floatx16 acc[RT][CT];
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int tb = 0; tb < CT; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ta][tb][r] = 0.f;
  float bsum[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) bsum[t] = 0.f;

  // ---- prologue: chunks c0 .. c0 + 2 in flight, c0 complete (and its tail zeroed) before the first read
#pragma unroll
  for (int k = 0; k < WG_NS - 1; ++k) {
    dma_begin();
    WgPieces<MODE, 0, NP>::run(d, va0, col0_4, step_a, ld_b4, wave, lane);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
  __builtin_amdgcn_s_barrier();
  zero_tail(c0, 0);
  if (stamps && tid == 0) { stamps[2] = __builtin_amdgcn_s_memtime(); stamps[6] = (unsigned long long)(c1 - c0); }

  // this lane's operand addresses inside a slot (floats): row 2 kg + h of the chunk, first tile of the wave
  const int offA = (2 * kg + h) * 256 + row_base + l31;
  const int offB = WG_PA + (2 * kg + h) * PWB + colp_base + l31;
  int slot = 0;
  float fa[2][RT], fb[2][CT];     // operands of the current / next k-step
  {
    const float *S = lds;
#pragma unroll
    for (int t = 0; t < RT; ++t) fa[0][t] = S[offA + 32 * t];
#pragma unroll
    for (int t = 0; t < CT; ++t) fb[0][t] = S[offB + 32 * t];
  }
  for (int c = c0; c < c1; ++c) {
    const float *S = lds + slot * SLOT;
    const int slot_next = slot + 1 == WG_NS ? 0 : slot + 1;
    const float *S_next = lds + slot_next * SLOT;
#pragma unroll
    for (int sl = 0; sl < KPC; ++sl) {
      constexpr int dummy = 0; (void)dummy;
      const int cur = sl & 1, nxt = cur ^ 1;
      // operands of the next k-step of this wave: same chunk, 2 KS rows further down -- or the first of the next chunk
      const float *Sn = sl + 1 < KPC ? S + (sl + 1) * 2 * KS * 256 : S_next;
      const float *SnB = sl + 1 < KPC ? S + (sl + 1) * 2 * KS * PWB : S_next;
#pragma unroll
      for (int ta = 0; ta < RT; ++ta) {
#pragma unroll
        for (int tb = 0; tb < CT; ++tb) {
          const int i = ta * CT + tb;               // MFMA i of the k-step; everything below runs in its shadow
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][ta], fb[cur][tb], acc[ta][tb], 0, 0, 0);
          if (i < RT) fa[nxt][i] = Sn[offA + 32 * i];
          else if (i < RT + CT) fb[nxt][i - RT] = SnB[offB + 32 * (i - RT)];
          if (tb == CT - 1) bsum[ta] += fa[cur][ta];
          if (sl >= KPC / 2 && i >= NM - PPK) {     // second half of the chunk: the pieces of the chunk three ahead
            const int piece = (sl - KPC / 2) * PPK + (i - (NM - PPK));
            switch (piece) {      // (compile-time after unrolling)
              case 0: wg_dma_piece<MODE, 0>(d, va0, col0_4, step_a, ld_b4, wave, lane); break;
              case 1: wg_dma_piece<MODE, 1>(d, va0, col0_4, step_a, ld_b4, wave, lane); break;
              case 2: wg_dma_piece<MODE, 2>(d, va0, col0_4, step_a, ld_b4, wave, lane); break;
              case 3: wg_dma_piece<MODE, 3>(d, va0, col0_4, step_a, ld_b4, wave, lane); break;
              case 4: wg_dma_piece<MODE, 4>(d, va0, col0_4, step_a, ld_b4, wave, lane); break;
              case 5: if (NP > 5) wg_dma_piece<MODE, (NP > 5 ? 5 : 0)>(d, va0, col0_4, step_a, ld_b4, wave, lane); break;
              case 6: if (NP > 6) wg_dma_piece<MODE, (NP > 6 ? 6 : 0)>(d, va0, col0_4, step_a, ld_b4, wave, lane); break;
              case 7: if (NP > 7) wg_dma_piece<MODE, (NP > 7 ? 7 : 0)>(d, va0, col0_4, step_a, ld_b4, wave, lane); break;
              default: break;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (sl == KPC / 2 - 1) {
        // middle of the chunk: the next chunk (its DMA was issued 1.5 chunks ago; only the newest chunk's NP pieces may still
        // be in flight) is complete for everybody past this barrier, and everybody has left the previous chunk, whose slot
        // the chunk three ahead now overwrites
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        __builtin_amdgcn_s_barrier();
        if (c + 1 < c1) zero_tail(c + 1, slot_next);
        dma_begin();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    slot = slot_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stamps && tid == 0) stamps[3] = __builtin_amdgcn_s_memtime();
  // ---- flush: dW block and bias sums, fp32 atomics (two 128-byte row segments per instruction) -- or, store mode, plain
  // stores of the same shape into this wave group's slice
  float *const slice = b.slab ? b.slab + ((int64_t)j_in_block * KS + kg) * b.n_out * b.cols : nullptr;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta) {
#pragma unroll
    for (int tb = 0; tb < CT; ++tb) {
      const int pc = colp_base + 32 * tb + l31;          // column inside the block
      const int col = b.col0 + pc;
      if (col < b.n_in) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = row_base + 32 * ta + 8 * (r >> 2) + 4 * h + (r & 3);
          if (n < b.n_out) {
            if (slice) slice[(int64_t)n * b.cols + pc] = acc[ta][tb][r];
            else atomicAdd(b.dW + (int64_t)n * b.ld_dw + col, acc[ta][tb][r]);
          }
        }
      }
    }
    const int n = row_base + 32 * ta + l31;
    if (b.dbias && wc == 0 && n < b.n_out) atomicAdd(b.dbias + n, bsum[ta]);
  }
}

__global__ __launch_bounds__(WG_THREADS, 1) void k_mlp_wgrad(WgArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[WG_LDS_FLOATS];
  const int64_t M = fgs_rows(a.M, a.m_dev);
  if (M <= 0) return;
  int blk = 0;
  while (blk + 1 < a.n_blocks && (int)blockIdx.x >= a.B[blk + 1].wg0) ++blk;
  const WgBlock &b = a.B[blk];
  const int j = (int)blockIdx.x - b.wg0;
  if (j >= b.n_wg) return;
  unsigned long long *stamps = fgs_stamp_wg(a.stamps);
  if (stamps && threadIdx.x == 0) {
    stamps[0] = __builtin_amdgcn_s_memtime(); stamps[1] = __builtin_amdgcn_s_memrealtime(); stamps[7] = (unsigned long long)blk;
  }
  switch (b.mode) {
    case 6: wgrad_block<6>(b, M, j, lds, stamps); break;
    case 5: wgrad_block<5>(b, M, j, lds, stamps); break;
    case 4: wgrad_block<4>(b, M, j, lds, stamps); break;
    case 3: wgrad_block<3>(b, M, j, lds, stamps); break;
    case 2: wgrad_block<2>(b, M, j, lds, stamps); break;
    case 1: wgrad_block<1>(b, M, j, lds, stamps); break;
    default: wgrad_block<0>(b, M, j, lds, stamps); break;
  }
  if (stamps) {                           // (every lane's flush atomics performed, not merely issued, before the end stamp)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) { stamps[4] = __builtin_amdgcn_s_memtime(); stamps[5] = __builtin_amdgcn_s_memrealtime(); }
  }
}

// Store mode, second launch: dW[n][col0 + c] += slice_0[n][c] + slice_1[n][c] + ... in slice order.  A slice is a flat array of
// n_out x cols floats (a multiple of 4, 16-byte aligned): one thread per float4 of it, eight 16-byte loads in flight, so that a
// workgroup reads 4 KB of every slice in one piece (one thread per float measured 40 us beside the scatter kernels; 256-byte pieces
// per wave 107 us: the slices lie 100..256 KB apart and DRAM pages want longer runs).  blockIdx.y = block.
__global__ __launch_bounds__(FGS_BLOCK) void k_wgrad_reduce(WgArgs a) {
  const WgBlock &b = a.B[blockIdx.y];
  const int per = b.n_out * b.cols, per4 = per >> 2;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x * blockDim.x >= per4 || !b.slab) return;
  if (q < per4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 s = {0.f, 0.f, 0.f, 0.f};
    const f4 *p = reinterpret_cast<const f4 *>(b.slab) + q;
    int k = 0;
    for (; k + 8 <= b.n_slices; k += 8) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + (int64_t)(k + u) * per4);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < b.n_slices; ++k) s += __builtin_nontemporal_load(p + (int64_t)k * per4);
    const float sv[4] = {s[0], s[1], s[2], s[3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = 4 * q + e, n = i / b.cols, c = i - n * b.cols;
      b.dW[(int64_t)n * b.ld_dw + b.col0 + c] += sv[e];
    }
  }
  // (measurement: the launch pair counts as ONE weight-gradient launch -- every workgroup of the reduction pushes the end reading of
  // the first record of fgs_dyn_t.stamps, which bench.py takes the maximum of, to its own end)
  unsigned long long *st = fgs_stamp_base(a.stamps);
  if (st && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicMax(st + 5, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
}

// M too small for 1 KB DMA pieces to stay inside the tensors: one thread per dW element
__global__ __launch_bounds__(FGS_BLOCK) void k_wgrad_small(WgArgs a) {
  const int64_t M = fgs_rows(a.M, a.m_dev);
  const WgBlock &b = a.B[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int cols = b.cols;
  if (i >= b.n_out * cols) return;
  const int n = i / cols, k = b.col0 + i % cols;
  float s = 0.f, sb = 0.f;
  for (int64_t m = 0; m < M; ++m) {
    const float g = b.dY[m * b.ld_dy + n];
    s = fmaf(g, b.X[m * b.ld_x + k], s);
    sb += g;
  }
  atomicAdd(b.dW + (int64_t)n * b.ld_dw + k, s);
  if (b.dbias && k == b.col0) atomicAdd(b.dbias + n, sb);
}

unsigned long long *g_wg_stamps = nullptr;

bool wg_aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }
bool wg_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Diagnostics: while a device buffer of >= 8 * 256 uint64 is set, workgroup w records into stamps[8 w ..] the shader clock at
// its start [0], after the prologue [2], after the sample loop [3] and after issuing the flush [4], the 100 MHz wall clock at
// start [1] and end [5], its number of chunks [6] and its block [7].  NULL switches it off.
FGS_API int fgs_mlp_wgrad_debug_stamps(unsigned long long *stamps) {
  g_wg_stamps = stamps;
  return 0;
}

namespace {
int wgrad_launch(int64_t M, int n_items, const fgs_wgrad_item_t *items, float *ws, int64_t ws_floats, const fgs_dyn_t *dyn,
                 fgs_stream_t stream);
}

FGS_API int fgs_mlp_wgrad(int64_t M, int n_items, const fgs_wgrad_item_t *items, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  return wgrad_launch(M, n_items, items, nullptr, 0, dyn, stream);
}

// Upper bound of the workspace fgs_mlp_wgrad_ws needs (floats), whatever the shapes: one 256 x 256 partial per workgroup.
FGS_API int64_t fgs_mlp_wgrad_ws_floats(void) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  return (int64_t)(cus + WG_MAXBLK) * 65536;
}

// The same products with the partials written by plain stores into `ws` and summed in a fixed order by a second launch: no float
// atomics (weight gradients bit-reproducible; bias sums keep theirs).  ws too small for the shapes at hand, or NULL: the atomic form.
FGS_API int fgs_mlp_wgrad_ws(int64_t M, int n_items, const fgs_wgrad_item_t *items, float *ws, int64_t ws_floats,
                             const fgs_dyn_t *dyn, fgs_stream_t stream) {
  return wgrad_launch(M, n_items, items, ws, ws_floats, dyn, stream);
}

namespace {
int wgrad_launch(int64_t M, int n_items, const fgs_wgrad_item_t *items, float *ws, int64_t ws_floats, const fgs_dyn_t *dyn,
                 fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && n_items >= 1 && n_items <= WG_MAXBLK, FGS_E_RANGE,
              "fgs_mlp_wgrad: M=%lld n_items=%d (1..%d)", (long long)M, n_items, WG_MAXBLK);
  if (M == 0) return 0;
  FGS_REQUIRE(items, FGS_E_INVALID, "fgs_mlp_wgrad: null pointer");
  WgArgs a;
  a.M = M; a.m_dev = fgs_dyn_rows(dyn); a.stamps = fgs_dyn_stamps(dyn, g_wg_stamps);
  int nb = 0, cost_total = 0;
  int cost[WG_MAXBLK];
  for (int i = 0; i < n_items; ++i) {
    const fgs_wgrad_item_t &U = items[i];
    FGS_REQUIRE(U.dY && U.X && U.dW && wg_aligned16(U.dY) && wg_aligned16(U.X) && wg_aligned4(U.dW) && U.n_out > 0 &&
                    U.n_out <= 256 && U.n_in > 0 && U.n_in <= 320 && U.ld_dy >= U.n_out && U.ld_x >= U.n_in &&
                    U.ld_dw >= U.n_in && U.ld_dy <= 320 && U.ld_x <= 320 && (U.ld_dy % 4) == 0 && (U.ld_x % 4) == 0,
                FGS_E_INVALID, "fgs_mlp_wgrad: item %d: bad pointer or shape (n_out <= 256, n_in <= 320, leading dimensions "
                               "of dY / X multiples of 4 and <= 320, 16-byte aligned)", i);
    // column blocks: 256 wide for layers of more than 192 rows, 192 wide for the others (their waves hold 3 x 3 tiles)
    const int blk_cols = U.n_out > 192 ? 256 : (U.n_out > 128 ? 192 : 128);
    for (int col0 = 0; col0 < U.n_in; col0 += blk_cols) {
      FGS_REQUIRE(nb < WG_MAXBLK, FGS_E_RANGE, "fgs_mlp_wgrad: more than %d column blocks", WG_MAXBLK);
      const int cols = U.n_in - col0 < blk_cols ? U.n_in - col0 : blk_cols;
      WgBlock &b = a.B[nb];
      b.dY = U.dY; b.X = U.X; b.dW = U.dW; b.dbias = col0 == 0 ? U.dbias : nullptr;
      b.ld_dy = U.ld_dy; b.ld_x = U.ld_x; b.ld_dw = U.ld_dw; b.n_out = U.n_out; b.n_in = U.n_in; b.col0 = col0; b.cols = cols;
      // cost = time per chunk of 16 samples: k-steps of the wave x MFMAs per k-step x 64 cycles + the chunk's fixed work
      if (U.n_out > 192) {
        b.mode = cols <= 64 ? 2 : cols <= 128 ? 1 : 0;
        cost[nb] = b.mode == 0 ? 33 : b.mode == 1 ? 18 : 10;
      } else if (U.n_out > 128) {
        b.mode = cols <= 64 ? 5 : cols <= 128 ? 4 : 3;
        cost[nb] = b.mode == 3 ? 20 : b.mode == 4 ? 14 : 8;
      } else {
        b.mode = 6;
        cost[nb] = 10;
      }
      cost_total += cost[nb];
      ++nb;
    }
  }
  a.n_blocks = nb;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  // FGS_WGRAD_CUS: fewer workgroups than CUs (shares re-dealt over them) leaves whole CUs to the kernels of the other graph
  // branch that run beside this launch (fused.py _wgrad); measured values in DESIGN.md section 3
  static const int cus_env = fgs_env_int("FGS_WGRAD_CUS", 0);
  if (cus_env > 0 && cus_env < cus) cus = cus_env;
  // one workgroup per CU (256 accumulator registers per lane); blocks get workgroups in proportion to their MFMA count,
  // at least one each, never more than one per 64 samples
  const int64_t max_per_block = (M + 63) / 64;
  static const float stagger = getenv("FGS_WGRAD_STAGGER") ? (float)atof(getenv("FGS_WGRAD_STAGGER")) : 0.04f;
  // (measured at M = 58 430: a = 0: 441 us, 0.03: 436, 0.06: 435, 0.10: 444, 0.15: 461 -- scripts/diag/wg_stagger.sh)
  int wg = 0, given = 0, cost_seen = 0;
  for (int i = 0; i < nb; ++i) {
    cost_seen += cost[i];
    int n = (int)((int64_t)cus * cost_seen / cost_total) - given;    // cumulative rounding: the shares add up to cus
    if (n < 1) n = 1;
    if (n > max_per_block) n = (int)max_per_block;
    given += n;
    a.B[i].wg0 = wg; a.B[i].n_wg = n; a.B[i].stagger = stagger;
    a.B[i].slab = nullptr; a.B[i].n_slices = 0;
    wg += n;
  }
  // store mode: slices of every block, back to back
  bool store_mode = false;
  bool rows4 = true;
  for (int i = 0; i < nb; ++i) rows4 = rows4 && (a.B[i].n_out * a.B[i].cols) % 4 == 0;
  if (ws && rows4 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && !(M < 64 && !a.m_dev)) {
    static const int ks_of_mode[7] = {1, 2, 4, 1, 2, 4, 4};
    int64_t need = 0;
    for (int i = 0; i < nb; ++i) need += (int64_t)a.B[i].n_wg * ks_of_mode[a.B[i].mode] * a.B[i].n_out * a.B[i].cols;
    if (need <= ws_floats) {
      int64_t off = 0;
      for (int i = 0; i < nb; ++i) {
        a.B[i].slab = ws + off;
        a.B[i].n_slices = a.B[i].n_wg * ks_of_mode[a.B[i].mode];
        off += (int64_t)a.B[i].n_slices * a.B[i].n_out * a.B[i].cols;
      }
      store_mode = true;
    }
  }
  if (M < 64 && !a.m_dev) {      // tiny batches: not worth a 256-register workgroup per block
    hipLaunchKernelGGL(k_wgrad_small, dim3(fgs_blocks(256 * 256), (unsigned)nb), dim3(FGS_BLOCK), 0, fgs_s(stream), a);
    FGS_LAUNCH_OK("fgs_mlp_wgrad (small)");
    return 0;
  }
  hipLaunchKernelGGL(k_mlp_wgrad, dim3((unsigned)wg), dim3(WG_THREADS), 0, fgs_s(stream), a);
  FGS_LAUNCH_OK("fgs_mlp_wgrad");
  if (store_mode) {
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(256 * 256 / 4 / FGS_BLOCK, (unsigned)nb), dim3(FGS_BLOCK), 0, fgs_s(stream), a);
    FGS_LAUNCH_OK("fgs_mlp_wgrad (reduce)");
  }
  return 0;
}
}  // namespace
