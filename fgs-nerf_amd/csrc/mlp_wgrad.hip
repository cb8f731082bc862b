// mlp_wgrad.hip -- every weight (and bias) gradient of the tiny MLPs in ONE launch:  dW_l += dY_l^T X_l,  db_l += colsum(dY_l)
// for all layers l of rgbnet + refnet (model/nerf.py:125-142; the products autograd runs as seven separate addmm calls).
//
// Shape of the problem: outputs are tiny (256 x 108..308), the reduction runs over the M ~ 50-60 K surviving samples.  So the
// SAMPLES are split over the chip: a workgroup takes one (layer, block of <= 256 weight columns) and a contiguous range of
// samples, keeps the whole 256 x 256 output block in its accumulators (4 waves as 2 x 2, 128 x 128 = 16 MFMA tiles = 256
// accumulator registers per wave) and streams its sample range straight from HBM/L2 into MFMA operands:
//   A operand (32x32x2: lane (j, h) holds A[i = j][k = h])  = dY[sample 2s + h][out-feature tile + j]
//   B operand                                               = X [sample 2s + h][in-feature tile + j]
// -- both are 128-byte row segments of the row-major activations, so no transpose is needed anywhere: panels of 16 sample
// rows are copied by LDS-DMA exactly as they lie in memory (a contiguous byte range per panel, three slots in rotation, one
// barrier per 16 samples, DMA issued piece by piece between the MFMAs) and the operands are conflict-free ds_read_b32.
// (Fetching the operands straight from global memory into registers, four k-steps ahead, ran at 45 TFLOP/s: hipcc's in-order
// vmcnt waits collapse the prefetch.)  At the end a workgroup adds its block to dW with fp32 atomics (two 128-byte
// row segments per instruction: the full-rate shape) -- one partial per CU, 256 KB, instead of one per 128 x 128 tile and
// K slice.  All layers share the launch, so every CU is busy from start to end and the per-layer launch gaps are gone;
// workgroups are dealt to (layer, block) in proportion to the block's MFMA count.
// Bias gradients ride along: the A fragments ARE dY, so a wave adds them up (4 VALU adds per 16 MFMAs) and the column-block-0
// workgroups add the sums to db.
#include "fgs_common.h"

#include <type_traits>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int WG_THREADS = 256;
constexpr int WG_MAXBLK = 24;       // (layer, column block) pairs per launch

struct WgBlock {
  const float *dY;
  const float *X;
  float *dW;
  float *dbias;          // null unless this is column block 0 of a layer with a bias gradient
  int64_t ld_dy, ld_x, ld_dw;
  int n_out, n_in;       // valid rows / columns of dW
  int col0, nctw;        // first column of the block; column tiles per wave (1, 2 or 4; the block spans 2 * nctw tiles)
  int wg0, n_wg;         // workgroups [wg0, wg0 + n_wg) split the samples of this block
};

struct WgArgs {
  int64_t M;
  const int64_t *m_dev;
  int n_blocks;
  WgBlock B[WG_MAXBLK];
};

// LDS: 3 slots x {A panel, B panel}; a panel = 16 consecutive sample rows of dY / X exactly as they lie in memory (pitch =
// the matrix's leading dimension <= 320 floats), copied by LDS-DMA as a contiguous byte range in 1 KB pieces.
constexpr int WG_ROWS = 16;                       // samples per chunk = 8 k-steps
constexpr int WG_PANEL = WG_ROWS * 320;           // floats per panel (20 KB)
constexpr int WG_SLOT = 2 * WG_PANEL;
constexpr int WG_SLOTS = 3;

struct WgDma {
  const float *src_a, *src_b;     // this lane's source of piece 0 of the chunk being fetched (A panel, B panel)
  float *dst;                     // slot base
  int p, np_a, np_b;              // next piece of this wave; pieces per panel
  int64_t lim_a, lim_b;           // float offset of the tensors' last float4 (keeps every lane's copy inside them)
  int64_t off_a, off_b;           // float offset of the chunk in dY / X
};

// one 1 KB piece of the chunk being fetched (branch-free; a wave without a piece left re-fetches piece 0 into the dump area)
__device__ __forceinline__ void wg_dma_piece(WgDma &d, const float *dY, const float *X, float *dump, int lane) {
  const int np = d.np_a + d.np_b;
  const bool ok = d.p < np;
  const int p = ok ? d.p : 0;
  const bool is_b = p >= d.np_a;
  const int q = is_b ? p - d.np_a : p;
  // per-LANE clamp to the tensor's last 16 bytes: the lanes of a piece that straddles the end still bring the rows that
  // exist to their places; the others re-read the last float4 into rows beyond M, which the MFMA loop zeroes
  int64_t off = (is_b ? d.off_b : d.off_a) + (int64_t)q * 256 + lane * 4;
  const int64_t lim = is_b ? d.lim_b : d.lim_a;
  off = off < lim ? off : lim;
  const float *src = (is_b ? X : dY) + off;
  float *dst = ok ? d.dst + (is_b ? WG_PANEL : 0) + q * 256 : dump;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  d.p += 4;
}

template <int NCTW>
__device__ __forceinline__ void wgrad_block(const WgBlock &b, int64_t M, int j_in_block, float *lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  // chunk range of this workgroup (chunks of 16 samples: every panel starts 16-byte aligned)
  const int64_t NC = (M + WG_ROWS - 1) / WG_ROWS;
  const int64_t c0 = NC * j_in_block / b.n_wg, c1 = NC * (j_in_block + 1) / b.n_wg;
  if (c0 >= c1) return;
  float *dump = lds + WG_SLOTS * WG_SLOT + wave * 256;
  const int ld_a = (int)b.ld_dy, ld_b = (int)b.ld_x;

  int offA[4], offB[NCTW];          // float offset of this lane's operand inside a panel row pair (row h)
  bool okA[4], okB[NCTW];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = 128 * wr + 32 * t + l31;
    okA[t] = n < b.n_out;
    offA[t] = h * ld_a + (okA[t] ? n : 0);
  }
#pragma unroll
  for (int t = 0; t < NCTW; ++t) {
    const int c = b.col0 + 32 * (NCTW * wc + t) + l31;
    okB[t] = c < b.n_in;
    offB[t] = WG_PANEL + h * ld_b + (okB[t] ? c : 0);
  }
  floatx16 acc[4][NCTW];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < NCTW; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ta][tb][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  WgDma d;
  d.np_a = (WG_ROWS * ld_a + 255) / 256; d.np_b = (WG_ROWS * ld_b + 255) / 256;
  d.lim_a = M * b.ld_dy - 4; d.lim_b = M * b.ld_x - 4;
  int64_t issue_c = c0;
  int issue_slot = 0;
  auto dma_begin = [&]() __attribute__((always_inline)) {
    d.off_a = issue_c * WG_ROWS * b.ld_dy; d.off_b = issue_c * WG_ROWS * b.ld_x;
    d.dst = lds + issue_slot * WG_SLOT;
    d.p = wave;
    ++issue_c;
    issue_slot = issue_slot + 1 == WG_SLOTS ? 0 : issue_slot + 1;
  };
  // prologue: chunks c0 and c0 + 1 in flight, chunk c0 complete before the first read
  dma_begin();
  while (d.p < d.np_a + d.np_b) wg_dma_piece(d, b.dY, b.X, dump, lane);
  if (issue_c < c1) {
    dma_begin();
    while (d.p < d.np_a + d.np_b) wg_dma_piece(d, b.dY, b.X, dump, lane);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  d.p = d.np_a + d.np_b;          // nothing pending

  int slot = 0;
  float fa[2][4], fb[2][NCTW];    // operands of the current / next k-step
  {
    const float *S = lds;
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[0][t] = S[offA[t]];
#pragma unroll
    for (int t = 0; t < NCTW; ++t) fb[0][t] = S[offB[t]];
  }
  // FAST form of a chunk: both leading dimensions are 256 (LDS row offsets become instruction immediates: no address
  // arithmetic per k-step), every row and column of the block exists and the chunk lies fully inside the sample range (no
  // zeroing selects) -- 8 ds_read_b32 and 4 adds beside the 16 MFMAs of a k-step instead of ~40 instructions.  That is the
  // shape of five of the seven fine-stage products; everything else, and the last partial chunk, takes the general form.
  const bool block_fast = NCTW == 4 && ld_a == 256 && ld_b == 256 && b.n_out == 256 && b.col0 + 256 <= b.n_in;
  auto chunk = [&](int64_t c, auto fast_tag) __attribute__((always_inline)) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int lda = FAST ? 256 : ld_a, ldb = FAST ? 256 : ld_b;
    const float *S = lds + slot * WG_SLOT;
    slot = slot + 1 == WG_SLOTS ? 0 : slot + 1;
    const float *S_next = (c + 1 < c1) ? lds + slot * WG_SLOT : nullptr;
    const int64_t rows_left = M - c * WG_ROWS;          // sample rows of this chunk that exist (>= 1)
#pragma unroll
    for (int s = 0; s < WG_ROWS / 2; ++s) {
      // order of a k-step: (1) zeroing selects on THIS k-step's operands -- hipcc waits lgkmcnt(0) for them, and the only
      // LDS reads outstanding at that point are these, issued a whole k-step ago; (2) the reads of the NEXT k-step (of the
      // next chunk for the last one) and this k-step's share of DMA pieces; (3) the 16 MFMAs, under which (2) completes.
      // (With the reads in front of the selects the lgkmcnt(0) waited for the reads just issued: ~500 cycles per k-step.)
      __builtin_amdgcn_sched_barrier(0);
      const bool live = 2 * s + h < rows_left;           // rows beyond M hold whatever the clamped DMA brought: zero them
      float av[4], bv[NCTW];
#pragma unroll
      for (int ta = 0; ta < 4; ++ta) av[ta] = (FAST || (live && okA[ta])) ? fa[s & 1][ta] : 0.f;
#pragma unroll
      for (int tb = 0; tb < NCTW; ++tb) bv[tb] = (FAST || (live && okB[tb])) ? fb[s & 1][tb] : 0.f;
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < WG_ROWS / 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[(s + 1) & 1][t] = S[(2 * s + 2) * lda + offA[t]];
#pragma unroll
        for (int t = 0; t < NCTW; ++t) fb[(s + 1) & 1][t] = S[(2 * s + 2) * ldb + offB[t]];
      } else if (S_next) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[(s + 1) & 1][t] = S_next[offA[t]];
#pragma unroll
        for (int t = 0; t < NCTW; ++t) fb[(s + 1) & 1][t] = S_next[offB[t]];
      }
      if (s >= WG_ROWS / 4) {       // second half of the chunk: the DMA pieces of the chunk after next, a few per k-step
#pragma unroll
        for (int k = 0; k < 3; ++k) wg_dma_piece(d, b.dY, b.X, dump, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ta = 0; ta < 4; ++ta) {
        bsum[ta] += av[ta];
#pragma unroll
        for (int tb = 0; tb < NCTW; ++tb)
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ta], bv[tb], acc[ta][tb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s == WG_ROWS / 4 - 1) {
        // middle of the chunk: the next chunk (DMA issued a chunk ago) is complete for everybody past this barrier, and
        // everybody has left the previous chunk, whose slot the chunk after next now overwrites
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (issue_c < c1) dma_begin();
        else d.p = d.np_a + d.np_b;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  {
    int64_t c = c0;
    if (NCTW == 4 && block_fast) {
      const int64_t c_full = M / WG_ROWS < c1 ? M / WG_ROWS : c1;     // chunks [c0, c_full) have all 16 sample rows
      for (; c < c_full; ++c) chunk(c, std::integral_constant<bool, NCTW == 4>{});
    }
    for (; c < c1; ++c) chunk(c, std::integral_constant<bool, false>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- flush: dW block and bias sums, fp32 atomics (two 128-byte row segments per instruction)
#pragma unroll
  for (int ta = 0; ta < 4; ++ta) {
#pragma unroll
    for (int tb = 0; tb < NCTW; ++tb) {
      if (okB[tb]) {
        const int col = b.col0 + 32 * (NCTW * wc + tb) + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = 128 * wr + 32 * ta + 8 * (r >> 2) + 4 * h + (r & 3);
          if (n < b.n_out) atomicAdd(b.dW + (int64_t)n * b.ld_dw + col, acc[ta][tb][r]);
        }
      }
    }
    if (b.dbias && wc == 0 && okA[ta]) atomicAdd(b.dbias + 128 * wr + 32 * ta + l31, bsum[ta]);
  }
}

__global__ __launch_bounds__(WG_THREADS, 1) void k_mlp_wgrad(WgArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[WG_SLOTS * WG_SLOT + 4 * 256];    // + 1 KB dump area per wave
  const int64_t M = fgs_rows(a.M, a.m_dev);
  if (M <= 0) return;
  int blk = 0;
  while (blk + 1 < a.n_blocks && (int)blockIdx.x >= a.B[blk + 1].wg0) ++blk;
  const WgBlock &b = a.B[blk];
  const int j = (int)blockIdx.x - b.wg0;
  if (j >= b.n_wg) return;
  switch (b.nctw) {
    case 1: wgrad_block<1>(b, M, j, lds); break;
    case 2: wgrad_block<2>(b, M, j, lds); break;
    default: wgrad_block<4>(b, M, j, lds); break;
  }
}

// M too small for 1 KB DMA pieces to stay inside the tensors: one thread per dW element
__global__ __launch_bounds__(FGS_BLOCK) void k_wgrad_small(WgArgs a) {
  const int64_t M = fgs_rows(a.M, a.m_dev);
  const WgBlock &b = a.B[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int cols = b.n_in - b.col0 < 256 ? b.n_in - b.col0 : 256;
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

bool wg_aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }
bool wg_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

FGS_API int fgs_mlp_wgrad(int64_t M, int n_items, const fgs_wgrad_item_t *items, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && n_items >= 1 && n_items <= WG_MAXBLK, FGS_E_RANGE,
              "fgs_mlp_wgrad: M=%lld n_items=%d (1..%d)", (long long)M, n_items, WG_MAXBLK);
  if (M == 0) return 0;
  FGS_REQUIRE(items, FGS_E_INVALID, "fgs_mlp_wgrad: null pointer");
  WgArgs a;
  a.M = M; a.m_dev = fgs_row_ptr();
  int nb = 0, cost_total = 0;
  int cost[WG_MAXBLK];
  for (int i = 0; i < n_items; ++i) {
    const fgs_wgrad_item_t &U = items[i];
    FGS_REQUIRE(U.dY && U.X && U.dW && wg_aligned16(U.dY) && wg_aligned16(U.X) && wg_aligned4(U.dW) && U.n_out > 0 &&
                    U.n_out <= 256 && U.n_in > 0 && U.n_in <= 320 && U.ld_dy >= U.n_out && U.ld_x >= U.n_in &&
                    U.ld_dw >= U.n_in && U.ld_dy <= 320 && U.ld_x <= 320 && (U.ld_dy % 4) == 0 && (U.ld_x % 4) == 0,
                FGS_E_INVALID, "fgs_mlp_wgrad: item %d: bad pointer or shape (n_out <= 256, n_in <= 320, leading dimensions "
                               "of dY / X multiples of 4 and <= 320, 16-byte aligned)", i);
    for (int col0 = 0; col0 < U.n_in; col0 += 256) {
      FGS_REQUIRE(nb < WG_MAXBLK, FGS_E_RANGE, "fgs_mlp_wgrad: more than %d column blocks", WG_MAXBLK);
      const int cols = U.n_in - col0 < 256 ? U.n_in - col0 : 256;
      const int tiles = (cols + 31) / 32;                   // column tiles of the block
      WgBlock &b = a.B[nb];
      b.dY = U.dY; b.X = U.X; b.dW = U.dW; b.dbias = col0 == 0 ? U.dbias : nullptr;
      b.ld_dy = U.ld_dy; b.ld_x = U.ld_x; b.ld_dw = U.ld_dw; b.n_out = U.n_out; b.n_in = U.n_in; b.col0 = col0;
      b.nctw = tiles <= 2 ? 1 : tiles <= 4 ? 2 : 4;
      // time per k-step, measured: 16 MFMAs (1024 cycles) + ~250 for a full block; a narrow block is bound by the per-sample
      // streaming work (panel DMA, operand reads, barrier), ~700 cycles, not by its 4 or 8 MFMAs
      cost[nb] = b.nctw == 4 ? 10 : 7;
      cost_total += cost[nb];
      ++nb;
    }
  }
  a.n_blocks = nb;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  // one workgroup per CU (256 accumulator registers per lane); blocks get workgroups in proportion to their MFMA count,
  // at least one each, never more than one per 64 samples
  const int64_t max_per_block = (M + 63) / 64;
  int wg = 0, given = 0, cost_seen = 0;
  for (int i = 0; i < nb; ++i) {
    cost_seen += cost[i];
    int n = (int)((int64_t)cus * cost_seen / cost_total) - given;    // cumulative rounding: the shares add up to cus
    if (n < 1) n = 1;
    if (n > max_per_block) n = (int)max_per_block;
    given += n;
    a.B[i].wg0 = wg; a.B[i].n_wg = n;
    wg += n;
  }
  if (M < 64 && !a.m_dev) {      // tiny batches: not worth a 256-register workgroup per block
    hipLaunchKernelGGL(k_wgrad_small, dim3(fgs_blocks(256 * 256), (unsigned)nb), dim3(FGS_BLOCK), 0, fgs_s(stream), a);
    FGS_LAUNCH_OK("fgs_mlp_wgrad (small)");
    return 0;
  }
  hipLaunchKernelGGL(k_mlp_wgrad, dim3((unsigned)wg), dim3(WG_THREADS), 0, fgs_s(stream), a);
  FGS_LAUNCH_OK("fgs_mlp_wgrad");
  return 0;
}
