// mlp_fused.hip -- the whole forward chain of the two tiny MLPs (rgbnet 106 -> 256^3 -> 256, refnet 307 -> 256^3,
// model/nerf.py:125-142,877,884) as ONE persistent kernel.
//
// Why: launched layer by layer (fgs_gemm_f32) every product is a short kernel -- K = 256 is 8 chunks of MFMA work per
// tile -- that pays its own launch gap, first-chunk latency, 64 KB-per-tile epilogue and partially filled last round
// of tiles (measured: 84 us per layer at M_s = 50 K where the steady-state K loop would need 51).  Here a 512-thread
// workgroup (8 waves, two per SIMD, one workgroup per CU) owns a block of 64 samples and walks ALL layers with the
// activation block resident in LDS:
//   * H [64][324] (83 KB): the current layer's input rows; a layer's output overwrites columns 0..255 in place once its
//     K loop is done; columns 256..319 hold the reflection encoding the second network appends (refnet input =
//     [rgbnet output | reflect PE], model/nerf.py:883), loaded once per block;
//   * the weight chunks stream through a double-buffered [256][36] image (74 KB) with register-staged prefetch two chunks
//     ahead that runs ACROSS layer boundaries, so the matrix cores never wait for a layer's first operand;
//   * every layer's output is also written to HBM (coalesced float4 rows) because the backward pass needs it.
// Per accumulator the k order is exactly fgs_gemm_f32's (chunks ascending, within a chunk the k pairs (s, 16+s) of the
// first then of the second half), so the results are bit-identical to the layer-by-layer path.
#include "fgs_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int MB = 64;          // sample rows per block
constexpr int NW = 256;         // width of every layer
constexpr int LDH = 324;        // floats per H row: 320 columns + 4 (rows land 4 banks apart: conflict-free ds_read_b128)
constexpr int LDK = 36;         // floats per staged weight row
constexpr int BK = 32;
constexpr int MAXL = 8;
constexpr int THREADS = 512;

struct MlpLayer {
  const float *W;      // [n_rows <= NW][ldw], K valid columns; output column n = row n of W
  const float *bias;   // [NW] or null
  float *out;          // [M][ldo]: the first n_store columns of the layer output (next layer's input / backward's operand)
  int64_t ldw, ldo;
  int K, relu;
  // used by the backward data-gradient chain (same kernel, transposed weights):
  const float *mask;   // [M][ldm] or null: output (m, n) is zeroed where mask[m, n] <= 0 (the ReLU input saved by the forward)
  int64_t ldm;
  float *colsum;       // [NW] or null: colsum[n] += sum_m output[m, n] (bias gradient of the layer below), atomically
  int n_rows, n_store; // valid weight rows (= meaningful output columns); columns written to `out` (multiple of 4)
};

struct MlpArgs {
  int64_t M;
  int n_layers;
  const float *X0; int64_t ldx0; int k0;               // layer-0 input rows -> H[:, 0:k0)
  const float *T; int64_t ldt; int t_cols;             // appended columns -> H[:, 256:256+t_cols) (NULL: none)
  int64_t rows_per_wg;                                 // contiguous rows per workgroup (multiple of 32)
  MlpLayer L[MAXL];
};

struct WStage {
  float4 v[4];
};

// weight chunk c of layer l: rows n = (tid>>3) + 64 p, k = 32 c + 4 (tid & 7)
// GEN = false: the forward-pass form (full 256-row weights, no mask, no column sums); the general code is compiled out
template <bool GEN>
__device__ __forceinline__ void w_load(WStage &s, const MlpLayer &L, int c, int tid) {
  const int k = c * BK + 4 * (tid & 7);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int n = (tid >> 3) + 64 * p;
    s.v[p] = (k < L.K && (!GEN || n < L.n_rows)) ? *reinterpret_cast<const float4 *>(L.W + (int64_t)n * L.ldw + k)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

__device__ __forceinline__ void w_store(const WStage &s, float *__restrict__ wst, int tid) {
#pragma unroll
  for (int p = 0; p < 4; ++p)
    *reinterpret_cast<float4 *>(wst + ((tid >> 3) + 64 * p) * LDK + 4 * (tid & 7)) = s.v[p];
}

// 8 operand values k = 16 h + 8 half .. + 7 of row `row` (pitch ld) starting at column k0
__device__ __forceinline__ void half8(float (&f)[8], const float *__restrict__ base, int row, int ld, int k0, int h, int half) {
  const float4 *p = reinterpret_cast<const float4 *>(base + row * ld + k0 + 16 * h + 8 * half);
  const float4 u = p[0], v = p[1];
  f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w;
  f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
}

// One block of `ROWS` (64 or 32) sample rows through all layers.  TN = MFMA column tiles per wave:
//   TN = 2: 64 rows, waves 2 (rows) x 4 (columns), a wave owns 32 rows x 64 columns;
//   TN = 1: 32 rows, waves 1 x 8, a wave owns 32 rows x 32 columns -- the half-size block that ends a workgroup's row range,
//           so that the ranges can be balanced to 32 rows instead of 64 (M = 49 920 on 256 CUs: 3.5 block-times, not 4).
template <int TN, bool GEN>
__device__ __forceinline__ void mlp_block(const MlpArgs &a, int64_t m0, float *__restrict__ H, float (*Wst)[NW * LDK]) {
  constexpr int ROWS = 32 * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (TN == 2) ? wave >> 2 : 0, h = lane >> 5, l31 = lane & 31;
  const int arow = wm * 32 + l31;                                      // this lane's A row inside the block
  const int bcol0 = ((TN == 2) ? (wave & 3) * 64 : wave * 32) + l31;   // this lane's output columns bcol0 (+ 32)

  // ---- block inputs -> H (previous block's readers are past its trailing barrier)
  for (int q = tid; q < ROWS * (NW / 4); q += THREADS) {       // columns 0..255: layer-0 input (k0 of them), zero padded
    const int row = q >> 6, c4 = q & 63;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + row < a.M && 4 * c4 < a.k0) v = *reinterpret_cast<const float4 *>(a.X0 + (m0 + row) * a.ldx0 + 4 * c4);
    *reinterpret_cast<float4 *>(H + row * LDH + 4 * c4) = v;
  }
  for (int q = tid; q < ROWS * 16; q += THREADS) {             // columns 256..319: appended encoding, zero padded
    const int row = q >> 4, c4 = q & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.T && m0 + row < a.M && 4 * c4 < a.t_cols) v = *reinterpret_cast<const float4 *>(a.T + (m0 + row) * a.ldt + 4 * c4);
    *reinterpret_cast<float4 *>(H + row * LDH + NW + 4 * c4) = v;
  }
  // ---- weight stream: chunk (0,0) into the image, chunk +1 into the staging registers
  WStage sw;
  w_load<GEN>(sw, a.L[0], 0, tid);
  w_store(sw, Wst[0], tid);
  {
    const int nch0 = (a.L[0].K + BK - 1) / BK;
    if (nch0 > 1) w_load<GEN>(sw, a.L[0], 1, tid);
    else if (a.n_layers > 1) w_load<GEN>(sw, a.L[1], 0, tid);
  }
  __syncthreads();
  int buf = 0;
  // The HBM copy of a layer's output (needed by the backward pass) is not issued in one burst after the layer -- written
  // that way the seven 64-row x 1 KB copies of all workgroups hit HBM at the same moments and cost 56 us of a 586 us launch
  // (measured by removing them) -- but one 512-float4 slice per K chunk of the NEXT layer, whose K loop only reads H.
  constexpr int COPY_SLICES = ROWS * (NW / 4) / THREADS;   // 8 (64 rows) or 4 (32 rows); every later layer has >= 8 chunks
  float *pend_out = nullptr;
  int64_t pend_ldo = 0;

  for (int l = 0; l < a.n_layers; ++l) {
    const MlpLayer &L = a.L[l];
    const int nch = (L.K + BK - 1) / BK;
    const bool more_layers = l + 1 < a.n_layers;
    const int nch_next = more_layers ? (a.L[l + 1].K + BK - 1) / BK : 0;
    floatx16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float mk[TN][16];                                     // ReLU-input mask of this lane's outputs, fetched ahead of use
    if (GEN && L.mask) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          mk[j][r] = (row < a.M) ? L.mask[row * L.ldm + bcol0 + 32 * j] : 0.f;
        }
    }
    float f0a[8], f1a[8], f0b[TN][8], f1b[TN][8];
    half8(f0a, H, arow, LDH, 0, h, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) half8(f0b[j], Wst[buf], bcol0 + 32 * j, LDK, 0, h, 0);

    for (int c = 0; c < nch; ++c) {
      // chunk +1 (already in registers) -> the other image; chunk +2 -> registers.  "+1/+2" run across the layer end.
      const bool has1 = (c + 1 < nch) || more_layers;
      if (has1) {
        w_store(sw, Wst[buf ^ 1], tid);
        if (c + 2 < nch) w_load<GEN>(sw, L, c + 2, tid);
        else if (more_layers) {
          const int c2 = c + 2 - nch;                      // 0 or 1 in the next layer
          if (c2 < nch_next) w_load<GEN>(sw, a.L[l + 1], c2, tid);
          else if (l + 2 < a.n_layers) w_load<GEN>(sw, a.L[l + 2], 0, tid);   // next layer has a single chunk
        }
      }
      if (pend_out && c < COPY_SLICES) {                   // slice c of the previous layer's output -> HBM
        const int q = tid + THREADS * c, row = q >> 6, c4 = q & 63;
        if (m0 + row < a.M)
          *reinterpret_cast<float4 *>(pend_out + (m0 + row) * pend_ldo + 4 * c4) =
              *reinterpret_cast<const float4 *>(H + row * LDH + 4 * c4);
      }
      half8(f1a, H, arow, LDH, c * BK, h, 1);
#pragma unroll
      for (int j = 0; j < TN; ++j) half8(f1b[j], Wst[buf], bcol0 + 32 * j, LDK, 0, h, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0a[s], f0b[j][s], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (c + 1 < nch) {
        half8(f0a, H, arow, LDH, (c + 1) * BK, h, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) half8(f0b[j], Wst[buf ^ 1], bcol0 + 32 * j, LDK, 0, h, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1a[s], f1b[j][s], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      buf ^= 1;
    }
    // ---- layer output: nobody reads H any more in this layer (every wave's last fragments were fetched before the last
    // barrier), so the tile goes straight back into columns 0..255; bias + ReLU on the way
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = bcol0 + 32 * j;
      const float b = L.bias ? L.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[j][r] + b;
        if (L.relu) v = fmaxf(v, 0.f);
        if (GEN && L.mask && !(mk[j][r] > 0.f)) v = 0.f;
        H[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDH + col] = v;
      }
    }
    __syncthreads();
    if (GEN && L.colsum && tid < NW) {              // column sums of the block (rows beyond M are zero), one atomic per column
      float cs = 0.f;
#pragma unroll 8
      for (int row = 0; row < ROWS; ++row) cs += H[row * LDH + tid];
      if (tid < L.n_rows) atomicAdd(L.colsum + tid, cs);
    }
    const int nch_after = more_layers ? nch_next : 0;
    if (nch_after >= COPY_SLICES && (!GEN || L.n_store == NW)) {   // deferred: trickles out under the next layer's K loop
      pend_out = L.out;
      pend_ldo = L.ldo;
    } else {                                 // last layer (or a short next layer, or a narrow output): float4 rows, now
      pend_out = nullptr;
      for (int q = tid; q < ROWS * (NW / 4); q += THREADS) {
        const int row = q >> 6, c4 = q & 63;
        if (m0 + row < a.M && (!GEN || 4 * c4 < L.n_store))
          *reinterpret_cast<float4 *>(L.out + (m0 + row) * L.ldo + 4 * c4) = *reinterpret_cast<const float4 *>(H + row * LDH + 4 * c4);
      }
    }
  }
  __syncthreads();   // H and the weight image are free for the next block
}

template <bool GEN>
__global__ __launch_bounds__(THREADS, 1) void k_mlp_fwd(MlpArgs a) {
  __shared__ __attribute__((aligned(16))) float H[MB * LDH];
  __shared__ __attribute__((aligned(16))) float Wst[2][NW * LDK];
  // contiguous row range per workgroup, a multiple of 32 rows: 64-row blocks, then at most one 32-row block
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_wg;
  const int64_t r1 = (r0 + a.rows_per_wg < a.M) ? r0 + a.rows_per_wg : a.M;
  for (int64_t m0 = r0; m0 < r1;) {
    if (r1 - m0 > 32) {
      mlp_block<2, GEN>(a, m0, H, Wst);
      m0 += 64;
    } else {
      mlp_block<1, GEN>(a, m0, H, Wst);
      m0 += 32;
    }
  }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// A chain of 256-wide products over M sample rows in one launch.  Forward pass: Linear(+bias)(+ReLU) layers.  Backward
// pass (data gradients): the same kernel on TRANSPOSED weights, dY of the top layer as the input, per layer the ReLU mask
// of the layer below (its saved input) and the column sums that are that layer's bias gradient.
//   layer 0 input  : X0[M, k0]                                   (k0 <= 256, multiple of 4)
//   layer l input  : output of layer l-1 (256 columns) followed by T[M, t_cols] when K_l > 256   (t_cols <= 64)
//   W[l]           : [n_rows[l] <= 256, ldw[l]] with K[l] valid columns; output column n uses row n (missing rows: zero)
//   outs[l][M, ld] : the first n_store[l] columns of layer l's output (after bias / ReLU / mask)
// All arrays are HOST arrays of n_layers entries; bias / mask / colsum entries may be NULL.
FGS_API int fgs_mlp_chain_f32(int64_t M, int n_layers, const float *X0, int64_t ldx0, int k0, const float *T, int64_t ldt,
                              int t_cols, const float *const *W, const int64_t *ldw, const int *K, const int *n_rows,
                              const float *const *bias, const int *relu, const float *const *mask, const int64_t *ldm,
                              float *const *colsum, float *const *outs, const int64_t *ldo, const int *n_store,
                              fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31) && n_layers >= 1 && n_layers <= MAXL, FGS_E_RANGE,
              "fgs_mlp_chain_f32: M=%lld n_layers=%d (1..%d)", (long long)M, n_layers, MAXL);
  if (M == 0) return 0;
  FGS_REQUIRE(X0 && W && ldw && K && n_rows && bias && relu && mask && ldm && colsum && outs && ldo && n_store, FGS_E_INVALID,
              "fgs_mlp_chain_f32: null pointer");
  FGS_REQUIRE(k0 > 0 && k0 <= NW && (k0 % 4) == 0 && (ldx0 % 4) == 0 && aligned16(X0) && K[0] == k0, FGS_E_INVALID,
              "fgs_mlp_chain_f32: layer-0 input must have k0 = K[0] <= 256 columns, multiple of 4, 16-byte aligned rows");
  FGS_REQUIRE(t_cols >= 0 && t_cols <= 64 && (t_cols % 4) == 0 && (!T || ((ldt % 4) == 0 && aligned16(T))), FGS_E_INVALID,
              "fgs_mlp_chain_f32: appended columns: at most 64, multiple of 4, 16-byte aligned rows");
  MlpArgs a;
  a.M = M; a.n_layers = n_layers; a.X0 = X0; a.ldx0 = ldx0; a.k0 = k0; a.T = T; a.ldt = ldt; a.t_cols = t_cols;
  for (int l = 0; l < n_layers; ++l) {
    FGS_REQUIRE(W[l] && outs[l] && aligned16(W[l]) && aligned16(outs[l]) && (ldw[l] % 4) == 0 && (ldo[l] % 4) == 0 &&
                    (!bias[l] || aligned16(bias[l])) && n_rows[l] > 0 && n_rows[l] <= NW && n_store[l] > 0 &&
                    n_store[l] <= NW && (n_store[l] % 4) == 0 && ldo[l] >= n_store[l] && (!mask[l] || ldm[l] >= n_rows[l]),
                FGS_E_INVALID, "fgs_mlp_chain_f32: layer %d: bad pointer / alignment / leading dimension / column counts", l);
    FGS_REQUIRE(K[l] > 0 && (K[l] % 4) == 0 && ldw[l] >= K[l] && (l == 0 || K[l] == NW || (T && K[l] == NW + t_cols)),
                FGS_E_INVALID, "fgs_mlp_chain_f32: layer %d: K=%d (expected %d or %d)", l, K[l], NW, NW + t_cols);
    FGS_REQUIRE(l + 1 == n_layers || n_rows[l] == NW, FGS_E_INVALID,
                "fgs_mlp_chain_f32: layer %d: only the last layer may produce fewer than 256 columns", l);
    MlpLayer &L = a.L[l];
    L.W = W[l]; L.bias = bias[l]; L.out = outs[l]; L.ldw = ldw[l]; L.ldo = ldo[l]; L.K = K[l]; L.relu = relu[l];
    L.mask = mask[l]; L.ldm = ldm[l]; L.colsum = colsum[l]; L.n_rows = n_rows[l]; L.n_store = n_store[l];
  }
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  const int64_t n_blocks = (M + MB - 1) / MB;
  const unsigned grid = (unsigned)(n_blocks < cus ? n_blocks : cus);
  a.rows_per_wg = ((M + grid - 1) / grid + 31) / 32 * 32;
  bool general = false;
  for (int l = 0; l < n_layers; ++l) general |= mask[l] || colsum[l] || n_rows[l] != NW || n_store[l] != NW;
  if (general) hipLaunchKernelGGL(k_mlp_fwd<true>, dim3(grid), dim3(THREADS), 0, fgs_s(stream), a);
  else hipLaunchKernelGGL(k_mlp_fwd<false>, dim3(grid), dim3(THREADS), 0, fgs_s(stream), a);
  FGS_LAUNCH_OK("fgs_mlp_chain_f32");
  return 0;
}

// The forward-pass form of fgs_mlp_chain_f32: full 256-row weights, bias + ReLU epilogues, every output stored in full.
FGS_API int fgs_mlp_fwd_f32(int64_t M, int n_layers, const float *X0, int64_t ldx0, int k0, const float *T, int64_t ldt,
                            int t_cols, const float *const *W, const int64_t *ldw, const int *K, const float *const *bias,
                            const int *relu, float *const *outs, const int64_t *ldo, fgs_stream_t stream) {
  FGS_REQUIRE(n_layers >= 1 && n_layers <= MAXL, FGS_E_RANGE, "fgs_mlp_fwd_f32: n_layers=%d (1..%d)", n_layers, MAXL);
  int full[MAXL];
  const float *no_mask[MAXL];
  float *no_colsum[MAXL];
  int64_t zero_ld[MAXL];
  for (int l = 0; l < MAXL; ++l) { full[l] = NW; no_mask[l] = nullptr; no_colsum[l] = nullptr; zero_ld[l] = 0; }
  return fgs_mlp_chain_f32(M, n_layers, X0, ldx0, k0, T, ldt, t_cols, W, ldw, K, full, bias, relu, no_mask, zero_ld, no_colsum,
                           outs, ldo, full, stream);
}

namespace {

struct TransposeArgs {
  int n;
  const float *src[MAXL];
  float *dst[MAXL];
  int rows[MAXL], cols[MAXL];
  int64_t ld_src[MAXL], ld_dst[MAXL];
  int tile0[MAXL + 1];   // first 32x32 tile of each matrix in the flat grid
};

// dst[c][r] = src[r][c] for up to 8 small matrices in one launch (the transposed weights the backward chain multiplies by)
__global__ __launch_bounds__(256) void k_transpose_multi(TransposeArgs a) {
  __shared__ float t[32][33];
  int m = 0;
  while (m + 1 < a.n && (int)blockIdx.x >= a.tile0[m + 1]) ++m;
  const int local = (int)blockIdx.x - a.tile0[m];
  const int tiles_c = (a.cols[m] + 31) / 32;
  const int r0 = (local / tiles_c) * 32, c0 = (local % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    t[ty + 8 * i][tx] = (r < a.rows[m] && c < a.cols[m]) ? a.src[m][(int64_t)r * a.ld_src[m] + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < a.cols[m] && r < a.rows[m]) a.dst[m][(int64_t)c * a.ld_dst[m] + r] = t[tx][ty + 8 * i];
  }
}

}  // namespace

// dst[i] [cols[i], ld_dst[i]] = transpose of src[i] [rows[i], ld_src[i]] (cols[i] valid columns), i < n <= 8; HOST arrays.
FGS_API int fgs_transpose_multi(int n, const float *const *src, const int *rows, const int *cols, const int64_t *ld_src,
                                float *const *dst, const int64_t *ld_dst, fgs_stream_t stream) {
  FGS_REQUIRE(n >= 0 && n <= MAXL, FGS_E_RANGE, "fgs_transpose_multi: n=%d (0..%d)", n, MAXL);
  if (n == 0) return 0;
  FGS_REQUIRE(src && rows && cols && ld_src && dst && ld_dst, FGS_E_INVALID, "fgs_transpose_multi: null pointer");
  TransposeArgs a;
  a.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    FGS_REQUIRE(src[i] && dst[i] && rows[i] > 0 && cols[i] > 0 && ld_src[i] >= cols[i] && ld_dst[i] >= rows[i], FGS_E_INVALID,
                "fgs_transpose_multi: matrix %d: bad pointer or shape", i);
    a.src[i] = src[i]; a.dst[i] = dst[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i]; a.ld_src[i] = ld_src[i]; a.ld_dst[i] = ld_dst[i];
    a.tile0[i] = tiles;
    tiles += ((rows[i] + 31) / 32) * ((cols[i] + 31) / 32);
  }
  a.tile0[n] = tiles;
  hipLaunchKernelGGL(k_transpose_multi, dim3((unsigned)tiles), dim3(256), 0, fgs_s(stream), a);
  FGS_LAUNCH_OK("fgs_transpose_multi");
  return 0;
}

// dst[i] [rows[i], ld_dst[i]] = src[i] [rows[i], cols[i]] with the columns cols[i] .. ld_dst[i]-1 zero-filled, i < n <= 8, in
// ONE launch (the K-padded first-layer weights of both MLPs: torch's F.pad is a fill + a copy launch per matrix and step).
namespace {
struct PadArgs {
  int n;
  const float *src[MAXL];
  float *dst[MAXL];
  int rows[MAXL], cols[MAXL], width[MAXL];     // width: columns WRITTEN per destination row (cols copied, the rest zero)
  int64_t ld_src[MAXL], ld_dst[MAXL];
  int64_t elem0[MAXL + 1];   // first destination element of each matrix in the flat range
};
__global__ __launch_bounds__(FGS_BLOCK) void k_pad_cols_multi(PadArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.elem0[a.n]) return;
  int m = 0;
  while (m + 1 < a.n && i >= a.elem0[m + 1]) ++m;
  const int64_t e = i - a.elem0[m];
  const int64_t r = e / a.width[m], c = e - r * a.width[m];
  a.dst[m][r * a.ld_dst[m] + c] = c < a.cols[m] ? a.src[m][r * a.ld_src[m] + c] : 0.f;
}
}  // namespace

// width[i] columns are written per destination row (pitch ld_dst[i] >= width[i] >= cols[i]): destinations may be column
// ranges of a wider matrix, i.e. several sources gathered side by side into one.  width == NULL: width[i] = ld_dst[i].
FGS_API int fgs_copy_cols_multi(int n, const float *const *src, const int *rows, const int *cols, const int64_t *ld_src,
                                float *const *dst, const int64_t *ld_dst, const int *width, fgs_stream_t stream) {
  FGS_REQUIRE(n >= 1 && n <= MAXL, FGS_E_RANGE, "fgs_copy_cols_multi: n=%d (1..%d)", n, MAXL);
  FGS_REQUIRE(src && rows && cols && ld_src && dst && ld_dst, FGS_E_INVALID, "fgs_copy_cols_multi: null pointer");
  PadArgs a;
  a.n = n;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t w = width ? width[i] : ld_dst[i];
    FGS_REQUIRE(src[i] && dst[i] && rows[i] > 0 && cols[i] > 0 && ld_src[i] >= cols[i] && w >= cols[i] && ld_dst[i] >= w &&
                    w < ((int64_t)1 << 31), FGS_E_INVALID, "fgs_copy_cols_multi: matrix %d: bad pointer or shape", i);
    a.src[i] = src[i]; a.dst[i] = dst[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i]; a.width[i] = (int)w;
    a.ld_src[i] = ld_src[i]; a.ld_dst[i] = ld_dst[i];
    a.elem0[i] = total;
    total += (int64_t)rows[i] * w;
  }
  a.elem0[n] = total;
  hipLaunchKernelGGL(k_pad_cols_multi, dim3(fgs_blocks(total)), dim3(FGS_BLOCK), 0, fgs_s(stream), a);
  FGS_LAUNCH_OK("fgs_copy_cols_multi");
  return 0;
}

// Diagnostics (tests/test_mlp_rc_gpu.py, DESIGN.md section 4, "The round-2 abort"): ONE matrix copied with the index expression
// this kernel had up to commit a5aee36 -- dst[e] for e < rows * ld_dst, which takes the destination for a whole [rows, ld_dst]
// matrix -- so that the canary test can be shown to catch it when the destination is a column slice of a wider tensor.  The
// caller guarantees rows * ld_dst floats behind `dst` are its own memory.
namespace {
__global__ __launch_bounds__(FGS_BLOCK) void k_pad_cols_old_indexing(const float *__restrict__ src, float *__restrict__ dst, int rows,
                                                                     int cols, int64_t ld_src, int64_t ld_dst) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)rows * ld_dst) return;
  const int64_t r = e / ld_dst, c = e - r * ld_dst;
  dst[e] = c < cols ? src[r * ld_src + c] : 0.f;
}
}  // namespace

FGS_API int fgs_debug_pad_cols_old_indexing(const float *src, int rows, int cols, int64_t ld_src, float *dst, int64_t ld_dst,
                                            fgs_stream_t stream) {
  FGS_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols, FGS_E_INVALID,
              "fgs_debug_pad_cols_old_indexing: bad argument");
  hipLaunchKernelGGL(k_pad_cols_old_indexing, dim3(fgs_blocks((int64_t)rows * ld_dst)), dim3(FGS_BLOCK), 0, fgs_s(stream), src, dst,
                     rows, cols, ld_src, ld_dst);
  FGS_LAUNCH_OK("fgs_debug_pad_cols_old_indexing");
  return 0;
}

FGS_API int fgs_pad_cols_multi(int n, const float *const *src, const int *rows, const int *cols, const int64_t *ld_src,
                               float *const *dst, const int64_t *ld_dst, fgs_stream_t stream) {
  return fgs_copy_cols_multi(n, src, rows, cols, ld_src, dst, ld_dst, nullptr, stream);
}

