// mlp_rc2.hip -- the tiny-MLP chains (rgbnet + refnet, model/nerf.py:125-142,877,884,1009), second form: the four waves of a
// workgroup SPLIT A LAYER'S OUTPUT FEATURES, the slab's activations live in LDS.
//
// Why a second form.  mlp_rc.hip gives every wave 32 samples and all 256 output features; its unit of work per SIMD is a whole
// 32-sample tile through all layers, and 57 K survivors are 1.73 such rounds on 1024 SIMDs -- which cost 2 (0.865 of the launch is
// useful).  Here a workgroup (one per CU) owns a SLAB of S <= 4 sample tiles (32 samples each); wave w computes output features
// 64 w .. 64 w + 63 of every layer for ALL S tiles, so the four SIMDs of a CU finish a slab together whatever S is, and a CU that
// gets 7 tiles (two slabs, 4 + 3) costs 7 tile-times instead of 8: the launch is quantised in units of 1 / 4 tile-time per SIMD.
//
// Formulation (as mlp_rc.hip): D^T[feature, sample] = W[feature, k] * X^T[k, sample] on v_mfma_f32_32x32x2_f32, weights as the A
// operand, activations as B.  Reduction in groups of 8 columns: lane (j = lane & 31, h = lane >> 5) holds 16 bytes of each operand,
// columns 8 g + 4 h + 0..3, and MFMA i of a group multiplies component i -- the pairs (8 g + i, 8 g + 4 + i).
//   A: straight from L2 into registers.  A pack kernel writes each layer's weights in FRAGMENT ORDER -- [feature tile][k-group][lane]
//      x 16 bytes -- so a wave's global_load_dwordx4 is one contiguous 1 KB piece that nobody else on the CU needs: no LDS ring, no
//      LDS-DMA issue cost, no per-chunk barrier.  One 16-byte load feeds 4 S MFMAs.
//   B: ds_read_b128 from the slab's activation buffer X[sample][260] in LDS (pitch / 4 odd: every lane group of the read hits 16
//      distinct 4-bank groups); one read feeds 8 MFMAs.
//   D: 2 feature tiles x S sample tiles of accumulators per wave (128 registers at S = 4).  A layer's epilogue (ReLU / mask, sign
//      bits, HBM copy for the weight-gradient kernel) writes the new activations over X between two barriers.
// Narrow products of the carried input that the old form left to k_gemm (backward: the reflection-encoding columns of dZ, the
// compact dX0) are SIDE layers: 1..2 feature tiles dealt to the waves by (tile, sample tile), output to HBM only, X untouched.
//
// Numerics: every sum runs over k in the order of the groups above, a fixed order per output element (deterministic); the bias is
// the accumulators' initial value.  Not bit-identical to mlp_rc.hip or to fgs_gemm_f32 (other k orders), same error bound.
#include "fgs_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int R2_MAXL = 10;           // layers per chain (main + side)
constexpr int R2_THREADS = 256;
constexpr int R2_SMAX = 4;            // sample tiles per slab
constexpr int R2_P = 260;             // floats per sample row of X (256 + 4)
constexpr int R2_PE = 52;             // floats per sample row of E, the appended columns (forward refnet layer 0)
constexpr int R2_X_FLOATS = R2_SMAX * 32 * R2_P;
constexpr int R2_E_FLOATS = R2_SMAX * 32 * R2_PE + 64;     // + what the last row's padded / prefetched k-groups read beyond it
constexpr int R2_SINK_FLOATS = 512;   // where lanes without a sample store (behind the weight image; never read)
constexpr int R2_ZERO_FLOATS = 256;   // zeros behind the sink (written by the pack launch)

struct R2Layer {
  int64_t img;                 // float offset of the layer's fragment image
  int k8x, k8e;                // k-groups over X (the carried input) and over E (appended columns); both multiples of 4
  int n_ft;                    // output feature tiles: 8 (main) or 1..2 (side)
  int side;                    // 0: main, 1: side (plain), 2: side + bias + sigmoid, the first n_store <= 4 columns stored one by one
  int relu;
  const float *bias;           // forward main layers (may be null)
  unsigned *mask_w;            // forward + relu: sign bits of the output, [tile][wave][lane] words
  const unsigned *mask_r;      // backward: bits applied to this layer's output
  float *out;
  int64_t ldo;
  int n_store;                 // output columns stored (multiple of 4)
};

struct R2Args {
  int64_t M;
  const int64_t *m_dev;
  int n_layers;
  FgsStamps stamps;
  const float *img;
  float *sink;
  const float *zeros;          // 256 zero floats behind the image (the bias of a layer without one; the all-ones mask's source)
  const float *in0;            // the chain's input [M][ld_in0]
  int64_t ld_in0;
  int in0_cols, in0_valid, in0_zero_to;      // columns of the buffer / columns that carry data / columns of X to zero-fill up to
  const float *ext;            // columns appended to the carried input of the one layer with k8e > 0
  int64_t ld_ext;
  int ext_cols, ext_valid;
  R2Layer L[R2_MAXL];
};

// ------------------------------------------------------------------------------------------------ weight image
struct P2Layer {
  const float *W;
  int64_t ldw;
  int n_out, n_in;      // W is [n_out][ldw] with n_in valid columns
  int n_ft, k8;         // image geometry
  int transpose;
  float *dst;           // the layer's fragments
  int64_t f4_begin;     // first float4 of this layer in the flat work range of the launch
};
struct P2Args {           // up to two chains' images in one launch (fgs_mlp_rc2_pack: the forward and the backward chain of a step)
  int n_layers;
  float *zeros[2];      // R2_ZERO_FLOATS floats each, zero-filled by the first threads (second may be null)
  int64_t f4_total;
  P2Layer L[2 * R2_MAXL];
};

// image element (row, k) = transpose ? W[k][row] : W[row][k]; zero outside.  One thread per float4 = one lane's operand.
__global__ __launch_bounds__(FGS_BLOCK) void k_rc2_pack(P2Args a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 64) reinterpret_cast<float4 *>(a.zeros[0])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  else if (i < 128 && a.zeros[1]) reinterpret_cast<float4 *>(a.zeros[1])[i - 64] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i >= a.f4_total) return;
  int l = 0;
  while (l + 1 < a.n_layers && i >= a.L[l + 1].f4_begin) ++l;
  const P2Layer &L = a.L[l];
  const int64_t q = i - L.f4_begin;              // [ft][g][lane]
  const int lane = (int)(q & 63);
  const int g = (int)((q >> 6) % L.k8), ft = (int)((q >> 6) / L.k8);
  const int row = 32 * ft + (lane & 31), k0 = 8 * g + 4 * (lane >> 5);
  const int n_rows = L.transpose ? L.n_in : L.n_out, n_k = L.transpose ? L.n_out : L.n_in;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    float x = 0.f;
    if (row < n_rows && k < n_k) x = L.transpose ? L.W[(int64_t)k * L.ldw + row] : L.W[(int64_t)row * L.ldw + k];
    v[j] = x;
  }
  *reinterpret_cast<float4 *>(L.dst + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
}

// ------------------------------------------------------------------------------------------------ the chain
// (explicit address spaces: through pointer arrays that a loop advances, hipcc loses track of where a generic pointer points and
// emits flat_load -- which counts in vmcnt AND lgkmcnt, returns out of order, and is waited for with a full drain)
typedef const __attribute__((address_space(1))) floatx4 *r2_gptr;
typedef const __attribute__((address_space(3))) floatx4 *r2_lptr;
__device__ __forceinline__ r2_gptr r2_g(const float *p) { return (r2_gptr)(uintptr_t)p; }
__device__ __forceinline__ r2_lptr r2_l(const float *p) {
  return (r2_lptr)(const __attribute__((address_space(3))) float *)p;
}
__device__ __forceinline__ floatx4 r2_ldg(const float *p) { return *r2_g(p); }

// One group of 8 reduction columns: 4 MFMAs per (feature tile, sample tile), component i of both operands.  The accumulators are
// visited round-robin, so one of them is touched every NF NS MFMAs (>= 2: a dependent 32x32x2 issues after 64 cycles).
template <int NF, int NS>
__device__ __forceinline__ void r2_group(floatx16 (&acc)[2][R2_SMAX], const floatx4 (&a)[2], const floatx4 (&b)[R2_SMAX]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int n = 0; n < NS; ++n)
        acc[f][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[f][i], b[n][i], acc[f][n], 0, 0, 0);
}

// The HBM copy of the PREVIOUS layer's output (what X holds while this layer reduces over it) leaves from inside this loop:
// a wave stores whole 1 KB rows -- one ds_read_b128 + one global_store_dwordx4 per k-group, fully coalesced -- instead of the
// epilogue's 32-byte pieces (a 128 KB burst of those per layer and workgroup is store-issue bound: the first form measured ~12 K
// idle cycles per layer for it).
struct R2Stream {
  r2_lptr lds;          // this lane's 16 bytes of the wave's next row in X
  float *g;             // ... and where they go (rows 4 i + wave of the slab)
  float *sink;          // this lane's 16 bytes of the sink: where a row beyond M goes (same instruction stream)
  int64_t g_step;       // floats between two rows of this wave (4 ldo)
  int valid;            // rows of this wave that exist (uniform)
  int i;                // next row
  bool on;              // a copy is pending (uniform)
};

__device__ __forceinline__ void r2_stream_put(R2Stream &st, const floatx4 &v) {      // row st.i - 1, read a group ago
  float *dst = (st.i - 1 < st.valid) ? st.g : st.sink;      // uniform select: scalar registers
  *reinterpret_cast<__attribute__((address_space(1))) floatx4 *>((uintptr_t)dst) = v;
  st.g += st.g_step;
}
__device__ __forceinline__ floatx4 r2_stream_get(R2Stream &st) {
  const floatx4 v = *st.lds;
  st.lds += 4 * (R2_P / 4);
  ++st.i;
  return v;
}

// G k-groups (a multiple of 4) of the reduction.  STR: every group also moves one row of the pending output copy.
template <int NF, int NS, bool STR>
__device__ __forceinline__ void r2_groups(floatx16 (&acc)[2][R2_SMAX], floatx4 (&a)[4][2], floatx4 (&b)[2][R2_SMAX], r2_gptr (&Ap)[2],
                                          r2_lptr (&xp)[R2_SMAX], int G, R2Stream &st, floatx4 &sv) {
  for (int g = 0; g < G; g += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < NF; ++f) a[(u + 2) & 3][f] = Ap[f][u * 64];
#pragma unroll
      for (int n = 0; n < NS; ++n) b[(u + 1) & 1][n] = xp[n][2 * u];
      if (STR) {
        if (g + u > 0) r2_stream_put(st, sv);      // (g + u > 0: compile-time for u > 0, one scalar test for u = 0)
        sv = r2_stream_get(st);
      }
      r2_group<NF, NS>(acc, a[u & 3], b[u & 1]);
      // Schedule of the group: its memory instructions are dealt ONE AT A TIME between its MFMAs.  Issued in one run in front of
      // the group they cost ~130 cycles of matrix-pipe idle (6 % of the reduction); in the 64-cycle shadow of a running MFMA
      // they are free.  hipcc's own counted waits (vmcnt / lgkmcnt) stay exact.
      constexpr int MF = 4 * NF * NS, MEM = NF + NS + (STR ? 2 : 0), PER = MF / MEM > 0 ? MF / MEM : 1;
#pragma unroll
      for (int k = 0; k < MEM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                              // PER MFMAs
        if (k < NF) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    // a fragment load (VMEM read)
        else if (k < NF + NS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // an activation read (DS read)
        else if (k == NF + NS) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);         // the copy's store (VMEM write)
        else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           // the copy's row read (DS read)
      }
      if (MF - PER * MEM > 0) __builtin_amdgcn_sched_group_barrier(0x008, MF - PER * MEM > 0 ? MF - PER * MEM : 1, 0);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) Ap[f] += 4 * 64;
#pragma unroll
    for (int n = 0; n < NS; ++n) xp[n] += 8;
  }
}

// The reduction of one layer for this wave: NF feature tiles (fragment streams A[f], 256 floats per k-group) x NS sample tiles
// (B rows xb[n] in LDS).  `k8` groups, a multiple of 4.  The A operands run two groups ahead of the MFMAs (L2 latency), the B
// operands one (LDS latency); `pre` holds the fragments of groups 0 and 1 on entry.  sched_barrier(0) pins "this group's loads,
// then the previous group's MFMAs": left alone, hipcc sinks every load to its use and waits for it there (vmcnt(0) eight MFMAs
// after the issue); with the order pinned its own counted waits (vmcnt(2 NF), lgkmcnt(NS)) are exactly right.
// The last groups fetch beyond the layer (the next feature tile's / layer's fragments, the sink behind the image; the next
// sample's columns in LDS): valid memory, never used -- branch-free.
// The HBM copy of the PREVIOUS layer's output (what X holds while this layer reduces over it) leaves from inside this loop:
// a wave stores whole 1 KB rows -- one ds_read_b128 + one global_store_dwordx4 per k-group, fully coalesced -- instead of the
// epilogue's 32-byte pieces (a 128 KB burst of those per layer and workgroup is store-issue bound: the first form measured ~12 K
// idle cycles per layer for it).  NROWS rows per wave (8 per sample tile of the slab) go out during the first NROWS groups.
template <int NF, int NS, int NROWS>
__device__ __forceinline__ void r2_reduce(floatx16 (&acc)[2][R2_SMAX], const float *const (&A)[2], const float *const (&xb)[R2_SMAX],
                                          int k8, floatx4 (&pre)[2][2], R2Stream &st) {
  floatx4 a[4][2], b[2][R2_SMAX];
  r2_gptr Ap[2] = {r2_g(A[0]) + 2 * 64, r2_g(A[1]) + 2 * 64};      // group g + 2 (64 float4 per group)
  r2_lptr xp[R2_SMAX];
#pragma unroll
  for (int n = 0; n < R2_SMAX; ++n) xp[n] = r2_l(xb[n]) + 2;       // group g + 1 (2 float4 per group)
#pragma unroll
  for (int f = 0; f < NF; ++f) { a[0][f] = pre[0][f]; a[1][f] = pre[1][f]; }
#pragma unroll
  for (int n = 0; n < NS; ++n) b[0][n] = *r2_l(xb[n]);
  floatx4 sv = {0.f, 0.f, 0.f, 0.f};
  if (NROWS > 0 && st.on && k8 >= NROWS) {
    r2_groups<NF, NS, true>(acc, a, b, Ap, xp, NROWS, st, sv);
    __builtin_amdgcn_sched_barrier(0);
    r2_stream_put(st, sv);
    st.on = false;
    if (k8 > NROWS) r2_groups<NF, NS, false>(acc, a, b, Ap, xp, k8 - NROWS, st, sv);
  } else {
    r2_groups<NF, NS, false>(acc, a, b, Ap, xp, k8, st, sv);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// a pending output copy that no reduction carried (behind the last main layer of a slab): all rows at once
__device__ __forceinline__ void r2_stream_flush(R2Stream &st, int n_rows) {
  if (!st.on) return;
  for (int i = 0; i < n_rows; ++i) {
    const floatx4 v = r2_stream_get(st);
    r2_stream_put(st, v);
  }
  st.on = false;
}

#ifdef FGS_RC2_PHASE_STAMPS
#define R2_T(x) (x) = __builtin_amdgcn_s_memtime()
#define R2_ACC(dst, t0) (dst) += __builtin_amdgcn_s_memtime() - (t0)
#else
#define R2_T(x) (void)0
#define R2_ACC(dst, t0) (void)0
#endif

struct R2Wave {
  unsigned long long t_load, t_init, t_red, t_epi;      // diagnostics (-DFGS_RC2_PHASE_STAMPS): shader cycles per phase
  int wave, lane, j, h;
  int64_t tile0;             // first sample tile of the slab
  int S;
  int64_t M;
};

// fragment stream of feature tile `ft` of layer L for this lane
__device__ __forceinline__ const float *r2_frag(const R2Args &a, const R2Layer &L, int ft, int lane) {
  return a.img + L.img + ((int64_t)ft * (L.k8x + L.k8e) * 64 + lane) * 4;
}

// which (feature tile, first sample tile, sample tiles) of a SIDE layer this wave computes in a slab of S tiles
__device__ __forceinline__ void r2_side_share(int n_ft, int S, int wave, int &ft, int &n0, int &ns) {
  if (n_ft >= 2) {
    ft = wave & 1;
    if (S >= 3) { n0 = 2 * (wave >> 1); ns = (S - n0 >= 2) ? 2 : (S - n0); }
    else { n0 = wave >> 1; ns = (n0 < S) ? 1 : 0; }
  } else {
    ft = 0; n0 = wave; ns = (wave < S) ? 1 : 0;
  }
}

// fragments of groups 0 and 1 of layer `l` (the ones its reduction starts from) for this wave; layer index beyond the chain: none
template <int NFT>
__device__ __forceinline__ void r2_prefetch(const R2Args &a, int l, const R2Wave &w, floatx4 (&pre)[2][2]) {
  if (l >= a.n_layers) return;
  const R2Layer &L = a.L[l];
  int ft0 = (NFT == 8 ? 2 : 1) * w.wave, ft1 = NFT == 8 ? ft0 + 1 : ft0;
  if (L.side) {
    int n0, ns;
    r2_side_share(L.n_ft, w.S, w.wave, ft0, n0, ns);
    ft1 = ft0;
  }
  const float *A0 = r2_frag(a, L, ft0, w.lane), *A1 = r2_frag(a, L, ft1, w.lane);
  pre[0][0] = r2_ldg(A0); pre[1][0] = r2_ldg(A0 + 256);
  pre[0][1] = r2_ldg(A1); pre[1][1] = r2_ldg(A1 + 256);
}

// NFT: feature tiles of the chain's main layers -- 8 (width 256: two per wave), 6 (192: one per wave + tiles 4, 5 dealt by
// (tile, sample tile) like a side layer: 2 S pairs over four waves) or 4 (128: one per wave).
template <bool BWD, int S, int NFT>
__device__ __forceinline__ void r2_slab(const R2Args &a, float *X, float *E, R2Wave &w) {
  constexpr int NF = NFT == 8 ? 2 : 1;
  constexpr bool REM = NFT == 6;
  const int tid = w.wave * 64 + w.lane;
  const int64_t row0 = w.tile0 * 32;
  unsigned long long tp = 0; (void)tp;
  R2_T(tp);
  // ---- the slab's input -> X (and the appended columns -> E).  Every wave has left the previous slab's last reduction (and its
  // own LDS writes -- the zero fill at kernel start -- have landed: s_barrier alone does not wait for them).
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  floatx4 pre[2][2];
  r2_prefetch<NFT>(a, 0, w, pre);
  {
    // 16 bytes per thread and step, row = idx / 64, column group = idx % 64; loads in batches of 8 (one exposed memory latency
    // per batch: issued one by one, each float4 waited ~1.5 K cycles for its own round trip -- 60 K cycles per slab)
    const int c4 = a.in0_cols >> 2, z4 = a.in0_zero_to >> 2;
    const int c = tid & 63;
    const bool c_in = c < c4, c_wr = c < z4;
    const int col = 4 * c;
    float4 keep = make_float4(1.f, 1.f, 1.f, 1.f);         // padding columns of the buffer may hold anything: select, not multiply
    const bool k0 = col < a.in0_valid, k1 = col + 1 < a.in0_valid, k2 = col + 2 < a.in0_valid, k3 = col + 3 < a.in0_valid;
    (void)keep;
    {      // ALL of the thread's 8 S rows (r0, r0 + 4, ...) are requested before the first one is written: one memory round trip
      const int r0 = tid >> 6;
      float4 v[8 * S];
#pragma unroll
      for (int k = 0; k < 8 * S; ++k) {
        int64_t row = row0 + r0 + 4 * k;
        if (row >= w.M) row = w.M - 1;                     // padding samples compute on a valid row; their results are dropped
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_in) v[k] = *reinterpret_cast<const float4 *>(a.in0 + row * a.ld_in0 + col);
      }
#pragma unroll
      for (int k = 0; k < 8 * S; ++k) {
        float4 o = v[k];
        o.x = k0 ? o.x : 0.f; o.y = k1 ? o.y : 0.f; o.z = k2 ? o.z : 0.f; o.w = k3 ? o.w : 0.f;
        if (c_wr) *reinterpret_cast<float4 *>(X + (r0 + 4 * k) * R2_P + col) = o;
      }
    }
    if (!BWD && a.ext) {
      const int e4 = a.ext_cols >> 2;
      const int ce = tid & 15;
      const bool e_in = ce < e4, e_wr = ce < R2_PE / 4;
      const int ecol = 4 * ce;
      const bool e0 = ecol < a.ext_valid, e1 = ecol + 1 < a.ext_valid, e2 = ecol + 2 < a.ext_valid, e3 = ecol + 3 < a.ext_valid;
      for (int r0 = tid >> 4; r0 < S * 32; r0 += 128) {     // 16 rows per step, 8 steps per batch
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = r0 + 16 * k;
          int64_t row = row0 + (r < S * 32 ? r : 0);
          if (row >= w.M) row = w.M - 1;
          v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e_in) v[k] = *reinterpret_cast<const float4 *>(a.ext + row * a.ld_ext + ecol);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = r0 + 16 * k;
          float4 o = v[k];
          o.x = e0 ? o.x : 0.f; o.y = e1 ? o.y : 0.f; o.z = e2 ? o.z : 0.f; o.w = e3 ? o.w : 0.f;
          if (e_wr && r < S * 32) *reinterpret_cast<float4 *>(E + r * R2_PE + ecol) = o;
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  R2_ACC(w.t_load, tp);

  // this lane's B rows (sample j of each tile, columns 4 h ..)
  const float *xrow[R2_SMAX], *erow[R2_SMAX];
#pragma unroll
  for (int n = 0; n < R2_SMAX; ++n) {
    xrow[n] = X + ((n < S ? n : 0) * 32 + w.j) * R2_P + 4 * w.h;
    erow[n] = E + ((n < S ? n : 0) * 32 + w.j) * R2_PE + 4 * w.h;
  }
  // rows of the slab this wave copies to HBM (row 4 i + wave, i < 8 S) that exist: a uniform count
  int st_rows = 0;
  {
    const int64_t left = w.M - row0 - w.wave;
    st_rows = left <= 0 ? 0 : (int)((left + 3) / 4 < 8 * S ? (left + 3) / 4 : 8 * S);
    st_rows = __builtin_amdgcn_readfirstlane(st_rows);
  }
  R2Stream st;
  st.on = false; st.i = 0; st.valid = 0; st.lds = r2_l(X); st.g = a.sink; st.sink = a.sink + 4 * w.lane; st.g_step = 0;

  for (int l = 0; l < a.n_layers; ++l) {
    const R2Layer &L = a.L[l];
    const int k8x = __builtin_amdgcn_readfirstlane(L.k8x), k8e = __builtin_amdgcn_readfirstlane(L.k8e);
    floatx16 acc[2][R2_SMAX];
    if (!L.side) {
      // ---------------------------------------------------------------- main layer: feature tiles NF w .. of all S sample tiles
      // (+ for 6 tiles: tile 4 + (w & 1) of the sample tiles r2_side_share deals to this wave)
      const int ftb = NF * w.wave;
      floatx16 accR[2][R2_SMAX];
      int ftr = 0, n0r = 0, nsr = 0;
      if (REM) {
        r2_side_share(2, S, w.wave, ftr, n0r, nsr);
        ftr += 4;
        n0r = __builtin_amdgcn_readfirstlane(n0r); nsr = __builtin_amdgcn_readfirstlane(nsr);
      }
      R2_T(tp);
      // accumulators start from the bias (row 8 (r >> 2) + 4 h + (r & 3) of each feature tile); no bias: the zeros behind the image
      {
        const float *bias = (!BWD && L.bias) ? L.bias : a.zeros;
#pragma unroll
        for (int f = 0; f < NF + (REM ? 1 : 0); ++f) {
          const int ft = f < NF ? ftb + f : ftr;
          floatx16 b0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const floatx4 bq = r2_ldg(bias + 32 * ft + 8 * q + 4 * w.h);
            b0[4 * q] = bq[0]; b0[4 * q + 1] = bq[1]; b0[4 * q + 2] = bq[2]; b0[4 * q + 3] = bq[3];
          }
          if (f < NF) {
#pragma unroll
            for (int n = 0; n < S; ++n) acc[f][n] = b0;
          } else {
            accR[0][0] = b0; accR[0][1] = b0;
          }
        }
      }
      const float *const A[2] = {r2_frag(a, L, ftb, w.lane), r2_frag(a, L, ftb + (NF == 2 ? 1 : 0), w.lane)};
      R2_ACC(w.t_init, tp); R2_T(tp);
      {
        const float *const xb[R2_SMAX] = {xrow[0], xrow[1], xrow[2], xrow[3]};
        r2_reduce<NF, S, 8 * S>(acc, A, xb, k8x, pre, st);
      }
      r2_stream_flush(st, 8 * S);      // (a reduction with fewer groups than rows to copy: not in the shipped chains)
      if (NFT == 8 && k8e > 0) {      // the appended columns: the fragment stream simply continues
        const float *const A2[2] = {A[0] + (int64_t)k8x * 256, A[1] + (int64_t)k8x * 256};
        floatx4 pre2[2][2] = {{r2_ldg(A2[0]), r2_ldg(A2[1])}, {r2_ldg(A2[0] + 256), r2_ldg(A2[1] + 256)}};
        const float *const eb[R2_SMAX] = {erow[0], erow[1], erow[2], erow[3]};
        r2_reduce<NF, S, 0>(acc, A2, eb, k8e, pre2, st);
      }
      if (REM && nsr > 0) {           // tiles 4, 5 of a 192-wide layer: this wave's (tile, sample tiles) share
        const float *Ar = r2_frag(a, L, ftr, w.lane);
        const float *const A2[2] = {Ar, Ar};
        floatx4 pre2[2][2] = {{r2_ldg(Ar), r2_ldg(Ar)}, {r2_ldg(Ar + 256), r2_ldg(Ar + 256)}};
        const float *x0 = X + (n0r * 32 + w.j) * R2_P + 4 * w.h;
        const float *const xr[R2_SMAX] = {x0, x0 + 32 * R2_P, x0, x0};
        if (nsr == 2) r2_reduce<1, 2, 0>(accR, A2, xr, k8x, pre2, st);
        else r2_reduce<1, 1, 0>(accR, A2, xr, k8x, pre2, st);
      }
      R2_ACC(w.t_red, tp); R2_T(tp);
      // every wave has read X for the last time in this layer
      __builtin_amdgcn_s_barrier();
      r2_prefetch<NFT>(a, l + 1, w, pre);
      __builtin_amdgcn_sched_barrier(0);
      // ---- epilogue, straight-line: ReLU / mask, sign bits, the new activations over X; their HBM copy leaves from inside the
      // next reduction (R2Stream), the sign bits here (one word per sample tile)
      const float lo = __builtin_amdgcn_readfirstlane(L.relu) ? 0.f : -INFINITY;
      // (a layer without bits: the words go to / come from one 256-byte spot of the sink / the zeros -- same instruction stream)
      const bool has_mw = !BWD && L.mask_w != nullptr, has_mr = BWD && L.mask_r != nullptr;
      unsigned *const mw = has_mw ? L.mask_w : reinterpret_cast<unsigned *>(a.sink);
      const unsigned *const mr = has_mr ? L.mask_r : reinterpret_cast<const unsigned *>(a.zeros);
      const unsigned all_ones = has_mr ? 0u : ~0u;
      int64_t mi[R2_SMAX];
      unsigned mb[R2_SMAX];
#pragma unroll
      for (int n = 0; n < S; ++n) {
        mi[n] = (has_mw || has_mr) ? ((w.tile0 + n) * 4 + w.wave) * 64 + w.lane : w.lane;
        mb[n] = BWD ? (mr[mi[n]] | all_ones) : 0u;
      }
#pragma unroll
      for (int n = 0; n < S; ++n) {
        unsigned bits = 0u;
        const bool rem_here = REM && n >= n0r && n < n0r + nsr;      // (uniform)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          if (f >= NF && !REM) { if (!BWD) bits <<= 16; continue; }   // (one tile per wave: the word's second half stays empty)
          floatx16 v;
          if (f < NF) v = acc[f][n];
          else {
            if (!rem_here) { if (!BWD) bits <<= 16; continue; }
            v = (n - n0r == 0) ? accR[0][0] : accR[0][1];
          }
          const int ft = f < NF ? ftb + f : ftr;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (!BWD) {
              v[r] = fmaxf(v[r], lo);
              // element e = 16 f + r of the word ends up at bit 31 - e: "v > 0" of a ReLU output is "bit pattern non-zero"
              bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(v[r]) + 0x7fffffffu, 31);
            } else {
              const int keep = (int)(mb[n] << (16 * f + r)) >> 31;
              v[r] = __uint_as_float(__float_as_uint(v[r]) & (unsigned)keep);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const floatx4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            *(__attribute__((address_space(3))) floatx4 *)r2_l(X + (n * 32 + w.j) * R2_P + 32 * ft + 8 * q + 4 * w.h) = o;
          }
        }
        if (!BWD) mw[mi[n]] = bits;
      }
      // the copy of this output: rows 4 i + wave of the slab, from X
      st.on = true; st.i = 0; st.valid = st_rows;
      st.lds = r2_l(X + w.wave * R2_P + 4 * w.lane);
      {   // (a row is 32 NFT floats: the lanes beyond it keep storing into the sink)
        const bool lane_ok = 4 * w.lane < 32 * NFT;
        st.g = lane_ok ? L.out + (row0 + w.wave) * L.ldo + 4 * w.lane : st.sink;
        st.g_step = lane_ok ? 4 * L.ldo : 0;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the LDS writes have landed before anybody is released
      __builtin_amdgcn_s_barrier();
      R2_ACC(w.t_epi, tp);
    } else {
      // ---------------------------------------------------------------- side layer: (feature tile, sample tiles) per wave
      int ft, n0, ns;
      r2_side_share(L.n_ft, S, w.wave, ft, n0, ns);
      ns = __builtin_amdgcn_readfirstlane(ns); n0 = __builtin_amdgcn_readfirstlane(n0);
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][n][r] = 0.f;
      const float *const A[2] = {r2_frag(a, L, ft, w.lane), r2_frag(a, L, ft, w.lane)};
      const float *x0 = X + (n0 * 32 + w.j) * R2_P + 4 * w.h;
      const float *const xb[R2_SMAX] = {x0, x0 + 32 * R2_P, x0, x0};
      if (ns == 2) r2_reduce<1, 2, 8 * S>(acc, A, xb, k8x, pre, st);
      else if (ns == 1) r2_reduce<1, 1, 8 * S>(acc, A, xb, k8x, pre, st);
      r2_stream_flush(st, 8 * S);
      r2_prefetch<NFT>(a, l + 1, w, pre);
      __builtin_amdgcn_sched_barrier(0);
      const int n_store = __builtin_amdgcn_readfirstlane(L.n_store);
      if (__builtin_amdgcn_readfirstlane(L.side) == 2) {
        // the 256 -> 3 output head (model/nerf.py:884): rgb = sigmoid(h . V^T + c).  Features 0 .. 3 of the tile are registers
        // 0 .. 3 of the lanes with h = 0: a lane stores its sample's n_store values, 12 contiguous bytes per lane
        const float4 bq = L.bias ? *reinterpret_cast<const float4 *>(L.bias) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          if (n < ns) {
            const int64_t gr = row0 + (n0 + n) * 32 + w.j;
            if (gr < w.M && w.h == 0) {
#pragma unroll
              for (int c = 0; c < 4; ++c)
                if (c < n_store) L.out[gr * L.ldo + c] = 1.f / (1.f + expf(-(acc[0][n][c] + bv[c])));
            }
          }
        }
        continue;
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        if (n < ns) {
          const int64_t gr = row0 + (n0 + n) * 32 + w.j;
          float *orow = (gr < w.M) ? L.out + gr * L.ldo : a.sink;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 32 * ft + 8 * q + 4 * w.h;
            if (col < n_store)
              *reinterpret_cast<float4 *>(orow + col) = make_float4(acc[0][n][4 * q], acc[0][n][4 * q + 1], acc[0][n][4 * q + 2],
                                                                   acc[0][n][4 * q + 3]);
          }
        }
      }
    }
  }
  r2_stream_flush(st, 8 * S);       // the last main layer's output (X is not touched again before the next slab's opening barrier)
}

template <bool BWD, int NFT>
__global__ __launch_bounds__(R2_THREADS, 1) void k_mlp_rc2(R2Args a) {
  __shared__ __attribute__((aligned(16))) float lds[R2_X_FLOATS + R2_E_FLOATS];
  float *X = lds, *E = lds + R2_X_FLOATS;
  const int tid = threadIdx.x;
  R2Wave w;
  w.lane = tid & 63;
  w.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  w.j = w.lane & 31; w.h = w.lane >> 5;
  w.t_load = w.t_init = w.t_red = w.t_epi = 0;
  w.M = fgs_rows(a.M, a.m_dev);
  const int64_t T = (w.M + 31) / 32;
  const int64_t t0 = (int64_t)blockIdx.x * T / gridDim.x, t1 = ((int64_t)blockIdx.x + 1) * T / gridDim.x;
  if (t0 >= t1) return;
  unsigned long long *const stamps = fgs_stamp_wg(a.stamps);
  if (stamps && tid == 0) {
    stamps[0] = __builtin_amdgcn_s_memtime();
    stamps[1] = __builtin_amdgcn_s_memrealtime();
  }
  // LDS holds finite values from here on: padded k-groups multiply whatever lies behind a row's data by zero weights
  for (int i = tid; i < (R2_X_FLOATS + R2_E_FLOATS) / 4; i += R2_THREADS)
    reinterpret_cast<float4 *>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t tb = t0; tb < t1;) {
    const int64_t left = t1 - tb;
    // slabs of 4, the rest split evenly when it is 5 or 6 (3 + 2, 3 + 3: two slabs of similar length keep the A stream's reuse)
    const int S = (int)(left >= 7 ? 4 : left >= 5 ? 3 : left);
    w.tile0 = tb; w.S = S;
    switch (S) {
      case 4: r2_slab<BWD, 4, NFT>(a, X, E, w); break;
      case 3: r2_slab<BWD, 3, NFT>(a, X, E, w); break;
      case 2: r2_slab<BWD, 2, NFT>(a, X, E, w); break;
      default: r2_slab<BWD, 1, NFT>(a, X, E, w); break;
    }
    tb += S;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stamps && tid == 0) {
    stamps[2] = __builtin_amdgcn_s_memtime();
    stamps[3] = __builtin_amdgcn_s_memrealtime();
    stamps[4] = w.t_init; stamps[5] = w.t_red; stamps[6] = w.t_epi; stamps[7] = w.t_load;
  }
}

bool r2_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
int r2_pad4(int v) { return (v + 3) / 4 * 4; }

// image geometry of one layer: (feature tiles, k-groups over X, k-groups over E)
void r2_geometry(int backward, const fgs_rc2_layer_t &U, int &n_ft, int &k8x, int &k8e) {
  const int rows = backward ? U.n_in : U.n_out, k = backward ? U.n_out : U.n_in;
  const int ext = backward ? 0 : U.ext_cols;
  n_ft = (rows + 31) / 32;
  if (ext > 0) { k8x = 32; k8e = r2_pad4((k - 256 + 7) / 8); }
  else { k8x = r2_pad4((k + 7) / 8); k8e = 0; }
}

}  // namespace

FGS_API int64_t fgs_mlp_rc2_image_floats(int backward, int n_layers, const fgs_rc2_layer_t *layers) {
  if (!layers || n_layers < 1 || n_layers > R2_MAXL) return -1;
  int64_t total = 0;
  for (int l = 0; l < n_layers; ++l) {
    int n_ft, k8x, k8e;
    r2_geometry(backward, layers[l], n_ft, k8x, k8e);
    total += (int64_t)n_ft * (k8x + k8e) * 256;
  }
  return total + R2_SINK_FLOATS + R2_ZERO_FLOATS;
}

namespace {
// validates a chain and fills the kernel arguments and the pack records of its layers (appended to p.L from p.n_layers on)
int r2_build(const char *who, int backward, int64_t M, int n_layers, const fgs_rc2_layer_t *layers, const float *in0, int64_t ld_in0,
             int in0_cols, float *image_ws, int64_t image_ws_floats, const fgs_dyn_t *dyn, R2Args &a, P2Args &p, int chain_slot,
             int &main_rows_out) {
  FGS_REQUIRE(n_layers >= 1 && n_layers <= R2_MAXL, FGS_E_RANGE, "%s: n_layers=%d (1..%d)", who, n_layers, R2_MAXL);
  FGS_REQUIRE(layers && image_ws && (reinterpret_cast<uintptr_t>(image_ws) & 15) == 0, FGS_E_INVALID, "%s: null / unaligned pointer", who);
  FGS_REQUIRE(in0_cols > 0 && in0_cols <= 256 && (in0_cols % 4) == 0 && (ld_in0 % 4) == 0 && ld_in0 >= in0_cols,
              FGS_E_INVALID, "%s: first input: 4..256 columns, multiple of 4", who);
  const int64_t need = fgs_mlp_rc2_image_floats(backward, n_layers, layers);
  FGS_REQUIRE(image_ws_floats >= need, FGS_E_INVALID, "%s: image workspace %lld floats, need %lld", who, (long long)image_ws_floats,
              (long long)need);
  a.M = M; a.m_dev = fgs_dyn_rows(dyn); a.n_layers = n_layers; a.img = image_ws; a.stamps = fgs_dyn_stamps(dyn);
  a.sink = image_ws + (need - R2_SINK_FLOATS - R2_ZERO_FLOATS);
  a.zeros = image_ws + (need - R2_ZERO_FLOATS);
  a.in0 = in0; a.ld_in0 = ld_in0; a.in0_cols = in0_cols;
  a.ext = nullptr; a.ld_ext = 0; a.ext_cols = 0; a.ext_valid = 0;
  p.zeros[chain_slot] = const_cast<float *>(a.zeros);
  int carried = in0_cols;            // columns of the carried input (what X holds)
  int main_rows = 0;                 // width of the chain's main layers
  int64_t base = 0;
  for (int l = 0; l < n_layers; ++l) {
    const fgs_rc2_layer_t &U = layers[l];
    FGS_REQUIRE(U.W && U.n_out > 0 && U.n_in > 0 && U.ldw >= U.n_in, FGS_E_INVALID, "%s: layer %d: bad weight", who, l);
    const int rows = backward ? U.n_in : U.n_out, k = backward ? U.n_out : U.n_in;
    const int ext_cols = backward ? 0 : U.ext_cols;
    FGS_REQUIRE(k <= carried + ext_cols && carried + ext_cols < k + 4, FGS_E_INVALID,
                "%s: layer %d reduces over %d columns but its input has %d (+%d appended)", who, l, k, carried, ext_cols);
    FGS_REQUIRE(ext_cols >= 0 && ext_cols <= R2_PE && (ext_cols % 4) == 0 &&
                    (ext_cols == 0 || (U.ext && !a.ext && carried == 256 && (U.ld_ext % 4) == 0 && r2_aligned16(U.ext) && !U.side)),
                FGS_E_INVALID, "%s: layer %d: appended columns need a full 256-column carried input, <= %d of them, "
                               "multiple of 4, aligned, at most one such layer", who, l, R2_PE);
    if (U.side == 2)
      FGS_REQUIRE(!backward && rows <= 4 && U.out && U.n_store >= 1 && U.n_store <= rows && U.ldo >= U.n_store && !U.relu &&
                      !U.mask_bits && (!U.bias || r2_aligned16(U.bias)), FGS_E_INVALID,
                  "%s: head layer %d: forward, <= 4 output columns, an output, 16-byte aligned bias", who, l);
    else if (U.side)
      FGS_REQUIRE(rows <= 64 && U.out && !U.bias && !U.relu && !U.mask_bits, FGS_E_INVALID,
                  "%s: side layer %d: <= 64 output columns, an output, no bias / activation", who, l);
    else
      FGS_REQUIRE((rows == 256 || rows == 192 || rows == 128) && (main_rows == 0 || rows == main_rows) && U.out &&
                      U.n_store == rows && (rows == 256 || ext_cols == 0), FGS_E_INVALID,
                  "%s: layer %d produces %d columns (stores %d): the main layers of one chain are all 256, "
                  "192 or 128 wide, their output is stored whole, appended columns need 256", who, l, rows, U.n_store);
    if (!U.side) main_rows = rows;
    FGS_REQUIRE(!U.out || U.side == 2 || ((U.ldo % 4) == 0 && r2_aligned16(U.out) && (U.n_store % 4) == 0 &&
                                           U.n_store <= (U.side ? 64 : 256) && U.ldo >= U.n_store),
                FGS_E_INVALID, "%s: layer %d: bad output", who, l);
    FGS_REQUIRE((!U.bias || r2_aligned16(U.bias)) && (!U.mask_bits || (reinterpret_cast<uintptr_t>(U.mask_bits) & 3) == 0),
                FGS_E_INVALID, "%s: layer %d: bias must be 16-byte aligned", who, l);
    R2Layer &L = a.L[l];
    r2_geometry(backward, U, L.n_ft, L.k8x, L.k8e);
    FGS_REQUIRE(L.n_ft >= 1 && L.k8x * 8 <= 256, FGS_E_RANGE, "%s: layer %d: reduction over %d columns", who, l, k);
    L.img = base;
    L.side = U.side == 2 ? 2 : (U.side ? 1 : 0);
    L.relu = backward ? 0 : U.relu;
    L.bias = backward ? nullptr : U.bias;      // (main layers and the head)
    L.mask_w = (!backward && U.relu) ? reinterpret_cast<unsigned *>(U.mask_bits) : nullptr;
    L.mask_r = backward ? reinterpret_cast<const unsigned *>(U.mask_bits) : nullptr;
    L.out = U.out; L.ldo = U.ldo; L.n_store = U.n_store;
    if (ext_cols) { a.ext = U.ext; a.ld_ext = U.ld_ext; a.ext_cols = ext_cols; a.ext_valid = k - carried; }
    if (l == 0) { a.in0_valid = ext_cols ? in0_cols : (k < in0_cols ? k : in0_cols); a.in0_zero_to = L.k8x * 8; }
    P2Layer &P = p.L[p.n_layers++];
    P.W = U.W; P.ldw = U.ldw; P.n_out = U.n_out; P.n_in = U.n_in; P.n_ft = L.n_ft; P.k8 = L.k8x + L.k8e;
    P.transpose = backward ? 1 : 0; P.dst = image_ws + base; P.f4_begin = p.f4_total;
    p.f4_total += (int64_t)L.n_ft * (L.k8x + L.k8e) * 64;
    base += (int64_t)L.n_ft * (L.k8x + L.k8e) * 256;
    if (!U.side) carried = rows;
  }
  if (a.in0_zero_to < in0_cols) a.in0_zero_to = in0_cols;
  FGS_REQUIRE(main_rows > 0, FGS_E_INVALID, "%s: a chain needs a main layer", who);
  main_rows_out = main_rows;
  return 0;
}

int r2_launch(int backward, int64_t M, int main_rows, const R2Args &a, hipStream_t st) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      cus <= 0)
    cus = 256;
  const int64_t T = (M + 31) / 32;
  const unsigned grid = (unsigned)(T < cus ? T : cus);
#define R2_LAUNCH(B, N) hipLaunchKernelGGL((k_mlp_rc2<B, N>), dim3(grid), dim3(R2_THREADS), 0, st, a)
  if (backward) { if (main_rows == 256) R2_LAUNCH(true, 8); else if (main_rows == 192) R2_LAUNCH(true, 6); else R2_LAUNCH(true, 4); }
  else { if (main_rows == 256) R2_LAUNCH(false, 8); else if (main_rows == 192) R2_LAUNCH(false, 6); else R2_LAUNCH(false, 4); }
#undef R2_LAUNCH
  FGS_LAUNCH_OK("fgs_mlp_rc2_chain");
  return 0;
}
}  // namespace

// prepacked != 0: image_ws already holds this chain's fragments (fgs_mlp_rc2_pack with the same layer list and unchanged weights):
// no pack launch.
FGS_API int fgs_mlp_rc2_chain(int backward, int64_t M, int n_layers, const fgs_rc2_layer_t *layers, const float *in0,
                              int64_t ld_in0, int in0_cols, float *image_ws, int64_t image_ws_floats, int prepacked,
                              const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && M < ((int64_t)1 << 31), FGS_E_RANGE, "fgs_mlp_rc2_chain: M=%lld", (long long)M);
  if (M == 0) return 0;
  FGS_REQUIRE(in0 && r2_aligned16(in0), FGS_E_INVALID, "fgs_mlp_rc2_chain: first input null / unaligned");
  R2Args a;
  P2Args p;
  p.n_layers = 0; p.f4_total = 0; p.zeros[0] = p.zeros[1] = nullptr;
  int main_rows = 0;
  if (int e = r2_build("fgs_mlp_rc2_chain", backward, M, n_layers, layers, in0, ld_in0, in0_cols, image_ws, image_ws_floats, dyn, a, p,
                       0, main_rows))
    return e;
  hipStream_t st = fgs_s(stream);
  if (!prepacked) {
    hipLaunchKernelGGL(k_rc2_pack, dim3(fgs_blocks(p.f4_total < 128 ? 128 : p.f4_total)), dim3(FGS_BLOCK), 0, st, p);
    FGS_LAUNCH_OK("fgs_mlp_rc2_chain (pack)");
  }
  return r2_launch(backward, M, main_rows, a, st);
}

// The weight images of TWO chains (a step's forward and backward chain; the second may be absent: n_layers_b = 0) in ONE launch,
// for fgs_mlp_rc2_chain(..., prepacked = 1).  Only W / ldw / n_out / n_in / side / ext_cols of the layers are read.
FGS_API int fgs_mlp_rc2_pack(int backward_a, int n_layers_a, const fgs_rc2_layer_t *layers_a, int in0_cols_a, float *image_ws_a,
                             int64_t image_ws_floats_a, int backward_b, int n_layers_b, const fgs_rc2_layer_t *layers_b,
                             int in0_cols_b, float *image_ws_b, int64_t image_ws_floats_b, fgs_stream_t stream) {
  P2Args p;
  p.n_layers = 0; p.f4_total = 0; p.zeros[0] = p.zeros[1] = nullptr;
  const int bw[2] = {backward_a, backward_b}, nl[2] = {n_layers_a, n_layers_b}, ic[2] = {in0_cols_a, in0_cols_b};
  const fgs_rc2_layer_t *ls[2] = {layers_a, layers_b};
  float *ws[2] = {image_ws_a, image_ws_b};
  const int64_t wf[2] = {image_ws_floats_a, image_ws_floats_b};
  for (int c = 0; c < 2; ++c) {
    if (nl[c] <= 0) continue;
    FGS_REQUIRE(nl[c] <= R2_MAXL && ls[c], FGS_E_INVALID, "fgs_mlp_rc2_pack: chain %d: bad layer list", c);
    // (the pack needs no outputs: validate on a copy whose outputs point at the image itself)
    fgs_rc2_layer_t tmp[R2_MAXL];
    for (int l = 0; l < nl[c]; ++l) {
      tmp[l] = ls[c][l];
      tmp[l].out = ws[c]; tmp[l].ldo = 256; tmp[l].n_store = tmp[l].side == 2 ? 1 : tmp[l].side ? 4 : (bw[c] ? tmp[l].n_in : tmp[l].n_out);
      tmp[l].bias = nullptr; tmp[l].mask_bits = nullptr; tmp[l].relu = 0;
      if (tmp[l].ext_cols && !tmp[l].ext) tmp[l].ext = ws[c];
      if (tmp[l].ext_cols && !tmp[l].ld_ext) tmp[l].ld_ext = 256;
    }
    R2Args a;
    int main_rows = 0;
    if (int e = r2_build("fgs_mlp_rc2_pack", bw[c], 1, nl[c], tmp, ws[c], ic[c], ic[c], ws[c], wf[c], nullptr, a, p, c, main_rows))
      return e;
  }
  if (p.f4_total == 0) return 0;
  hipLaunchKernelGGL(k_rc2_pack, dim3(fgs_blocks(p.f4_total < 128 ? 128 : p.f4_total)), dim3(FGS_BLOCK), 0, fgs_s(stream), p);
  FGS_LAUNCH_OK("fgs_mlp_rc2_pack");
  return 0;
}
