// gemm_f32.hip -- exact-fp32 matrix-core GEMM for the tiny MLPs (model/nerf.py:125-142: rgbnet / refnet).
//
// The reference runs these layers as cuBLAS SGEMMs via nn.Linear.  Here they are v_mfma_f32_32x32x2_f32
// (f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain, so the 1e-5 rel-L2 bar holds; gfx950 has
// no reduced-precision f32 path) in one kernel template serving the three products of a Linear layer:
//
//   FGS_GEMM_NT  forward        C[m,n]  = sum_k A[m,k] * B[n,k]   (+ bias[n]) (ReLU)        A = X, B = W[N,K]
//   FGS_GEMM_NN  data gradient  C[m,n]  = sum_k A[m,k] * B[k,n]   (* (mask[m,n] > 0))       A = dY, B = W[K,N]
//   FGS_GEMM_TN  weight grad    C[m,n] += sum_k A[k,m] * B[k,n]   (split-K, fp32 atomics)   A = dY, B = X
//
// Tiling for CDNA4: 256-thread workgroup = 4 wavefronts (one per SIMD), 128x128 output tile, each wave a
// 64x64 quadrant = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs), K consumed in chunks of 32 through a
// double-buffered LDS image (<= 73.7 KB, two workgroups per CU).  Operands that are contiguous along K are
// staged as [row][36] (144-byte rows: the four ds_read_b128 of a lane's 16-value K run are conflict-free);
// operands contiguous along the output index are staged as [k][128] and read with conflict-free
// ds_read_b32.  Lane half h owns k = 16h..16h+15 of a chunk for both operands, so one staged chunk feeds
// 16 MFMAs per tile with no cross-lane movement.  Workgroup ids are dealt so that the column tiles of one
// row tile land on the same XCD (ids b, b+8, ...) and share its L2.
#include "fgs_common.h"
#include <stdlib.h>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4n __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = 36;   // floats per staged row of a K-contiguous operand
constexpr int LDI = 128;  // floats per staged k-row of an index-contiguous operand
constexpr int TILE_FLOATS = 128 * LDK;  // >= 32 * LDI

struct GemmArgs {
  int64_t M, N, K;
  const int64_t *m_dev;       // fgs_dyn_t.row_count (NT / NN only): M is the capacity, the kernel clamps to *m_dev
  const float *A; int64_t lda;
  const float *B; int64_t ldb;
  float *C; int64_t ldc;
  const float *bias;          // NT: [N] or null
  int relu;                   // NT: apply max(.,0)
  const float *mask; int64_t ldm;   // NN: zero C where mask <= 0 (saved activation), or null
  float *colsum;              // NT/NN: if non-null, colsum[n] += sum_m C[m,n] (after the epilogue) -> bias gradients
  int64_t k_per_split;        // TN: reduction rows handled by one workgroup (multiple of BK)
  int tiles_m, tiles_n;
  FgsStamps stamps;           // fgs_dyn_t.stamps (k_gemm only): {~min start, max end} of the launch's workgroups, 100 MHz wall clock
};

enum { EPI_STORE = 0, EPI_ATOMIC = 1 };

// ---- global -> registers -> LDS staging ----------------------------------------------------------------
// K-contiguous operand: tile = 128 rows x 32 k.  Thread t: k4 = t & 7 (float4 along k), rows t>>3 + 32*p.
template <bool KC>
struct Stage {
  float4 v[4];
};

__device__ __forceinline__ float4 ld4_guard(const float *p, bool ok) {
  return ok ? *reinterpret_cast<const float4 *>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// CLAMP: the caller guarantees that every k of the chunk exists (K a multiple of BK); rows / indices beyond the operand are
// CLAMPED to its last row / last float4 instead of guarded -- what they feed are accumulator rows / columns the epilogue never
// stores.  No per-load test: a guarded load is a 64-bit vector compare, an exec-mask round trip through the scalar unit (which
// waits for the compare behind whatever MFMA is in the pipe) and a select, eight times per chunk.
template <bool KC>
__device__ __forceinline__ void stage_load_clamped(Stage<KC> &s, const float *__restrict__ base, int64_t ld, int64_t row0,
                                                   int64_t n_rows, int64_t k0, int tid) {
  if (KC) {
    const int k4 = tid & 7;
    const int last = (int)n_rows - 1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int r = (int)row0 + (tid >> 3) + 32 * p;
      r = r < last ? r : last;
      const floatx4n t = *reinterpret_cast<const floatx4n *>(base + (int64_t)r * ld + k0 + 4 * k4);
      s.v[p] = make_float4(t.x, t.y, t.z, t.w);      // (element-wise: a float4 struct copy kept the staging arrays in scratch)
    }
  } else {
    const int i4 = tid & 31;
    int i = (int)row0 + 4 * i4;
    i = i < (int)n_rows - 4 ? i : (int)n_rows - 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const floatx4n t = *reinterpret_cast<const floatx4n *>(base + (k0 + (tid >> 5) + 8 * p) * ld + i);
      s.v[p] = make_float4(t.x, t.y, t.z, t.w);
    }
  }
}

template <bool KC, bool CLAMP = false>
__device__ __forceinline__ void stage_load(Stage<KC> &s, const float *__restrict__ base, int64_t ld, int64_t row0,
                                           int64_t n_rows, int64_t k0, int64_t k_end, int tid) {
  if constexpr (CLAMP) {
    stage_load_clamped<KC>(s, base, ld, row0, n_rows, k0, tid);
    return;
  } else {
  if (KC) {
    // element (row0 + r, k0 + 4*k4 ..)
    const int k4 = tid & 7;
    const int64_t k = k0 + 4 * k4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = row0 + (tid >> 3) + 32 * p;
      s.v[p] = ld4_guard(base + r * ld + k, r < n_rows && k < k_end);
    }
  } else {
    // operand stored [k][index]: element (k0 + kk, row0 + 4*i4 ..)
    const int i4 = tid & 31;
    const int64_t i = row0 + 4 * i4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t k = k0 + (tid >> 5) + 8 * p;
      s.v[p] = ld4_guard(base + k * ld + i, k < k_end && i < n_rows);
    }
  }
  }
}

template <bool KC>
__device__ __forceinline__ void stage_store(const Stage<KC> &s, float *__restrict__ lds, int tid) {
  if (KC) {
    const int k4 = tid & 7;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      *reinterpret_cast<float4 *>(lds + ((tid >> 3) + 32 * p) * LDK + 4 * k4) = s.v[p];
  } else {
    const int i4 = tid & 31;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      *reinterpret_cast<float4 *>(lds + ((tid >> 5) + 8 * p) * LDI + 4 * i4) = s.v[p];
  }
}

// 16 operand values (k = 16h .. 16h+15 of the chunk) of tile row/col `idx` (0..127) for this lane.
template <bool KC>
__device__ __forceinline__ void frag_load(float (&f)[16], const float *__restrict__ lds, int idx, int h) {
  if (KC) {
    const float4 *p = reinterpret_cast<const float4 *>(lds + idx * LDK + 16 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = p[q];
      f[4 * q + 0] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) f[s] = lds[(16 * h + s) * LDI + idx];
  }
}

// Half of that run: the 8 values k = 16h + 8*half .. + 7 (one pipeline stage of the K loop).
template <bool KC>
__device__ __forceinline__ void half_load(float (&f)[8], const float *__restrict__ lds, int idx, int h, int half) {
  if (KC) {
    const float4 *p = reinterpret_cast<const float4 *>(lds + idx * LDK + 16 * h + 8 * half);
    const float4 u = p[0], v = p[1];
    f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w;
    f[4] = v.x; f[5] = v.y; f[6] = v.z; f[7] = v.w;
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = lds[(16 * h + 8 * half + s) * LDI + idx];
  }
}

typedef float LdsImage[2][2][TILE_FLOATS];  // [buffer][A|B]

__device__ __forceinline__ void acc_zero(floatx16 (&acc)[2][2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// acc += A[m0.., k_begin..k_end) * B[n0.., k_begin..k_end) for one 128x128 tile.  Leaves every wave past its last LDS read.
// Barrier for LDS hand-overs: waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for
// every global store the wave has issued to be acknowledged -- in the store epilogue that put the 16 row stores of a tile
// on the critical path of the column-sum reduction behind them: measured with per-workgroup wall-clock stamps, the
// epilogue of a data-gradient tile took 23 us on average (10..46) next to a 32 us main loop.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NARROW (N <= 64, chosen by the launcher): the four waves take 32 rows x 64 columns each (1 x 2 MFMA tiles) instead of 64 x 64
// quadrants -- half the MFMAs; in the 2 x 2 arrangement the two waves of the right-hand quadrants would multiply columns
// that do not exist (the 52-column encoding part of dZ ran at 41 % useful matrix work).
template <bool A_KC, bool B_KC, bool CLAMP = false, bool NARROW = false>
__device__ __forceinline__ void tile_mainloop(const GemmArgs &g, int64_t m0, int64_t n0, int64_t k_begin, int64_t k_end,
                                              floatx16 (&acc)[2][2], LdsImage &lds) {
  constexpr int RT = NARROW ? 1 : 2;          // A row tiles per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1, h = lane >> 5, l31 = lane & 31;

  // Software pipeline, three stages deep, one raw barrier per K chunk:
  //   global -> staging registers : chunk c+2 (issued at the top of iteration c, a full iteration to land)
  //   staging registers -> LDS    : chunk c+1 (top of iteration c, into the buffer chunk c-1 has vacated)
  //   LDS -> fragment registers   : the next HALF chunk (8 k-steps) is fetched while the current half's 32 MFMAs issue,
  //                                 so a wave's MFMA stream is not interrupted by LDS latency.
  // The barrier is s_waitcnt lgkmcnt(0) + s_barrier (no vmcnt wait: __syncthreads() would drain the prefetch).
  Stage<A_KC> sa;
  Stage<B_KC> sb;
  const int ia0 = NARROW ? wave * 32 + l31 : wave_m * 64 + l31, ib0 = NARROW ? l31 : wave_n * 64 + l31;
  const int n_chunks = (int)((k_end - k_begin + BK - 1) / BK);      // (32-bit: loop tests stay on the scalar unit)
  stage_load<A_KC, CLAMP>(sa, g.A, g.lda, m0, g.M, k_begin, k_end, tid);
  stage_load<B_KC, CLAMP>(sb, g.B, g.ldb, n0, g.N, k_begin, k_end, tid);
  stage_store<A_KC>(sa, lds[0][0], tid);
  stage_store<B_KC>(sb, lds[0][1], tid);
  if (n_chunks > 1) {
    stage_load<A_KC, CLAMP>(sa, g.A, g.lda, m0, g.M, k_begin + BK, k_end, tid);
    stage_load<B_KC, CLAMP>(sb, g.B, g.ldb, n0, g.N, k_begin + BK, k_end, tid);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  float f0a[2][8], f0b[2][8], f1a[2][8], f1b[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < RT) half_load<A_KC>(f0a[t], lds[0][0], ia0 + t * 32, h, 0);
    half_load<B_KC>(f0b[t], lds[0][1], ib0 + t * 32, h, 0);
  }
  for (int c = 0; c < n_chunks; ++c) {
    const int cur = c & 1;
    if (c + 1 < n_chunks) {
      stage_store<A_KC>(sa, lds[cur ^ 1][0], tid);
      stage_store<B_KC>(sb, lds[cur ^ 1][1], tid);
      if (c + 2 < n_chunks) {
        stage_load<A_KC, CLAMP>(sa, g.A, g.lda, m0, g.M, k_begin + (c + 2) * BK, k_end, tid);
        stage_load<B_KC, CLAMP>(sb, g.B, g.ldb, n0, g.N, k_begin + (c + 2) * BK, k_end, tid);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < RT) half_load<A_KC>(f1a[t], lds[cur][0], ia0 + t * 32, h, 1);
      half_load<B_KC>(f1b[t], lds[cur][1], ib0 + t * 32, h, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0a[i][s], f0b[j][s], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (c + 1 < n_chunks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t < RT) half_load<A_KC>(f0a[t], lds[cur ^ 1][0], ia0 + t * 32, h, 0);
        half_load<B_KC>(f0b[t], lds[cur ^ 1][1], ib0 + t * 32, h, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f1a[i][s], f1b[j][s], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  lds_barrier();  // every wave is past its last LDS fragment read before the epilogue reuses the image
}

// ---- epilogue: accumulator (reg r, lane) -> C[row, col]; col = lane & 31, row = (r&3) + 8*(r>>2) + 4*h
template <int EPI, bool NARROW = false>
__device__ __forceinline__ void tile_epilogue(const GemmArgs &g, int64_t m0, int64_t n0, floatx16 (&acc)[2][2],
                                              LdsImage &lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1, h = lane >> 5, l31 = lane & 31;
  if (EPI == EPI_ATOMIC) {
    // one atomic wave-instruction = two 128-byte row segments (the full-rate shape for global_atomic_add_f32)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wave_n * 64 + j * 32 + l31;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wave_m * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (col < g.N && row < g.M) atomicAdd(g.C + row * g.ldc + col, acc[i][j][r]);
        }
    }
    return;
  }
  // Stage the 128x128 tile through LDS (the K loop is done with it) so that bias / mask reads and the stores are
  // 16-byte accesses over whole 512-byte row segments instead of 64 scalar stores per lane.
  constexpr int LDC = 132;
  float *ct = &lds[0][0][0];  // 128 * 132 floats = 67.6 KB <= the 73.7 KB image
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < (NARROW ? 1 : 2); ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        ct[((NARROW ? wave * 32 : wave_m * 64 + i * 32) + (r & 3) + 8 * (r >> 2) + 4 * h) * LDC +
           (NARROW ? 0 : wave_n * 64) + j * 32 + l31] = acc[i][j][r];
  lds_barrier();
  const int c4 = tid & 31;
  const int64_t col = n0 + 4 * c4;
  const bool col_ok = col < g.N;  // N % 4 == 0 on every path that reaches here
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias && col_ok) bv = *reinterpret_cast<const float4 *>(g.bias + col);
  // Rows: one uniform test per tile (all 128 rows exist, true for every tile but the last row tile) instead of sixteen per-lane
  // ones -- each a 64-bit vector compare and an exec mask held in scalar registers across the loads: the guarded form alone
  // kept this kernel at 106 scalar registers with 156 more spilled.  Columns: one per-lane test around everything.
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool rows_full = m0 + BM <= g.M;
  if (col_ok) {
    float *crow = g.C + (m0 + (tid >> 5)) * g.ldc + col;
    const float *mrow = g.mask ? g.mask + (m0 + (tid >> 5)) * g.ldm + col : nullptr;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const int rl = (tid >> 5) + 8 * p;
      if (!rows_full && m0 + rl >= g.M) continue;
      float4 v = *reinterpret_cast<const float4 *>(ct + rl * LDC + 4 * c4);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      if (g.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (mrow) {
        const float4 mkp = *reinterpret_cast<const float4 *>(mrow + (int64_t)8 * p * g.ldm);
        if (!(mkp.x > 0.f)) v.x = 0.f;
        if (!(mkp.y > 0.f)) v.y = 0.f;
        if (!(mkp.z > 0.f)) v.z = 0.f;
        if (!(mkp.w > 0.f)) v.w = 0.f;
      }
      *reinterpret_cast<float4 *>(crow + (int64_t)8 * p * g.ldc) = v;
      cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
    }
  }
  if (g.colsum) {  // 8 threads (tid >> 5) share a column quad: reduce through LDS, one atomic per column
    lds_barrier();
    float *red = ct;  // [8][128]
    *reinterpret_cast<float4 *>(red + (tid >> 5) * 128 + 4 * c4) = cs;
    lds_barrier();
    if (tid < 128 && n0 + tid < g.N) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += red[q * 128 + tid];
      atomicAdd(g.colsum + n0 + tid, s);
    }
  }
}

// One workgroup per output tile (EPI_STORE) or per (tile, K slice) (EPI_ATOMIC).
template <bool A_KC, bool B_KC, int EPI, bool CLAMP = false, bool NARROW = false>
__device__ __forceinline__ void gemm_block(const GemmArgs &g, int b, LdsImage &lds) {
  // XCD-aware tile assignment: ids b, b+8, b+16.. (same XCD) walk the column tiles of one row tile
  int tile_m, tile_n;
  int64_t k_begin = 0, k_end = g.K;
  if (EPI == EPI_ATOMIC) {
    // split-K: few output tiles, many K slices.  Consecutive ids (dealt round-robin over the 8 XCDs) take the
    // tiles of one slice, so every XCD works and the 2-4 tiles sharing a slice of A / B run at the same time.
    const int tiles = g.tiles_m * g.tiles_n;
    const int split = b / tiles, tile = b - split * tiles;
    tile_m = tile % g.tiles_m;
    tile_n = tile / g.tiles_m;
    k_begin = (int64_t)split * g.k_per_split;
    k_end = k_begin + g.k_per_split < g.K ? k_begin + g.k_per_split : g.K;
    if (k_begin >= k_end) return;
  } else {
    const int grp = b / (8 * g.tiles_n);
    const int within = b - grp * 8 * g.tiles_n;
    tile_m = grp * 8 + (within & 7);
    tile_n = within >> 3;
    if (tile_m >= g.tiles_m || (int64_t)tile_m * BM >= g.M) return;
  }
  const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
  floatx16 acc[2][2];
  acc_zero(acc);
  tile_mainloop<A_KC, B_KC, CLAMP, NARROW>(g, m0, n0, k_begin, k_end, acc, lds);
  tile_epilogue<EPI, NARROW>(g, m0, n0, acc, lds);
}

// CLAMP (chosen by the launcher for store-epilogue products with K a multiple of 32 and, for an index-contiguous B, N a
// multiple of 4): operand loads without guards, see stage_load.  A kernel of its own: the guarded main loop keeps ~16 exec
// masks and 64-bit row bounds alive in scalar registers -- 106 used and 156 more spilled to vector lanes in the guarded
// instantiation -- and both loops in one kernel pushed the spills to scratch.
template <bool A_KC, bool B_KC, int EPI, bool CLAMP = false, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) LdsImage lds;
  if (EPI != EPI_ATOMIC) g.M = fgs_rows(g.M, g.m_dev);       // row tiles beyond the device-side count return at once
  unsigned long long *const stamps = fgs_stamp_base(g.stamps);
  if (stamps && threadIdx.x == 0) atomicMax(stamps, ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
  gemm_block<A_KC, B_KC, EPI, CLAMP, NARROW>(g, (int)blockIdx.x, lds);
  if (stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(stamps + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
}

// Backward of one Linear layer in ONE launch: the data-gradient product (NN, store epilogue with ReLU mask / column sums)
// and the weight-gradient product (TN, split-K atomics) both depend only on dY.  Launched separately each pays its own
// launch gap, first-chunk latency and partially filled last round of tiles (780 tiles on 512 slots = 1.52 rounds for the
// price of 2); launched together the split-K workgroups of the weight gradient and the data-gradient tiles share the
// rounds.
//
// Order and slice length matter (measured per layer at M = 49 920, 780 data tiles of ~43 us on 512 resident slots):
//   data tiles first, then 512 equal slices (13 chunks, ~65 us each)        179 us -- the slices start in rounds 2 and 3
//   192 long slices FIRST (33 chunks, ~145 us), the data tiles behind them   166 us -- the long workgroups start at t = 0
//     and the short data tiles pack into the other 320 slots (2.4 rounds); 160..224 slices are within 3 us of each other
//   one persistent workgroup per slot with its data tiles and a slice sized to even out the finish times: 170 us.
__global__ __launch_bounds__(256, 2) void k_linear_bwd(GemmArgs nn, GemmArgs tn, int nn_blocks, int tn_blocks, int tn_first) {
  __shared__ __attribute__((aligned(16))) LdsImage lds;
  const int b = (int)blockIdx.x;
  if (tn_first) {
    if (b < tn_blocks) gemm_block<false, false, EPI_ATOMIC>(tn, b, lds);
    else gemm_block<true, false, EPI_STORE>(nn, b - tn_blocks, lds);
  } else {
    if (b < nn_blocks) gemm_block<true, false, EPI_STORE>(nn, b, lds);
    else gemm_block<false, false, EPI_ATOMIC>(tn, b - nn_blocks, lds);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stream-K form of the EPI_STORE products -- OPT-IN (the caller passes a workspace), NOT the default: measured on MI355X it
// only ties with the tile-per-workgroup kernel (M = 49 920, K = N = 256: 83.0 vs 82.4 us; M = 16 384: 48 vs 29 us;
// M = 105 000, K = 192: 126 vs 114 us).  The balance it buys (ideal 51 us of MFMA time at M = 49 920 instead of two
// rounds) is eaten by what every extra tile SEGMENT costs here: ~2.5 us first-chunk latency, ~3 us LDS-staged epilogue,
// ~3 us of the 64 KB output burst, the 64 KB partial-tile hand-over and its agent-scope release / acquire.  It stays as
// the tested base for the next step (overlapping a segment's epilogue with the next segment's first loads).
// One tile's K loop is only n_chunks = K/32 (8 for the 256-wide layers) long and
// a launch has T = tiles_m * tiles_n tiles for 2 x 256 workgroup slots: T = 780 (M_s = 50 K) costs two full rounds, the
// second one 52 % occupied (measured: 88 us, the same as T = 1024).  Here the T * n_chunks chunk-units are cut into
// G = 2 x CUs equal contiguous ranges, one per resident workgroup.  A range covers the tail of one tile, zero or more
// whole tiles and the head of another.  The workgroup that holds a tile's FIRST chunk owns it: it adds the partial
// accumulators of the following workgroup(s) (parked in a per-workgroup 64 KB workspace slot, announced through an
// agent-scope release/acquire flag) and runs the ordinary bias / ReLU / mask / column-sum epilogue.  A non-owned partial is
// always the first thing a workgroup computes and never waits on anybody, so there is no circular wait, and with
// G <= resident capacity every producer is running.  The k-order of a split tile's sum is (head chunks) + (tail chunks)
// instead of one chain: fp32-level reassociation only.  Flags are reset by their consumer: the workspace stays clean.
struct StreamK {
  int units, n_chunks;  // total chunk-units (< 2^31), chunks per tile
  int G;                // workgroups
  float *ws;            // [G][64][256] partial accumulators in register layout
  int *flags;           // [G], zero outside a launch
};

__device__ __forceinline__ int sk_range_begin(const StreamK &s, int w) { return (int)(((int64_t)s.units * w) / s.G); }

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void k_gemm_sk(GemmArgs g, StreamK s) {
  __shared__ __attribute__((aligned(16))) LdsImage lds;
  // consecutive ranges on the same XCD (block ids b, b+8, ... share an XCD): neighbours share A row tiles in its L2 and
  // hand partial tiles over through it
  const int w = (blockIdx.x & 7) * (s.G >> 3) + (blockIdx.x >> 3);   // G is a multiple of 8
  const int tid = threadIdx.x;
  int u = sk_range_begin(s, w);
  const int u_end = sk_range_begin(s, w + 1);
  floatx16 acc[2][2];
  while (u < u_end) {
    const int tile = u / s.n_chunks, c0 = u - tile * s.n_chunks;
    const int c1 = (u_end - tile * s.n_chunks < s.n_chunks) ? u_end - tile * s.n_chunks : s.n_chunks;
    const int tile_m = tile / g.tiles_n, tile_n = tile - tile_m * g.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int64_t k_begin = (int64_t)c0 * BK, k_end = ((int64_t)c1 * BK < g.K) ? (int64_t)c1 * BK : g.K;
    acc_zero(acc);
    tile_mainloop<A_KC, B_KC>(g, m0, n0, k_begin, k_end, acc, lds);
    if (c0 != 0) {
      // not the owner: park the partial sums (register layout, coalesced) and announce them
      float *slot = s.ws + (int64_t)w * (64 * 256) + tid;
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        const floatx16 v = (q == 0) ? acc[0][0] : (q == 1) ? acc[0][1] : (q == 2) ? acc[1][0] : acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(q * 16 + r) * 256] = v[r];
      }
      // one agent-scope release per workgroup: the barrier orders every thread's stores before thread 0's release
      // (a per-thread __threadfence() costs an L2 write-back per wave and made split tiles 3x slower)
      __syncthreads();
      if (tid == 0) __hip_atomic_store(s.flags + w, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // owner: collect what the following workgroups hold of this tile
      int have = c1, wn = w;
      while (have < s.n_chunks) {
        ++wn;
        const int nb = sk_range_begin(s, wn), ne = sk_range_begin(s, wn + 1), tile_end = (tile + 1) * s.n_chunks;
        have += ((ne < tile_end) ? ne : tile_end) - nb;
        // poll with relaxed loads (an acquire load invalidates caches on every iteration), then ONE acquire fence
        if (tid == 0)
          while (__hip_atomic_load(s.flags + wn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float *slot = s.ws + (int64_t)wn * (64 * 256) + tid;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = slot[((i * 2 + j) * 16 + r) * 256];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += t[r];
          }
        __syncthreads();   // all four waves have read the slot
        if (tid == 0) __hip_atomic_store(s.flags + wn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      tile_epilogue<EPI_STORE>(g, m0, n0, acc, lds);
    }
    __syncthreads();  // the LDS image (epilogue staging) is free again
    u = tile * s.n_chunks + c1;
  }
}

template <bool A_KC, bool B_KC, int EPI>
int launch(const GemmArgs &g, unsigned splits, hipStream_t st) {
  const unsigned groups = (unsigned)((g.tiles_m + 7) / 8);
  dim3 grid(EPI == EPI_ATOMIC ? (unsigned)(g.tiles_m * g.tiles_n) * splits : groups * 8 * (unsigned)g.tiles_n, 1, 1);
  if (EPI == EPI_STORE && (g.K % BK) == 0 && (B_KC || ((g.N & 3) == 0 && g.N >= 4)) && g.M >= 1) {
    if (g.N <= 64) hipLaunchKernelGGL((k_gemm<A_KC, B_KC, EPI, EPI == EPI_STORE, EPI == EPI_STORE>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((k_gemm<A_KC, B_KC, EPI, EPI == EPI_STORE, false>), grid, dim3(256), 0, st, g);
  } else
    hipLaunchKernelGGL((k_gemm<A_KC, B_KC, EPI, false>), grid, dim3(256), 0, st, g);
  FGS_LAUNCH_OK("fgs_gemm_f32");
  return 0;
}

template <bool A_KC, bool B_KC>
int launch_sk(const GemmArgs &g, const StreamK &s, hipStream_t st) {
  hipLaunchKernelGGL((k_gemm_sk<A_KC, B_KC>), dim3((unsigned)s.G), dim3(256), 0, st, g, s);
  FGS_LAUNCH_OK("fgs_gemm_f32 (stream-K)");
  return 0;
}

int resident_slots() {  // 2 workgroups of k_gemm_sk per CU (73.7 KB LDS each, <= 128 VGPRs)
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0)
      cus = 256;
    slots = 2 * cus;
  }
  return slots;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// split the long reduction of a weight-gradient product so that ~2 workgroups per CU are in flight; returns the splits
// (measured on MI355X at M_s = 64 K, 256x256 outputs: 256-512 workgroups 96 us, 128: 159 us, 2048: 147 us)
unsigned setup_splitk(GemmArgs &g, int default_wgs = 512) {
  const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n;
  static const int env_wgs = getenv("FGS_TN_WGS") ? atoi(getenv("FGS_TN_WGS")) : 0;
  const int target_wgs = env_wgs > 0 ? env_wgs : default_wgs;
  int64_t want = (target_wgs + tiles - 1) / tiles;
  const int64_t chunks = (g.K + BK - 1) / BK;
  if (want > chunks) want = chunks;
  if (want < 1) want = 1;
  g.k_per_split = ((chunks + want - 1) / want) * BK;
  return (unsigned)((g.K + g.k_per_split - 1) / g.k_per_split);
}

GemmArgs make_args(int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B, int64_t ldb, float *C,
                   int64_t ldc, const float *bias, int relu, const float *mask, int64_t ldm, float *colsum) {
  GemmArgs g;
  g.M = M; g.N = N; g.K = K; g.m_dev = nullptr; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.bias = bias; g.relu = relu; g.mask = mask; g.ldm = ldm; g.colsum = colsum; g.k_per_split = 0;
  g.tiles_m = (int)((M + BM - 1) / BM);
  g.tiles_n = (int)((N + BN - 1) / BN);
  g.stamps = fgs_dyn_stamps(nullptr);
  return g;
}

}  // namespace

// Bytes of scratch the stream-K form wants: G partial-tile slots of 64 KB + G flags (the flag words must be ZERO when
// first handed in; the kernels leave them zero).
FGS_API int64_t fgs_gemm_workspace_bytes(void) { return (int64_t)resident_slots() * (64 * 256 * 4 + 4); }

FGS_API int fgs_gemm_f32(int op, int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B,
                         int64_t ldb, float *C, int64_t ldc, const float *bias, int relu, const float *mask, int64_t ldm,
                         float *colsum, void *workspace, int64_t workspace_bytes, const fgs_dyn_t *dyn, fgs_stream_t stream) {
  FGS_REQUIRE(op >= 0 && op <= 2, FGS_E_INVALID, "fgs_gemm_f32: op=%d", op);
  FGS_REQUIRE(M >= 0 && N >= 0 && K >= 0 && M < ((int64_t)1 << 31) && N < ((int64_t)1 << 31) && K < ((int64_t)1 << 31),
              FGS_E_RANGE, "fgs_gemm_f32: M=%lld N=%lld K=%lld", (long long)M, (long long)N, (long long)K);
  if (M == 0 || N == 0) return 0;
  if (K == 0 && op == FGS_GEMM_TN) return 0;
  FGS_REQUIRE(A && B && C, FGS_E_INVALID, "fgs_gemm_f32: null pointer");
  FGS_REQUIRE(aligned16(A) && aligned16(B) && (lda % 4) == 0 && (ldb % 4) == 0, FGS_E_INVALID,
              "fgs_gemm_f32: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  // vector loads walk K (K-contiguous operands) or the output index (index-contiguous operands) 4 at a time
  if (op == FGS_GEMM_NT) FGS_REQUIRE(K % 4 == 0 && N % 4 == 0, FGS_E_INVALID, "fgs_gemm_f32(NT): K and N must be multiples of 4");
  if (op != FGS_GEMM_TN)
    FGS_REQUIRE(aligned16(C) && (ldc % 4) == 0 && (!bias || aligned16(bias)) && (!mask || (aligned16(mask) && (ldm % 4) == 0)),
                FGS_E_INVALID, "fgs_gemm_f32: C / bias / mask must be 16-byte aligned with leading dimensions multiple of 4");
  if (op == FGS_GEMM_NN) FGS_REQUIRE(K % 4 == 0 && N % 4 == 0, FGS_E_INVALID, "fgs_gemm_f32(NN): K and N must be multiples of 4");
  if (op == FGS_GEMM_TN) FGS_REQUIRE(M % 4 == 0 && N % 4 == 0, FGS_E_INVALID, "fgs_gemm_f32(TN): M and N must be multiples of 4");

  GemmArgs g = make_args(M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldm, colsum);
  hipStream_t st = fgs_s(stream);
  // device-side row count (fgs_dyn_t.row_count): supported by the one-tile-per-workgroup NT / NN form (rows = M); the
  // split-K reduction over the rows (TN) and the stream-K grid take their partition from the host count
  if (fgs_dyn_rows(dyn)) {
    FGS_REQUIRE(op != FGS_GEMM_TN && !workspace, FGS_E_INVALID,
                "fgs_gemm_f32: TN / stream-K are not available under fgs_dyn_t.row_count (use fgs_mlp_wgrad)");
    FGS_REQUIRE(!colsum, FGS_E_INVALID, "fgs_gemm_f32: colsum is not available under fgs_dyn_t.row_count");
    g.m_dev = fgs_dyn_rows(dyn);
  }
  g.stamps = fgs_dyn_stamps(dyn);
  if (op != FGS_GEMM_TN && workspace) {
    // the caller asked for stream-K (by passing a workspace); used when it can balance: more than a handful of tiles,
    // a K loop worth cutting, every range non-empty
    const int sk_mode = 1;
    const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n, n_chunks = (K + BK - 1) / BK;
    StreamK s;
    const int64_t units = tiles * n_chunks;
    s.n_chunks = (int)n_chunks;
    s.units = (int)units;
    int64_t G = resident_slots();
    if (G > units / 2) G = units / 2;
    G &= ~(int64_t)7;   // a multiple of 8: whole ranges per XCD
    s.G = (int)G;
    if (sk_mode && tiles >= 32 && n_chunks >= 2 && G >= 16 && units < ((int64_t)1 << 30)) {
      FGS_REQUIRE(workspace_bytes >= fgs_gemm_workspace_bytes() && aligned16(workspace), FGS_E_INVALID,
                  "fgs_gemm_f32: workspace must be 16-byte aligned and hold fgs_gemm_workspace_bytes() bytes");
      s.ws = reinterpret_cast<float *>(workspace);   // slots first, the flag words behind ALL resident_slots() slots
      s.flags = reinterpret_cast<int *>(reinterpret_cast<char *>(workspace) + (int64_t)resident_slots() * (64 * 256 * 4));
      return op == FGS_GEMM_NT ? launch_sk<true, true>(g, s, st) : launch_sk<true, false>(g, s, st);
    }
  }
  switch (op) {
    case FGS_GEMM_NT: return launch<true, true, EPI_STORE>(g, 1, st);
    case FGS_GEMM_NN: return launch<true, false, EPI_STORE>(g, 1, st);
    default: {
      const unsigned splits = setup_splitk(g);
      return launch<false, false, EPI_ATOMIC>(g, splits, st);
    }
  }
}

// Backward of y = x W^T (+ b) for one Linear layer, both products in one launch (k_linear_bwd):
//   dX[M, K_in]      = (dY[M, N_out] . W[N_out, K_in]) * (mask[M, K_in] > 0)        (+ colsum[K_in] += column sums)
//   dW[N_out, K_in] += dY^T . X[M, K_in]                                            (fp32 atomics: dW zero-initialised)
FGS_API int fgs_linear_bwd_f32(int64_t M, int64_t N_out, int64_t K_in, const float *dY, int64_t lddy, const float *W,
                               int64_t ldw, const float *X, int64_t ldx, float *dX, int64_t lddx, const float *mask,
                               int64_t ldm, float *colsum, float *dW, int64_t lddw, fgs_stream_t stream) {
  FGS_REQUIRE(M >= 0 && N_out > 0 && K_in > 0 && M < ((int64_t)1 << 31) && N_out < ((int64_t)1 << 31) && K_in < ((int64_t)1 << 31),
              FGS_E_RANGE, "fgs_linear_bwd_f32: M=%lld N_out=%lld K_in=%lld", (long long)M, (long long)N_out, (long long)K_in);
  if (M == 0) return 0;
  FGS_REQUIRE(dY && W && X && dX && dW, FGS_E_INVALID, "fgs_linear_bwd_f32: null pointer");
  FGS_REQUIRE(aligned16(dY) && aligned16(W) && aligned16(X) && aligned16(dX) && (!mask || aligned16(mask)) && (lddy % 4) == 0 &&
                  (ldw % 4) == 0 && (ldx % 4) == 0 && (lddx % 4) == 0 && (!mask || (ldm % 4) == 0) && (N_out % 4) == 0 &&
                  (K_in % 4) == 0,
              FGS_E_INVALID, "fgs_linear_bwd_f32: operands must be 16-byte aligned; sizes and leading dimensions multiples of 4");
  GemmArgs nn = make_args(M, K_in, N_out, dY, lddy, W, ldw, dX, lddx, nullptr, 0, mask, ldm, colsum);
  GemmArgs tn = make_args(N_out, K_in, M, dY, lddy, X, ldx, dW, lddw, nullptr, 0, nullptr, 0, nullptr);
  static const int tn_first = getenv("FGS_TN_FIRST") ? atoi(getenv("FGS_TN_FIRST")) : 1;
  // weight-gradient slices: 192 long ones dispatched first (see k_linear_bwd); 512 short ones when they come last
  unsigned splits = setup_splitk(tn, tn_first ? 192 : 512);
  const unsigned tn_tiles = (unsigned)(tn.tiles_m * tn.tiles_n);
  if (tn_first && ((tn_tiles * splits) & 7u)) {   // keep the data tiles' XCD-aware numbering: a multiple of 8 in front
    while ((tn_tiles * splits) & 7u) ++splits;
    const int64_t chunks = (tn.K + BK - 1) / BK;
    tn.k_per_split = ((chunks + splits - 1) / splits) * BK;
  }
  const unsigned nn_blocks = (unsigned)((nn.tiles_m + 7) / 8) * 8 * (unsigned)nn.tiles_n;
  const unsigned tn_blocks = tn_tiles * splits;
  hipLaunchKernelGGL(k_linear_bwd, dim3(nn_blocks + tn_blocks), dim3(256), 0, fgs_s(stream), nn, tn, (int)nn_blocks,
                     (int)tn_blocks, tn_first);
  FGS_LAUNCH_OK("fgs_linear_bwd_f32");
  return 0;
}
