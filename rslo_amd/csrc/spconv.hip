// Sparse 3-D convolution arithmetic for gfx950: output-stationary implicit GEMM.
//
// A workgroup (4 wave64) owns a tile of 64*RB output rows, each wave 16*RB of them.  For every
// kernel offset k the wave gathers the neighbour rows named by the table straight into MFMA
// A-fragments (16-byte loads, four lanes cover one 64-byte row segment), W_k is staged once
// per workgroup in LDS and read as B-fragments, and v_mfma_f32_16x16x4_f32 accumulates the
// [16 x Cout] result in registers across all K offsets: one coalesced-ish store per output
// row, no atomics, no scatter.  Offsets for which none of the wave's 16 rows has a neighbour
// are skipped (about half of them on LiDAR data), so the MFMA work follows the rulebook
// density.  Bias and LeakyReLU are fused into the epilogue.
//
// K-slab trick: lane (i = lane&15, g = lane>>4) loads the float4 in[row_i][16j + 4g .. +3];
// slab (j,t) feeds element t of it as A[i][g], so the slab's k index g stands for input
// channel 16j+4g+t, and the B fragment is read from LDS at that same channel.  The
// contraction order differs from a plain loop only by a permutation of the channel sum.
#include <stdlib.h>

#include "rslo_common.h"
#include "wgrad_reduce.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SPC_THREADS 256
#define SPC_WAVES 4
#define SPC_MAXK 27

// XCD-aware tile order (MI355X: 8 XCDs, block b runs on XCD b % 8, each XCD has its own 4 MB L2).  Consecutive
// row tiles gather overlapping neighbour rows, so give every XCD one contiguous chunk of tiles: launch
// 8 * ceil(n/8) blocks, block b works on tile (b % 8) * ceil(n/8) + b / 8 (tiles >= n are idle).  A wrong
// guess about placement only costs speed.
__device__ __forceinline__ int64_t xcd_tile(int64_t n_tiles) {
  static const int NX = 8;
  const int64_t chunk = (n_tiles + NX - 1) / NX;
  return (int64_t)(blockIdx.x % NX) * chunk + blockIdx.x / NX;
}
static inline unsigned xcd_grid(int64_t n_tiles) { return (unsigned)(8 * ((n_tiles + 7) / 8)); }
// The same order over the first n_live of the launch's tiles (capacity-laid-out tables: the tiles past the live rows are
// padding): every XCD gets a contiguous eighth of the LIVE tiles, the surplus workgroups (-1) return at once.  Mapping the
// capacity instead leaves the XCDs that own the tail with nothing but padding (measured: 20 000 live rows of 40 000 took
// 89 us against 92).
__device__ __forceinline__ int64_t xcd_tile_live(int64_t n_live) {
  static const int NX = 8;
  const int64_t chunk = (n_live + NX - 1) / NX, j = blockIdx.x / NX;
  if (j >= chunk) return -1;
  const int64_t t = (int64_t)(blockIdx.x % NX) * chunk + j;
  return t < n_live ? t : -1;
}

template <int CIN_T>
struct AFrag {
  float v[CIN_T >= 16 ? CIN_T / 4 : 2];
};

// Loads the A fragments of one neighbour row.  cin is the true channel count.
template <int CIN_T>
__device__ __forceinline__ void load_a(const float *__restrict__ in, int32_t r, int cin, int g,
                                       AFrag<CIN_T> &a) {
  if constexpr (CIN_T >= 16) {
#pragma unroll
    for (int j = 0; j < CIN_T / 16; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r >= 0) v = *reinterpret_cast<const float4 *>(in + (int64_t)r * cin + 16 * j + 4 * g);
      a.v[4 * j + 0] = v.x;
      a.v[4 * j + 1] = v.y;
      a.v[4 * j + 2] = v.z;
      a.v[4 * j + 3] = v.w;
    }
  } else {  // CIN_T == 8: channels 4s+g, guarded
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = 4 * s + g;
      a.v[s] = (r >= 0 && c < cin) ? in[(int64_t)r * cin + c] : 0.f;
    }
  }
}

// LDS channel row of slab `s` for lane group g
template <int CIN_T>
__device__ __forceinline__ int slab_channel(int s, int g) {
  if constexpr (CIN_T >= 16)
    return 16 * (s >> 2) + 4 * g + (s & 3);
  else
    return 4 * s + g;
}

// out[o] = act(bias + sum_k in[nbr[o][k]] . B_k),  B_k[ci][co] = TRANS ? W[kk][co][ci] : W[kk][ci][co]
template <int CIN_T, int COUT_T, int RB, bool TRANS>
__global__ __launch_bounds__(SPC_THREADS) void k_spconv(const float *__restrict__ in, int cin,
                                                        const float *__restrict__ W,
                                                        const float *__restrict__ bias,
                                                        const int32_t *__restrict__ nbr, int64_t n_out,
                                                        int K, int cout, int flip_k, float slope,
                                                        float *__restrict__ out, const int32_t *__restrict__ n_live) {
  constexpr int LDW = COUT_T + 4;
  constexpr int NSLAB = CIN_T / 4;
  constexpr int NB = COUT_T / 16;
  constexpr int TILE = 64 * RB;
  __shared__ __attribute__((aligned(16))) float Wl[2][CIN_T * LDW];
  __shared__ int32_t nbl[TILE * SPC_MAXK];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int64_t n_tiles = (n_out + TILE - 1) / TILE;
  // (n_live: only the tiles that hold live rows of a capacity-laid-out table, rslo_spconv_set_live_rows)
  const int64_t tile_id = n_live ? xcd_tile_live(((int64_t)*n_live + TILE - 1) / TILE) : xcd_tile(n_tiles);
  if (tile_id < 0 || tile_id >= n_tiles) return;
  const int64_t row0 = tile_id * TILE;

  // neighbour rows of the tile: one contiguous slab of the table
  {
    const int64_t lim = (n_out - row0) * K;
    for (int e = tid; e < TILE * K; e += SPC_THREADS) nbl[e] = (e < lim) ? nbr[row0 * K + e] : -1;
  }

  auto stage = [&](int k, int buf) {
    const int kk = flip_k ? (K - 1 - k) : k;
    const float *wk = W + (int64_t)kk * cin * cout;
    for (int e = tid; e < CIN_T * COUT_T; e += SPC_THREADS) {
      const int ci = e / COUT_T, co = e - ci * COUT_T;
      float v = 0.f;
      if (ci < cin && co < cout) v = TRANS ? wk[co * cin + ci] : wk[ci * cout + co];
      Wl[buf][ci * LDW + co] = v;
    }
  };

  f32x4 acc[RB][NB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  __syncthreads();

  for (int k = 0; k < K; ++k) {
    const int buf = k & 1;
    if (k + 1 < K) stage(k + 1, buf ^ 1);  // other buffer: readers of it finished before the last barrier
    int32_t r[RB];
    bool any = false;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      r[rb] = nbl[((wid * RB + rb) * 16 + li) * K + k];
      any |= (__ballot(r[rb] >= 0) != 0ull);
    }
    if (any) {
      AFrag<CIN_T> a[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) load_a<CIN_T>(in, r[rb], cin, g, a[rb]);
      const float *wl = Wl[buf];
#pragma unroll
      for (int s = 0; s < NSLAB; ++s) {
        const int ch = slab_channel<CIN_T>(s, g);
        float b[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = wl[ch * LDW + nb * 16 + li];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb].v[s], b[nb], acc[rb][nb], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: D[row = 4g + j][col = li] per 16x16 block
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = nb * 16 + li;
      if (col >= cout) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t row = row0 + (wid * RB + rb) * 16 + 4 * g + j;
        if (row < n_out) {
          float v = acc[rb][nb][j] + bv;
          v = v > 0.f ? v : v * slope;
          out[row * cout + col] = v;
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------
// v3: the production kernel for channel counts that are multiples of 16.
//   * wave-granular: a wave owns 16*RBW output rows x all COUT_T channels and walks ONLY its active kernel
//     offsets (a 27-bit mask built once from the wave-private LDS copy of its neighbour rows), so the offset
//     loop has no data-dependent branch around the MFMAs and the accumulators stay in AGPRs;
//   * every operand load is 16 bytes: A rows by plain float4 gathers (lanes without a neighbour read row 0 and
//     zero the fragment; the b128 buffer-load builtin with its free bounds check mis-selects to a dword load on
//     this toolchain), B straight from the L2-resident weight tensor using two index
//     permutations: K-slab (j,t) <-> input channel 16j+4g+t, and column li of block nb <-> output channel
//     NB*li+nb, which makes both the forward (contiguous along Cout) and the transposed (contiguous along
//     Cin) weight reads float4 and turns the epilogue into one 16-byte store per lane and row;
//   * MFMA order is slab-major so consecutive MFMAs hit different accumulators (16x16x4 f32 needs >= 2).
// ---------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Tile setup shared by the wave-granular kernels (v3 / v6 / bf16): the ROWS output rows of a wave's tile -- taken through
// the optional row order (rslo_rulebook_row_order: rows with similar neighbour masks adjacent, so that a tile's union
// mask, i.e. the offsets it has to issue MFMAs for, shrinks from ~1.5x to ~1.15x of the real neighbour count) -- and
// their neighbour rows go into the wave's LDS slab.  Results do not depend on the order: an offset a row lacks
// contributes exact zeros.
template <int ROWS>
__device__ __forceinline__ void spc_load_tile(const int32_t *__restrict__ nbr, const int32_t *__restrict__ order,
                                              int64_t row0, int64_t n_out, int K, int lane,
                                              int32_t *__restrict__ nbl, int32_t *__restrict__ orow) {
  if (order == nullptr) {
    const int64_t lim = (n_out - row0) * K;
    for (int e = lane; e < ROWS * K; e += 64) nbl[e] = (e < lim) ? nbr[row0 * K + e] : -1;
    for (int r = lane; r < ROWS; r += 64) orow[r] = (row0 + r < n_out) ? (int32_t)(row0 + r) : -1;
  } else {
    for (int r = lane; r < ROWS; r += 64) orow[r] = (row0 + r < n_out) ? order[row0 + r] : -1;
    const int half = lane >> 5, k = lane & 31;     // two table rows per load instruction
#pragma unroll 4
    for (int r = 0; r < ROWS; r += 2) {
      const int64_t q = row0 + r + half;
      const int32_t src = (q < n_out) ? order[q] : -1;
      if (k < K) nbl[(r + half) * K + k] = (src >= 0) ? nbr[(int64_t)src * K + k] : -1;
    }
  }
}


template <int N>
struct VecF {
  float v[N];
};

template <int NB>
__device__ __forceinline__ VecF<NB> load_vec(const float *p) {
  VecF<NB> r;
  if constexpr (NB == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else if constexpr (NB == 2) {
    const float2 t = *reinterpret_cast<const float2 *>(p);
    r.v[0] = t.x; r.v[1] = t.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}

template <int CIN_T, int COUT_T, int RBW, bool TRANS>
__global__ __launch_bounds__(SPC_THREADS) void k_spconv_v3(const float *__restrict__ in,
                                                           const float *__restrict__ W,
                                                           const float *__restrict__ bias,
                                                           const int32_t *__restrict__ nbr,
                                                           const int32_t *__restrict__ order, int64_t n_out,
                                                           int K, int flip_k, float slope,
                                                           float *__restrict__ out, const int32_t *__restrict__ n_live) {
  constexpr int NJ = CIN_T / 16;
  constexpr int NB = COUT_T / 16;
  constexpr int ROWS = 16 * RBW;
  __shared__ int32_t nbl[SPC_WAVES][ROWS * SPC_MAXK];
  __shared__ int32_t orow[SPC_WAVES][ROWS];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int64_t n_tiles = (n_out + ROWS - 1) / ROWS;
  const int64_t n_blocks = (n_tiles + SPC_WAVES - 1) / SPC_WAVES;
  // (n_live: only the workgroups that hold live rows of a capacity-laid-out table, rslo_spconv_set_live_rows)
  const bool use_live = n_live != nullptr && order == nullptr;
  const int64_t vb = use_live ? xcd_tile_live(((int64_t)*n_live + SPC_WAVES * ROWS - 1) / (SPC_WAVES * ROWS)) : xcd_tile(n_blocks);
  if (vb < 0) return;
  const int64_t tile = vb * SPC_WAVES + wid;
  const bool active = vb < n_blocks && tile < n_tiles;
  const int64_t row0 = tile * ROWS;

  if (active) spc_load_tile<ROWS>(nbr, order, row0, n_out, K, lane, nbl[wid], orow[wid]);
  __syncthreads();
  if (!active) return;

  // active-offset mask of this wave
  unsigned mask = 0;
  for (int k = 0; k < K; ++k) {
    bool any = false;
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) any |= nbl[wid][(rb * 16 + li) * K + k] >= 0;
    if (__ballot(any) != 0ull) mask |= 1u << k;
  }
  mask = __builtin_amdgcn_readfirstlane(mask);

  f32x4 acc[RBW][NB];
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  while (mask) {
    const int k = __builtin_ctz(mask);
    mask &= mask - 1;
    const int kk = flip_k ? (K - 1 - k) : k;
    const float *wk = W + (int64_t)kk * CIN_T * COUT_T;
    const float *ap[RBW];
    bool ok[RBW];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
      const int32_t r = nbl[wid][(rb * 16 + li) * K + k];
      ok[rb] = r >= 0;                      // "no neighbour": read row 0 and zero the fragment
      ap[rb] = in + (int64_t)(ok[rb] ? r : 0) * CIN_T + 4 * g;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // A: channels 16j+4g .. +3 of the neighbour rows
      float a[RBW][4];
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) {
        const float4 t = *reinterpret_cast<const float4 *>(ap[rb] + 16 * j);
        a[rb][0] = ok[rb] ? t.x : 0.f;
        a[rb][1] = ok[rb] ? t.y : 0.f;
        a[rb][2] = ok[rb] ? t.z : 0.f;
        a[rb][3] = ok[rb] ? t.w : 0.f;
      }
      // B[slab t][nb]: input channel 16j+4g+t, output channel NB*li+nb
      float b[4][NB];
      if constexpr (TRANS) {   // wk[co * CIN_T + ci]: float4 along ci (= along t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float4 v = *reinterpret_cast<const float4 *>(wk + (NB * li + nb) * CIN_T + 16 * j + 4 * g);
          b[0][nb] = v.x; b[1][nb] = v.y; b[2][nb] = v.z; b[3][nb] = v.w;
        }
      } else {                 // wk[ci * COUT_T + co]: NB contiguous floats along co (= along nb)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const VecF<NB> v = load_vec<NB>(wk + (16 * j + 4 * g + t) * COUT_T + NB * li);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) b[t][nb] = v.v[nb];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][t], b[t][nb], acc[rb][nb], 0, 0, 0);
    }
  }

  // epilogue: lane (g, li) holds rows 4g+j, output channels NB*li .. NB*li+NB-1
  VecF<NB> bv;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bv.v[nb] = bias ? bias[NB * li + nb] : 0.f;
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = orow[wid][rb * 16 + 4 * g + j];
      if (row < 0) continue;
      float o[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v = acc[rb][nb][j] + bv.v[nb];
        o[nb] = v > 0.f ? v : v * slope;
      }
      float *dst = out + row * COUT_T + NB * li;
      if constexpr (NB == 4)
        *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      else if constexpr (NB == 2)
        *reinterpret_cast<float2 *>(dst) = make_float2(o[0], o[1]);
      else
        dst[0] = o[0];
    }
}

// ---------------------------------------------------------------------------------------
// v6: fp32-accurate sparse conv on the bf16 matrix cores (channel counts 32 / 64).
// gfx950 issues v_mfma_f32_16x16x4_f32 at 1/16 of the bf16 MFMA rate (2048 flop / 32 cycles against 16384 flop / ~17
// cycles for v_mfma_f32_16x16x32_bf16).  An fp32 number splits EXACTLY into three bf16 pieces by truncation
// (a = hi + mid + lo, 8 significant bits each); products of pieces are exact in fp32 and the matrix core accumulates
// in fp32, so   a*b = hh + (hm + mh) + (hl + mm + lh) + O(2^-24 |ab|)   -- six bf16 MFMAs with K = 32 replace
// eight fp32 MFMAs with K = 4 per 16x16x32 block: 2.5x fewer matrix-core cycles at fp32-level accuracy (the three
// dropped terms are below one fp32 ulp of the product).  Weights are split once per call by k_weight_split (which
// also transposes for the data gradient); activations are split in registers after the gather (2 and + 2 sub per
// value, v_perm to pack pairs).
// Operand layout of v_mfma_f32_16x16x32_bf16: lane (g = l>>4, li = l&15) holds A[row li][8 k-values of group g] and
// B[the same 8 k-values][col li]; k-value e of group g in K-step s is input channel 32 s + 8 g + e, so a lane gathers
// 32 contiguous bytes of its row per K-step.  C/D as in the fp32 form (col li, rows 4g..4g+3).
// ---------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Ws[plane][k][s][g][co][e] (bf16 bits), plane 0/1/2 = hi/mid/lo, input channel ci = 32 s + 8 g + e.
// src index: TRANSPOSE ? W[k][co_out = ci][..]: the conv computed is  out[:, co] = sum_ci in[:, ci] * M[ci][co]  with
// M = W[k] (forward) or W[k]^T (data gradient: ci runs over Cout of W, co over Cin of W).
__device__ __forceinline__ void weight_split_elem(const float *__restrict__ W, int K, int cin_op, int cout_op,
                                                  int transpose, unsigned short *__restrict__ Ws, int64_t i) {
  const int64_t n = (int64_t)K * cin_op * cout_op;
  if (i >= n) return;
  // i enumerates the destination (k, s, g, co, e)
  const int e = (int)(i & 7);
  int64_t r = i >> 3;
  const int co = (int)(r % cout_op);
  r /= cout_op;
  const int g = (int)(r & 3);
  r >>= 2;
  const int S = cin_op / 32;
  const int sk = (int)(r % S);
  const int k = (int)(r / S);
  const int ci = 32 * sk + 8 * g + e;
  // position co = 16 nb + li holds output channel NB li + nb (a lane's NB accumulator columns are then NB consecutive
  // channels: one 16-byte store per row in the epilogue), NB = cout_op / 16
  const int ch = (cout_op / 16) * (co & 15) + (co >> 4);
  const float w = transpose ? W[((int64_t)k * cout_op + ch) * cin_op + ci]      // W is [K, Cin_w = cout_op, Cout_w = cin_op]
                            : W[((int64_t)k * cin_op + ci) * cout_op + ch];
  rslo_split1(w, Ws[i], Ws[n + i], Ws[2 * n + i]);
}

__global__ void k_weight_split(const float *__restrict__ W, int K, int cin_op, int cout_op, int transpose,
                               unsigned short *__restrict__ Ws) {
  weight_split_elem(W, K, cin_op, cout_op, transpose, Ws, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// every MFMA-shaped layer of a model, both orientations, in one launch: grid (ceil(max_n / 256), 2 * n_layers); row
// 2 l + t writes layer l's operand for the forward (t = 0) or the data-gradient (t = 1) kernel call
__global__ void k_weight_split_many(const RsloWeightSplitDesc *__restrict__ desc) {
  const RsloWeightSplitDesc d = desc[blockIdx.y >> 1];
  const int t = blockIdx.y & 1;
  weight_split_elem(d.W, d.K, t ? d.cout : d.cin, t ? d.cin : d.cout, t, (unsigned short *)(t ? d.ws_dgrad : d.ws_fwd),
                    (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

struct Split8 {
  u32x4 h, m, l;
};

// 8 fp32 values (two float4) -> three bf16x8 operands
__device__ __forceinline__ Split8 split8(const float4 a, const float4 b, bool ok) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  Split8 o;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const RsloSplit2 s = rslo_split2(ok ? v[2 * p] : 0.f, ok ? v[2 * p + 1] : 0.f);
    o.h[p] = s.h;
    o.m[p] = s.m;
    o.l[p] = s.l;
  }
  return o;
}

#ifndef SPC_PREW_ALL
#define SPC_PREW_ALL 0
#endif
#define SPC_WLOAD(p) (*reinterpret_cast<const u32x4 *>(p))
#define MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

// KS = 2: the active offsets of a tile are divided between TWO waves (lowest half of the set bits / the rest), whose
// accumulators are added through LDS before the epilogue (first half + second half: a fixed order).  A launch of the
// level-2 / level-3 layers is ONE generation of waves (about 100k rows = 3.1 tiles of 32 rows per SIMD): each SIMD holds
// 3 or 4 waves from start to end, the kernel's time is the gather latency chain of the SIMDs that got 4, and nothing
// else is in flight to hide it.  Half-length chains in twice as many waves put 6.2 waves on a SIMD (quantisation 6.2 vs
// 7 instead of 3.1 vs 4) at the same weight and row traffic per product.
// SKIPB (round 6; always on for 32-row tiles): a 16-row block none of whose rows has the offset is skipped -- no gather, no
// split, no MFMAs for it (the tile's offset union is what the loop walks; 30 % of the issued products are such padding on C3,
// 42 % on C5).  A lone live block interleaves the accumulators of two column blocks instead of two row blocks.  A row's sum
// keeps its order (offsets ascending, the same six products): same bits.
template <int CIN_T, int COUT_T, int RBW, int KS = 1, bool SKIPB = false>
__global__ __launch_bounds__(SPC_THREADS, (SKIPB ? 4 : 1)) void k_spconv_v6(const float *__restrict__ in,
                                                           const unsigned short *__restrict__ Ws,
                                                           const float *__restrict__ bias,
                                                           const int32_t *__restrict__ nbr,
                                                           const int32_t *__restrict__ order, int64_t n_out,
                                                           int K, int flip_k, float slope,
                                                           float *__restrict__ out, const int32_t *__restrict__ n_live) {
  constexpr int NS = CIN_T / 32;       // K-steps of 32 input channels
  constexpr int NB = COUT_T / 16;      // 16-column blocks: column li of block nb = output channel NB li + nb
  constexpr int ROWS = 16 * RBW;
  constexpr int TPB = SPC_WAVES / KS;  // tiles per workgroup
  __shared__ int32_t nbl[TPB][ROWS * SPC_MAXK];
  __shared__ int32_t orow[TPB][ROWS];
  __shared__ __attribute__((aligned(16))) VecF<NB> part[KS > 1 ? TPB * (KS - 1) * RBW * 4 * 64 : 1];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int tw = wid / KS, half = wid % KS;
  const int64_t n_tiles = (n_out + ROWS - 1) / ROWS;
  const int64_t n_blocks = (n_tiles + TPB - 1) / TPB;
  // n_live (rslo_spconv_set_live_rows: a capacity-laid-out table whose live-row count stays on the device): only the
  // workgroups that hold live rows run, spread over the XCDs like a launch of that size; a workgroup of padding rows has
  // nothing to compute and nothing anyone reads to write
  const bool use_live = n_live != nullptr && order == nullptr;
  const int64_t vb = use_live ? xcd_tile_live(((int64_t)*n_live + TPB * ROWS - 1) / (TPB * ROWS)) : xcd_tile(n_blocks);
  if (vb < 0) return;
  const int64_t tile = vb * TPB + tw;
  const bool active = vb < n_blocks && tile < n_tiles;
  const int64_t row0 = tile * ROWS;

  if (active && half == 0) spc_load_tile<ROWS>(nbr, order, row0, n_out, K, lane, nbl[tw], orow[tw]);
  __syncthreads();
  if (KS == 1 && !active) return;

  unsigned mask = 0;
  if (active) {
    for (int k = 0; k < K; ++k) {
      bool any = false;
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) any |= nbl[tw][(rb * 16 + li) * K + k] >= 0;
      if (__ballot(any) != 0ull) mask |= 1u << k;
    }
  }
  mask = __builtin_amdgcn_readfirstlane(mask);
  if constexpr (KS > 1) {
    // part `half` owns the FIXED offsets k = half (mod KS): a row's sum is ((part 0's offsets in order) + (part 1's) ..)
    // whatever tile the row order puts it in -- results do not depend on the row order.  Interleaved, not ranges: the
    // mask-sorted order makes tiles of rows that lack the same side of the neighbourhood (a contiguous range of
    // offsets), which ranges would hand to one wave (measured: 161 vs 149 us on the 64 -> 64 level-2 layer)
    mask &= (KS == 2 ? 0x55555555u : 0x11111111u) << half;
  }

  f32x4 acc[RBW][NB];
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t plane = (int64_t)K * CIN_T * COUT_T;          // bf16 elements per plane
  while (mask) {
    const int k = __builtin_ctz(mask);
    mask &= mask - 1;
#ifdef SPC_EXPERIMENT_ALIAS_W      // scripts/variant_spconv.sh only: every offset reads offset 0's 24 KB slice (L1-resident) -- the
    const int kk = 0;              // timing upper bound of any weight-stationary scheme, results are wrong by construction
#else
    const int kk = flip_k ? (K - 1 - k) : k;
#endif
    const float *ap[RBW];
    bool ok[RBW];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
      const int32_t r = nbl[tw][(rb * 16 + li) * K + k];
      ok[rb] = r >= 0;
      ap[rb] = in + (int64_t)(ok[rb] ? r : 0) * CIN_T + 8 * g;
    }
    if constexpr (SKIPB && RBW == 2) {
      const bool l0 = __ballot(ok[0]) != 0ull, l1 = __ballot(ok[1]) != 0ull;
      if (!(l0 && l1)) {
        // exactly one live row block (the offset is in the mask: at least one is)
#define SPC6_ONE(R)                                                                                              \
  _Pragma("unroll") for (int sk = 0; sk < NS; ++sk) {                                                            \
    const unsigned short *wb = Ws + ((((int64_t)kk * NS + sk) * 4 + g) * COUT_T + li) * 8;                       \
    const float4 x0 = *reinterpret_cast<const float4 *>(ap[R] + 32 * sk);                                        \
    const float4 x1 = *reinterpret_cast<const float4 *>(ap[R] + 32 * sk + 4);                                    \
    const Split8 a1 = split8(x0, x1, ok[R]);                                                                     \
    _Pragma("unroll") for (int nb = 0; nb < NB; nb += 2) {                                                       \
      const u32x4 bh0 = SPC_WLOAD(wb + nb * 16 * 8), bh1 = SPC_WLOAD(wb + (nb + 1) * 16 * 8);                    \
      const u32x4 bm0 = SPC_WLOAD(wb + plane + nb * 16 * 8), bm1 = SPC_WLOAD(wb + plane + (nb + 1) * 16 * 8);    \
      const u32x4 bl0 = SPC_WLOAD(wb + 2 * plane + nb * 16 * 8), bl1 = SPC_WLOAD(wb + 2 * plane + (nb + 1) * 16 * 8); \
      acc[R][nb] = MFMA_BF16(a1.l, bh0, acc[R][nb]); acc[R][nb + 1] = MFMA_BF16(a1.l, bh1, acc[R][nb + 1]);      \
      acc[R][nb] = MFMA_BF16(a1.m, bm0, acc[R][nb]); acc[R][nb + 1] = MFMA_BF16(a1.m, bm1, acc[R][nb + 1]);      \
      acc[R][nb] = MFMA_BF16(a1.h, bl0, acc[R][nb]); acc[R][nb + 1] = MFMA_BF16(a1.h, bl1, acc[R][nb + 1]);      \
      acc[R][nb] = MFMA_BF16(a1.m, bh0, acc[R][nb]); acc[R][nb + 1] = MFMA_BF16(a1.m, bh1, acc[R][nb + 1]);      \
      acc[R][nb] = MFMA_BF16(a1.h, bm0, acc[R][nb]); acc[R][nb + 1] = MFMA_BF16(a1.h, bm1, acc[R][nb + 1]);      \
      acc[R][nb] = MFMA_BF16(a1.h, bh0, acc[R][nb]); acc[R][nb + 1] = MFMA_BF16(a1.h, bh1, acc[R][nb + 1]);      \
    }                                                                                                            \
  }
        if (l0) { SPC6_ONE(0) } else { SPC6_ONE(1) }
#undef SPC6_ONE
        continue;
      }
    }
#pragma unroll
    for (int sk = 0; sk < NS; ++sk) {
      // Ws[plane][kk][sk][g][co][8]: 16 bytes per lane, consecutive li -> consecutive 16 bytes
      const unsigned short *wb = Ws + ((((int64_t)kk * NS + sk) * 4 + g) * COUT_T + li) * 8;
      // one-row-block tiles (the small-problem tiling: a launch of a few waves per SIMD, nothing to hide latency behind):
      // ALL weight operands of the step are requested before the gathered rows are waited for -- one memory round trip
      // per step instead of one for the rows plus one per column block (C2: 26 -> ... us per 64 -> 64 layer)
      constexpr bool PREW = RBW == 1 || SPC_PREW_ALL;
      u32x4 pw[PREW ? NB : 1][3];
      if constexpr (PREW) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          pw[nb][0] = SPC_WLOAD(wb + nb * 16 * 8);
          pw[nb][1] = SPC_WLOAD(wb + plane + nb * 16 * 8);
          pw[nb][2] = SPC_WLOAD(wb + 2 * plane + nb * 16 * 8);
        }
      }
      Split8 a[RBW];
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) {
        const float4 x0 = *reinterpret_cast<const float4 *>(ap[rb] + 32 * sk);
        const float4 x1 = *reinterpret_cast<const float4 *>(ap[rb] + 32 * sk + 4);
        a[rb] = split8(x0, x1, ok[rb]);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const u32x4 bh = PREW ? pw[PREW ? nb : 0][0] : SPC_WLOAD(wb + nb * 16 * 8);
        const u32x4 bm = PREW ? pw[PREW ? nb : 0][1] : SPC_WLOAD(wb + plane + nb * 16 * 8);
        const u32x4 bl = PREW ? pw[PREW ? nb : 0][2] : SPC_WLOAD(wb + 2 * plane + nb * 16 * 8);
        // smallest terms first; consecutive MFMAs alternate between the row blocks' accumulators
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acc[rb][nb] = MFMA_BF16(a[rb].l, bh, acc[rb][nb]);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acc[rb][nb] = MFMA_BF16(a[rb].m, bm, acc[rb][nb]);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acc[rb][nb] = MFMA_BF16(a[rb].h, bl, acc[rb][nb]);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acc[rb][nb] = MFMA_BF16(a[rb].m, bh, acc[rb][nb]);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acc[rb][nb] = MFMA_BF16(a[rb].h, bm, acc[rb][nb]);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acc[rb][nb] = MFMA_BF16(a[rb].h, bh, acc[rb][nb]);
      }
    }
  }

  if constexpr (KS > 1) {      // parts 1 .. KS-1 -> LDS -> added to part 0 in part order (same lane, same registers)
    VecF<NB> *slot = part + tw * ((KS - 1) * RBW * 4 * 64);
    if (half > 0 && active) {
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          VecF<NB> v;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) v.v[nb] = acc[rb][nb][j];
          slot[((half - 1) * RBW * 4 + rb * 4 + j) * 64 + lane] = v;
        }
    }
    __syncthreads();
    if (half > 0 || !active) return;
#pragma unroll
    for (int h = 1; h < KS; ++h)
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const VecF<NB> v = slot[((h - 1) * RBW * 4 + rb * 4 + j) * 64 + lane];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[rb][nb][j] += v.v[nb];
        }
  }
  // epilogue: lane (g, li) holds rows 4g+j, output channels NB*li .. NB*li+NB-1
  VecF<NB> bv;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bv.v[nb] = bias ? bias[NB * li + nb] : 0.f;
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = orow[tw][rb * 16 + 4 * g + j];
      if (row < 0) continue;
      float o[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v = acc[rb][nb][j] + bv.v[nb];
        o[nb] = v > 0.f ? v : v * slope;
      }
      float *dst = out + row * COUT_T + NB * li;
      if constexpr (NB == 4)
        *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
      else
        *reinterpret_cast<float2 *>(dst) = make_float2(o[0], o[1]);
    }
}

// ---------------------------------------------------------------------------------------
// What bounds v6, measured (round 2; scripts/ablate_spconv.sh compiles parts of the kernel out; 64 -> 64, 8 frames,
// 166 us in that harness): no weight loads 120 us, no row gathers 97 us, neither 92 us (matrix cores + operand split),
// no operand split 158 us.  Two restructurings were built, verified bit-identical and measured, then removed:
//   * v7: every operand fetched one offset ahead (whole weight set of the current offset in registers, each
//     fragment refilled for the next offset right after its MFMAs; next offset's rows gathered before the current
//     offset's first MFMA; 2 waves per SIMD): 157.5 vs 159.6 us -- latency is not the limiter;
//   * v8: the weight operands through a two-slot LDS ring, staged once per workgroup and offset (one barrier per
//     offset, gathers prefetched): 232 us (64 -> 64), 84 vs 71 us (32 -> 32) -- the barrier couples four waves whose
//     tiles need different offsets, and the ring leaves 2 workgroups per CU;
//   * a non-temporal hint on the weight loads: 194 us (they do hit in the vector L1 today).
// Tile orders (scripts/bench_spconv.py ORDER=...): rows sorted by neighbour mask inside windows of 2048 cut the issued
// MFMAs by 17 % on 64 -> 64 but scatter the gathers (163 vs 160 us); Morton order 193 us; random 199 us: contiguous
// neighbour rows (raster order) matter more than fewer offsets.  The mask order IS a gain on the sparse transposed
// tables (inverse convolutions, strided data gradients: 8 of 27 offsets per row): 103 -> 72 us, 102 -> 75, 67 -> 51.
// ---------------------------------------------------------------------------------------
// W [K, Cin, Cout] -> Wt [K, Cout, Cin]: lets the data gradient run through the FORWARD kernel (dgrad = conv of
// dout with the per-offset transposed weights), whose weight reads are contiguous along the lane index.  The
// TRANS instantiations read W with a Cin*4-byte lane stride (64 cache lines per wave load) and measured
// 12-50 % slower on the 64-channel layers; the transpose is 442 KB per layer.
__global__ void k_weight_transpose(const float *__restrict__ W, int cin, int cout, float *__restrict__ Wt) {
  __shared__ float tile[32][33];
  const float *src = W + (size_t)blockIdx.z * cin * cout;
  float *dst = Wt + (size_t)blockIdx.z * cin * cout;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;     // c over cout, r over cin
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < cin && c < cout) ? src[(size_t)r * cout + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (c < cout && r < cin) dst[(size_t)c * cin + r] = tile[threadIdx.x][i];
  }
}

extern "C" int rslo_weight_transpose(const float *W, int K, int cin, int cout, float *Wt, void *stream) {
  RSLO_CHECK_ARG(W && Wt && K >= 1 && cin >= 1 && cout >= 1, "rslo_weight_transpose: bad arguments");
  dim3 grid((unsigned)rslo_cdiv(cout, 32), (unsigned)rslo_cdiv(cin, 32), (unsigned)K);
  hipLaunchKernelGGL(k_weight_transpose, grid, dim3(32, 8), 0, (hipStream_t)stream, W, cin, cout, Wt);
  RSLO_CHECK_LAUNCH("k_weight_transpose");
  return RSLO_OK;
}


// ---------------------------------------------------------------------------------------
// bf16 feature path (BASELINE config C4: bf16 features, 32-bit rulebook indices, fp32 accumulation).
// Same wave-granular structure as v6 without the operand split: the gathered bf16 rows ARE the MFMA operands
// (16 bytes per lane and K-step), weights are rounded once per call to bf16 in operand order (k_weight_bf16), the
// accumulators, bias and activation stay fp32, the output is rounded to bf16 (RNE).  One MFMA where v6 issues six:
// the kernel is bound by the row gather (half the bytes of fp32), not by the matrix cores.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__global__ void k_weight_bf16(const float *__restrict__ W, int K, int cin_op, int cout_op, int transpose,
                              unsigned short *__restrict__ Wb) {
  const int64_t n = (int64_t)K * cin_op * cout_op;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int e = (int)(i & 7);
  int64_t r = i >> 3;
  const int co = (int)(r % cout_op);
  r /= cout_op;
  const int g = (int)(r & 3);
  r >>= 2;
  const int S = cin_op / 32;
  const int sk = (int)(r % S);
  const int k = (int)(r / S);
  const int ci = 32 * sk + 8 * g + e;
  const int ch = (cout_op / 16) * (co & 15) + (co >> 4);
  const float w = transpose ? W[((int64_t)k * cout_op + ch) * cin_op + ci] : W[((int64_t)k * cin_op + ci) * cout_op + ch];
  Wb[i] = f32_to_bf16_rne(w);
}

// KS waves per tile as in k_spconv_v6 (offsets k = wave (mod KS), fp32 accumulators added through LDS in wave order)
template <int CIN_T, int COUT_T, int RBW, int KS = 1>
__global__ __launch_bounds__(SPC_THREADS) void k_spconv_bf16(const unsigned short *__restrict__ in,
                                                             const unsigned short *__restrict__ Wb,
                                                             const float *__restrict__ bias,
                                                             const int32_t *__restrict__ nbr,
                                                             const int32_t *__restrict__ order, int64_t n_out, int K,
                                                             int flip_k, float slope, unsigned short *__restrict__ out) {
  constexpr int NS = CIN_T / 32;
  constexpr int NB = COUT_T / 16;
  constexpr int ROWS = 16 * RBW;
  constexpr int TPB = SPC_WAVES / KS;  // tiles per workgroup
  __shared__ int32_t nbl[TPB][ROWS * SPC_MAXK];
  __shared__ int32_t orow[TPB][ROWS];
  __shared__ __attribute__((aligned(16))) VecF<NB> part[KS > 1 ? TPB * (KS - 1) * RBW * 4 * 64 : 1];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int64_t n_tiles = (n_out + ROWS - 1) / ROWS;
  const int tw = wid / KS, half = wid % KS;
  const int64_t n_blocks = (n_tiles + TPB - 1) / TPB;
  const int64_t vb = xcd_tile(n_blocks);
  const int64_t tile = vb * TPB + tw;
  const bool active = vb < n_blocks && tile < n_tiles;
  const int64_t row0 = tile * ROWS;

  if (active && half == 0) spc_load_tile<ROWS>(nbr, order, row0, n_out, K, lane, nbl[tw], orow[tw]);
  __syncthreads();
  if (KS == 1 && !active) return;

  unsigned mask = 0;
  if (active) {
    for (int k = 0; k < K; ++k) {
      bool any = false;
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) any |= nbl[tw][(rb * 16 + li) * K + k] >= 0;
      if (__ballot(any) != 0ull) mask |= 1u << k;
    }
  }
  mask = __builtin_amdgcn_readfirstlane(mask);
  if constexpr (KS > 1) mask &= (KS == 2 ? 0x55555555u : 0x11111111u) << half;

  f32x4 acc[RBW][NB];
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[rb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  while (mask) {
    const int k = __builtin_ctz(mask);
    mask &= mask - 1;
    const int kk = flip_k ? (K - 1 - k) : k;
    const unsigned short *ap[RBW];
    bool ok[RBW];
#pragma unroll
    for (int rb = 0; rb < RBW; ++rb) {
      const int32_t r = nbl[tw][(rb * 16 + li) * K + k];
      ok[rb] = r >= 0;
      ap[rb] = in + (int64_t)(ok[rb] ? r : 0) * CIN_T + 8 * g;
    }
#pragma unroll
    for (int sk = 0; sk < NS; ++sk) {
      u32x4 a[RBW];
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(ap[rb] + 32 * sk);
        a[rb] = ok[rb] ? v : (u32x4){0u, 0u, 0u, 0u};
      }
      const unsigned short *wb = Wb + ((((int64_t)kk * NS + sk) * 4 + g) * COUT_T + li) * 8;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const u32x4 b = *reinterpret_cast<const u32x4 *>(wb + nb * 16 * 8);
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) acc[rb][nb] = MFMA_BF16(a[rb], b, acc[rb][nb]);
      }
    }
  }

  if constexpr (KS > 1) {
    VecF<NB> *slot = part + tw * ((KS - 1) * RBW * 4 * 64);
    if (half > 0 && active) {
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          VecF<NB> v;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) v.v[nb] = acc[rb][nb][j];
          slot[((half - 1) * RBW * 4 + rb * 4 + j) * 64 + lane] = v;
        }
    }
    __syncthreads();
    if (half > 0 || !active) return;
#pragma unroll
    for (int h = 1; h < KS; ++h)
#pragma unroll
      for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const VecF<NB> v = slot[((h - 1) * RBW * 4 + rb * 4 + j) * 64 + lane];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[rb][nb][j] += v.v[nb];
        }
  }
  VecF<NB> bv;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bv.v[nb] = bias ? bias[NB * li + nb] : 0.f;
#pragma unroll
  for (int rb = 0; rb < RBW; ++rb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = orow[tw][rb * 16 + 4 * g + j];
      if (row < 0) continue;
      unsigned short o[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float v = acc[rb][nb][j] + bv.v[nb];
        o[nb] = f32_to_bf16_rne(v > 0.f ? v : v * slope);
      }
      unsigned short *dst = out + row * COUT_T + NB * li;
      if constexpr (NB == 4)
        *reinterpret_cast<uint2 *>(dst) = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
      else
        *reinterpret_cast<unsigned *>(dst) = (unsigned)o[0] | ((unsigned)o[1] << 16);
    }
}

// k_spconv_v6 tiling: rows per tile = 16 rbw (1, 2, 4), waves per tile ks (1, 2, 4; rbw 4 only with ks 1); 0 = choose
#define spc_force_rbw rslo_tune(RSLO_TUNE_SPCONV_RBW)
#define spc_force_ks rslo_tune(RSLO_TUNE_SPCONV_KS)

extern "C" int rslo_weight_to_bf16(const float *W, int K, int cin_op, int cout_op, int transpose, void *Wb, void *stream) {
  RSLO_CHECK_ARG(W && Wb && K >= 1, "rslo_weight_to_bf16: bad arguments");
  RSLO_CHECK_ARG((cin_op == 32 || cin_op == 64) && (cout_op == 32 || cout_op == 64),
                 "rslo_weight_to_bf16: channel counts must be 32 or 64");
  const int64_t n = (int64_t)K * cin_op * cout_op;
  hipLaunchKernelGGL(k_weight_bf16, dim3((unsigned)rslo_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, W, K, cin_op,
                     cout_op, transpose, (unsigned short *)Wb);
  RSLO_CHECK_LAUNCH("k_weight_bf16");
  return RSLO_OK;
}

extern "C" int rslo_spconv_fwd_bf16(const void *in, int cin, const void *Wb, const float *bias, const int32_t *nbr,
                                    const int32_t *row_order, int64_t n_out, int K, int cout, int flip_k,
                                    float act_slope, void *out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG((cin == 32 || cin == 64) && (cout == 32 || cout == 64), "spconv_fwd_bf16: channels must be 32 or 64");
  RSLO_CHECK_ARG(K >= 1 && K <= SPC_MAXK, "spconv_fwd_bf16: K must be in 1..27");
  if (n_out == 0) return RSLO_OK;
  const unsigned short *x = (const unsigned short *)in, *w = (const unsigned short *)Wb;
  unsigned short *o = (unsigned short *)out;
  // rslo_spconv_set_tiling applies here too.  Measured (profiles/r02_spconv_offset_split.txt): the bf16 kernel moves half
  // the bytes per product and gains less from a second wave per tile -- 64 -> 64 level 3 35.9 -> 32.0 us (16-row tiles),
  // level 2 67.0 -> 65.8 us (32-row tiles), 32 -> 32 loses (27.4 -> 29.7 us)
  const int rbw = spc_force_rbw == 1 || spc_force_rbw == 2 ? spc_force_rbw : ((n_out >= 256 * 32 * 8) ? 2 : 1);
  const int ks = spc_force_ks ? spc_force_ks : ((cin == 32 && cout == 32) ? 1 : 2);
#define SPCB_LAUNCH(CI, CO, RB, KSv)                                                                           \
  hipLaunchKernelGGL((k_spconv_bf16<CI, CO, RB, KSv>), dim3(xcd_grid(rslo_cdiv(rslo_cdiv(n_out, 16 * RB), 4 / KSv))), \
                     dim3(SPC_THREADS), 0, st, x, w, bias, nbr, row_order, n_out, K, flip_k, act_slope, o)
#define SPCB_CASE(CI, CO)                                                                                   \
  if (cin == CI && cout == CO) {                                                                            \
    if (rbw == 2 && ks == 4) SPCB_LAUNCH(CI, CO, 2, 4);                                                     \
    else if (rbw == 2 && ks == 2) SPCB_LAUNCH(CI, CO, 2, 2);                                                \
    else if (rbw == 2) SPCB_LAUNCH(CI, CO, 2, 1);                                                           \
    else if (ks == 4) SPCB_LAUNCH(CI, CO, 1, 4);                                                            \
    else if (ks == 2) SPCB_LAUNCH(CI, CO, 1, 2);                                                            \
    else SPCB_LAUNCH(CI, CO, 1, 1);                                                                         \
  }
  SPCB_CASE(32, 32) SPCB_CASE(32, 64) SPCB_CASE(64, 32) SPCB_CASE(64, 64)
#undef SPCB_LAUNCH
#undef SPCB_CASE
  RSLO_CHECK_LAUNCH("spconv_bf16");
  return RSLO_OK;
}

extern "C" size_t rslo_weight_split_bytes(int K, int cin, int cout) { return (size_t)3 * K * cin * cout * sizeof(unsigned short); }

extern "C" int rslo_weight_split(const float *W, int K, int cin_op, int cout_op, int transpose, void *Ws, void *stream) {
  RSLO_CHECK_ARG(W && Ws && K >= 1, "rslo_weight_split: bad arguments");
  RSLO_CHECK_ARG((cin_op == 32 || cin_op == 64) && (cout_op == 32 || cout_op == 64),
                 "rslo_weight_split: channel counts must be 32 or 64");
  const int64_t n = (int64_t)K * cin_op * cout_op;
  hipLaunchKernelGGL(k_weight_split, dim3((unsigned)rslo_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, W, K, cin_op,
                     cout_op, transpose, (unsigned short *)Ws);
  RSLO_CHECK_LAUNCH("k_weight_split");
  return RSLO_OK;
}

extern "C" int rslo_weight_split_many(const RsloWeightSplitDesc *desc_dev, int n_layers, int64_t max_weight_elems,
                                      void *stream) {
  RSLO_CHECK_ARG(desc_dev && n_layers > 0 && n_layers < 32768 && max_weight_elems > 0, "rslo_weight_split_many: bad sizes");
  hipLaunchKernelGGL(k_weight_split_many, dim3((unsigned)rslo_cdiv(max_weight_elems, 256), (unsigned)(2 * n_layers)),
                     dim3(256), 0, (hipStream_t)stream, desc_dev);
  RSLO_CHECK_LAUNCH("k_weight_split_many");
  return RSLO_OK;
}

// Live-row count of the NEXT forward launch (rslo_spconv_fwd / rslo_spconv_fwd_split), consumed by it: see include/rslo_hip.h
static thread_local const int32_t *g_spc_live_rows = nullptr;      // per thread: set and consumed by the same caller
extern "C" void rslo_spconv_set_live_rows(const int32_t *n_live_dev) { g_spc_live_rows = n_live_dev; }
static inline const int32_t *spc_take_live() {
  const int32_t *p = g_spc_live_rows;
  g_spc_live_rows = nullptr;
  return p;
}

extern "C" void rslo_spconv_set_tiling(int rbw, int ks) {
  g_rslo_tune[RSLO_TUNE_SPCONV_RBW] = (rbw == 1 || rbw == 2 || rbw == 4) ? rbw : 0;
  g_rslo_tune[RSLO_TUNE_SPCONV_KS] = (ks == 1 || ks == 2 || ks == 4) ? ks : 0;
}

extern "C" int rslo_spconv_fwd_split(const float *in, int cin, const void *Ws, const float *bias, const int32_t *nbr,
                                     const int32_t *row_order, int64_t n_out, int K, int cout, int flip_k,
                                     float act_slope, float *out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  const int32_t *live = spc_take_live();
  RSLO_CHECK_ARG((cin == 32 || cin == 64) && (cout == 32 || cout == 64), "spconv_fwd_split: channels must be 32 or 64");
  RSLO_CHECK_ARG(K >= 1 && K <= SPC_MAXK, "spconv_fwd_split: K must be in 1..27");
  if (n_out == 0) return RSLO_OK;
  const unsigned short *ws = (const unsigned short *)Ws;
  const int force_rbw = spc_force_rbw, force_ks = spc_force_ks;
  int rbw = force_rbw ? force_rbw : ((n_out >= 256 * 32 * 8) ? 2 : 1);
  // measured per layer shape (profiles/r02_spconv_offset_split.txt): two waves per 32-row tile win everywhere except
  // 32 -> 32 (the cheapest products per gathered row: the LDS hand-over costs more than the shorter chain saves)
  int ks = force_ks ? force_ks : ((cin == 32 && cout == 32) ? 1 : 2);
  if (!force_rbw && ks == 2) rbw = 2;
  // Small problems (a single frame: BASELINE config C2, levels of 2.6 k .. 18 k rows): even with two waves per tile most
  // SIMDs hold no wave at all and a launch lasts as long as ONE tile's chain of ~11 offsets x (weights + gather ->
  // split -> MFMA).  16-row tiles shared by four waves quarter the chain: whole forward pass of one frame 0.83 ->
  // 0.68 ms (scripts/bench_encoder.py c2; R1K2 0.69, R2K4 0.67, R1K1 0.96).  Above ~20 k rows the LDS hand-over of four
  // waves costs more than the shorter chain returns (profiles/r02_spconv_offset_split.txt).
  if (!force_rbw && !force_ks && n_out < 20000) {
    rbw = 1;
    ks = 4;
  }
#define SPC6_LAUNCH(CI, CO, RB, KSv)                                                                        \
  hipLaunchKernelGGL((k_spconv_v6<CI, CO, RB, KSv>), dim3(xcd_grid(rslo_cdiv(rslo_cdiv(n_out, 16 * RB), 4 / KSv))), \
                     dim3(SPC_THREADS), 0, st, in, ws, bias, nbr, row_order, n_out, K, flip_k, act_slope, out, live)
#define SPC6_LAUNCH_SKIP(CI, CO, KSv)                                                                       \
  hipLaunchKernelGGL((k_spconv_v6<CI, CO, 2, KSv, true>), dim3(xcd_grid(rslo_cdiv(rslo_cdiv(n_out, 32), 4 / KSv))), \
                     dim3(SPC_THREADS), 0, st, in, ws, bias, nbr, row_order, n_out, K, flip_k, act_slope, out, live)
  // measured (profiles/r06_spconv_skip.txt, 8 frames, us without | with, the skipping kernel compiled for 4 waves per SIMD like the
  // other one): 64 -> 64 level 2 155.3 | 152.6, level 3 70.7 | 67.5, strided 64 -> 64 72.4 | 66.5, inverse 64 -> 64 94.2 | 84.3,
  // inverse 64 -> 32 96.0 | 93.3, 32 -> 32 71.2 | 68.4, strided 32 -> 64 66.7 | 62.0.  The 32-row kernels without the skip are
  // deleted (their bits: tests/golden/kernel_bits.json).
  // tilings kept: 32-row tiles by two waves or one, 16-row tiles by four waves (the small-problem tiling).  The other
  // combinations of (row blocks, waves) were measured and closed in rounds 2-5 (profiles/NOTES.md): (4,1) / (4,2) / (4,4)
  // register-bound, (2,4) / (1,2) / (1,1) never ahead; a forced value outside the kept set takes the nearest kept one.
  if (rbw == 4) rbw = 2;
  if (rbw == 2 && ks == 4) ks = 2;
  if (rbw == 1) ks = 4;
#define SPC6_CASE(CI, CO)                                                                                    \
  if (cin == CI && cout == CO) {                                                                             \
    if (rbw == 2 && ks == 2) SPC6_LAUNCH_SKIP(CI, CO, 2);                                                    \
    else if (rbw == 2) SPC6_LAUNCH_SKIP(CI, CO, 1);                                                          \
    else SPC6_LAUNCH(CI, CO, 1, 4);                                                                          \
  }
  SPC6_CASE(32, 32) SPC6_CASE(32, 64) SPC6_CASE(64, 32) SPC6_CASE(64, 64)
#undef SPC6_CASE
#undef SPC6_LAUNCH
#undef SPC6_LAUNCH_SKIP
  RSLO_CHECK_LAUNCH("spconv_v6");
  return RSLO_OK;
}

static int pad_cin(int c) { return c <= 8 ? 8 : (c <= 16 ? 16 : (c <= 32 ? 32 : 64)); }
static int pad_cout(int c) { return c <= 16 ? 16 : (c <= 32 ? 32 : 64); }

template <bool TRANS>
static int launch_spconv(const float *in, int cin, const float *W, const float *bias, const int32_t *nbr,
                         const int32_t *order, int64_t n_out, int K, int cout, int flip_k, float slope, float *out,
                         hipStream_t st) {
  const int32_t *live = spc_take_live();
  RSLO_CHECK_ARG(cin >= 1 && cin <= 64 && cout >= 1 && cout <= 64, "spconv: channels must be in 1..64");
  RSLO_CHECK_ARG(K >= 1 && K <= SPC_MAXK, "spconv: K must be in 1..27");
  RSLO_CHECK_ARG(cin <= 8 || cin % 4 == 0, "spconv: cin > 8 must be a multiple of 4");
  if (n_out == 0) return RSLO_OK;
  const int ci = pad_cin(cin), co = pad_cout(cout);
  RSLO_CHECK_ARG(ci == 8 || ci == cin, "spconv: cin must be <=8, 16, 32 or 64");
  const int variant = rslo_tune(RSLO_TUNE_SPCONV_V);
  if ((variant == 0 || variant >= 100) && ci == cin && co == cout && cin % 16 == 0 && cout % 16 == 0) {
    // v3 (variant 100 + RBW forces the row blocking; default: 2 row blocks when the launch still fills the chip)
    int rbw = variant >= 100 ? variant - 100 : ((n_out >= 256 * 32 * 8) ? 2 : 1);
#define SPC3_CASE(CI, CO)                                                                                  \
    if (ci == CI && co == CO) {                                                                            \
      if (rbw == 2)                                                                                        \
        hipLaunchKernelGGL((k_spconv_v3<CI, CO, 2, TRANS>), dim3(xcd_grid(rslo_cdiv(rslo_cdiv(n_out, 32), 4))), \
                           dim3(SPC_THREADS), 0, st, in, W, bias, nbr, order, n_out, K, flip_k, slope, out, live); \
      else                                                                                                 \
        hipLaunchKernelGGL((k_spconv_v3<CI, CO, 1, TRANS>), dim3(xcd_grid(rslo_cdiv(rslo_cdiv(n_out, 16), 4))), \
                           dim3(SPC_THREADS), 0, st, in, W, bias, nbr, order, n_out, K, flip_k, slope, out, live); \
    }
    SPC3_CASE(16, 16) SPC3_CASE(16, 32) SPC3_CASE(16, 64)
    SPC3_CASE(32, 16) SPC3_CASE(32, 32) SPC3_CASE(32, 64)
    SPC3_CASE(64, 16) SPC3_CASE(64, 32) SPC3_CASE(64, 64)
#undef SPC3_CASE
    RSLO_CHECK_LAUNCH("spconv_v3");
    return RSLO_OK;
  }
  // one 16-row block per wave while the launch would not fill the chip, two otherwise
  const bool rb2 = (n_out >= 256 * 128 * 2) && co >= 32;
#define SPC_CASE(CI, CO)                                                                             \
  if (ci == CI && co == CO) {                                                                        \
    if (rb2)                                                                                         \
      hipLaunchKernelGGL((k_spconv<CI, CO, 2, TRANS>), dim3(xcd_grid(rslo_cdiv(n_out, 128))),        \
                         dim3(SPC_THREADS), 0, st, in, cin, W, bias, nbr, n_out, K, cout, flip_k,    \
                         slope, out, live);                                                          \
    else                                                                                             \
      hipLaunchKernelGGL((k_spconv<CI, CO, 1, TRANS>), dim3(xcd_grid(rslo_cdiv(n_out, 64))),         \
                         dim3(SPC_THREADS), 0, st, in, cin, W, bias, nbr, n_out, K, cout, flip_k,    \
                         slope, out, live);                                                          \
  }
  SPC_CASE(8, 16) SPC_CASE(8, 32) SPC_CASE(8, 64)
  SPC_CASE(16, 16) SPC_CASE(16, 32) SPC_CASE(16, 64)
  SPC_CASE(32, 16) SPC_CASE(32, 32) SPC_CASE(32, 64)
  SPC_CASE(64, 16) SPC_CASE(64, 32) SPC_CASE(64, 64)
#undef SPC_CASE
  RSLO_CHECK_LAUNCH("spconv");
  return RSLO_OK;
}

extern "C" int rslo_spconv_fwd(const float *in, int cin, const float *W, const float *bias,
                               const int32_t *nbr, const int32_t *row_order, int64_t n_out, int K, int cout,
                               int flip_k, float act_slope, float *out, void *stream) {
  return launch_spconv<false>(in, cin, W, bias, nbr, row_order, n_out, K, cout, flip_k, act_slope, out,
                              (hipStream_t)stream);
}

// din[i][a] = sum_k sum_b dout[nbrT[i][k]][b] W[kk][a][b]: the same kernel with the roles of the
// channel counts swapped and W_k read transposed.
extern "C" int rslo_spconv_dgrad(const float *dout, int cout, const float *W, const int32_t *nbrT,
                                 const int32_t *row_order, int64_t n_in, int K, int cin, int flip_k, float *din,
                                 void *stream) {
  RSLO_CHECK_ARG(cout <= 8 || cout % 4 == 0, "spconv_dgrad: cout > 8 must be a multiple of 4");
  return launch_spconv<true>(dout, cout, W, nullptr, nbrT, row_order, n_in, K, cin, flip_k, 1.0f, din,
                             (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// wgrad: dW[k] = sum_o in[nbr[o][k]]^T dout[o].  Grid (chunk, k): the workgroup compacts the
// valid (in,out) pairs of its row chunk for offset k (ballot + prefix, deterministic), then
// the four waves take 4-pair groups round-robin and accumulate Cin x Cout in MFMA registers
// (A = in^T: lane (ci, g) reads in[pair g][ci]; B = dout: lane (g, co)).  Wave partials are
// summed through LDS and written to ws[chunk][k]; a second kernel reduces the chunks in a
// fixed order -- no atomics, bit-reproducible.
// ---------------------------------------------------------------------------------------
#define WG_CHUNK 2048

template <int CIN_T, int COUT_T>
__global__ __launch_bounds__(SPC_THREADS) void k_wgrad(const float *__restrict__ in, int cin,
                                                       const float *__restrict__ dout, int cout,
                                                       const int32_t *__restrict__ nbr, int64_t n_out, int K,
                                                       float *__restrict__ ws, float *__restrict__ ws_bias) {
  constexpr int CB = CIN_T / 16, NB = COUT_T / 16;
  __shared__ int32_t p_in[WG_CHUNK], p_out[WG_CHUNK];
  __shared__ int32_t wcnt[SPC_WAVES];
  __shared__ int32_t base_s;
  __shared__ __attribute__((aligned(16))) float red[CIN_T * COUT_T];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int k = blockIdx.y;
  const int64_t o0 = (int64_t)blockIdx.x * WG_CHUNK;
  const int64_t o1 = (o0 + WG_CHUNK < n_out) ? o0 + WG_CHUNK : n_out;

  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int64_t ob = o0; ob < o1; ob += SPC_THREADS) {
    const int64_t o = ob + tid;
    const int32_t r = (o < o1) ? nbr[o * K + k] : -1;
    const unsigned long long m = __ballot(r >= 0);
    if (lane == 0) wcnt[wid] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wid; ++w) off += wcnt[w];
    if (r >= 0) {
      const int p = off + __popcll(m & ((1ull << lane) - 1ull));
      p_in[p] = r;
      p_out[p] = (int32_t)(o - 0);
    }
    __syncthreads();
    if (tid == 0) base_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  const int np = base_s;

  f32x4 acc[CB][NB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int q = wid * 4; q < np; q += SPC_WAVES * 4) {
    const int p = q + g;
    const bool ok = p < np;
    const int32_t ri = ok ? p_in[p] : 0;
    const int64_t ro = ok ? (int64_t)p_out[p] : 0;
    float a[CB], b[NB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const int c = cb * 16 + li;
      a[cb] = (ok && c < cin) ? in[(int64_t)ri * cin + c] : 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int c = nb * 16 + li;
      b[nb] = (ok && c < cout) ? dout[ro * cout + c] : 0.f;
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb], b[nb], acc[cb][nb], 0, 0, 0);
  }

  // fixed-order reduction of the 4 wave partials through LDS
  for (int w = 0; w < SPC_WAVES; ++w) {
    if (wid == w) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ci = cb * 16 + 4 * g + j, co = nb * 16 + li;
            float v = acc[cb][nb][j];
            if (w) v += red[ci * COUT_T + co];
            red[ci * COUT_T + co] = v;
          }
    }
    __syncthreads();
  }
  float *dst = ws + ((int64_t)blockIdx.x * K + k) * cin * cout;
  for (int e = tid; e < cin * cout; e += SPC_THREADS) {
    const int ci = e / cout, co = e - ci * cout;
    dst[e] = red[ci * COUT_T + co];
  }
  if (k == 0 && ws_bias) {  // bias gradient partial: column sums of dout over the chunk
    __syncthreads();
    const int c = tid & 63, part = tid >> 6;
    float s = 0.f;
    if (c < cout)
      for (int64_t o = o0 + part; o < o1; o += 4) s += dout[o * cout + c];
    red[tid] = s;
    __syncthreads();
    if (part == 0 && c < cout)
      ws_bias[(int64_t)blockIdx.x * cout + c] = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
  }
}

__global__ void k_wgrad_reduce(const float *__restrict__ ws, int nchunk, int64_t n, float *__restrict__ dW) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += ws[(int64_t)c * n + e];
  dW[e] = s;
}

static int wgrad_chunks(int64_t n_out) { return (int)rslo_cdiv(n_out > 0 ? n_out : 1, WG_CHUNK); }

extern "C" size_t rslo_spconv_wgrad_ws_bytes(int64_t n_out, int K, int cin, int cout) {
  return (size_t)wgrad_chunks(n_out) * ((size_t)K * cin * cout + cout) * sizeof(float);
}

extern "C" int rslo_spconv_wgrad(const float *in, int cin, const float *dout, int cout,
                                 const int32_t *nbr, int64_t n_out, int K, void *ws, size_t ws_bytes,
                                 float *dW, float *dbias, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(cin >= 1 && cin <= 64 && cout >= 1 && cout <= 64, "wgrad: channels must be in 1..64");
  RSLO_CHECK_ARG(K >= 1 && K <= SPC_MAXK, "wgrad: K must be in 1..27");
  const int64_t nW = (int64_t)K * cin * cout;
  if (n_out == 0) {
    RSLO_HIP(hipMemsetAsync(dW, 0, nW * sizeof(float), st));
    if (dbias) RSLO_HIP(hipMemsetAsync(dbias, 0, cout * sizeof(float), st));
    return RSLO_OK;
  }
  if (ws_bytes < rslo_spconv_wgrad_ws_bytes(n_out, K, cin, cout)) {
    rslo_set_error("wgrad: workspace too small");
    return RSLO_EWS;
  }
  const int nch = wgrad_chunks(n_out);
  const int ci = cin <= 16 ? 16 : (cin <= 32 ? 32 : 64), co = pad_cout(cout);
  dim3 grid((unsigned)nch, (unsigned)K);
  float *ws_bias = (float *)ws + (int64_t)nch * nW;
#define WG_CASE(CI, CO)                                                                               \
  if (ci == CI && co == CO)                                                                           \
    hipLaunchKernelGGL((k_wgrad<CI, CO>), grid, dim3(SPC_THREADS), 0, st, in, cin, dout, cout, nbr,   \
                       n_out, K, (float *)ws, dbias ? ws_bias : nullptr);
  WG_CASE(16, 16) WG_CASE(16, 32) WG_CASE(16, 64)
  WG_CASE(32, 16) WG_CASE(32, 32) WG_CASE(32, 64)
  WG_CASE(64, 16) WG_CASE(64, 32) WG_CASE(64, 64)
#undef WG_CASE
  hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)rslo_cdiv(nW, 256)), dim3(256), 0, st,
                     (const float *)ws, nch, nW, dW);
  if (dbias)
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(1), dim3(256), 0, st, (const float *)ws_bias, nch,
                       (int64_t)cout, dbias);
  RSLO_CHECK_LAUNCH("wgrad");
  return RSLO_OK;
}


// ---------------------------------------------------------------------------------------
// wgrad v2 on explicit pair lists.  rslo_rulebook_pairs() turns a neighbour table into the spconv-style
// rulebook once per indice_key (pairs of offset k contiguous, ascending output row; koff[K+1] on the device),
// so the gradient kernels do no compaction.  One workgroup per (pair chunk, offset), dealt to the XCDs by wg2_assign()
// below: the 4 waves of a workgroup split a
// chunk of WG2_CHUNK pairs; lane (li, g) loads CB contiguous input channels of pair g and NB contiguous output
// channels (16-byte loads for 64 channels, the same column-permutation trick as the forward kernel) and
// v_mfma_f32_16x16x4_f32 accumulates dW[ci = CB*i + cb][co = NB*li + nb] with the pair index as the
// contraction dimension.  Wave partials are summed through LDS in a fixed order, chunk partials by a second
// kernel in chunk order: deterministic, no atomics.
// ---------------------------------------------------------------------------------------
#ifndef WG2_CHUNK
#define WG2_CHUNK 2048      /* pairs per workgroup, fp32-MFMA kernel (16-channel layers) */
#endif
#ifndef WG3_CHUNK
#define WG3_CHUNK 1024      /* split-bf16 / bf16 kernels (32 / 64 channels): 55 vs 61 us (32->32), 64-66 vs 71 us (64->64, 40 k rows) */
#endif
#define WG_MIN_CHUNK (WG2_CHUNK < WG3_CHUNK ? WG2_CHUNK : WG3_CHUNK)

// Which (offset k, pair chunk c) a workgroup of the weight-gradient kernels takes.  The grid is one-dimensional and
// workgroups are dealt to the 8 XCDs round-robin (id & 7), each XCD with its own L2.  XCD x takes, of EVERY offset, the
// chunks c in [x n_k / 8, (x + 1) n_k / 8) (n_k = chunks of offset k): pair lists ascend in the output row, so these are
// the pairs of the same eighth of the rows for all 27 offsets -- a row of `in` / `dout` is pulled into ONE L2 and serves
// its ~15 pairs from there, where the (chunk, offset) grid spread the 27 readers of a row over all XCDs (353 MB through
// L2 misses per 64->64 launch for 46 MB of distinct rows).  The chunks themselves and the slot a partial is written to
// are unchanged, so the results keep their bits.  wg2_grid() is the matching launch size (slots per XCD >= what any
// XCD can be dealt: sum_k (n_k / 8 + 1)).
template <int CHUNK>
__device__ __forceinline__ bool wg2_assign(const int32_t *__restrict__ koff, int K, int legacy_nch, int &k, int &c) {
  if (legacy_nch > 0) {          // the (chunk, offset) order of the earlier grid, kept for A/B runs (RSLO_WGRAD_XCD=0)
    k = (int)blockIdx.x / legacy_nch;
    c = (int)blockIdx.x - k * legacy_nch;
    return k < K && c * CHUNK < koff[k + 1] - koff[k];
  }
  const int x = blockIdx.x & 7;
  int j = blockIdx.x >> 3;
  for (int kk = 0; kk < K; ++kk) {
    const int n = (koff[kk + 1] - koff[kk] + CHUNK - 1) / CHUNK;
    const int c0 = (x * n) >> 3, c1 = ((x + 1) * n) >> 3;
    if (j < c1 - c0) {
      k = kk;
      c = c0 + j;
      return true;
    }
    j -= c1 - c0;
  }
  return false;
}
static inline bool wg2_xcd() { return rslo_tune(RSLO_TUNE_WGRAD_XCD) != 0; }
static inline unsigned wg2_grid(int nch, int K, bool xcd) {
  return xcd ? 8u * (unsigned)((nch * K + 7) / 8 + K) : (unsigned)(nch * K);
}
static inline int wg2_legacy(int nch, bool xcd) { return xcd ? 0 : nch; }

template <int CIN_T, int COUT_T, bool EXACT>
__global__ __launch_bounds__(SPC_THREADS) void k_wgrad2(const float *__restrict__ in, int cin,
                                                        const float *__restrict__ dout, int cout,
                                                        const int32_t *__restrict__ pin,
                                                        const int32_t *__restrict__ pout,
                                                        const int32_t *__restrict__ koff, int K,
                                                        int legacy_nch, float *__restrict__ ws) {
  constexpr int CB = CIN_T / 16, NB = COUT_T / 16;
  __shared__ __attribute__((aligned(16))) float red[CIN_T * COUT_T];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  int k, chunk;
  if (!wg2_assign<WG2_CHUNK>(koff, K, legacy_nch, k, chunk)) return;
  const int pk0 = koff[k], pk1 = koff[k + 1];
  const int p0 = pk0 + chunk * WG2_CHUNK;
  const int p1 = (p0 + WG2_CHUNK < pk1) ? p0 + WG2_CHUNK : pk1;

  f32x4 acc[CB][NB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int w0 = p0 + wid * (WG2_CHUNK / SPC_WAVES);
  const int w1 = (w0 + WG2_CHUNK / SPC_WAVES < p1) ? w0 + WG2_CHUNK / SPC_WAVES : p1;
  constexpr int UN = (CB * NB >= 8) ? 4 : 8;     // pair groups whose gathers are in flight together
  for (int q = w0; q < w1; q += 64) {
    // one coalesced index load per lane for 64 pairs, handed to the (li, g) fragment lanes by cross-lane reads
    const int myp = q + lane;
    const int32_t my_i = (myp < w1) ? pin[myp] : -1;
    const int32_t my_o = (myp < w1) ? pout[myp] : 0;
#pragma unroll
    for (int h = 0; h < 16 / UN; ++h) {
      if (q + h * UN * 4 >= w1) break;
      float a[UN][CB], b[UN][NB];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int src = (h * UN + u) * 4 + g;
        const int32_t ri = __shfl(my_i, src, 64);
        const int32_t ro = __shfl(my_o, src, 64);
        const bool ok = ri >= 0;
        if constexpr (EXACT) {
          const VecF<CB> va = load_vec<CB>(in + (int64_t)(ok ? ri : 0) * CIN_T + CB * li);
          const VecF<NB> vb = load_vec<NB>(dout + (int64_t)ro * COUT_T + NB * li);
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) a[u][cb] = ok ? va.v[cb] : 0.f;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) b[u][nb] = vb.v[nb];
        } else {
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) {
            const int c = CB * li + cb;
            a[u][cb] = (ok && c < cin) ? in[(int64_t)ri * cin + c] : 0.f;
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int c = NB * li + nb;
            b[u][nb] = (c < cout) ? dout[(int64_t)ro * cout + c] : 0.f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[cb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][cb], b[u][nb], acc[cb][nb], 0, 0, 0);
    }
  }

  for (int w = 0; w < SPC_WAVES; ++w) {
    if (wid == w) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ci = CB * (4 * g + j) + cb, co = NB * li + nb;
            float v = acc[cb][nb][j];
            if (w) v += red[ci * COUT_T + co];
            red[ci * COUT_T + co] = v;
          }
    }
    __syncthreads();
  }
  float *dst = ws + ((int64_t)chunk * K + k) * cin * cout;
  for (int e = tid; e < cin * cout; e += SPC_THREADS) {
    const int ci = e / cout, co = e - ci * cout;
    dst[e] = red[ci * COUT_T + co];
  }
}

// Weight gradient on the bf16 matrix cores with exactly split fp32 operands (the wgrad counterpart of k_spconv_v6; channel
// counts 32 / 64).  The contraction index of v_mfma_f32_16x16x32_bf16 runs over 32 pairs; lane group g takes pairs
// {4 e + g : e = 0..7} of a 32-pair chunk as its 8 k-values -- exactly the pairs whose rows this lane gathers anyway
// (one float4 / float2 of its own channels per pair), so no transposition is needed: the 8 gathered values of a
// channel are split (hi + mid + lo) and packed straight into the A / B operands.  6 MFMAs per 16x16 block and 32 pairs
// instead of 8 fp32 MFMAs: 2.5x fewer matrix-core cycles, same fixed-order reductions as k_wgrad2.
#ifndef WG3_WPE
#define WG3_WPE 2      /* 3 fits 64->64 into 168 registers with 19 spilled: 142 vs 130 us */
#endif
template <int CIN_T, int COUT_T>
__global__ __launch_bounds__(SPC_THREADS) __attribute__((amdgpu_waves_per_eu(WG3_WPE))) void k_wgrad3(const float *__restrict__ in, const float *__restrict__ dout,
                                                        const int32_t *__restrict__ pin,
                                                        const int32_t *__restrict__ pout,
                                                        const int32_t *__restrict__ koff, int K,
                                                        int legacy_nch, float *__restrict__ ws) {
  constexpr int CB = CIN_T / 16, NB = COUT_T / 16;
  __shared__ __attribute__((aligned(16))) float red[CIN_T * COUT_T];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  int k, chunk;
  if (!wg2_assign<WG3_CHUNK>(koff, K, legacy_nch, k, chunk)) return;
  const int pk0 = koff[k], pk1 = koff[k + 1];
  const int p0 = pk0 + chunk * WG3_CHUNK;
  const int p1 = (p0 + WG3_CHUNK < pk1) ? p0 + WG3_CHUNK : pk1;

  f32x4 acc[CB][NB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int w0 = p0 + wid * (WG3_CHUNK / SPC_WAVES);
  const int w1 = (w0 + WG3_CHUNK / SPC_WAVES < p1) ? w0 + WG3_CHUNK / SPC_WAVES : p1;
  for (int q = w0; q < w1; q += 64) {
    const int myp = q + lane;
    const int32_t my_i = (myp < w1) ? pin[myp] : -1;
    const int32_t my_o = (myp < w1) ? pout[myp] : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (q + h * 32 >= w1) break;
      // B operands (dout) are split and packed for all NB blocks; the A side keeps the 8 gathered fp32 rows and is split
      // one 16-channel block at a time inside the product loop (register budget: 2 waves per SIMD)
      u32x4 bh[NB], bm[NB], bl[NB];
      float araw[8][CB];
#pragma unroll
      for (int p = 0; p < 4; ++p) {                      // k-values e = 2p, 2p+1 -> dword p of every operand
        float xb[2][NB];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int src = h * 32 + 4 * (2 * p + t) + g;
          const int32_t ri = __shfl(my_i, src, 64);
          const int32_t ro = __shfl(my_o, src, 64);
          const bool ok = ri >= 0;
          const VecF<CB> va = load_vec<CB>(in + (int64_t)(ok ? ri : 0) * CIN_T + CB * li);
          const VecF<NB> vb = load_vec<NB>(dout + (int64_t)ro * COUT_T + NB * li);
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) araw[2 * p + t][cb] = ok ? va.v[cb] : 0.f;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) xb[t][nb] = ok ? vb.v[nb] : 0.f;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const RsloSplit2 sb = rslo_split2(xb[0][nb], xb[1][nb]);
          bh[nb][p] = sb.h;
          bm[nb][p] = sb.m;
          bl[nb][p] = sb.l;
        }
      }
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        u32x4 ah, am, al;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const RsloSplit2 sa = rslo_split2(araw[2 * p][cb], araw[2 * p + 1][cb]);
          ah[p] = sa.h;
          am[p] = sa.m;
          al[p] = sa.l;
        }
        // six products per block, smallest first; consecutive MFMAs hit different accumulators (the NB column blocks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = MFMA_BF16(al, bh[nb], acc[cb][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = MFMA_BF16(am, bm[nb], acc[cb][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = MFMA_BF16(ah, bl[nb], acc[cb][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = MFMA_BF16(am, bh[nb], acc[cb][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = MFMA_BF16(ah, bm[nb], acc[cb][nb]);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = MFMA_BF16(ah, bh[nb], acc[cb][nb]);
      }
    }
  }

  for (int w = 0; w < SPC_WAVES; ++w) {
    if (wid == w) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ci = CB * (4 * g + j) + cb, co = NB * li + nb;
            float v = acc[cb][nb][j];
            if (w) v += red[ci * COUT_T + co];
            red[ci * COUT_T + co] = v;
          }
    }
    __syncthreads();
  }
  float *dst = ws + ((int64_t)chunk * K + k) * CIN_T * COUT_T;
  for (int e = tid; e < CIN_T * COUT_T; e += SPC_THREADS) dst[e] = red[e];
}

// bf16 feature path (BASELINE config C4): the same pair-list weight gradient with bf16 rows on both sides.  The rows ARE
// the MFMA operands -- one v_mfma_f32_16x16x32_bf16 per 16x16 block and 32 pairs where k_wgrad3 issues six -- and a
// gathered row is half the bytes.  Lane (g, li) gathers pairs {4e+g}: CB consecutive input channels (CB*li..) and NB
// consecutive output channels as 2-byte values; element e of the operand for channel cb is the 16-bit half `cb & 1` of
// dword `cb >> 1` of pair e's load, so two pairs pack into one operand dword with a single v_perm.  fp32 accumulation,
// fp32 partials, same deterministic two-stage reduction as the fp32 path.
template <int N>
struct VecH;          // N bf16 values as dwords
template <>
struct VecH<4> {
  unsigned v[2];
};
template <>
struct VecH<2> {
  unsigned v[1];
};
template <int N>
__device__ __forceinline__ VecH<N> load_vech(const unsigned short *p) {
  VecH<N> r;
  if constexpr (N == 4) {
    const uint2 t = *reinterpret_cast<const uint2 *>(p);
    r.v[0] = t.x;
    r.v[1] = t.y;
  } else {
    r.v[0] = *reinterpret_cast<const unsigned *>(p);
  }
  return r;
}
// 16-bit half `c & 1` of dword c >> 1 of two loads -> (lo = first, hi = second)
#define PACK_H(A, B, C) __builtin_amdgcn_perm((B).v[(C) >> 1], (A).v[(C) >> 1], ((C)&1) ? 0x07060302u : 0x05040100u)

template <int CIN_T, int COUT_T>
__global__ __launch_bounds__(SPC_THREADS) void k_wgrad3_bf16(const unsigned short *__restrict__ in,
                                                             const unsigned short *__restrict__ dout,
                                                             const int32_t *__restrict__ pin,
                                                             const int32_t *__restrict__ pout,
                                                             const int32_t *__restrict__ koff, int K,
                                                             int legacy_nch, float *__restrict__ ws) {
  constexpr int CB = CIN_T / 16, NB = COUT_T / 16;
  __shared__ __attribute__((aligned(16))) float red[CIN_T * COUT_T];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  int k, chunk;
  if (!wg2_assign<WG3_CHUNK>(koff, K, legacy_nch, k, chunk)) return;
  const int pk0 = koff[k], pk1 = koff[k + 1];
  const int p0 = pk0 + chunk * WG3_CHUNK;
  const int p1 = (p0 + WG3_CHUNK < pk1) ? p0 + WG3_CHUNK : pk1;

  f32x4 acc[CB][NB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int w0 = p0 + wid * (WG3_CHUNK / SPC_WAVES);
  const int w1 = (w0 + WG3_CHUNK / SPC_WAVES < p1) ? w0 + WG3_CHUNK / SPC_WAVES : p1;
  for (int q = w0; q < w1; q += 64) {
    const int myp = q + lane;
    const int32_t my_i = (myp < w1) ? pin[myp] : -1;
    const int32_t my_o = (myp < w1) ? pout[myp] : 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (q + h * 32 >= w1) break;
      VecH<CB> xa[8];
      VecH<NB> xb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int src = h * 32 + 4 * e + g;
        const int32_t ri = __shfl(my_i, src, 64);
        const int32_t ro = __shfl(my_o, src, 64);
        const bool ok = ri >= 0;
        xa[e] = load_vech<CB>(in + (int64_t)(ok ? ri : 0) * CIN_T + CB * li);
        xb[e] = load_vech<NB>(dout + (int64_t)ro * COUT_T + NB * li);
        if (!ok) {
#pragma unroll
          for (int d = 0; d < (CB + 1) / 2; ++d) xa[e].v[d] = 0u;
        }
      }
      u32x4 b[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < 4; ++p) b[nb][p] = PACK_H(xb[2 * p], xb[2 * p + 1], nb);
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        u32x4 a;
#pragma unroll
        for (int p = 0; p < 4; ++p) a[p] = PACK_H(xa[2 * p], xa[2 * p + 1], cb);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[cb][nb] = MFMA_BF16(a, b[nb], acc[cb][nb]);
      }
    }
  }

  for (int w = 0; w < SPC_WAVES; ++w) {
    if (wid == w) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ci = CB * (4 * g + j) + cb, co = NB * li + nb;
            float v = acc[cb][nb][j];
            if (w) v += red[ci * COUT_T + co];
            red[ci * COUT_T + co] = v;
          }
    }
    __syncthreads();
  }
  float *dst = ws + ((int64_t)chunk * K + k) * CIN_T * COUT_T;
  for (int e = tid; e < CIN_T * COUT_T; e += SPC_THREADS) dst[e] = red[e];
}

// grid (ceil(cc / 32), K [+ 1]): the chunk partials added in chunk order (+ the bias-gradient rows); block body in
// wgrad_reduce.h (shared with the one-launch form over many layers, rslo_wgrad_reduce_many)
__global__ void k_wgrad2_reduce(const float *__restrict__ ws, const int32_t *__restrict__ koff, int K, int chunk, int cc,
                                float *__restrict__ dW, const float *__restrict__ bpart, int n_bpart, int cout,
                                float *__restrict__ dbias) {
  __shared__ float red[256];
  wr_sparse_block((int)blockIdx.x, (int)blockIdx.y, ws, koff, K, chunk, cc, dW, bpart, n_bpart, cout, dbias, red);
}

// Bias gradient = column sums of the (activation-masked) output gradient, in two deterministic stages without fences:
// stage 1 writes one partial row per block -- either k_colsum_partial below or, for layers with a fused activation,
// k_leaky_bwd_colsum, which forms g = dout * act'(y) anyway --, stage 2 is the extra grid row of k_wgrad2_reduce.
#define CS1_MAXBLK 512
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ x, int64_t rows, int rows_per_blk,
                                                        int cols, int cols_pad, float *__restrict__ partial) {
  __shared__ float red[256];
  const int c = threadIdx.x % cols_pad, part = threadIdx.x / cols_pad, nparts = 256 / cols_pad;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk, r1 = (r0 + rows_per_blk < rows) ? r0 + rows_per_blk : rows;
  float s = 0.f;
  if (c < cols) {
    for (int64_t r = r0 + part; r < r1; r += 8 * nparts) {      // 8 independent loads in flight, ordered adds
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t rr = r + (int64_t)u * nparts;
        v[u] = rr < r1 ? x[rr * cols + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (part == 0 && c < cols) {
    float t = 0.f;
    for (int p = 0; p < nparts; ++p) t += red[p * cols_pad + c];
    partial[(int64_t)blockIdx.x * cols + c] = t;
  }
}

extern "C" size_t rslo_spconv_wgrad_pairs_ws_bytes(int64_t n_out, int K, int cin, int cout) {
  const int64_t nch = rslo_cdiv(n_out > 0 ? n_out : 1, WG_MIN_CHUNK);
  return ((size_t)nch * (size_t)K * cin * cout + (size_t)(CS1_MAXBLK + 8) * cout) * sizeof(float);
}

extern "C" int rslo_spconv_wgrad_pairs(const float *in, int cin, const float *dout, int cout,
                                       const int32_t *pairs_in, const int32_t *pairs_out, const int32_t *koff,
                                       int64_t n_out, int K, void *ws, size_t ws_bytes, float *dW, float *dbias,
                                       const float *bias_partial, int n_bias_partial, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(cin >= 1 && cin <= 64 && cout >= 1 && cout <= 64, "wgrad: channels must be in 1..64");
  RSLO_CHECK_ARG(K >= 1 && K <= SPC_MAXK, "wgrad: K must be in 1..27");
  const int64_t nW = (int64_t)K * cin * cout;
  if (n_out == 0) {
    RSLO_HIP(hipMemsetAsync(dW, 0, nW * sizeof(float), st));
    if (dbias) RSLO_HIP(hipMemsetAsync(dbias, 0, cout * sizeof(float), st));
    return RSLO_OK;
  }
  if (ws_bytes < rslo_spconv_wgrad_pairs_ws_bytes(n_out, K, cin, cout)) {
    rslo_set_error("wgrad_pairs: workspace too small");
    return RSLO_EWS;
  }
  const int ci = cin <= 16 ? 16 : (cin <= 32 ? 32 : 64), co = pad_cout(cout);
  const bool exact = (ci == cin && co == cout);
  const bool split_on = rslo_tune(RSLO_TUNE_SPCONV_WGRAD_SPLIT) != 0;
  const bool use3 = split_on && exact && (cin == 32 || cin == 64) && (cout == 32 || cout == 64);
  const int chunk = use3 ? WG3_CHUNK : WG2_CHUNK;
  const int nch = (int)rslo_cdiv(n_out, chunk);   // P_k <= n_out: upper bound on chunks per offset
  const bool xcd = wg2_xcd();
  dim3 grid(wg2_grid(nch, K, xcd));
  if (use3) {
#define WG3_CASE(CI, CO)                                                                                  \
    if (cin == CI && cout == CO)                                                                          \
      hipLaunchKernelGGL((k_wgrad3<CI, CO>), grid, dim3(SPC_THREADS), 0, st, in, dout, pairs_in, pairs_out, koff, K, \
                         wg2_legacy(nch, xcd), (float *)ws);
    WG3_CASE(32, 32) WG3_CASE(32, 64) WG3_CASE(64, 32) WG3_CASE(64, 64)
#undef WG3_CASE
  } else {
#define WG2_CASE(CI, CO)                                                                                  \
  if (ci == CI && co == CO) {                                                                             \
    if (exact)                                                                                            \
      hipLaunchKernelGGL((k_wgrad2<CI, CO, true>), grid, dim3(SPC_THREADS), 0, st, in, cin, dout, cout,   \
                         pairs_in, pairs_out, koff, K, wg2_legacy(nch, xcd), (float *)ws);                                      \
    else                                                                                                  \
      hipLaunchKernelGGL((k_wgrad2<CI, CO, false>), grid, dim3(SPC_THREADS), 0, st, in, cin, dout, cout,  \
                         pairs_in, pairs_out, koff, K, wg2_legacy(nch, xcd), (float *)ws);                                      \
  }
  WG2_CASE(16, 16) WG2_CASE(16, 32) WG2_CASE(16, 64)
  WG2_CASE(32, 16) WG2_CASE(32, 32) WG2_CASE(32, 64)
  WG2_CASE(64, 16) WG2_CASE(64, 32) WG2_CASE(64, 64)
#undef WG2_CASE
  }
  const int cc = cin * cout;
  const float *bpart = nullptr;
  int n_bpart = 0;
  if (dbias) {
    if (bias_partial) {          // stage 1 already done by rslo_leaky_bwd_colsum
      bpart = bias_partial;
      n_bpart = n_bias_partial;
    } else {
      float *wsb = (float *)ws + (int64_t)nch * nW;
      int rpb = (int)rslo_cdiv(n_out, CS1_MAXBLK);
      rpb = rpb < 64 ? 64 : rpb;
      n_bpart = (int)rslo_cdiv(n_out, rpb);
      hipLaunchKernelGGL(k_colsum_partial, dim3((unsigned)n_bpart), dim3(256), 0, st, dout, n_out, rpb, cout, co, wsb);
      bpart = wsb;
    }
  }
  if (wr_defer(wr_sparse_desc(ws, koff, K, chunk, cc, dW, bpart, n_bpart, cout, dbias))) return RSLO_OK;
  hipLaunchKernelGGL(k_wgrad2_reduce, dim3((unsigned)rslo_cdiv(cc, 32), (unsigned)(K + (dbias ? 1 : 0))), dim3(256), 0, st,
                     (const float *)ws, koff, K, chunk, cc, dW, bpart, n_bpart, cout, dbias);
  RSLO_CHECK_LAUNCH("wgrad_pairs");
  return RSLO_OK;
}

extern "C" int rslo_spconv_wgrad_pairs_bf16(const void *in, int cin, const void *dout, int cout,
                                            const int32_t *pairs_in, const int32_t *pairs_out, const int32_t *koff,
                                            int64_t n_out, int K, void *ws, size_t ws_bytes, float *dW, float *dbias,
                                            const float *bias_partial, int n_bias_partial, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG((cin == 32 || cin == 64) && (cout == 32 || cout == 64), "wgrad_pairs_bf16: channels must be 32 or 64");
  RSLO_CHECK_ARG(K >= 1 && K <= SPC_MAXK, "wgrad_pairs_bf16: K must be in 1..27");
  RSLO_CHECK_ARG(!dbias || bias_partial, "wgrad_pairs_bf16: the bias gradient needs the column partials of "
                                         "rslo_leaky_bwd_colsum_bf16");
  const int64_t nW = (int64_t)K * cin * cout;
  if (n_out == 0) {
    RSLO_HIP(hipMemsetAsync(dW, 0, nW * sizeof(float), st));
    if (dbias) RSLO_HIP(hipMemsetAsync(dbias, 0, cout * sizeof(float), st));
    return RSLO_OK;
  }
  if (ws_bytes < rslo_spconv_wgrad_pairs_ws_bytes(n_out, K, cin, cout)) {
    rslo_set_error("wgrad_pairs_bf16: workspace too small");
    return RSLO_EWS;
  }
  const int nch = (int)rslo_cdiv(n_out, WG3_CHUNK);
  const bool xcd = wg2_xcd();
  dim3 grid(wg2_grid(nch, K, xcd));
  const unsigned short *x = (const unsigned short *)in, *g = (const unsigned short *)dout;
#define WG3B_CASE(CI, CO)                                                                                      \
  if (cin == CI && cout == CO)                                                                                 \
    hipLaunchKernelGGL((k_wgrad3_bf16<CI, CO>), grid, dim3(SPC_THREADS), 0, st, x, g, pairs_in, pairs_out, koff, K, \
                       wg2_legacy(nch, xcd), (float *)ws);
  WG3B_CASE(32, 32) WG3B_CASE(32, 64) WG3B_CASE(64, 32) WG3B_CASE(64, 64)
#undef WG3B_CASE
  const int cc = cin * cout;
  if (wr_defer(wr_sparse_desc(ws, koff, K, WG3_CHUNK, cc, dW, bias_partial, dbias ? n_bias_partial : 0, cout, dbias))) return RSLO_OK;
  hipLaunchKernelGGL(k_wgrad2_reduce, dim3((unsigned)rslo_cdiv(cc, 32), (unsigned)(K + (dbias ? 1 : 0))), dim3(256), 0, st,
                     (const float *)ws, koff, K, WG3_CHUNK, cc, dW, bias_partial, dbias ? n_bias_partial : 0, cout, dbias);
  RSLO_CHECK_LAUNCH("wgrad_pairs_bf16");
  return RSLO_OK;
}

__global__ void k_leaky_bwd(const float *__restrict__ y, const float *__restrict__ dout, int64_t n,
                            float slope, float *__restrict__ g) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 yv = *reinterpret_cast<const float4 *>(y + i);
    float4 d = *reinterpret_cast<const float4 *>(dout + i);
    d.x = yv.x > 0.f ? d.x : d.x * slope;
    d.y = yv.y > 0.f ? d.y : d.y * slope;
    d.z = yv.z > 0.f ? d.z : d.z * slope;
    d.w = yv.w > 0.f ? d.w : d.w * slope;
    *reinterpret_cast<float4 *>(g + i) = d;
  } else {
    for (; i < n; ++i) g[i] = y[i] > 0.f ? dout[i] : dout[i] * slope;
  }
}

// g = dout * act'(y) AND the per-block column sums of g (stage 1 of the bias gradient).  A block owns LB_TILES tiles of
// 1024 consecutive floats; 1024 % cols == 0, so a thread's 4 columns are the same in every tile.
#define LB_TILES 8
__global__ __launch_bounds__(256) void k_leaky_bwd_colsum(const float *__restrict__ y, const float *__restrict__ dout,
                                                          int64_t n, int cols, float slope, float *__restrict__ g,
                                                          float *__restrict__ partial) {
  __shared__ float red[1024];
  const int64_t base = (int64_t)blockIdx.x * (1024 * LB_TILES) + 4 * threadIdx.x;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < LB_TILES; ++t) {
    const int64_t i = base + (int64_t)t * 1024;
    if (i < n) {               // n % 4 == 0 (cols % 4 == 0)
      const float4 yv = *reinterpret_cast<const float4 *>(y + i);
      float4 d = *reinterpret_cast<const float4 *>(dout + i);
      d.x = yv.x > 0.f ? d.x : d.x * slope;
      d.y = yv.y > 0.f ? d.y : d.y * slope;
      d.z = yv.z > 0.f ? d.z : d.z * slope;
      d.w = yv.w > 0.f ? d.w : d.w * slope;
      *reinterpret_cast<float4 *>(g + i) = d;
      s[0] += d.x; s[1] += d.y; s[2] += d.z; s[3] += d.w;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[4 * threadIdx.x + k] = s[k];
  __syncthreads();
  if ((int)threadIdx.x < cols) {      // position p of the 1024-float tile holds column p % cols
    float t = 0.f;
    for (int p = threadIdx.x; p < 1024; p += cols) t += red[p];
    partial[(int64_t)blockIdx.x * cols + threadIdx.x] = t;
  }
}

// bf16 rows: g = dout * act'(y) written in bf16 (round to nearest), column sums accumulated in fp32 from the ROUNDED g --
// the values the weight-gradient kernel multiplies.  A block owns LB_TILES tiles of 2048 consecutive values (8 per
// thread); 2048 % cols == 0.
__global__ __launch_bounds__(256) void k_leaky_bwd_colsum_bf16(const unsigned short *__restrict__ y,
                                                               const unsigned short *__restrict__ dout, int64_t n,
                                                               int cols, float slope, unsigned short *__restrict__ g,
                                                               float *__restrict__ partial) {
  __shared__ float red[2048];
  const int64_t base = (int64_t)blockIdx.x * (2048 * LB_TILES) + 8 * threadIdx.x;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < LB_TILES; ++t) {
    const int64_t i = base + (int64_t)t * 2048;
    if (i < n) {               // n % 8 == 0 (cols % 8 == 0)
      const uint4 yv = *reinterpret_cast<const uint4 *>(y + i);
      const uint4 dv = *reinterpret_cast<const uint4 *>(dout + i);
      const unsigned yw[4] = {yv.x, yv.y, yv.z, yv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
      unsigned ow[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        float g0 = __uint_as_float(dw[d] << 16), g1 = __uint_as_float(dw[d] & 0xffff0000u);
        // bf16 sign bit: y > 0  <=>  not negative and not zero
        const bool p0 = (yw[d] & 0x8000u) == 0 && (yw[d] & 0x7fffu) != 0;
        const bool p1 = (yw[d] & 0x80000000u) == 0 && (yw[d] & 0x7fff0000u) != 0;
        g0 = p0 ? g0 : g0 * slope;
        g1 = p1 ? g1 : g1 * slope;
        ow[d] = rslo_pk_bf16_rn(g0, g1);
        s[2 * d] += __uint_as_float(ow[d] << 16);
        s[2 * d + 1] += __uint_as_float(ow[d] & 0xffff0000u);
      }
      *reinterpret_cast<uint4 *>(g + i) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[8 * threadIdx.x + k] = s[k];
  __syncthreads();
  if (partial && (int)threadIdx.x < cols) {
    float t = 0.f;
    for (int p = threadIdx.x; p < 2048; p += cols) t += red[p];
    partial[(int64_t)blockIdx.x * cols + threadIdx.x] = t;
  }
}

extern "C" int64_t rslo_leaky_bwd_colsum_bf16_blocks(int64_t rows, int cols) {
  return rslo_cdiv(rows * cols > 0 ? rows * cols : 1, 2048 * LB_TILES);
}

extern "C" int rslo_leaky_bwd_colsum_bf16(const void *y, const void *dout, int64_t rows, int cols, float slope, void *g,
                                          float *partial /*[blocks, cols] or NULL*/, void *stream) {
  RSLO_CHECK_ARG(y && dout && g, "rslo_leaky_bwd_colsum_bf16: bad arguments");
  RSLO_CHECK_ARG(cols >= 8 && cols <= 256 && 2048 % cols == 0, "rslo_leaky_bwd_colsum_bf16: cols must divide 2048");
  if (rows == 0) return RSLO_OK;
  const int64_t nblk = rslo_leaky_bwd_colsum_bf16_blocks(rows, cols);
  hipLaunchKernelGGL(k_leaky_bwd_colsum_bf16, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short *)y, (const unsigned short *)dout, rows * cols, cols, slope, (unsigned short *)g,
                     partial);
  RSLO_CHECK_LAUNCH("leaky_bwd_colsum_bf16");
  return RSLO_OK;
}

extern "C" int64_t rslo_leaky_bwd_colsum_blocks(int64_t rows, int cols) {
  return rslo_cdiv(rows * cols > 0 ? rows * cols : 1, 1024 * LB_TILES);
}

extern "C" int rslo_leaky_bwd_colsum(const float *y, const float *dout, int64_t rows, int cols, float slope, float *g,
                                     float *partial /*[blocks, cols]*/, void *stream) {
  RSLO_CHECK_ARG(y && dout && g && partial, "rslo_leaky_bwd_colsum: bad arguments");
  RSLO_CHECK_ARG(cols >= 4 && cols <= 256 && 1024 % cols == 0, "rslo_leaky_bwd_colsum: cols must divide 1024");
  if (rows == 0) return RSLO_OK;
  const int64_t nblk = rslo_leaky_bwd_colsum_blocks(rows, cols);
  hipLaunchKernelGGL(k_leaky_bwd_colsum, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, y, dout, rows * cols,
                     cols, slope, g, partial);
  RSLO_CHECK_LAUNCH("leaky_bwd_colsum");
  return RSLO_OK;
}

extern "C" int rslo_leaky_bwd(const float *y, const float *dout, int64_t n, float slope, float *g,
                              void *stream) {
  if (n == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_leaky_bwd, dim3((unsigned)rslo_cdiv(rslo_cdiv(n, 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, y, dout, n, slope, g);
  RSLO_CHECK_LAUNCH("leaky_bwd");
  return RSLO_OK;
}
