// Exact nearest neighbour with spatial pruning (same contract and the same bits as the brute-force
// kernels in chamfer.hip: reference arithmetic of chamfer_distance.cpp:116-144, lowest index wins ties).
//
// Both clouds are bucketed by a coarse Morton cell key (counting sort: histogram, scan, scatter), so that 64
// consecutive sorted points are spatially compact.  Targets are cut into tiles of 64 with an exact bounding box.
// A wave owns 64 sorted queries; for every target tile each lane forms the box lower bound
//     lb = (ex^2 + ey^2) + ez^2,  e = max(lo - q, q - hi, 0)
// in the SAME operation order as the distance.  fp32 rounding is monotone, so every point p of the tile has a
// computed distance d(p) >= lb bit for bit: a tile whose lb exceeds the current best of all 64 lanes cannot hold a
// winner (nor a tie) and is skipped; surviving tiles are evaluated point by point (staged in the wave's LDS slot and
// read as broadcasts) under the order-independent rule "smaller distance, then smaller original index".
// The result is therefore independent of the sort order and identical to the exhaustive scan, at ~5 % of its
// distance evaluations for overlapping clouds.  Sorting only affects speed, never the answer: points outside the
// key range are clamped into border cells.
#include "rslo_common.h"

#pragma clang fp contract(off)   /* distances and bounds in the reference's per-operation rounding */

#define CG_XB 7
#define CG_YB 7
#define CG_ZB 3
#define CG_BINS (1 << (CG_XB + CG_YB + CG_ZB))
#define CG_CELL_XY 1.25f      /* 128 cells: [-80, 80) m */
#define CG_CELL_Z 2.5f        /* 8 cells:   [-10, 10) m */
#define CG_TILE 64
#define CG_LDS_TILES 1024

__device__ __forceinline__ unsigned cg_spread7(unsigned v) {   // 7 bits -> every second bit
  v &= 0x7f;
  v = (v | (v << 4)) & 0x070f;
  v = (v | (v << 2)) & 0x1333;
  v = (v | (v << 1)) & 0x1555;
  return v;
}

__device__ __forceinline__ int cg_key(float x, float y, float z) {
  // NaN / inf compare false everywhere and land in cell 0 / the last cell: harmless, the key is only a sort hint
  int cx = (int)floorf((x + 80.0f) * (1.0f / CG_CELL_XY));
  int cy = (int)floorf((y + 80.0f) * (1.0f / CG_CELL_XY));
  int cz = (int)floorf((z + 10.0f) * (1.0f / CG_CELL_Z));
  cx = cx < 0 ? 0 : (cx > 127 ? 127 : cx);
  cy = cy < 0 ? 0 : (cy > 127 ? 127 : cy);
  cz = cz < 0 ? 0 : (cz > 7 ? 7 : cz);
  if (!(x == x)) cx = 0;
  if (!(y == y)) cy = 0;
  if (!(z == z)) cz = 0;
  return (int)(((cg_spread7((unsigned)cx) | (cg_spread7((unsigned)cy) << 1)) << CG_ZB) | (unsigned)cz);
}

// blockIdx.z: cloud (0 = queries, 1 = targets); blockIdx.y: pair
__global__ void k_cg_hist(const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int M,
                          const int32_t *__restrict__ ncnt, const int32_t *__restrict__ mcnt,
                          int32_t *__restrict__ hist, int B, float *__restrict__ dist, int32_t *__restrict__ idx) {
  const int cloud = blockIdx.z, b = blockIdx.y;
  const int n = cloud ? M : N;
  const int cnt = cloud ? (mcnt ? mcnt[b] : M) : (ncnt ? ncnt[b] : N);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i >= cnt) {
    if (!cloud) {   // padding row of a ragged batch: never inside any ROI
      dist[(int64_t)b * N + i] = __builtin_inff();
      idx[(int64_t)b * N + i] = 0;
    }
    return;
  }
  const float *p = (cloud ? xyz2 : xyz1) + ((int64_t)b * n + i) * 3;
  atomicAdd(&hist[((int64_t)cloud * B + b) * CG_BINS + cg_key(p[0], p[1], p[2])], 1);
}

// exclusive scan of every CG_BINS-long segment in place, in two fully parallel launches:
// (1) sums of 1024-bin chunks, (2) per chunk: carry = sum of the earlier chunk sums, then a block scan.
#define CG_CHUNK 1024
#define CG_NCHUNK (CG_BINS / CG_CHUNK)
__global__ __launch_bounds__(256) void k_cg_chunk_sums(const int32_t *__restrict__ hist, int32_t *__restrict__ sums) {
  __shared__ int32_t red[4];
  const int32_t *h = hist + ((int64_t)blockIdx.y * CG_NCHUNK + blockIdx.x) * CG_CHUNK;
  int32_t s = h[threadIdx.x] + h[threadIdx.x + 256] + h[threadIdx.x + 512] + h[threadIdx.x + 768];
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.y * CG_NCHUNK + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void k_cg_scan(int32_t *__restrict__ hist, const int32_t *__restrict__ sums) {
  __shared__ int32_t wsum[4];
  __shared__ int32_t carry_s;
  int32_t *h = hist + ((int64_t)blockIdx.y * CG_NCHUNK + blockIdx.x) * CG_CHUNK;
  // carry of this chunk
  int32_t c = 0;
  for (int k = threadIdx.x; k < (int)blockIdx.x; k += 256) c += sums[blockIdx.y * CG_NCHUNK + k];
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) carry_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  // each thread owns 4 consecutive bins
  const int4 v = *reinterpret_cast<const int4 *>(h + threadIdx.x * 4);
  const int32_t tot = v.x + v.y + v.z + v.w;
  int32_t inc = tot;
  const int lane = threadIdx.x & 63;
  for (int d = 1; d < 64; d <<= 1) {
    const int32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  __syncthreads();
  if (lane == 63) wsum[threadIdx.x >> 6] = inc;
  __syncthreads();
  int32_t base = carry_s + inc - tot;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
  *reinterpret_cast<int4 *>(h + threadIdx.x * 4) = make_int4(base, base + v.x, base + v.x + v.y, base + v.x + v.y + v.z);
}

// sorted[cloud][b][pos] = (x, y, z, original index); tails up to the padded length get +inf / index -1
__global__ void k_cg_scatter(const float *__restrict__ xyz1, const float *__restrict__ xyz2, int N, int M, int Npad,
                             int Mpad, const int32_t *__restrict__ ncnt, const int32_t *__restrict__ mcnt,
                             const int32_t *__restrict__ off, int32_t *__restrict__ cursor, int B,
                             float4 *__restrict__ sq, float4 *__restrict__ st) {
  const int cloud = blockIdx.z, b = blockIdx.y;
  const int n = cloud ? M : N, npad = cloud ? Mpad : Npad;
  const int cnt = cloud ? (mcnt ? mcnt[b] : M) : (ncnt ? ncnt[b] : N);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float4 *dst = (cloud ? st : sq) + (int64_t)b * npad;
  if (i >= npad) return;
  if (i >= cnt) {
    const float inf = __builtin_inff();
    dst[i] = make_float4(inf, inf, inf, __int_as_float(-1));   // slots [cnt, npad) are exactly the unsorted tail
    return;
  }
  const float *p = (cloud ? xyz2 : xyz1) + ((int64_t)b * n + i) * 3;
  const int64_t slot = ((int64_t)cloud * B + b) * CG_BINS + cg_key(p[0], p[1], p[2]);
  const int pos = off[slot] + atomicAdd(&cursor[slot], 1);
  dst[pos] = make_float4(p[0], p[1], p[2], __int_as_float(i));
}

// exact bounding box of every target tile (one wave per tile); empty tiles get an inverted box (lb = +inf)
__global__ __launch_bounds__(256) void k_cg_tiles(const float4 *__restrict__ st, int Mpad, int T,
                                                  float *__restrict__ boxes) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= T) return;
  const float4 p = st[(int64_t)b * Mpad + t * CG_TILE + lane];
  const bool ok = __float_as_int(p.w) >= 0;
  const float inf = __builtin_inff();
  float lo[3] = {ok ? p.x : inf, ok ? p.y : inf, ok ? p.z : inf};
  float hi[3] = {ok ? p.x : -inf, ok ? p.y : -inf, ok ? p.z : -inf};
  for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], d, 64));     // fminf / fmaxf drop NaN coordinates: such points never win
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d, 64));
    }
  }
  if (lane == 0) {
    float *o = boxes + ((int64_t)b * T + t) * 6;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2];
    o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
  }
}

// The running best of a lane is ONE 64-bit key (distance bits << 32 | original index): non-negative floats order like
// their bit patterns, so "smaller distance, then smaller original index" is a single unsigned 64-bit minimum --
// order-independent, NaN distances (0x7fc00000 > +inf) and padding points (index 0xffffffff at +inf) never win,
// and the initial key (+inf, 0) is the reference's answer when nothing compares smaller.
typedef unsigned long long u64;
#define CG_KEY_INIT ((u64)0x7f800000u << 32)

// all 64 lanes against the 64 points of one target tile, staged in the wave's LDS slot (broadcast reads).
// The slot holds the tile as 32 point PAIRS (x0 x1 y0 y1 | z0 z1 w0 w1): two ds_read_b128 deliver the pair's coordinates
// in adjacent registers, so the three subtractions, three squares and two additions of BOTH points are packed fp32
// instructions (v_pk_add_f32 / v_pk_mul_f32: same IEEE results per element, no contraction) -- 8 arithmetic issues per
// pair instead of 16; the key comparisons stay in point order.
typedef float cg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cg_eval_tile(const float4 *__restrict__ tile, float4 *slot, int lane, float qx, float qy,
                                             float qz, u64 &best) {
  const float4 mine = tile[lane];
  float *sf = reinterpret_cast<float *>(slot) + (lane >> 1) * 8 + (lane & 1);
  sf[0] = mine.x; sf[2] = mine.y; sf[4] = mine.z; sf[6] = mine.w;
  __builtin_amdgcn_wave_barrier();
  const cg_f2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};
#pragma unroll 8
  for (int j = 0; j < CG_TILE / 2; ++j) {
    const float4 A = slot[2 * j], Bv = slot[2 * j + 1];
    const cg_f2 px = {A.x, A.y}, py = {A.z, A.w}, pz = {Bv.x, Bv.y};
    const cg_f2 dx = px - qx2, dy = py - qy2, dz = pz - qz2;
    const cg_f2 d = (dx * dx + dy * dy) + dz * dz;
    const u64 k0 = ((u64)(unsigned)__float_as_int(d.x) << 32) | (unsigned)__float_as_int(Bv.z);
    const u64 k1 = ((u64)(unsigned)__float_as_int(d.y) << 32) | (unsigned)__float_as_int(Bv.w);
    best = k0 < best ? k0 : best;
    best = k1 < best ? k1 : best;
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float cg_box_lb(const float *bx, float qx, float qy, float qz) {
  const float ex = fmaxf(fmaxf(bx[0] - qx, qx - bx[3]), 0.0f);
  const float ey = fmaxf(fmaxf(bx[1] - qy, qy - bx[4]), 0.0f);
  const float ez = fmaxf(fmaxf(bx[2] - qz, qz - bx[5]), 0.0f);
  return (ex * ex + ey * ey) + ez * ez;
}

#define CG_GROUP 8      /* target tiles per group box */
// grid: (query tiles / 4, segments S, pairs B).  Segment s scans target tiles [s * seg_tiles, (s+1) * seg_tiles) after
// seeding from the lanes' own key-rank tiles (anywhere in the cloud); partial keys go to pkey[b][s][sorted query].
__global__ __launch_bounds__(256) void k_cg_search(const float4 *__restrict__ sq, const float4 *__restrict__ st,
                                                   const float *__restrict__ boxes, const int32_t *__restrict__ off,
                                                   int Npad, int Mpad, int T, int B, int S, int seg_tiles,
                                                   const int32_t *__restrict__ mcnt, int M, u64 *__restrict__ pkey) {
  __shared__ float sb[CG_LDS_TILES * 6];
  __shared__ float gb[(CG_LDS_TILES / CG_GROUP) * 6];
  __shared__ float4 slots[4][CG_TILE];
  const int b = blockIdx.z, seg = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int qt = blockIdx.x * 4 + wid;
  const bool wave_on = qt * CG_TILE < Npad;
  const float4 q = wave_on ? sq[(int64_t)b * Npad + qt * CG_TILE + lane] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
  const bool on = __float_as_int(q.w) >= 0;
  u64 best = CG_KEY_INIT;
  const int Mb = mcnt ? mcnt[b] : M;
  const int Tb = (Mb + CG_TILE - 1) / CG_TILE;
  const float4 *tb = st + (int64_t)b * Mpad;

  // seeds: every lane's own cell rank in the target ordering names a tile; the wave evaluates the distinct ones
  // (a handful, the queries are sorted by the same key) so each lane starts from a near-final bound
  if (wave_on && Tb > 0) {
    int ts = on ? off[((int64_t)B + b) * CG_BINS + cg_key(q.x, q.y, q.z)] / CG_TILE : -1;
    ts = ts >= Tb ? Tb - 1 : ts;
    u64 todo = __ballot(on);
    int guard = 0;
    while (todo && guard++ < 12) {
      const int l = __builtin_ctzll(todo);
      const int t = __builtin_amdgcn_readlane(ts, l);
      cg_eval_tile(tb + t * CG_TILE, slots[wid], lane, q.x, q.y, q.z, best);
      todo &= ~__ballot(ts == t);
    }
  }

  const int tbeg = seg * seg_tiles, tend = (tbeg + seg_tiles < Tb) ? tbeg + seg_tiles : Tb;
  for (int t0 = tbeg; t0 < tend; t0 += CG_LDS_TILES) {
    const int nt = (tend - t0 < CG_LDS_TILES) ? (tend - t0) : CG_LDS_TILES;
    const int ng = (nt + CG_GROUP - 1) / CG_GROUP;
    __syncthreads();
    for (int e = threadIdx.x; e < nt * 6; e += 256) sb[e] = boxes[((int64_t)b * T + t0) * 6 + e];
    __syncthreads();
    for (int e = threadIdx.x; e < ng * 6; e += 256) {      // union box of each group of CG_GROUP tiles
      const int g = e / 6, k = e % 6;
      const int t1 = (g + 1) * CG_GROUP < nt ? (g + 1) * CG_GROUP : nt;
      float v = sb[g * CG_GROUP * 6 + k];
      for (int t = g * CG_GROUP + 1; t < t1; ++t) v = k < 3 ? fminf(v, sb[t * 6 + k]) : fmaxf(v, sb[t * 6 + k]);
      gb[e] = v;
    }
    __syncthreads();
    if (!wave_on) continue;
    for (int g = 0; g < ng; ++g) {
      // keep unless lb > best on every lane with a query; NaN queries (lb = NaN) keep everything, like the full scan
      float bd = __int_as_float((int)(best >> 32));
      if (__ballot(on && !(cg_box_lb(gb + g * 6, q.x, q.y, q.z) > bd)) == 0ull) continue;
      const int t1 = (g + 1) * CG_GROUP < nt ? (g + 1) * CG_GROUP : nt;
      for (int t = g * CG_GROUP; t < t1; ++t) {
        bd = __int_as_float((int)(best >> 32));
        if (__ballot(on && !(cg_box_lb(sb + t * 6, q.x, q.y, q.z) > bd)) == 0ull) continue;
        cg_eval_tile(tb + (t0 + t) * CG_TILE, slots[wid], lane, q.x, q.y, q.z, best);
      }
    }
  }
  if (wave_on) pkey[((int64_t)b * S + seg) * Npad + qt * CG_TILE + lane] = best;
}

// minimum key over the segments -> dist / idx at the query's original position
__global__ void k_cg_merge(const float4 *__restrict__ sq, const u64 *__restrict__ pkey, int N, int Npad, int S,
                           float *__restrict__ dist, int32_t *__restrict__ idx) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Npad) return;
  const int qi = __float_as_int(sq[(int64_t)b * Npad + i].w);
  if (qi < 0) return;
  u64 best = pkey[((int64_t)b * S) * Npad + i];
  for (int s = 1; s < S; ++s) {
    const u64 k = pkey[((int64_t)b * S + s) * Npad + i];
    best = k < best ? k : best;
  }
  dist[(int64_t)b * N + qi] = __int_as_float((int)(best >> 32));
  idx[(int64_t)b * N + qi] = (int32_t)(unsigned)best;
}

static int cg_segments(int B, int N, int M) {
  const int forced = rslo_tune(RSLO_TUNE_CHAMFER_SEGMENTS);
  if (forced >= 1 && forced <= 8) {
    const int cap = (int)rslo_cdiv(rslo_cdiv(M > 0 ? M : 1, CG_TILE), CG_GROUP);
    return forced > cap ? (cap < 1 ? 1 : cap) : forced;
  }
  const int64_t qwaves = rslo_cdiv(N > 0 ? N : 1, CG_TILE) * (B > 0 ? B : 1);
  int S = (int)rslo_cdiv(8192, qwaves);            // enough waves to balance the heavy (far-query) ones
  const int maxS = (int)rslo_cdiv(rslo_cdiv(M > 0 ? M : 1, CG_TILE), 4 * CG_GROUP);
  if (S > maxS) S = maxS;
  if (S > 8) S = 8;
  return S < 1 ? 1 : S;
}

static inline int cg_pad(int n) { return (int)(rslo_cdiv(n > 0 ? n : 1, CG_TILE) * CG_TILE); }

extern "C" size_t rslo_chamfer_grid_ws_bytes(int B, int N, int M) {
  const size_t b = (size_t)(B > 0 ? B : 1);
  const size_t hist = 2 * b * CG_BINS * sizeof(int32_t);
  const size_t sorted = b * ((size_t)cg_pad(N) + (size_t)cg_pad(M)) * sizeof(float4);
  const size_t boxes = b * (size_t)(cg_pad(M) / CG_TILE) * 6 * sizeof(float);
  const size_t keys = b * (size_t)cg_segments(B, N, M) * (size_t)cg_pad(N) * sizeof(u64);
  return 2 * hist + sorted + boxes + 2 * b * CG_NCHUNK * sizeof(int32_t) + keys + 512;
}

extern "C" int rslo_chamfer_grid_nn(const float *xyz1, const float *xyz2, int B, int N, int M, const int32_t *ncnt,
                                    const int32_t *mcnt, float *dist, int32_t *idx, void *ws, size_t ws_bytes,
                                    void *stream) {
  hipStream_t s = (hipStream_t)stream;
  RSLO_CHECK_ARG(B >= 0 && N >= 0 && M >= 1, "chamfer_grid_nn: need M >= 1");
  if (B == 0 || N == 0) return RSLO_OK;
  if (ws_bytes < rslo_chamfer_grid_ws_bytes(B, N, M)) {
    rslo_set_error("chamfer_grid_nn: workspace too small");
    return RSLO_EWS;
  }
  const int Npad = cg_pad(N), Mpad = cg_pad(M), T = Mpad / CG_TILE;
  const size_t hist_n = (size_t)2 * B * CG_BINS;
  int32_t *hist = (int32_t *)ws;
  int32_t *cursor = hist + hist_n;
  float4 *sq = (float4 *)(((uintptr_t)(cursor + hist_n) + 15) & ~(uintptr_t)15);
  float4 *st = sq + (size_t)B * Npad;
  float *boxes = (float *)(st + (size_t)B * Mpad);
  int32_t *sums = (int32_t *)(boxes + (size_t)B * T * 6);
  u64 *pkey = (u64 *)(((uintptr_t)(sums + (size_t)2 * B * CG_NCHUNK) + 15) & ~(uintptr_t)15);
  const int S = cg_segments(B, N, M);
  const int seg_tiles = (int)(rslo_cdiv(rslo_cdiv(T, S), CG_GROUP) * CG_GROUP);
  RSLO_HIP(hipMemsetAsync(hist, 0, 2 * hist_n * sizeof(int32_t), s));
  const int nmax = N > M ? N : M, pmax = Npad > Mpad ? Npad : Mpad;
  hipLaunchKernelGGL(k_cg_hist, dim3((unsigned)rslo_cdiv(nmax, 256), (unsigned)B, 2), dim3(256), 0, s, xyz1, xyz2, N,
                     M, ncnt, mcnt, hist, B, dist, idx);
  hipLaunchKernelGGL(k_cg_chunk_sums, dim3(CG_NCHUNK, (unsigned)(2 * B)), dim3(256), 0, s, hist, sums);
  hipLaunchKernelGGL(k_cg_scan, dim3(CG_NCHUNK, (unsigned)(2 * B)), dim3(256), 0, s, hist, sums);
  hipLaunchKernelGGL(k_cg_scatter, dim3((unsigned)rslo_cdiv(pmax, 256), (unsigned)B, 2), dim3(256), 0, s, xyz1, xyz2,
                     N, M, Npad, Mpad, ncnt, mcnt, hist, cursor, B, sq, st);
  hipLaunchKernelGGL(k_cg_tiles, dim3((unsigned)rslo_cdiv(T, 4), (unsigned)B), dim3(256), 0, s, st, Mpad, T, boxes);
  hipLaunchKernelGGL(k_cg_search, dim3((unsigned)rslo_cdiv(Npad / CG_TILE, 4), (unsigned)S, (unsigned)B), dim3(256), 0, s,
                     sq, st, boxes, hist, Npad, Mpad, T, B, S, seg_tiles, mcnt, M, pkey);
  hipLaunchKernelGGL(k_cg_merge, dim3((unsigned)rslo_cdiv(Npad, 256), (unsigned)B), dim3(256), 0, s, sq, pkey, N, Npad, S,
                     dist, idx);
  RSLO_CHECK_LAUNCH("chamfer_grid_nn");
  return RSLO_OK;
}
