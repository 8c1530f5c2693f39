// Site index (hash), rulebook builders, voxelizer, VFE mean and dense scatter for gfx950.
// All kernels are HBM/L2-bound integer work: one thread per (site, kernel-offset) with the
// offset fastest so table writes are coalesced; the hash tables (<= a few MB) live in L2.
#include <stdarg.h>
#include <string.h>

#include "rslo_common.h"

// ---------------------------------------------------------------------------------------
// error text
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
extern "C" void rslo_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *rslo_last_error(void) { return g_err; }
extern "C" int rslo_abi_version(void) { return 1; }

// ---------------------------------------------------------------------------------------
// exclusive scan of int32 values (or popcounts of uint32 words), 1024 elements per block
// ---------------------------------------------------------------------------------------
#define SCAN_BLOCK 256
#define SCAN_ITEMS 4
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

template <bool POPC>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_block(const uint32_t *__restrict__ in,
                                                           int32_t *__restrict__ out,
                                                           int32_t *__restrict__ sums, int64_t n) {
  __shared__ int32_t wsum[SCAN_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int32_t tsum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    uint32_t x = (base + j < n) ? in[base + j] : 0u;
    v[j] = POPC ? __popc(x) : (int32_t)x;
    tsum += v[j];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int32_t inc = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  int32_t run = woff + inc - tsum;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) out[base + j] = run;
    run += v[j];
  }
  if (threadIdx.x == SCAN_BLOCK - 1) sums[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(int32_t *__restrict__ sums, int64_t nb,
                                                          int32_t *__restrict__ total) {
  __shared__ int32_t wsum[SCAN_BLOCK / 64];
  __shared__ int32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int64_t b0 = 0; b0 < nb; b0 += SCAN_BLOCK) {
    int64_t i = b0 + threadIdx.x;
    int32_t v = (i < nb) ? sums[i] : 0;
    int32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    int32_t woff = carry_s;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    if (i < nb) sums[i] = woff + inc - v;
    __syncthreads();
    if (threadIdx.x == SCAN_BLOCK - 1) carry_s = woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = carry_s;
}

__global__ void k_scan_add(int32_t *__restrict__ out, const int32_t *__restrict__ sums, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += sums[i / SCAN_TILE];
}

extern "C" size_t rslo_scan_ws_bytes(int64_t n) {
  return (size_t)(rslo_cdiv(n > 0 ? n : 1, SCAN_TILE) + 4) * sizeof(int32_t);
}

// in: n uint32 (flags or bitmap words), out: n exclusive prefix, total -> *d_total
template <bool POPC>
static int scan_exclusive(const uint32_t *in, int32_t *out, int64_t n, void *ws, size_t ws_bytes,
                          int32_t *d_total, hipStream_t st) {
  if (ws_bytes < rslo_scan_ws_bytes(n)) {
    rslo_set_error("scan workspace too small");
    return RSLO_EWS;
  }
  int32_t *sums = (int32_t *)ws;
  int64_t nb = rslo_cdiv(n > 0 ? n : 1, SCAN_TILE);
  hipLaunchKernelGGL(k_scan_block<POPC>, dim3((unsigned)nb), dim3(SCAN_BLOCK), 0, st, in, out, sums, n);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, st, sums, nb, d_total);
  if (n > 0)
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)rslo_cdiv(n, 256)), dim3(256), 0, st, out, sums, n);
  RSLO_CHECK_LAUNCH("scan");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// hash build
// ---------------------------------------------------------------------------------------
extern "C" int64_t rslo_hash_capacity(int64_t n) {
  int64_t c = 1024;
  while (c < 2 * n) c <<= 1;
  return c;
}

// Row counts of the structure kernels come either from the host (N) or, inside rslo_plan_encoder, from a device word
// (d_n != NULL: the count an earlier kernel of the same stream produced -- no host read in between).  Every kernel
// walks its work items in a grid-stride loop, so a launch sized for the CAPACITY of a level does the work of its
// actual size.
__device__ __forceinline__ int64_t rslo_rows(int64_t N, const int32_t *__restrict__ d_n) {
  return d_n ? (int64_t)*d_n : N;
}
#define RSLO_GRID_STRIDE(i, total) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)
static inline unsigned rb_grid(int64_t items) {
  const int64_t b = rslo_cdiv(items > 0 ? items : 1, 256);
  return (unsigned)(b < 65536 ? b : 65536);
}

__global__ void k_hash_insert(const int32_t *__restrict__ coords, int64_t N, const int32_t *__restrict__ d_n, Dims3 s,
                              uint32_t *__restrict__ keys, int32_t *__restrict__ vals, uint32_t mask,
                              int shift) {
  const int64_t n = rslo_rows(N, d_n);
  RSLO_GRID_STRIDE(i, n) {
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    const uint32_t key = rslo_lin(c.x, c.y, c.z, c.w, s);
    uint32_t p = rslo_hslot(key, shift);
    while (true) {
      uint32_t prev = atomicCAS(&keys[p], RSLO_EMPTY_KEY, key);
      if (prev == RSLO_EMPTY_KEY || prev == key) {
        vals[p] = (int32_t)i;
        break;
      }
      p = (p + 1) & mask;
    }
  }
}

static int check_volume(int B, const int32_t *d) {
  double v = (double)B * d[0] * d[1] * d[2];
  if (v >= 4294967295.0) {
    rslo_set_error("volume %g does not fit the 32-bit key space", v);
    return RSLO_ERANGE;
  }
  return RSLO_OK;
}

extern "C" int rslo_hash_build(const int32_t *coords, int64_t N, int B, const int32_t *d,
                               uint32_t *keys, int32_t *vals, int64_t cap, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(cap >= 2 * N && (cap & (cap - 1)) == 0, "hash_build: cap must be a power of two >= 2N");
  if (int rc = check_volume(B, d)) return rc;
  RSLO_HIP(hipMemsetAsync(keys, 0xFF, (size_t)cap * sizeof(uint32_t), st));
  if (N == 0) return RSLO_OK;
  const int shift = 32 - rslo_log2_i64(cap);
  hipLaunchKernelGGL(k_hash_insert, dim3(rb_grid(N)), dim3(256), 0, st, coords, N, (const int32_t *)nullptr,
                     Dims3{d[0], d[1], d[2]}, keys, vals, (uint32_t)(cap - 1), shift);
  RSLO_CHECK_LAUNCH("hash_insert");
  return RSLO_OK;
}

// (row, kz, ky, kx) of thread t = row * K + (kz * ks.b + ky) * ks.c + kx.  The generic form is one 64-bit and four 32-bit
// divisions by run-time values (~200 instructions on this ISA, more than the rest of these kernels); every layer of the
// path has 3x3x3 kernels, for which FAST decomposes with constant divisors in 32 bits (launchers check row * 27 < 2^31).
template <bool FAST>
__device__ __forceinline__ void rb_decompose(int64_t t, Int3 ks, int64_t &row, int &kz, int &ky, int &kx) {
  if (FAST) {
    const unsigned u = (unsigned)t, r = u / 27u, k = u - r * 27u;
    row = r;
    kz = (int)(k / 9u);
    ky = (int)((k / 3u) % 3u);
    kx = (int)(k % 3u);
  } else {
    const int K = ks.a * ks.b * ks.c;
    row = t / K;
    int k = (int)(t - row * K);
    kx = k % ks.c;
    k /= ks.c;
    ky = k % ks.b;
    kz = k / ks.b;
  }
}
// t / s and t % s == 0 for a stride component (FAST2: every stride is 2)
template <bool FAST2>
__device__ __forceinline__ bool rb_unstride(int t, int s, int &q) {
  if (FAST2) {
    q = t >> 1;
    return (t & 1) == 0;
  }
  q = t / s;
  return t - q * s == 0;
}
static inline bool rb_fast_k(const int32_t *ks, int64_t rows) {
  return ks[0] == 3 && ks[1] == 3 && ks[2] == 3 && rows * 27 < ((int64_t)1 << 31);
}
static inline bool rb_fast_s(const int32_t *st) { return st[0] == 2 && st[1] == 2 && st[2] == 2; }

// ---------------------------------------------------------------------------------------
// SubM rulebook: one thread per (row, offset)
// ---------------------------------------------------------------------------------------
template <bool FAST>
__global__ void k_rulebook_subm(const int32_t *__restrict__ coords, int64_t N, const int32_t *__restrict__ d_n,
                                Dims3 s, Int3 ks, const uint32_t *__restrict__ keys,
                                const int32_t *__restrict__ vals, uint32_t mask, int shift,
                                int32_t *__restrict__ nbr) {
  const int K = ks.a * ks.b * ks.c;
  const int64_t total = rslo_rows(N, d_n) * K;
  RSLO_GRID_STRIDE(t, total) {
    int64_t o;
    int kz, ky, kx;
    rb_decompose<FAST>(t, ks, o, kz, ky, kx);
    const int4 c = reinterpret_cast<const int4 *>(coords)[o];
    const int z = c.y + kz - ks.a / 2, y = c.z + ky - ks.b / 2, x = c.w + kx - ks.c / 2;
    int32_t r = -1;
    if (z >= 0 && z < s.d && y >= 0 && y < s.h && x >= 0 && x < s.w)
      r = rslo_hfind(keys, vals, mask, shift, rslo_lin(c.x, z, y, x, s));
    nbr[t] = r;
  }
}

extern "C" int rslo_rulebook_subm(const int32_t *coords, int64_t N, int B, const int32_t *d,
                                  const int32_t *ks, const uint32_t *keys, const int32_t *vals,
                                  int64_t cap, int32_t *nbr, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (int rc = check_volume(B, d)) return rc;
  if (N == 0) return RSLO_OK;
  const int K = ks[0] * ks[1] * ks[2];
  const int shift = 32 - rslo_log2_i64(cap);
  if (rb_fast_k(ks, N))
    hipLaunchKernelGGL(k_rulebook_subm<true>, dim3(rb_grid(N * K)), dim3(256), 0, st, coords, N, (const int32_t *)nullptr,
                       Dims3{d[0], d[1], d[2]}, Int3{ks[0], ks[1], ks[2]}, keys, vals, (uint32_t)(cap - 1), shift, nbr);
  else
    hipLaunchKernelGGL(k_rulebook_subm<false>, dim3(rb_grid(N * K)), dim3(256), 0, st, coords, N, (const int32_t *)nullptr,
                       Dims3{d[0], d[1], d[2]}, Int3{ks[0], ks[1], ks[2]}, keys, vals, (uint32_t)(cap - 1), shift, nbr);
  RSLO_CHECK_LAUNCH("rulebook_subm");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// strided conv: output set through a bitmap over the output volume (gives the ascending
// linear order without a sort), then the two neighbour tables by hash probes.
// ---------------------------------------------------------------------------------------
extern "C" int64_t rslo_conv_bitmap_words(int B, const int32_t *od) {
  int64_t vol = (int64_t)B * od[0] * od[1] * od[2];
  return (vol + 31) / 32;
}

// One thread per INPUT row: along every axis an input coordinate c feeds the outputs o with o * stride - pad + k = c, i.e.
// the k of the right parity -- 1 or 2 of 3 for a 3 / 2 / 1 convolution -- so a row marks 1..8 output sites.  (One thread
// per (row, offset) walked all 27 candidates of which ~3.4 survive: 81 us for 250 k rows, now ~15.)
__global__ void k_conv_mark(const int32_t *__restrict__ coords, int64_t N, const int32_t *__restrict__ d_n, Int3 ks,
                            Int3 st, Int3 pd, Dims3 od, uint32_t *__restrict__ bitmap) {
  const int64_t n = rslo_rows(N, d_n);
  RSLO_GRID_STRIDE(i, n) {
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    int zs[3], ys[3], xs[3], nz = 0, ny = 0, nx = 0;
    for (int k = 0; k < ks.a && k < 3; ++k) {
      const int t = c.y + pd.a - k;
      if (t >= 0 && t % st.a == 0 && t / st.a < od.d) zs[nz++] = t / st.a;
    }
    for (int k = 0; k < ks.b && k < 3; ++k) {
      const int t = c.z + pd.b - k;
      if (t >= 0 && t % st.b == 0 && t / st.b < od.h) ys[ny++] = t / st.b;
    }
    for (int k = 0; k < ks.c && k < 3; ++k) {
      const int t = c.w + pd.c - k;
      if (t >= 0 && t % st.c == 0 && t / st.c < od.w) xs[nx++] = t / st.c;
    }
    for (int a = 0; a < nz; ++a)
      for (int b = 0; b < ny; ++b)
        for (int e = 0; e < nx; ++e) {
          const uint32_t lin = rslo_lin(c.x, zs[a], ys[b], xs[e], od);
          const uint32_t bit = 1u << (lin & 31);
          // an output site has ~6 contributing inputs: skip the atomic when the bit is already there
          if (!(__builtin_nontemporal_load(&bitmap[lin >> 5]) & bit)) atomicOr(&bitmap[lin >> 5], bit);
        }
  }
}

extern "C" int rslo_conv_out_count(const int32_t *coords_in, int64_t N, int B, const int32_t *ks,
                                   const int32_t *stride, const int32_t *pad, const int32_t *od,
                                   uint32_t *bitmap, int32_t *word_prefix, int64_t words, void *scan_ws,
                                   size_t scan_ws_bytes, int32_t *d_count, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (int rc = check_volume(B, od)) return rc;
  RSLO_CHECK_ARG(words == rslo_conv_bitmap_words(B, od), "conv_out_count: wrong bitmap size");
  RSLO_HIP(hipMemsetAsync(bitmap, 0, (size_t)words * sizeof(uint32_t), st));
  const int K = ks[0] * ks[1] * ks[2];
  if (N > 0) {
    RSLO_CHECK_ARG(ks[0] <= 3 && ks[1] <= 3 && ks[2] <= 3, "conv_out_count: kernel extent > 3 unsupported");
    hipLaunchKernelGGL(k_conv_mark, dim3(rb_grid(N)), dim3(256), 0, st, coords_in, N, (const int32_t *)nullptr,
                       Int3{ks[0], ks[1], ks[2]}, Int3{stride[0], stride[1], stride[2]},
                       Int3{pad[0], pad[1], pad[2]}, Dims3{od[0], od[1], od[2]}, bitmap);
    RSLO_CHECK_LAUNCH("conv_mark");
  }
  return scan_exclusive<true>(bitmap, word_prefix, words, scan_ws, scan_ws_bytes, d_count, st);
}

__global__ void k_conv_emit(const uint32_t *__restrict__ bitmap, const int32_t *__restrict__ prefix,
                            int64_t words, Dims3 od, int32_t *__restrict__ out_coords, int64_t M) {
  int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= words) return;
  uint32_t bits = bitmap[w];
  if (!bits) return;
  int32_t r = prefix[w];
  if (r >= M) return;          // rslo_plan_encoder: M is the level's capacity, rows past it are dropped (and flagged)
  while (bits) {
    const int b = __ffs(bits) - 1;
    bits &= bits - 1;
    uint32_t lin = (uint32_t)(w * 32 + b);
    const int x = lin % od.w;
    lin /= od.w;
    const int y = lin % od.h;
    lin /= od.h;
    const int z = lin % od.d;
    const int bb = lin / od.d;
    if (r < M) reinterpret_cast<int4 *>(out_coords)[r] = make_int4(bb, z, y, x);
    ++r;
  }
}

extern "C" int rslo_conv_out_coords(const uint32_t *bitmap, const int32_t *word_prefix, int64_t words,
                                    int B, const int32_t *od, int32_t *out_coords, int64_t M,
                                    void *stream) {
  hipStream_t st = (hipStream_t)stream;
  (void)B;
  if (M == 0 || words == 0) return RSLO_OK;
  hipLaunchKernelGGL(k_conv_emit, dim3((unsigned)rslo_cdiv(words, 256)), dim3(256), 0, st, bitmap,
                     word_prefix, words, Dims3{od[0], od[1], od[2]}, out_coords, M);
  RSLO_CHECK_LAUNCH("conv_emit");
  return RSLO_OK;
}

template <bool FAST>
__global__ void k_rulebook_conv(const int32_t *__restrict__ coords_out, int64_t M, const int32_t *__restrict__ d_m,
                                Dims3 id, Int3 ks, Int3 st, Int3 pd, const uint32_t *__restrict__ keys,
                                const int32_t *__restrict__ vals, uint32_t mask, int shift,
                                int32_t *__restrict__ nbr) {
  const int K = ks.a * ks.b * ks.c;
  const int64_t total = rslo_rows(M, d_m) * K;
  RSLO_GRID_STRIDE(t, total) {
    int64_t o;
    int kz, ky, kx;
    rb_decompose<FAST>(t, ks, o, kz, ky, kx);
    const int4 c = reinterpret_cast<const int4 *>(coords_out)[o];
    const int z = c.y * st.a - pd.a + kz, y = c.z * st.b - pd.b + ky, x = c.w * st.c - pd.c + kx;
    int32_t r = -1;
    if (z >= 0 && z < id.d && y >= 0 && y < id.h && x >= 0 && x < id.w)
      r = rslo_hfind(keys, vals, mask, shift, rslo_lin(c.x, z, y, x, id));
    nbr[t] = r;
  }
}

extern "C" int rslo_rulebook_conv(const int32_t *coords_out, int64_t M, int B, const int32_t *id,
                                  const int32_t *ks, const int32_t *stride, const int32_t *pad,
                                  const uint32_t *in_keys, const int32_t *in_vals, int64_t in_cap,
                                  int32_t *nbr, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (int rc = check_volume(B, id)) return rc;
  if (M == 0) return RSLO_OK;
  const int K = ks[0] * ks[1] * ks[2];
  const int shift = 32 - rslo_log2_i64(in_cap);
#define RB_CONV(F)                                                                                            \
  hipLaunchKernelGGL(k_rulebook_conv<F>, dim3(rb_grid(M * K)), dim3(256), 0, st, coords_out,                  \
                     M, (const int32_t *)nullptr, Dims3{id[0], id[1], id[2]}, Int3{ks[0], ks[1], ks[2]},       \
                     Int3{stride[0], stride[1], stride[2]}, Int3{pad[0], pad[1], pad[2]}, in_keys,             \
                     in_vals, (uint32_t)(in_cap - 1), shift, nbr)
  if (rb_fast_k(ks, M)) RB_CONV(true);
  else RB_CONV(false);
#undef RB_CONV
  RSLO_CHECK_LAUNCH("rulebook_conv");
  return RSLO_OK;
}

template <bool FAST, bool FAST2>
__global__ void k_rulebook_conv_T(const int32_t *__restrict__ coords_in, int64_t N, const int32_t *__restrict__ d_n,
                                  Dims3 od, Int3 ks, Int3 st, Int3 pd, const uint32_t *__restrict__ keys,
                                  const int32_t *__restrict__ vals, uint32_t mask, int shift,
                                  int32_t *__restrict__ nbrT) {
  const int K = ks.a * ks.b * ks.c;
  const int64_t total = rslo_rows(N, d_n) * K;
  RSLO_GRID_STRIDE(t, total) {
    int64_t i;
    int kz, ky, kx;
    rb_decompose<FAST>(t, ks, i, kz, ky, kx);
    const int4 c = reinterpret_cast<const int4 *>(coords_in)[i];
    const int tz = c.y + pd.a - kz, ty = c.z + pd.b - ky, tx = c.w + pd.c - kx;
    int32_t r = -1;
    int z, y, x;
    if (tz >= 0 && ty >= 0 && tx >= 0 && rb_unstride<FAST2>(tz, st.a, z) && rb_unstride<FAST2>(ty, st.b, y) &&
        rb_unstride<FAST2>(tx, st.c, x)) {
      if (z < od.d && y < od.h && x < od.w)
        r = rslo_hfind(keys, vals, mask, shift, rslo_lin(c.x, z, y, x, od));
    }
    nbrT[t] = r;
  }
}

extern "C" int rslo_rulebook_conv_T(const int32_t *coords_in, int64_t N, int B, const int32_t *od,
                                    const int32_t *ks, const int32_t *stride, const int32_t *pad,
                                    const uint32_t *out_keys, const int32_t *out_vals, int64_t out_cap,
                                    int32_t *nbrT, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (int rc = check_volume(B, od)) return rc;
  if (N == 0) return RSLO_OK;
  const int K = ks[0] * ks[1] * ks[2];
  const int shift = 32 - rslo_log2_i64(out_cap);
#define RB_CONVT(F, F2)                                                                                       \
  hipLaunchKernelGGL((k_rulebook_conv_T<F, F2>), dim3(rb_grid(N * K)), dim3(256), 0, st, coords_in,           \
                     N, (const int32_t *)nullptr, Dims3{od[0], od[1], od[2]}, Int3{ks[0], ks[1], ks[2]},       \
                     Int3{stride[0], stride[1], stride[2]}, Int3{pad[0], pad[1], pad[2]}, out_keys,            \
                     out_vals, (uint32_t)(out_cap - 1), shift, nbrT)
  if (rb_fast_k(ks, N) && rb_fast_s(stride)) RB_CONVT(true, true);
  else RB_CONVT(false, false);
#undef RB_CONVT
  RSLO_CHECK_LAUNCH("rulebook_conv_T");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// voxelizer.  Point i -> voxel key; hash slot keeps the smallest point index (first-come
// order) and a linked list of its points.  Voxel id = exclusive scan over "is first point"
// flags (= rank of the first index), rank of a point inside its voxel = number of list
// members with a smaller index.  Everything is order-independent, hence deterministic.
// ---------------------------------------------------------------------------------------
struct VoxWs {
  uint32_t *keys;
  int32_t *first, *head, *vid, *next, *slot, *pos, *cutoff;
  uint32_t *flags;
  void *scan_ws;
  size_t scan_bytes;
  int64_t cap;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t vox_ws_layout(int64_t P, void *base, VoxWs *w) {
  const int64_t cap = rslo_hash_capacity(P);
  size_t off = 0;
  char *b = (char *)base;
  auto take = [&](size_t bytes) {
    void *p = b ? (void *)(b + off) : nullptr;
    off += align256(bytes);
    return p;
  };
  // keys | head are filled with 0xFF and first | cutoff with 0x7F: adjacent, so each pair is ONE memset (cap * 4 is a
  // multiple of 256: no padding in between)
  void *keys = take(cap * 4), *head = take(cap * 4), *first = take(cap * 4);
  void *cutoff = take(256);
  void *vid = take(cap * 4);
  void *next = take((size_t)P * 4), *slot = take((size_t)P * 4), *pos = take((size_t)P * 4),
       *flags = take((size_t)P * 4);
  size_t sb = rslo_scan_ws_bytes(P);
  void *sws = take(sb);
  if (w) {
    w->keys = (uint32_t *)keys; w->first = (int32_t *)first; w->head = (int32_t *)head;
    w->vid = (int32_t *)vid; w->next = (int32_t *)next; w->slot = (int32_t *)slot;
    w->pos = (int32_t *)pos; w->flags = (uint32_t *)flags; w->cutoff = (int32_t *)cutoff;
    w->scan_ws = sws; w->scan_bytes = sb; w->cap = cap;
  }
  return off;
}

extern "C" size_t rslo_voxelize_ws_bytes(int64_t P) { return vox_ws_layout(P > 0 ? P : 1, nullptr, nullptr); }

struct VoxGeom {
  float lo[3], vs[3];
  int g[3];
};

__device__ __forceinline__ bool vox_coord(const float *__restrict__ p, const VoxGeom &G, int c[3]) {
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float q = __fdiv_rn(__fsub_rn(p[j], G.lo[j]), G.vs[j]);
    const float fl = floorf(q);
    if (!(fl >= 0.0f) || !(fl < (float)G.g[j])) return false;
    c[j] = (int)fl;
  }
  return true;
}

__global__ void k_vox_insert(const float *__restrict__ pts, int64_t P, int F, VoxGeom G,
                             uint32_t *__restrict__ keys, int32_t *__restrict__ first,
                             int32_t *__restrict__ head, int32_t *__restrict__ next,
                             int32_t *__restrict__ slot, uint32_t mask, int shift) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int c[3];
  if (!vox_coord(pts + i * F, G, c)) {
    slot[i] = -1;
    return;
  }
  const uint32_t key = ((uint32_t)c[2] * G.g[1] + c[1]) * G.g[0] + c[0];
  uint32_t s = rslo_hslot(key, shift);
  while (true) {
    uint32_t prev = atomicCAS(&keys[s], RSLO_EMPTY_KEY, key);
    if (prev == RSLO_EMPTY_KEY || prev == key) break;
    s = (s + 1) & mask;
  }
  atomicMin(&first[s], (int32_t)i);
  next[i] = atomicExch(&head[s], (int32_t)i);
  slot[i] = (int32_t)s;
}

__global__ void k_vox_flags(const int32_t *__restrict__ slot, const int32_t *__restrict__ first, int64_t P,
                            uint32_t *__restrict__ flags) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int32_t s = slot[i];
  flags[i] = (s >= 0 && first[s] == (int32_t)i) ? 1u : 0u;
}

__global__ void k_vox_assign(const uint32_t *__restrict__ keys, const int32_t *__restrict__ slot,
                             const uint32_t *__restrict__ flags, const int32_t *__restrict__ pos, int64_t P,
                             VoxGeom G, int max_voxels, int32_t *__restrict__ vid,
                             int32_t *__restrict__ coords, int32_t *__restrict__ cutoff,
                             int32_t *__restrict__ d_nvox) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  if (i == P - 1) {
    const int32_t tot = pos[i] + (int32_t)flags[i];
    *d_nvox = tot < max_voxels ? tot : max_voxels;
  }
  if (!flags[i]) return;
  const int32_t s = slot[i];
  const int32_t v = pos[i];
  vid[s] = v;
  if (v < max_voxels) {
    uint32_t key = keys[s];
    const int x = key % G.g[0];
    key /= G.g[0];
    const int y = key % G.g[1];
    const int z = key / G.g[1];
    coords[v * 3 + 0] = z;
    coords[v * 3 + 1] = y;
    coords[v * 3 + 2] = x;
  } else if (v == max_voxels) {
    *cutoff = (int32_t)i;  // the reference loop `break`s here
  }
}

__global__ void k_vox_fill(const float *__restrict__ pts, int64_t P, int F, int T,
                           const int32_t *__restrict__ slot, const int32_t *__restrict__ head,
                           const int32_t *__restrict__ next, const int32_t *__restrict__ vid,
                           const int32_t *__restrict__ cutoff, float *__restrict__ voxels,
                           int32_t *__restrict__ num_points) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int32_t s = slot[i];
  if (s < 0 || i >= (int64_t)*cutoff) return;
  int rank = 0;
  for (int32_t j = head[s]; j >= 0; j = next[j]) rank += (j < (int32_t)i);
  if (rank >= T) return;
  const int32_t v = vid[s];
  float *dst = voxels + ((int64_t)v * T + rank) * F;
  const float *src = pts + i * F;
  for (int f = 0; f < F; ++f) dst[f] = src[f];
  atomicAdd(&num_points[v], 1);
}

// the voxelizer's launches for one cloud (outputs already zero-filled by the caller)
static int vox_run(const float *points, int64_t P, int F, const VoxGeom &G, int T, int max_voxels, void *ws,
                   float *voxels, int32_t *coords, int32_t *num_points, int32_t *d_nvox, hipStream_t st) {
  VoxWs w;
  vox_ws_layout(P, ws, &w);
  RSLO_HIP(hipMemsetAsync(w.keys, 0xFF, (size_t)w.cap * 8, st));                 // keys | head
  RSLO_HIP(hipMemsetAsync(w.first, 0x7F, (size_t)w.cap * 4 + sizeof(int32_t), st));    // first | cutoff
  const unsigned nb = (unsigned)rslo_cdiv(P, 256);
  const int shift = 32 - rslo_log2_i64(w.cap);
  hipLaunchKernelGGL(k_vox_insert, dim3(nb), dim3(256), 0, st, points, P, F, G, w.keys, w.first, w.head,
                     w.next, w.slot, (uint32_t)(w.cap - 1), shift);
  hipLaunchKernelGGL(k_vox_flags, dim3(nb), dim3(256), 0, st, w.slot, w.first, P, w.flags);
  RSLO_CHECK_LAUNCH("vox_insert");
  if (int rc = scan_exclusive<false>(w.flags, w.pos, P, w.scan_ws, w.scan_bytes, nullptr, st)) return rc;
  hipLaunchKernelGGL(k_vox_assign, dim3(nb), dim3(256), 0, st, w.keys, w.slot, w.flags, w.pos, P, G,
                     max_voxels, w.vid, coords, w.cutoff, d_nvox);
  hipLaunchKernelGGL(k_vox_fill, dim3(nb), dim3(256), 0, st, points, P, F, T, w.slot, w.head, w.next,
                     w.vid, w.cutoff, voxels, num_points);
  RSLO_CHECK_LAUNCH("vox_fill");
  return RSLO_OK;
}

extern "C" int rslo_voxelize(const float *points, int64_t P, int F, const float *range6,
                             const float *vsize3, const int32_t *grid_xyz, int T, int max_voxels,
                             void *ws, size_t ws_bytes, float *voxels, int32_t *coords,
                             int32_t *num_points, int32_t *d_nvox, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(F >= 3 && T >= 1 && max_voxels >= 1, "voxelize: bad F/T/max_voxels");
  RSLO_CHECK_ARG(P < (int64_t)2000000000, "voxelize: too many points");
  {
    int32_t d[3] = {grid_xyz[2], grid_xyz[1], grid_xyz[0]};
    if (int rc = check_volume(1, d)) return rc;
  }
  {
    // the four outputs are zero-filled; a caller that lays them out back to back (voxels | num_points | coords | d_nvox,
    // as rslo_amd.capi does) gets ONE fill instead of four
    const size_t bv = (size_t)max_voxels * T * F * sizeof(float), bn = (size_t)max_voxels * sizeof(int32_t),
                 bc = (size_t)max_voxels * 3 * sizeof(int32_t);
    char *v0 = (char *)voxels;
    if ((char *)num_points == v0 + bv && (char *)coords == v0 + bv + bn && (char *)d_nvox == v0 + bv + bn + bc) {
      RSLO_HIP(hipMemsetAsync(voxels, 0, bv + bn + bc + sizeof(int32_t), st));
    } else {
      RSLO_HIP(hipMemsetAsync(voxels, 0, bv, st));
      RSLO_HIP(hipMemsetAsync(num_points, 0, bn, st));
      RSLO_HIP(hipMemsetAsync(coords, 0, bc, st));
      RSLO_HIP(hipMemsetAsync(d_nvox, 0, sizeof(int32_t), st));
    }
  }
  if (P == 0) return RSLO_OK;
  if (ws_bytes < rslo_voxelize_ws_bytes(P)) {
    rslo_set_error("voxelize: workspace too small (%zu < %zu)", ws_bytes, rslo_voxelize_ws_bytes(P));
    return RSLO_EWS;
  }
  VoxGeom G;
  for (int j = 0; j < 3; ++j) {
    G.lo[j] = range6[j];
    G.vs[j] = vsize3[j];
    G.g[j] = grid_xyz[j];
  }
  return vox_run(points, P, F, G, T, max_voxels, ws, voxels, coords, num_points, d_nvox, st);
}

// ---------------------------------------------------------------------------------------
// VFE mean (a4)
// ---------------------------------------------------------------------------------------
__global__ void k_vfe_mean(const float *__restrict__ voxels, const int32_t *__restrict__ num, int64_t M,
                           int T, int F, float *__restrict__ out) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= M) return;
  const float inv_n = (float)num[v];
  float m[16];
  for (int f = 0; f < F; ++f) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s = __fadd_rn(s, voxels[((int64_t)v * T + t) * F + f]);
    m[f] = __fdiv_rn(s, inv_n);
  }
  if (F >= 7) {
    const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(m[4], m[4]), __fmul_rn(m[5], m[5])), __fmul_rn(m[6], m[6]));
    const float d = __fadd_rn(__fsqrt_rn(n2), 1e-12f);
    m[4] = __fdiv_rn(m[4], d);
    m[5] = __fdiv_rn(m[5], d);
    m[6] = __fdiv_rn(m[6], d);
  }
  for (int f = 0; f < F; ++f) out[v * F + f] = m[f];
}

// The same arithmetic with coalesced reads: a wave copies the T*F floats of its 64 voxels (one contiguous run of the
// voxel tensor) into its LDS slab with 16-byte loads and every lane then sums its own voxel from there in the same t
// order (identical bits).  One thread per voxel straight from memory walks 64 cache lines per load instruction:
// 50 us for 125 k voxels of 10 x 7 floats (0.7 TB/s).
__global__ __launch_bounds__(256) void k_vfe_mean_lds(const float *__restrict__ voxels, const int32_t *__restrict__ num,
                                                      int64_t M, int T, int F, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float vfe_sm[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int TF = T * F;
  float *my = vfe_sm + (size_t)wid * 64 * TF;
  const int64_t v0 = ((int64_t)blockIdx.x * 4 + wid) * 64;
  if (v0 >= M) return;
  const int nv = (int)((M - v0 < 64) ? (M - v0) : 64);
  const int nfl = nv * TF, n4 = nfl >> 2;
  const float *src = voxels + v0 * TF;            // 64 * TF floats per wave: 16-byte aligned whenever `voxels` is
  for (int i = lane; i < n4; i += 64) reinterpret_cast<float4 *>(my)[i] = reinterpret_cast<const float4 *>(src)[i];
  for (int i = (n4 << 2) + lane; i < nfl; i += 64) my[i] = src[i];
  __builtin_amdgcn_wave_barrier();                // LDS operations of one wave complete in order
  if (lane >= nv) return;
  const int64_t v = v0 + lane;
  const float inv_n = (float)num[v];
  const float *p = my + lane * TF;
  float m[16];
  for (int f = 0; f < F; ++f) {
    float s = 0.f;
    for (int t = 0; t < T; ++t) s = __fadd_rn(s, p[t * F + f]);
    m[f] = __fdiv_rn(s, inv_n);
  }
  if (F >= 7) {
    const float n2 = __fadd_rn(__fadd_rn(__fmul_rn(m[4], m[4]), __fmul_rn(m[5], m[5])), __fmul_rn(m[6], m[6]));
    const float d = __fadd_rn(__fsqrt_rn(n2), 1e-12f);
    m[4] = __fdiv_rn(m[4], d);
    m[5] = __fdiv_rn(m[5], d);
    m[6] = __fdiv_rn(m[6], d);
  }
  for (int f = 0; f < F; ++f) out[v * F + f] = m[f];
}

extern "C" int rslo_vfe_mean(const float *voxels, const int32_t *num_points, int64_t M, int T, int F,
                             float *out, void *stream) {
  RSLO_CHECK_ARG(F <= 16, "vfe_mean: F > 16 unsupported");
  if (M == 0) return RSLO_OK;
  const size_t lds = (size_t)4 * 64 * T * F * sizeof(float);
  const bool lds_on = rslo_tune(RSLO_TUNE_VFE_LDS) != 0;
  if (lds_on && lds <= 80 * 1024 && ((uintptr_t)voxels & 15) == 0) {
    static bool attr_set = false;
    if (!attr_set) {
      RSLO_HIP(hipFuncSetAttribute((const void *)k_vfe_mean_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      attr_set = true;
    }
    hipLaunchKernelGGL(k_vfe_mean_lds, dim3((unsigned)rslo_cdiv(M, 256)), dim3(256), lds, (hipStream_t)stream, voxels,
                       num_points, M, T, F, out);
    RSLO_CHECK_LAUNCH("vfe_mean");
    return RSLO_OK;
  }
  hipLaunchKernelGGL(k_vfe_mean, dim3((unsigned)rslo_cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream, voxels,
                     num_points, M, T, F, out);
  RSLO_CHECK_LAUNCH("vfe_mean");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// dense scatter / gather (a8)
// ---------------------------------------------------------------------------------------
// frames > 1: the batched encoder tensor holds frame t of sample b at batch index t * B + b, and the dense tensor is laid
// out [B, frames, C, D, H, W] -- viewed [B, frames * C * D, H, W] it IS the channel concatenation of a sample's frames that
// the BEV head builds (rslo/models/odom_pred.py:170 `torch.cat(xs, dim=1)`), with no copy.
template <bool GATHER>
__global__ void k_dense(float *__restrict__ feat, const int32_t *__restrict__ coords, int64_t M, int C,
                        Dims3 s, float *__restrict__ dense, int B, int frames) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * C) return;
  const int64_t r = t % M;
  const int ch = (int)(t / M);
  int4 c = reinterpret_cast<const int4 *>(coords)[r];
  if (c.x < 0) return;      // padding row of a capacity-sized plan (rslo_plan_encoder_pad_tails): no site
  if (frames > 1) c.x = (c.x % B) * frames + c.x / B;
  const int64_t vol = (int64_t)s.d * s.h * s.w;
  const int64_t off = ((int64_t)c.x * C + ch) * vol + ((int64_t)c.y * s.h + c.z) * s.w + c.w;
  if (GATHER)
    feat[r * C + ch] = dense[off];
  else
    dense[off] = feat[r * C + ch];
}

// Tiled form (round 5): 64 rows x 64 channels per workgroup through LDS, so that BOTH sides are walked in their own fast
// direction -- the row side row-contiguous (256 bytes per row), the dense side with consecutive rows on consecutive lanes (the
// rows are sorted by cell index: neighbours along x share cache lines).  The one-thread-per-element kernel strides the row
// side by C floats per lane (113 us for the backward gather of the C3 step's 21 k rows x 64 channels).
template <bool GATHER>
__global__ __launch_bounds__(256) void k_dense_tiled(float *__restrict__ feat, const int32_t *__restrict__ coords, int64_t M,
                                                     int C, Dims3 s, float *__restrict__ dense, int B, int frames) {
  __shared__ float tile[64][65];
  __shared__ int64_t base[64];
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int nc = (C - c0 < 64) ? (C - c0) : 64;
  const int tid = threadIdx.x;
  const int64_t vol = (int64_t)s.d * s.h * s.w;
  if (tid < 64) {
    const int64_t r = row0 + tid;
    int64_t b = -1;
    if (r < M) {
      int4 c = reinterpret_cast<const int4 *>(coords)[r];
      if (c.x >= 0) {      // (a negative batch index: padding row of a capacity-sized plan)
        if (frames > 1) c.x = (c.x % B) * frames + c.x / B;
        b = ((int64_t)c.x * C) * vol + ((int64_t)c.y * s.h + c.z) * s.w + c.w;
      }
    }
    base[tid] = b;
  }
  __syncthreads();
  const int lr = tid & 63, lq = tid >> 6;           // dense side: lane = row, 4 channel phases
  const int fc = tid & 63, fq = tid >> 6;           // row side:   lane = channel, 4 row phases
  if (GATHER) {
    const int64_t b = base[lr];
    for (int ch = lq; ch < nc; ch += 4) tile[lr][ch] = b >= 0 ? dense[b + (int64_t)(c0 + ch) * vol] : 0.f;
    __syncthreads();
    if (fc < nc)
      for (int r = fq; r < 64; r += 4)
        if (row0 + r < M) feat[(row0 + r) * C + c0 + fc] = tile[r][fc];
  } else {
    if (fc < nc)
      for (int r = fq; r < 64; r += 4)
        if (row0 + r < M) tile[r][fc] = feat[(row0 + r) * C + c0 + fc];
    __syncthreads();
    const int64_t b = base[lr];
    if (b >= 0)
      for (int ch = lq; ch < nc; ch += 4) dense[b + (int64_t)(c0 + ch) * vol] = tile[lr][ch];
  }
}

static int dense_run(bool gather, float *feat, const int32_t *coords, int64_t M, int C, int B, int frames,
                     const int32_t *d, float *dense, hipStream_t st) {
  RSLO_CHECK_ARG(frames >= 1 && B >= 1 && B % frames == 0, "dense: the batch must hold whole frames");
  if (!gather) RSLO_HIP(hipMemsetAsync(dense, 0, (size_t)B * C * d[0] * d[1] * d[2] * sizeof(float), st));
  if (M == 0) return RSLO_OK;
  if (rslo_tune(RSLO_TUNE_DENSE_TILED) && C >= 16) {
    const dim3 grid((unsigned)rslo_cdiv(M, 64), (unsigned)rslo_cdiv(C, 64));
    if (gather)
      hipLaunchKernelGGL(k_dense_tiled<true>, grid, dim3(256), 0, st, feat, coords, M, C, Dims3{d[0], d[1], d[2]}, dense,
                         B / frames, frames);
    else
      hipLaunchKernelGGL(k_dense_tiled<false>, grid, dim3(256), 0, st, feat, coords, M, C, Dims3{d[0], d[1], d[2]}, dense,
                         B / frames, frames);
    RSLO_CHECK_LAUNCH("dense(tiled)");
    return RSLO_OK;
  }
  const unsigned nb = (unsigned)rslo_cdiv(M * C, 256);
  if (gather)
    hipLaunchKernelGGL(k_dense<true>, dim3(nb), dim3(256), 0, st, feat, coords, M, C, Dims3{d[0], d[1], d[2]}, dense,
                       B / frames, frames);
  else
    hipLaunchKernelGGL(k_dense<false>, dim3(nb), dim3(256), 0, st, feat, coords, M, C, Dims3{d[0], d[1], d[2]}, dense,
                       B / frames, frames);
  RSLO_CHECK_LAUNCH("dense");
  return RSLO_OK;
}

extern "C" int rslo_dense_scatter(const float *feat, const int32_t *coords, int64_t M, int C, int B,
                                  const int32_t *d, float *out, void *stream) {
  return dense_run(false, const_cast<float *>(feat), coords, M, C, B, 1, d, out, (hipStream_t)stream);
}

extern "C" int rslo_dense_gather(const float *dense, const int32_t *coords, int64_t M, int C, int B,
                                 const int32_t *d, float *dfeat, void *stream) {
  return dense_run(true, dfeat, coords, M, C, B, 1, d, const_cast<float *>(dense), (hipStream_t)stream);
}

extern "C" int rslo_dense_scatter_frames(const float *feat, const int32_t *coords, int64_t M, int C, int B, int frames,
                                         const int32_t *d, float *out, void *stream) {
  return dense_run(false, const_cast<float *>(feat), coords, M, C, B, frames, d, out, (hipStream_t)stream);
}

extern "C" int rslo_dense_gather_frames(const float *dense, const int32_t *coords, int64_t M, int C, int B, int frames,
                                        const int32_t *d, float *dfeat, void *stream) {
  return dense_run(true, dfeat, coords, M, C, B, frames, d, const_cast<float *>(dense), (hipStream_t)stream);
}

// Per-cell channel sums of a BEV tensor [B, G * Cg, HW] over each of its G channel groups -> [B, G, HW] in ONE pass.
// The head and the logging extras of the training forward need, per frame of a sample, "is any channel non-zero"
// (odom_pred.py:165-168 input mask, voxel_odom_net.py:519-527 feature mask) and the channel mean (middle_feature): four
// torch reductions over 35-70 MB plus a 70 MB concatenation; here the tensor is read once.  Channels are added in
// ascending order by one thread per cell (coalesced across cells).
__global__ void k_bev_channel_sums(const float *__restrict__ in, int G, int Cg, int64_t HW, float *__restrict__ out,
                                   float *__restrict__ mask_f, unsigned char *__restrict__ mask_b,
                                   unsigned char *__restrict__ outside_b) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const int g = blockIdx.y, b = blockIdx.z;
  const float *src = in + ((int64_t)(b * G + g) * Cg) * HW + p;
  float s = 0.f;
  int c = 0;
  for (; c + 8 <= Cg; c += 8) {      // 8 independent loads in flight, added in channel order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(c + u) * HW];
#pragma unroll
    for (int u = 0; u < 8; ++u) s = __fadd_rn(s, v[u]);
  }
  for (; c < Cg; ++c) s = __fadd_rn(s, src[(int64_t)c * HW]);
  out[(int64_t)(b * G + g) * HW + p] = s;
  if (g == 0) {       // occupancy of the first frame of the pair: the head's input mask
    const bool occ = s != 0.f;
    const int64_t o = (int64_t)b * HW + p;
    if (mask_f) mask_f[o] = occ ? 1.f : 0.f;
    if (mask_b) mask_b[o] = occ ? 1 : 0;
    if (outside_b) outside_b[o] = occ ? 0 : 1;
  }
}

extern "C" int rslo_bev_channel_sums_masks(const float *in, int B, int G, int Cg, int64_t HW, float *out, float *mask_f,
                                           unsigned char *mask_b, unsigned char *outside_b, void *stream) {
  RSLO_CHECK_ARG(in && out && B >= 1 && G >= 1 && Cg >= 1 && HW >= 1 && B < 65536 && G < 65536,
                 "rslo_bev_channel_sums: bad arguments");
  hipLaunchKernelGGL(k_bev_channel_sums, dim3((unsigned)rslo_cdiv(HW, 256), (unsigned)G, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, in, G, Cg, HW, out, mask_f, mask_b, outside_b);
  RSLO_CHECK_LAUNCH("k_bev_channel_sums");
  return RSLO_OK;
}

extern "C" int rslo_bev_channel_sums(const float *in, int B, int G, int Cg, int64_t HW, float *out, void *stream) {
  return rslo_bev_channel_sums_masks(in, B, G, Cg, HW, out, nullptr, nullptr, nullptr, stream);
}

// The logged extras of the training forward from those sums [B, T, HW] (voxel_odom_net.py:455-464 of the reference):
//   mask[b][p]    = (sum_t sums[b][t][p]) != 0
//   disp[t][b][p] = (d - min d) / (max d - min d + 1e-12),  d = sums[b][t][p] * (1 / Cg)  (min / max over the frame's
//                   whole batch), the arithmetic torch does for `mean -> (x - x.min()) / (x.max() - x.min() + 1e-12)`
// Two launches instead of ~17 reductions / element-wise ops: BD_PARTS workgroups per frame leave partial extrema, the
// apply launch folds them (64 values) and writes the maps.  (One workgroup per frame doing both passes was a 94 us
// latency chain on the training stream.)
#define BD_THREADS 256
#define BD_PARTS 64
__global__ __launch_bounds__(BD_THREADS) void k_bev_display_minmax(const float *__restrict__ sums, int B, int T, int64_t HW,
                                                                   float inv_c, float *__restrict__ part) {
  // plain operators under the pragma: the __f*_rn device functions are inlined from the HIP headers WITH their
  // contraction flags
#pragma clang fp contract(off)
  __shared__ float s_mn[BD_THREADS / 64], s_mx[BD_THREADS / 64];
  const int t = blockIdx.y, tid = threadIdx.x;
  const int64_t n = (int64_t)B * HW;
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * BD_THREADS + tid; i < n; i += (int64_t)BD_PARTS * BD_THREADS) {
    const int64_t b = i / HW, p = i - b * HW;
    const float d = sums[(b * T + t) * HW + p] * inv_c;
    mn = fminf(mn, d);
    mx = fmaxf(mx, d);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  if ((tid & 63) == 0) { s_mn[tid >> 6] = mn; s_mx[tid >> 6] = mx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < BD_THREADS / 64; ++w) { mn = fminf(mn, s_mn[w]); mx = fmaxf(mx, s_mx[w]); }
    part[(t * BD_PARTS + blockIdx.x) * 2] = mn;
    part[(t * BD_PARTS + blockIdx.x) * 2 + 1] = mx;
  }
}

__global__ __launch_bounds__(BD_THREADS) void k_bev_display_apply(const float *__restrict__ sums, int B, int T, int64_t HW,
                                                                  float inv_c, const float *__restrict__ part,
                                                                  float *__restrict__ mask, float *__restrict__ disp) {
#pragma clang fp contract(off)
  const int t = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  float mn = part[(t * BD_PARTS + lane) * 2], mx = part[(t * BD_PARTS + lane) * 2 + 1];      // BD_PARTS = 64 = one wave
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  const float den = (mx - mn) + 1e-12f;
  const int64_t n = (int64_t)B * HW;
  const int64_t i = (int64_t)blockIdx.x * BD_THREADS + tid;
  if (i >= n) return;
  const int64_t b = i / HW, p = i - b * HW;
  const float d = sums[(b * T + t) * HW + p] * inv_c;
  disp[(int64_t)t * n + i] = (d - mn) / den;
  if (t == 0) {
    float s = 0.f;
    for (int u = 0; u < T; ++u) s = s + sums[(b * T + u) * HW + p];
    mask[i] = s != 0.f ? 1.f : 0.f;
  }
}

extern "C" size_t rslo_bev_display_ws_bytes(int T) { return (size_t)(T > 0 ? T : 1) * BD_PARTS * 2 * sizeof(float); }

extern "C" int rslo_bev_display(const float *sums, int B, int T, int Cg, int64_t HW, float *mask, float *disp, void *ws,
                                size_t ws_bytes, void *stream) {
  RSLO_CHECK_ARG(sums && mask && disp && ws && B >= 1 && T >= 1 && T < 65536 && Cg >= 1 && HW >= 1, "rslo_bev_display: bad arguments");
  RSLO_CHECK_ARG(ws_bytes >= rslo_bev_display_ws_bytes(T), "rslo_bev_display: workspace too small");
  const float inv_c = 1.0f / (float)Cg;
  hipLaunchKernelGGL(k_bev_display_minmax, dim3(BD_PARTS, (unsigned)T), dim3(BD_THREADS), 0, (hipStream_t)stream, sums, B, T,
                     HW, inv_c, (float *)ws);
  RSLO_CHECK_LAUNCH("k_bev_display_minmax");
  hipLaunchKernelGGL(k_bev_display_apply, dim3((unsigned)rslo_cdiv((int64_t)B * HW, BD_THREADS), (unsigned)T),
                     dim3(BD_THREADS), 0, (hipStream_t)stream, sums, B, T, HW, inv_c, (const float *)ws, mask, disp);
  RSLO_CHECK_LAUNCH("k_bev_display_apply");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// neighbour table -> spconv-style pair lists (for the weight-gradient kernels)
// ---------------------------------------------------------------------------------------
// Pairs of offset k come in ascending row order, offsets in ascending k (the order spconv's indice_pairs have and the
// fixed summation order of the weight-gradient kernels).  Three launches over row blocks of 256:
//   k_pair_count : per (offset, row block) the number of present neighbours (wave ballots; the block's table rows are
//                  staged through LDS so the global reads are coalesced: a thread's K entries are K * 4 bytes apart)
//   k_pair_scan  : exclusive scan of the [K][blocks] counts in offset-major order (one workgroup) + koff
//   k_pair_emit  : position = block offset + waves before + ballot rank; writes (input row, output row)
// The table is read twice and the lists written once (~70 MB for 250 k rows) where the flag / scan / emit form moved
// ~250 MB through five launches.
#define PR_ROWS 256

// Several tables per launch (grid.y = table): rslo_plan_encoder builds the pair lists of all its rulebooks in three
// launches at the end of the plan; the stand-alone entry point is a batch of one.
#define PR_MAXTAB 16
struct PairTable {
  const int32_t *nbr;
  const int32_t *d_n;      // device row count (NULL: n_rows)
  int64_t n_rows;
  int32_t *counts, *pin, *pout, *koff;
  int K, nblk;
};
struct PairBatch {
  PairTable t[PR_MAXTAB];
};

template <bool EMIT>
__global__ __launch_bounds__(PR_ROWS) void k_pair_pass(PairBatch pb) {
  __shared__ int32_t tab[PR_ROWS * 27];
  __shared__ int32_t wcnt[PR_ROWS / 64][27];
  const PairTable &T = pb.t[blockIdx.y];
  const int32_t *__restrict__ nbr = T.nbr;
  const int K = T.K;
  int nblk = T.nblk;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t N = rslo_rows(T.n_rows, T.d_n);
  if (T.d_n) nblk = (int)((N + PR_ROWS - 1) / PR_ROWS);      // capacity-sized launch: counts are laid out for the ACTUAL blocks
  if ((int)blockIdx.x >= nblk) return;
  const int64_t row0 = (int64_t)blockIdx.x * PR_ROWS;
  const int64_t lim = (N - row0 < PR_ROWS ? N - row0 : PR_ROWS) * K;      // table entries of this block
  for (int e = tid; e < PR_ROWS * K; e += PR_ROWS) tab[e] = e < lim ? nbr[row0 * K + e] : -1;
  __syncthreads();
  const int32_t *mine = tab + tid * K;       // stride K = 27 words: conflict-free for odd K
  for (int k = 0; k < K; ++k) {
    const unsigned long long m = __ballot(mine[k] >= 0);
    if (lane == 0) wcnt[wid][k] = __popcll(m);
  }
  __syncthreads();
  if (!EMIT) {
    if (tid < K) {
      int32_t c = 0;
#pragma unroll
      for (int w = 0; w < PR_ROWS / 64; ++w) c += wcnt[w][tid];
      T.counts[(int64_t)tid * nblk + blockIdx.x] = c;
    }
    return;
  }
  const int64_t row = row0 + tid;
  for (int k = 0; k < K; ++k) {
    const int32_t r = mine[k];
    const unsigned long long m = __ballot(r >= 0);
    if (r >= 0) {
      int32_t pos = T.counts[(int64_t)k * nblk + blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
      for (int w = 0; w < wid; ++w) pos += wcnt[w][k];
      T.pin[pos] = r;
      T.pout[pos] = (int32_t)row;
    }
  }
}

// exclusive scan of n = K * blocks values in place, ONE workgroup per table; koff[k] = offset of (k, block 0)
__global__ __launch_bounds__(1024) void k_pair_scan(PairBatch pb) {
  __shared__ int32_t wsum[16];
  const PairTable &T = pb.t[blockIdx.x];
  int32_t *__restrict__ v = T.counts;
  int32_t *__restrict__ koff = T.koff;
  const int K = T.K;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t N = rslo_rows(T.n_rows, T.d_n);
  const int nblk = T.d_n ? (int)((N + PR_ROWS - 1) / PR_ROWS) : T.nblk;
  const int n = K * nblk;
  if (nblk == 0) {
    if (tid <= K) koff[tid] = 0;
    return;
  }
  const int per = (n + 1023) / 1024;
  const int i0 = tid * per, i1 = (i0 + per < n) ? i0 + per : n;
  int32_t s = 0;
  for (int i = i0; i < i1; ++i) s += v[i];
  int32_t inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int32_t run = inc - s;
  for (int w = 0; w < wid; ++w) run += wsum[w];
  for (int i = i0; i < i1; ++i) {
    const int32_t x = v[i];
    v[i] = run;
    if (i % nblk == 0) koff[i / nblk] = run;
    run += x;
  }
  if (tid == 1023) koff[K] = run;        // the last thread's running sum ends at the total (empty chunks carry it)
}

static int pairs_run_batch(const PairBatch &pb, int n_tab, hipStream_t st) {
  if (n_tab == 0) return RSLO_OK;
  int max_blk = 1;
  for (int i = 0; i < n_tab; ++i)
    if (pb.t[i].nblk > max_blk) max_blk = pb.t[i].nblk;
  hipLaunchKernelGGL(k_pair_pass<false>, dim3((unsigned)max_blk, (unsigned)n_tab), dim3(PR_ROWS), 0, st, pb);
  hipLaunchKernelGGL(k_pair_scan, dim3((unsigned)n_tab), dim3(1024), 0, st, pb);
  hipLaunchKernelGGL(k_pair_pass<true>, dim3((unsigned)max_blk, (unsigned)n_tab), dim3(PR_ROWS), 0, st, pb);
  RSLO_CHECK_LAUNCH("rulebook_pairs");
  return RSLO_OK;
}

static int pairs_run(const int32_t *nbr, int64_t n_rows, const int32_t *d_n, int K, int nblk, int32_t *counts,
                     int32_t *pairs_in, int32_t *pairs_out, int32_t *koff, hipStream_t st) {
  PairBatch pb;
  memset(&pb, 0, sizeof(pb));
  pb.t[0] = PairTable{nbr, d_n, n_rows, counts, pairs_in, pairs_out, koff, K, nblk};
  return pairs_run_batch(pb, 1, st);
}

extern "C" size_t rslo_rulebook_pairs_ws_bytes(int64_t n_rows, int K) {
  return align256((size_t)K * (size_t)rslo_cdiv(n_rows > 0 ? n_rows : 1, PR_ROWS) * sizeof(int32_t));
}

extern "C" int rslo_rulebook_pairs(const int32_t *nbr, int64_t n_rows, int K, void *ws, size_t ws_bytes,
                                   int32_t *pairs_in, int32_t *pairs_out, int32_t *koff, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(K >= 1 && K <= 27, "rulebook_pairs: K must be in 1..27");
  if (n_rows == 0) {
    RSLO_HIP(hipMemsetAsync(koff, 0, (size_t)(K + 1) * sizeof(int32_t), st));
    return RSLO_OK;
  }
  const int64_t n = n_rows * K;
  RSLO_CHECK_ARG(n < (int64_t)2000000000, "rulebook_pairs: table too large");
  const int nblk = (int)rslo_cdiv(n_rows, PR_ROWS);
  if (ws_bytes < (size_t)K * nblk * sizeof(int32_t)) {
    rslo_set_error("rulebook_pairs: workspace too small");
    return RSLO_EWS;
  }
  int32_t *counts = (int32_t *)ws;
  return pairs_run(nbr, n_rows, nullptr, K, nblk, counts, pairs_in, pairs_out, koff, st);
}

// ---------------------------------------------------------------------------------------
// Row order for the tile-granular sparse-conv kernels.  A wave issues MFMAs for every kernel offset that ANY of its
// 16/32 rows has a neighbour at; in scan (level 0) or raster (strided levels) order a tile's union mask holds ~1.5x
// the offsets a row really has.  Sorting the rows of each window of 2048 by their 27-bit neighbour mask puts rows
// with equal / similar masks into the same tile: useful-MFMA fraction 0.48 -> 0.70 (level 0), 0.61 -> 0.81,
// 0.71 -> 0.86 (level 2) on KITTI-shaped scans, while rows stay inside their window (gather locality in L2 is kept).
// One workgroup per window: key = mask << 11 | local row, bitonic sort in LDS (unique keys -> deterministic).
// flip = 1 reverses the bit order (the mask a SubM data-gradient call walks with flip_k).
// ---------------------------------------------------------------------------------------
#ifndef RO_BITS
#define RO_BITS 11
#endif
#define RO_WINDOW (1 << RO_BITS)
#define RO_THREADS 1024

__global__ __launch_bounds__(RO_THREADS) void k_row_order(const int32_t *__restrict__ nbr, int64_t n_host,
                                                          const int32_t *__restrict__ d_n, int K, int flip,
                                                          int32_t *__restrict__ order) {
  __shared__ unsigned long long key[RO_WINDOW];
  const int64_t n = rslo_rows(n_host, d_n);
  const int64_t base = (int64_t)blockIdx.x * RO_WINDOW;
  if (base >= n) return;       // capacity-sized launch (rslo_plan_encoder): windows past the actual rows
  for (int r = threadIdx.x; r < RO_WINDOW; r += RO_THREADS) {
    const int64_t row = base + r;
    unsigned long long kv = ~0ull;
    if (row < n) {
      unsigned m = 0;
      for (int k = 0; k < K; ++k)
        if (nbr[row * K + k] >= 0) m |= 1u << (flip ? (K - 1 - k) : k);
      kv = ((unsigned long long)m << RO_BITS) | (unsigned)r;
    }
    key[r] = kv;
  }
  __syncthreads();
  for (int size = 2; size <= RO_WINDOW; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < RO_WINDOW / 2; t += RO_THREADS) {
        const int lo = 2 * t - (t & (stride - 1));       // index with bit `stride` cleared
        const int hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = key[lo], b = key[hi];
        if ((a > b) == up) {
          key[lo] = b;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int r = threadIdx.x; r < RO_WINDOW; r += RO_THREADS) {
    const int64_t row = base + r;
    if (row < n) order[row] = (int32_t)(base + (int64_t)(key[r] & (unsigned long long)(RO_WINDOW - 1)));
  }
}

extern "C" int rslo_rulebook_row_order(const int32_t *nbr, int64_t n_rows, int K, int flip_k, int32_t *order,
                                       void *stream) {
  RSLO_CHECK_ARG(K >= 1 && K <= 27, "rslo_rulebook_row_order: K must be in 1..27");
  if (n_rows == 0) return RSLO_OK;
  RSLO_CHECK_ARG(nbr && order, "rslo_rulebook_row_order: null pointer");
  hipLaunchKernelGGL(k_row_order, dim3((unsigned)rslo_cdiv(n_rows, RO_WINDOW)), dim3(RO_THREADS), 0,
                     (hipStream_t)stream, nbr, n_rows, (const int32_t *)nullptr, K, flip_k, order);
  RSLO_CHECK_LAUNCH("k_row_order");
  return RSLO_OK;
}

// ---------------------------------------------------------------------------------------
// rslo_plan_encoder: voxelization of all clouds of a step + the whole rulebook chain of a chain-structured sparse
// encoder in ONE call without a single host read.  Every data-dependent size stays on the device: the kernels above
// take their row counts from the counts block (d_n), launches are sized for the capacity of a level, and the counts
// reach the host through one asynchronous copy into pinned memory that the caller reads when it picks the plan up
// (a step later, in rslo_amd.workload.ExamplePrefetcher).  The Python-issued form of the same work was ~230 launches
// and 7 blocking reads per step on a helper thread.
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// The voxelizer for ALL clouds of a step in one set of launches (rslo_plan_encoder): grid (blocks of the largest cloud,
// cloud).  Same algorithm as vox_run -- per-cloud hash with atomicMin(first point) + linked list, voxel id = rank of the
// first point, in-voxel rank = list members with a smaller index -- with the per-point arrays of all clouds back to back
// (ONE scan over all "is first point" flags: a cloud's voxel ids are the global prefix minus the prefix at its first
// point) and the per-cloud tables back to back (one fill each for keys | head and first | cutoff).  8 clouds: 11
// launches instead of 80, and launches of ~4000 workgroups instead of ~480.
// ---------------------------------------------------------------------------------------
struct VoxBatch {
  const float *pts[RSLO_PLAN_MAX_CLOUDS];
  int32_t start[RSLO_PLAN_MAX_CLOUDS + 1];     // first global point index of cloud c (start[n] = total)
  int32_t tab[RSLO_PLAN_MAX_CLOUDS];           // first table slot of cloud c
  int8_t shift[RSLO_PLAN_MAX_CLOUDS];          // 32 - log2(table capacity of cloud c)
  int32_t n, clouds_per_frame;
};

struct VoxBatchWs {
  uint32_t *keys;
  int32_t *head, *first, *cutoff, *vid, *next, *slot, *pos, *pre;      // pre[c] = global prefix at the first point of cloud c
  uint32_t *flags;
};

__global__ void kb_vox_insert(VoxBatch B, int F, VoxGeom G, VoxBatchWs w) {
  const int c = blockIdx.y;
  const int P = B.start[c + 1] - B.start[c];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int gi = B.start[c] + i;
  int cc[3];
  if (!vox_coord(B.pts[c] + (int64_t)i * F, G, cc)) {
    w.slot[gi] = -1;
    return;
  }
  const uint32_t key = ((uint32_t)cc[2] * G.g[1] + cc[1]) * G.g[0] + cc[0];
  const uint32_t mask = (1u << (32 - B.shift[c])) - 1u;
  uint32_t *keys = w.keys + B.tab[c];
  uint32_t s = rslo_hslot(key, B.shift[c]);
  // plain reads first: 3 of 4 points land in a voxel that already has its slot, and in scan order most of them are not
  // its first point -- the atomics still decide, the reads only skip those that cannot change anything
  while (true) {
    uint32_t prev = __builtin_nontemporal_load(&keys[s]);
    if (prev == RSLO_EMPTY_KEY) prev = atomicCAS(&keys[s], RSLO_EMPTY_KEY, key);
    if (prev == RSLO_EMPTY_KEY || prev == key) break;
    s = (s + 1) & mask;
  }
  if (__builtin_nontemporal_load(&w.first[B.tab[c] + s]) > i) atomicMin(&w.first[B.tab[c] + s], i);
  w.next[gi] = atomicExch(&w.head[B.tab[c] + s], i);
  w.slot[gi] = (int32_t)s;
}

__global__ void kb_vox_flags(VoxBatch B, VoxBatchWs w) {
  const int c = blockIdx.y;
  const int P = B.start[c + 1] - B.start[c];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int gi = B.start[c] + i;
  const int32_t s = w.slot[gi];
  w.flags[gi] = (s >= 0 && w.first[B.tab[c] + s] == i) ? 1u : 0u;
}

// one block: per cloud the number of voxels (capped), the row base of the next cloud, the prefix at its first point
__global__ void kb_vox_bases(VoxBatch B, VoxBatchWs w, int max_voxels, int32_t *__restrict__ cnt) {
  if (threadIdx.x != 0) return;
  const int total_pts = B.start[B.n];
  const int32_t all = total_pts > 0 ? w.pos[total_pts - 1] + (int32_t)w.flags[total_pts - 1] : 0;
  int32_t base = 0;
  cnt[RSLO_PLAN_CNT_BASE] = 0;
  for (int c = 0; c < B.n; ++c) {
    const int32_t s0 = B.start[c] < total_pts ? w.pos[B.start[c]] : all;
    const int32_t s1 = B.start[c + 1] < total_pts ? w.pos[B.start[c + 1]] : all;
    const int32_t tot = s1 - s0;
    const int32_t nv = tot < max_voxels ? tot : max_voxels;
    w.pre[c] = s0;
    cnt[RSLO_PLAN_CNT_NVOX + c] = nv;
    base += nv;
    cnt[RSLO_PLAN_CNT_BASE + c + 1] = base;
  }
}

__global__ void kb_vox_assign(VoxBatch B, VoxGeom G, VoxBatchWs w, int max_voxels, const int32_t *__restrict__ cnt,
                              int32_t *__restrict__ coords_frame, int32_t *__restrict__ coords_all) {
  const int c = blockIdx.y;
  const int P = B.start[c + 1] - B.start[c];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int gi = B.start[c] + i;
  if (!w.flags[gi]) return;
  const int32_t s = w.slot[gi];
  const int32_t v = w.pos[gi] - w.pre[c];
  w.vid[B.tab[c] + s] = v;
  if (v < max_voxels) {
    uint32_t key = w.keys[B.tab[c] + s];
    const int x = key % G.g[0];
    key /= G.g[0];
    const int y = key % G.g[1];
    const int z = key / G.g[1];
    const int32_t row = cnt[RSLO_PLAN_CNT_BASE + c] + v;
    reinterpret_cast<int4 *>(coords_frame)[row] = make_int4(c % B.clouds_per_frame, z, y, x);
    reinterpret_cast<int4 *>(coords_all)[row] = make_int4(c, z, y, x);
  } else if (v == max_voxels) {
    w.cutoff[c] = i;      // the reference loop `break`s here
  }
}

__global__ void kb_vox_fill(VoxBatch B, int F, int T, VoxBatchWs w, const int32_t *__restrict__ cnt,
                            float *__restrict__ voxels, int32_t *__restrict__ num_points) {
  const int c = blockIdx.y;
  const int P = B.start[c + 1] - B.start[c];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int gi = B.start[c] + i;
  const int32_t s = w.slot[gi];
  if (s < 0 || i >= w.cutoff[c]) return;
  int rank = 0;
  for (int32_t j = w.head[B.tab[c] + s]; j >= 0; j = w.next[B.start[c] + j]) rank += (j < i);
  if (rank >= T) return;
  const int32_t v = w.vid[B.tab[c] + s] + cnt[RSLO_PLAN_CNT_BASE + c];
  float *dst = voxels + ((int64_t)v * T + rank) * F;
  const float *src = B.pts[c] + (int64_t)i * F;
  for (int f = 0; f < F; ++f) dst[f] = src[f];
  atomicAdd(&num_points[v], 1);
}

// workspace of the batched voxelizer: tables (per-cloud capacities back to back) + per-point arrays
static size_t voxb_ws_layout(int n, const int64_t *h_n_points, void *base, VoxBatchWs *w, VoxBatch *B, void **scan_ws,
                             size_t *scan_bytes, size_t *fill_ff, size_t *fill_7f) {
  int64_t caps = 0, pts = 0;
  for (int c = 0; c < n; ++c) {
    const int64_t cap = rslo_hash_capacity(h_n_points[c] > 0 ? h_n_points[c] : 1);
    if (B) {
      B->tab[c] = (int32_t)caps;
      B->start[c] = (int32_t)pts;
      B->shift[c] = (int8_t)(32 - rslo_log2_i64(cap));
    }
    caps += cap;
    pts += h_n_points[c];
  }
  if (B) B->start[n] = (int32_t)pts;
  const int64_t np = pts > 0 ? pts : 1;
  size_t off = 0;
  char *b = (char *)base;
  auto take = [&](size_t bytes) {
    void *p = b ? (void *)(b + off) : nullptr;
    off += align256(bytes);
    return p;
  };
  // keys | head: one 0xFF fill; first | cutoff: one 0x7F fill (caps * 4 is a multiple of 256: no padding in between)
  void *keys = take((size_t)caps * 4), *head = take((size_t)caps * 4), *first = take((size_t)caps * 4);
  void *cutoff = take(256), *vid = take((size_t)caps * 4), *pre = take(256);
  void *next = take((size_t)np * 4), *slot = take((size_t)np * 4), *pos = take((size_t)np * 4), *flags = take((size_t)np * 4);
  const size_t sb = rslo_scan_ws_bytes(np);
  void *sws = take(sb);
  if (w) {
    w->keys = (uint32_t *)keys; w->head = (int32_t *)head; w->first = (int32_t *)first; w->cutoff = (int32_t *)cutoff;
    w->vid = (int32_t *)vid; w->pre = (int32_t *)pre; w->next = (int32_t *)next; w->slot = (int32_t *)slot;
    w->pos = (int32_t *)pos; w->flags = (uint32_t *)flags;
  }
  if (scan_ws) *scan_ws = sws;
  if (scan_bytes) *scan_bytes = sb;
  if (fill_ff) *fill_ff = (size_t)caps * 8;
  if (fill_7f) *fill_7f = (size_t)caps * 4 + 256;
  return off;
}

static size_t plan_take(size_t &off, size_t bytes) {
  const size_t o = off;
  off += align256(bytes > 0 ? bytes : 1);
  return o;
}

static void plan_out_dims(const int32_t *in, const int32_t *ks, const int32_t *st, const int32_t *pd, int32_t *out) {
  for (int j = 0; j < 3; ++j) out[j] = (in[j] + 2 * pd[j] - ks[j]) / st[j] + 1;
}

extern "C" int rslo_plan_encoder_layout(const RsloEncoderSpec *spec, int n_clouds, const int64_t *h_n_points,
                                        RsloPlanLayout *lay) {
  RSLO_CHECK_ARG(spec && lay && h_n_points, "plan_encoder_layout: null argument");
  RSLO_CHECK_ARG(spec->n_levels >= 1 && spec->n_levels <= RSLO_PLAN_MAX_LEVELS, "plan_encoder_layout: 1..8 levels");
  RSLO_CHECK_ARG(n_clouds >= 1 && n_clouds <= RSLO_PLAN_MAX_CLOUDS, "plan_encoder_layout: 1..64 clouds");
  RSLO_CHECK_ARG(spec->max_points >= 1 && spec->max_voxels >= 1 && spec->n_features >= 3 && spec->n_features <= 16,
                 "plan_encoder_layout: bad voxelizer parameters");
  memset(lay, 0, sizeof(*lay));
  const int L = spec->n_levels, T = spec->max_points, F = spec->n_features;
  int64_t sumP = 0, maxP = 1;
  for (int c = 0; c < n_clouds; ++c) {
    RSLO_CHECK_ARG(h_n_points[c] >= 0 && h_n_points[c] < (int64_t)2000000000, "plan_encoder_layout: bad point count");
    sumP += h_n_points[c];
    if (h_n_points[c] > maxP) maxP = h_n_points[c];
  }
  for (int j = 0; j < 3; ++j) lay->dims[0][j] = spec->dims0[j];
  for (int l = 0; l + 1 < L; ++l)
    plan_out_dims(lay->dims[l], spec->conv_ks[l], spec->conv_stride[l], spec->conv_pad[l], lay->dims[l + 1]);
  for (int l = 0; l < L; ++l) {
    RSLO_CHECK_ARG(lay->dims[l][0] >= 1 && lay->dims[l][1] >= 1 && lay->dims[l][2] >= 1, "plan_encoder_layout: empty level");
    if (int rc = check_volume(n_clouds, lay->dims[l])) return rc;
    int64_t cap = spec->cap_rows[l];
    if (l == 0) {      // every cloud adds at most min(max_voxels, P) rows: a smaller capacity could be overrun
      int64_t need = (int64_t)n_clouds * spec->max_voxels;
      if (sumP < need) need = sumP;
      RSLO_CHECK_ARG(cap <= 0 || cap >= need, "plan_encoder_layout: level-0 capacity below n_clouds * max_voxels");
      if (cap <= 0) cap = need;
    } else if (cap <= 0) {
      {
        cap = lay->cap_rows[l - 1];      // LiDAR surfaces thin out under stride 2; an overflow is flagged, never silent
      }
    }
    const int64_t vol = (int64_t)n_clouds * lay->dims[l][0] * lay->dims[l][1] * lay->dims[l][2];
    if (cap > vol) cap = vol;
    if (cap < 1) cap = 1;
    RSLO_CHECK_ARG(cap * 27 < ((int64_t)1 << 31), "plan_encoder_layout: level capacity too large");
    lay->cap_rows[l] = cap;
    lay->hash_cap[l] = rslo_hash_capacity(cap);
  }
  size_t off = 0;
  lay->counts_off = plan_take(off, RSLO_PLAN_CNT_WORDS * sizeof(int32_t));
  const int64_t c0 = lay->cap_rows[0];
  // voxels | num_points back to back: one zero fill
  lay->voxels_off = plan_take(off, (size_t)c0 * T * F * sizeof(float));
  lay->num_points_off = plan_take(off, (size_t)c0 * sizeof(int32_t));
  lay->coords_frame_off = plan_take(off, (size_t)c0 * 16);
  // the hash keys of all levels back to back (ONE 0xFF fill), then the output bitmaps of all strided levels (ONE zero fill)
  for (int l = 0; l < L; ++l) lay->keys_off[l] = plan_take(off, (size_t)lay->hash_cap[l] * 4);
  lay->keys_end_off = off;
  for (int l = 0; l + 1 < L; ++l)
    lay->bitmap_level_off[l] = plan_take(off, (size_t)rslo_conv_bitmap_words(n_clouds, lay->dims[l + 1]) * 4);
  lay->bitmap_end_off = off;
  for (int l = 0; l < L; ++l) {
    const int64_t cap = lay->cap_rows[l];
    lay->coords_off[l] = plan_take(off, (size_t)cap * 16);
    lay->vals_off[l] = plan_take(off, (size_t)lay->hash_cap[l] * 4);
    const int Ks = spec->subm_ks[l][0] * spec->subm_ks[l][1] * spec->subm_ks[l][2];
    if (Ks > 0) {
      RSLO_CHECK_ARG(Ks <= 27, "plan_encoder_layout: SubM kernel volume > 27");
      lay->subm_nbr_off[l] = plan_take(off, (size_t)cap * Ks * 4);
      if (spec->want_pairs) {
        lay->subm_pin_off[l] = plan_take(off, (size_t)cap * Ks * 4);
        lay->subm_pout_off[l] = plan_take(off, (size_t)cap * Ks * 4);
        lay->subm_koff_off[l] = plan_take(off, (size_t)(Ks + 1) * 4);
      }
    }
    if (l + 1 < L) {
      const int K = spec->conv_ks[l][0] * spec->conv_ks[l][1] * spec->conv_ks[l][2];
      RSLO_CHECK_ARG(K >= 1 && K <= 27, "plan_encoder_layout: conv kernel volume must be in 1..27");
      const int64_t capo = lay->cap_rows[l + 1];
      lay->conv_nbr_off[l] = plan_take(off, (size_t)capo * K * 4);
      lay->conv_nbrT_off[l] = plan_take(off, (size_t)cap * K * 4);
      if ((spec->want_orders >> l) & 1) lay->conv_order_off[l] = plan_take(off, (size_t)cap * 4);
      if (spec->want_pairs) {
        lay->conv_pin_off[l] = plan_take(off, (size_t)capo * K * 4);
        lay->conv_pout_off[l] = plan_take(off, (size_t)capo * K * 4);
        lay->conv_koff_off[l] = plan_take(off, (size_t)(K + 1) * 4);
      }
    }
  }
  // scratch shared by the stages (stream order): voxelizer workspace, bitmap + word prefix + scan sums, pair counts
  int64_t words = 1, pair_ws = 256;
  for (int l = 0; l + 1 < L; ++l) {
    const int64_t w = rslo_conv_bitmap_words(n_clouds, lay->dims[l + 1]);
    if (w > words) words = w;
  }
  pair_ws = 0;
  for (int l = 0; l < L; ++l) {      // one counts region per rulebook: the pair lists of all of them are built together
    if (spec->subm_ks[l][0] > 0) pair_ws += (int64_t)rslo_rulebook_pairs_ws_bytes(lay->cap_rows[l], 27);
    if (l + 1 < L) pair_ws += (int64_t)rslo_rulebook_pairs_ws_bytes(lay->cap_rows[l + 1], 27);
  }
  if (pair_ws < 256) pair_ws = 256;
  lay->scratch_words = words;
  lay->vox_ws_off = plan_take(off, voxb_ws_layout(n_clouds, h_n_points, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
  lay->bitmap_off = lay->bitmap_level_off[0];      // (kept in the struct; the per-level bitmaps above are what is used)
  lay->prefix_off = plan_take(off, (size_t)words * 4);
  lay->scan_ws_off = plan_take(off, rslo_scan_ws_bytes(words));
  lay->pair_ws_off = plan_take(off, (size_t)pair_ws);
  lay->total_bytes = off;
  return RSLO_OK;
}

__global__ void kp_level0(int32_t *__restrict__ cnt, int n_clouds) {
  const int j = threadIdx.x;
  if (j <= n_clouds) cnt[RSLO_PLAN_CNT_BOFF + j] = cnt[RSLO_PLAN_CNT_BASE + j];
  if (j == 0) cnt[RSLO_PLAN_CNT_ROWS] = cnt[RSLO_PLAN_CNT_BASE + n_clouds];
}

__global__ void kp_level_finish(int32_t *__restrict__ cnt, int level, int cap) {
  const int32_t raw = cnt[RSLO_PLAN_CNT_RAW + level];
  cnt[RSLO_PLAN_CNT_ROWS + level] = raw < cap ? raw : cap;
  if (raw > cap) atomicOr((unsigned *)&cnt[RSLO_PLAN_CNT_OVERFLOW], 1u << level);
}

// rows are grouped by ascending batch index on every level: first row with batch >= j, j = 0..nb
__global__ void kp_batch_offsets(const int32_t *__restrict__ coords, const int32_t *__restrict__ d_n, int nb,
                                 int32_t *__restrict__ out) {
  const int j = threadIdx.x;
  if (j > nb) return;
  int lo = 0, hi = *d_n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (coords[(int64_t)mid * 4] < j) lo = mid + 1;
    else hi = mid;
  }
  out[j] = lo;
}

extern "C" int rslo_plan_encoder(const RsloEncoderSpec *spec, const RsloPlanLayout *lay, int n_clouds,
                                 int clouds_per_frame, const float *const *h_points, const int64_t *h_n_points,
                                 void *arena, size_t arena_bytes, int32_t *h_counts, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  RSLO_CHECK_ARG(spec && lay && h_points && h_n_points && arena, "plan_encoder: null argument");
  RSLO_CHECK_ARG(arena_bytes >= lay->total_bytes, "plan_encoder: arena too small");
  RSLO_CHECK_ARG(clouds_per_frame >= 1 && n_clouds % clouds_per_frame == 0, "plan_encoder: clouds must fill whole frames");
  RSLO_CHECK_ARG(((uintptr_t)arena & 255) == 0, "plan_encoder: arena must be 256-byte aligned");
  char *A = (char *)arena;
  const int L = spec->n_levels, T = spec->max_points, F = spec->n_features;
  int32_t *cnt = (int32_t *)(A + lay->counts_off);
  RSLO_HIP(hipMemsetAsync(cnt, 0, RSLO_PLAN_CNT_WORDS * sizeof(int32_t), st));
  const int64_t c0 = lay->cap_rows[0];
  float *voxels = (float *)(A + lay->voxels_off);
  int32_t *num = (int32_t *)(A + lay->num_points_off);
  RSLO_HIP(hipMemsetAsync(voxels, 0, (size_t)(lay->num_points_off - lay->voxels_off) + (size_t)c0 * 4, st));
  int32_t *coords_frame = (int32_t *)(A + lay->coords_frame_off), *coords0 = (int32_t *)(A + lay->coords_off[0]);
  VoxGeom G;
  for (int j = 0; j < 3; ++j) {
    G.lo[j] = spec->range6[j];
    G.vs[j] = spec->vsize3[j];
    G.g[j] = spec->grid_xyz[j];
  }
  {
    VoxBatch VB;
    VoxBatchWs W;
    void *sws;
    size_t sbytes, f_ff, f_7f;
    memset(&VB, 0, sizeof(VB));
    voxb_ws_layout(n_clouds, h_n_points, A + lay->vox_ws_off, &W, &VB, &sws, &sbytes, &f_ff, &f_7f);
    VB.n = n_clouds;
    VB.clouds_per_frame = clouds_per_frame;
    int64_t maxP = 0;
    for (int c = 0; c < n_clouds; ++c) {
      VB.pts[c] = h_points[c];
      RSLO_CHECK_ARG(h_n_points[c] == 0 || h_points[c], "plan_encoder: null cloud");
      if (h_n_points[c] > maxP) maxP = h_n_points[c];
    }
    const int64_t total = VB.start[n_clouds];
    RSLO_CHECK_ARG(total < (int64_t)2000000000, "plan_encoder: too many points");
    if (total > 0) {
      RSLO_HIP(hipMemsetAsync(W.keys, 0xFF, f_ff, st));       // keys | head
      RSLO_HIP(hipMemsetAsync(W.first, 0x7F, f_7f, st));      // first | cutoff
      const dim3 grid((unsigned)rslo_cdiv(maxP, 256), (unsigned)n_clouds);
      hipLaunchKernelGGL(kb_vox_insert, grid, dim3(256), 0, st, VB, F, G, W);
      hipLaunchKernelGGL(kb_vox_flags, grid, dim3(256), 0, st, VB, W);
      RSLO_CHECK_LAUNCH("plan vox_insert");
      if (int rc = scan_exclusive<false>(W.flags, W.pos, total, sws, sbytes, nullptr, st)) return rc;
      // base + nvox <= level-0 capacity by construction: every cloud adds at most min(max_voxels, P) rows (layout)
      hipLaunchKernelGGL(kb_vox_bases, dim3(1), dim3(64), 0, st, VB, W, spec->max_voxels, cnt);
      hipLaunchKernelGGL(kb_vox_assign, grid, dim3(256), 0, st, VB, G, W, spec->max_voxels, (const int32_t *)cnt,
                         coords_frame, coords0);
      hipLaunchKernelGGL(kb_vox_fill, grid, dim3(256), 0, st, VB, F, T, W, (const int32_t *)cnt, voxels, num);
      RSLO_CHECK_LAUNCH("plan vox_fill");
    }
  }
  hipLaunchKernelGGL(kp_level0, dim3(1), dim3(RSLO_PLAN_MAX_CLOUDS + 1), 0, st, cnt, n_clouds);
  RSLO_CHECK_LAUNCH("plan level 0");

  RSLO_HIP(hipMemsetAsync(A + lay->keys_off[0], 0xFF, (size_t)(lay->keys_end_off - lay->keys_off[0]), st));
  if (L > 1)
    RSLO_HIP(hipMemsetAsync(A + lay->bitmap_level_off[0], 0, (size_t)(lay->bitmap_end_off - lay->bitmap_level_off[0]), st));
  auto hash_level = [&](int l) -> int {
    const int64_t hc = lay->hash_cap[l];
    hipLaunchKernelGGL(k_hash_insert, dim3(rb_grid(lay->cap_rows[l])), dim3(256), 0, st,
                       (const int32_t *)(A + lay->coords_off[l]), (int64_t)0, (const int32_t *)(cnt + RSLO_PLAN_CNT_ROWS + l),
                       Dims3{lay->dims[l][0], lay->dims[l][1], lay->dims[l][2]}, (uint32_t *)(A + lay->keys_off[l]),
                       (int32_t *)(A + lay->vals_off[l]), (uint32_t)(hc - 1), 32 - rslo_log2_i64(hc));
    RSLO_CHECK_LAUNCH("plan hash");
    return RSLO_OK;
  };
  PairBatch pbatch;
  memset(&pbatch, 0, sizeof(pbatch));
  int n_ptab = 0;
  size_t pws = lay->pair_ws_off;
  auto pairs_of = [&](const int32_t *nbr, int l_rows, int K, size_t pin, size_t pout, size_t koff) -> int {
    RSLO_CHECK_ARG(n_ptab < PR_MAXTAB, "plan_encoder: too many rulebooks");
    const int nblk = (int)rslo_cdiv(lay->cap_rows[l_rows], PR_ROWS);
    pbatch.t[n_ptab++] = PairTable{nbr, cnt + RSLO_PLAN_CNT_ROWS + l_rows, 0, (int32_t *)(A + pws), (int32_t *)(A + pin),
                                   (int32_t *)(A + pout), (int32_t *)(A + koff), K, nblk};
    pws += rslo_rulebook_pairs_ws_bytes(lay->cap_rows[l_rows], 27);
    return RSLO_OK;
  };

  if (int rc = hash_level(0)) return rc;
  for (int l = 0; l < L; ++l) {
    const int64_t cap = lay->cap_rows[l];
    const int32_t *d_n = cnt + RSLO_PLAN_CNT_ROWS + l;
    const int32_t *coords = (const int32_t *)(A + lay->coords_off[l]);
    const Dims3 dl{lay->dims[l][0], lay->dims[l][1], lay->dims[l][2]};
    const uint32_t *keys = (const uint32_t *)(A + lay->keys_off[l]);
    const int32_t *vals = (const int32_t *)(A + lay->vals_off[l]);
    const int64_t hc = lay->hash_cap[l];
    const int32_t *sk = spec->subm_ks[l];
    const int Ks = sk[0] * sk[1] * sk[2];
    if (Ks > 0) {
      int32_t *nbr = (int32_t *)(A + lay->subm_nbr_off[l]);
      if (rb_fast_k(sk, cap))
        hipLaunchKernelGGL(k_rulebook_subm<true>, dim3(rb_grid(cap * Ks)), dim3(256), 0, st, coords, (int64_t)0, d_n, dl,
                           Int3{sk[0], sk[1], sk[2]}, keys, vals, (uint32_t)(hc - 1), 32 - rslo_log2_i64(hc), nbr);
      else
        hipLaunchKernelGGL(k_rulebook_subm<false>, dim3(rb_grid(cap * Ks)), dim3(256), 0, st, coords, (int64_t)0, d_n, dl,
                           Int3{sk[0], sk[1], sk[2]}, keys, vals, (uint32_t)(hc - 1), 32 - rslo_log2_i64(hc), nbr);
      RSLO_CHECK_LAUNCH("plan subm");
      if (spec->want_pairs)
        if (int rc = pairs_of(nbr, l, Ks, lay->subm_pin_off[l], lay->subm_pout_off[l], lay->subm_koff_off[l])) return rc;
    }
    if (l + 1 >= L) break;
    const int32_t *ks = spec->conv_ks[l], *sd = spec->conv_stride[l], *pd = spec->conv_pad[l];
    const int K = ks[0] * ks[1] * ks[2];
    const int64_t capo = lay->cap_rows[l + 1];
    const Dims3 od{lay->dims[l + 1][0], lay->dims[l + 1][1], lay->dims[l + 1][2]};
    const int64_t words = rslo_conv_bitmap_words(n_clouds, lay->dims[l + 1]);
    uint32_t *bitmap = (uint32_t *)(A + lay->bitmap_level_off[l]);
    int32_t *prefix = (int32_t *)(A + lay->prefix_off);
    const bool fk = rb_fast_k(ks, cap > capo ? cap : capo), fs = rb_fast_s(sd);
    RSLO_CHECK_ARG(ks[0] <= 3 && ks[1] <= 3 && ks[2] <= 3, "plan_encoder: kernel extent > 3 unsupported");
    hipLaunchKernelGGL(k_conv_mark, dim3(rb_grid(cap)), dim3(256), 0, st, coords, (int64_t)0, d_n,
                       Int3{ks[0], ks[1], ks[2]}, Int3{sd[0], sd[1], sd[2]}, Int3{pd[0], pd[1], pd[2]}, od, bitmap);
    RSLO_CHECK_LAUNCH("plan conv_mark");
    if (int rc = scan_exclusive<true>(bitmap, prefix, words, A + lay->scan_ws_off, rslo_scan_ws_bytes(lay->scratch_words),
                                      cnt + RSLO_PLAN_CNT_RAW + l + 1, st))
      return rc;
    hipLaunchKernelGGL(kp_level_finish, dim3(1), dim3(1), 0, st, cnt, l + 1, (int)capo);
    int32_t *coords_o = (int32_t *)(A + lay->coords_off[l + 1]);
    hipLaunchKernelGGL(k_conv_emit, dim3((unsigned)rslo_cdiv(words, 256)), dim3(256), 0, st, bitmap, prefix, words, od,
                       coords_o, capo);
    const int32_t *d_m = cnt + RSLO_PLAN_CNT_ROWS + l + 1;
    hipLaunchKernelGGL(kp_batch_offsets, dim3(1), dim3(RSLO_PLAN_MAX_CLOUDS + 1), 0, st, coords_o, d_m, n_clouds,
                       cnt + RSLO_PLAN_CNT_BOFF + (l + 1) * (RSLO_PLAN_MAX_CLOUDS + 1));
    RSLO_CHECK_LAUNCH("plan conv_emit");
    if (int rc = hash_level(l + 1)) return rc;
    int32_t *nbr = (int32_t *)(A + lay->conv_nbr_off[l]), *nbrT = (int32_t *)(A + lay->conv_nbrT_off[l]);
    const uint32_t *okeys = (const uint32_t *)(A + lay->keys_off[l + 1]);
    const int32_t *ovals = (const int32_t *)(A + lay->vals_off[l + 1]);
    const int64_t ohc = lay->hash_cap[l + 1];
#define PLAN_CONV(F)                                                                                             \
    hipLaunchKernelGGL(k_rulebook_conv<F>, dim3(rb_grid(capo * K)), dim3(256), 0, st, coords_o, (int64_t)0, d_m, dl, \
                       Int3{ks[0], ks[1], ks[2]}, Int3{sd[0], sd[1], sd[2]}, Int3{pd[0], pd[1], pd[2]}, keys, vals, \
                       (uint32_t)(hc - 1), 32 - rslo_log2_i64(hc), nbr)
    if (fk) PLAN_CONV(true);
    else PLAN_CONV(false);
#undef PLAN_CONV
#define PLAN_CONVT(F, F2)                                                                                        \
    hipLaunchKernelGGL((k_rulebook_conv_T<F, F2>), dim3(rb_grid(cap * K)), dim3(256), 0, st, coords, (int64_t)0, d_n, \
                       od, Int3{ks[0], ks[1], ks[2]}, Int3{sd[0], sd[1], sd[2]}, Int3{pd[0], pd[1], pd[2]}, okeys, \
                       ovals, (uint32_t)(ohc - 1), 32 - rslo_log2_i64(ohc), nbrT)
    if (fk && fs) PLAN_CONVT(true, true);
    else PLAN_CONVT(false, false);
#undef PLAN_CONVT
    RSLO_CHECK_LAUNCH("plan rulebook_conv");
    if ((spec->want_orders >> l) & 1) {
      hipLaunchKernelGGL(k_row_order, dim3((unsigned)rslo_cdiv(cap, RO_WINDOW)), dim3(RO_THREADS), 0, st,
                         (const int32_t *)nbrT, (int64_t)0, d_n, K, 0, (int32_t *)(A + lay->conv_order_off[l]));
      RSLO_CHECK_LAUNCH("plan row_order");
    }
    if (spec->want_pairs)
      if (int rc = pairs_of(nbr, l + 1, K, lay->conv_pin_off[l], lay->conv_pout_off[l], lay->conv_koff_off[l])) return rc;
  }
  if (int rc = pairs_run_batch(pbatch, n_ptab, st)) return rc;      // the pair lists of all rulebooks: three launches
  if (h_counts)
    RSLO_HIP(hipMemcpyAsync(h_counts, cnt, RSLO_PLAN_CNT_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  return RSLO_OK;
}


// ---- static (capacity-sized) view of a plan: rslo_plan_encoder_pad_tails ----------------------------------------------
// An inference loop replays ONE hipGraph per arena: every kernel of the encoder is launched for the CAPACITY of its level
// and the tables' rows past the level's device-side count must then be harmless.  This pass (same stream, behind
// rslo_plan_encoder) turns them into padding rows: coordinates -1 (rslo_dense_scatter skips them), every neighbour entry -1
// (a padding output row gathers nothing: its value is the bias, which nobody reads), tile orders continued as the identity.
// (Tried: orders that mark the padding rows as absent, -1, for EVERY table so that a padding tile neither gathers nor stores --
// 0.62 vs 0.60 ms per replayed pass: the order indirection costs what the skipped stores save.)
struct PadEntry {
  int32_t *base;
  const int32_t *d_n;
  int32_t cap, width, identity;
};
struct PadTable {
  PadEntry e[6 * RSLO_PLAN_MAX_LEVELS + 1];
};

__global__ void kp_pad_tails(PadTable T) {
  const PadEntry e = T.e[blockIdx.y];
  const int64_t n0 = (int64_t)(*e.d_n < e.cap ? *e.d_n : e.cap) * e.width, n1 = (int64_t)e.cap * e.width;
  for (int64_t i = n0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += (int64_t)gridDim.x * blockDim.x)
    e.base[i] = e.identity ? (int32_t)i : -1;
}

extern "C" int rslo_plan_encoder_pad_tails(const RsloEncoderSpec *spec, const RsloPlanLayout *lay, void *arena, void *stream) {
  RSLO_CHECK_ARG(spec && lay && arena, "plan_encoder_pad_tails: null argument");
  char *A = (char *)arena;
  const int L = spec->n_levels;
  const int32_t *cnt = (const int32_t *)(A + lay->counts_off);
  PadTable T;
  int n = 0;
  int64_t widest = 1;
  auto add = [&](uint64_t off, int level_rows, int width, int identity) {
    T.e[n++] = PadEntry{(int32_t *)(A + off), cnt + RSLO_PLAN_CNT_ROWS + level_rows, (int32_t)lay->cap_rows[level_rows], width, identity};
    if ((int64_t)lay->cap_rows[level_rows] * width > widest) widest = (int64_t)lay->cap_rows[level_rows] * width;
  };
  for (int l = 0; l < L; ++l) {
    add(lay->coords_off[l], l, 4, 0);
    const int Ks = spec->subm_ks[l][0] * spec->subm_ks[l][1] * spec->subm_ks[l][2];
    if (Ks > 0) add(lay->subm_nbr_off[l], l, Ks, 0);
    if (l + 1 < L) {
      const int K = spec->conv_ks[l][0] * spec->conv_ks[l][1] * spec->conv_ks[l][2];
      add(lay->conv_nbr_off[l], l + 1, K, 0);
      add(lay->conv_nbrT_off[l], l, K, 0);
      if ((spec->want_orders >> l) & 1) add(lay->conv_order_off[l], l, 1, 1);
    }
  }
  add(lay->coords_frame_off, 0, 4, 0);
  int64_t bx = rslo_cdiv(widest, 256 * 8);
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(kp_pad_tails, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, T);
  RSLO_CHECK_LAUNCH("kp_pad_tails");
  return RSLO_OK;
}
