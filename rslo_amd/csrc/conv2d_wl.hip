// Dense 3x3 / stride-1 / padding-1 convolution of the BEV head whose WEIGHT operands reach the matrix cores through LDS
// ("generation 4" of k_conv2d_fwd, conv2d.hip).  Reference layers: rslo/models/odom_pred_base.py:155-207 (blocks,
// deblocks, skip blocks, heads), custom_resnet_spc.py:224-298 (BasicBlock); forward and data gradient (transposed, flipped
// weight operand) as in k_conv2d_fwd.
//
// What bounds k_conv2d_fwd (profiles/NOTES.md round 5): every wave streams its own 27 KB of split weight operands per
// 32-channel chunk through the CU's vector L1 -- 16 cycles per 1 KB operand on the 64 B/clk L1 path = one MFMA's time, and
// the 32 KB L1 is thrashed (234 MB of L2->L1 requests per launch of a 24 MB problem).  Here a workgroup's weight operands are
// copied ONCE per workgroup into LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no registers, the
// pre-split operand of k_conv2d_wsplit is already in fragment order = lane-linear) and every wave reads its A operands
// with ds_read_b128 (4 cycles per KB):
//
//   workgroup = NW waves, out tile = 32 channels x 2 NW rows x 16 pixels; a wave = BOTH 16-channel blocks x 2 rows, so an
//   A fragment serves two pixel blocks and a B fragment two channel blocks (12 LDS reads per 24 MFMAs);
//   weight stage = one kernel row (3 taps x 2 blocks x 3 planes x 1 KB = 18 KB), ring of two stages: the DMA of stage
//   s + 1 is issued when stage s starts and has a whole stage (72 MFMAs per wave) to land;
//   activation halo staged + split once per chunk as in k_conv2d_fwd (single buffer, 224-byte pixel records);
//   LDS = 36 KB + (2 NW + 2) x 18 x 224 B (NW = 4: 76 KB -> two workgroups per CU).
//
// Arithmetic is k_conv2d_fwd's: same split, same six products per block smallest first, taps and chunks in the same
// order -> bit-identical results (tests/test_gpu_kernels.py::test_conv2d_lds_weight_kernel_keeps_the_bits).
#include <stdlib.h>

#include "conv2d_tile.h"

#define WL_PXB 224            // bytes per staged pixel (3 planes x 64 B + 32 B pad: conflict-free ds_read_b128, conv2d.hip)
#define WL_STAGE 18432        // bytes per weight stage: 3 taps x 2 channel blocks x 3 planes x 1 KB

// One 1 KB fragment global -> LDS (lane l's 16 bytes land at lds_dst + 16 l).  hipcc does not count the copy: the caller
// waits (s_waitcnt vmcnt) and then passes a workgroup barrier before anyone reads it (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void wl_dma16(const glb_u8 *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int NW, int ABL = 0>      // ABL: experiments only (conv2d_ablate), wrong results
__global__ __launch_bounds__(64 * NW, 2) void k_conv2d_wl(const float *__restrict__ in, const unsigned short *__restrict__ Ws,
                                                          const float *__restrict__ bias, Conv2dFwdGeom gm,
                                                          float *__restrict__ out) {
  constexpr int TR = 2 * NW, HR = TR + 2, NPX = HR * 18, NTH = 64 * NW;
  constexpr int NTASK = (NPX * 4 + NTH - 1) / NTH;
  constexpr int NFR = (18 + NW - 1) / NW;      // DMA fragments per wave and stage
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WL_STAGE + NPX * WL_PXB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  int bx, by;
  if (!conv2d_xcd_tile(gm.xsc, gm.npix, gm.ny, bx, by)) return;
  const int tx = bx % gm.tiles_x; bx /= gm.tiles_x;
  const int ty = bx % gm.tiles_y;
  const int b = bx / gm.tiles_y;
  const int x0 = tx * 16, y0 = ty * TR;
  const int H = gm.H, W = gm.W;
  const int64_t HW = (int64_t)H * W;
  const int n_mt = gm.cout / 16;
  const int mt0 = by * 2;                      // first 16-channel output block of this workgroup

  f32x4 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // activation staging tasks, as in k_conv2d_fwd: task = (channel octet o, halo pixel q), q fastest
  int tsrc[NTASK], tdst[NTASK];
  unsigned tok = 0;
  const glb_u8 *const in_b = (const glb_u8 *)in;
#pragma unroll
  for (int r = 0; r < NTASK; ++r) {
    const int task = tid + r * NTH;
    const int o = task / NPX, q = task - o * NPX;
    const int qy = q / 18, qx = q - qy * 18;
    const int y = y0 - 1 + qy, x = x0 - 1 + qx;
    const bool ok = task < NPX * 4 && y >= 0 && y < H && x >= 0 && x < W;
    // a task outside the image loads the plane's first pixel instead (no branch around the loads) and stages zeros
    tsrc[r] = (int)((((int64_t)b * gm.cin + (task < NPX * 4 ? 8 * o : 0)) * HW + (ok ? (int64_t)y * W + x : 0)) * 4);
    tdst[r] = task < NPX * 4 ? 2 * WL_STAGE + q * WL_PXB + o * 16 : -1;
    if (ok) tok |= 1u << r;
  }
#define WL_RAW(CH, J, OFF) (*(const glb_f32 *)(c2f_uniform(in_b + ((int64_t)(CH) * 32 + (J)) * HW * 4) + (unsigned)(OFF)))
  float raw[NTASK][8];
  const int n_chunks = gm.cin / 32;
  const int n_stages = 3 * n_chunks;
#pragma unroll
  for (int r = 0; r < NTASK; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[r][j] = WL_RAW(0, j, tsrc[r]);

  // weight stage s = (chunk, kernel row): fragment f = tap-in-row * 6 + (block * 3 + plane); the six fragments of a tap are
  // contiguous in the pre-split operand ([chunk][tap][n_mt][3][64][8] bf16) and in the stage buffer
  const glb_u8 *const wsrc = (const glb_u8 *)Ws + (int64_t)mt0 * 3072 + lane * 16;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_u8 *)lds;
#define WL_ISSUE(S)                                                                                            \
  {                                                                                                            \
    const int s_ = (S);                                                                                        \
    const int64_t row_ = (int64_t)s_ * 3;      /* first tap of the stage: (chunk * 9 + ky * 3) */               \
    _Pragma("unroll") for (int i = 0; i < NFR; ++i) {                                                          \
      const int f = wid + i * NW;                                                                              \
      if (f < 18) {                                                                                            \
        const int tl = f / 6, j = f - 6 * tl;                                                                  \
        wl_dma16(wsrc + ((row_ + tl) * n_mt) * 3072 + j * 1024, lds0 + (unsigned)((s_ & 1) * WL_STAGE + f * 1024)); \
      }                                                                                                        \
    }                                                                                                          \
  }
  if (!(ABL & 1)) WL_ISSUE(0)

  const lds_u8 *const bbase = (const lds_u8 *)lds + 2 * WL_STAGE + (wid * 2 * 18 + li) * WL_PXB + g * 16;
  const lds_u8 *const abase = (const lds_u8 *)lds + lane * 16;
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int s = chunk * 3 + ky;
      if (ky == 0) {
        if (chunk > 0 && !(ABL & 4)) C2F_LDS_BARRIER();      // everyone is done with the previous chunk's halo
#pragma unroll
        for (int r = 0; r < NTASK; ++r) {
          if (tdst[r] >= 0 && !(ABL & 2)) {
            const Split3 sp = split_masked(raw[r], ((tok >> r) & 1u) ? 0xffu : 0u);
            unsigned char *dst = lds + tdst[r];
            *(u32x4 *)(dst) = sp.h;
            *(u32x4 *)(dst + 64) = sp.m;
            *(u32x4 *)(dst + 128) = sp.l;
          }
        }
      }
      // this wave's part of stage s has landed; behind the barrier everyone's has, the halo is visible, and nobody reads
      // the other stage buffer any more
      if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(ABL & 4)) C2F_LDS_BARRIER();
      // the next chunk's values first: hipcc orders a reload of `raw` behind the previous loads with a vmcnt(0) of its own,
      // which must not find the stage DMA (uncounted by it) in the queue
      // (one task per stage: eight loads get a whole stage to land instead of 24 sharing it)
      if (chunk + 1 < n_chunks && !(ABL & 2)) {
#pragma unroll
        for (int r = ky; r < NTASK; r += 3)
#pragma unroll
          for (int j = 0; j < 8; ++j) raw[r][j] = WL_RAW(chunk + 1, j, tsrc[r]);
      }
      if (s + 1 < n_stages && !(ABL & 1)) WL_ISSUE(s + 1)
      const lds_u8 *const wa = abase + (s & 1) * WL_STAGE;
      u32x4 pa[2][3], pb[2][3];
#define WL_LOAD(KX)                                                                     \
  _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                       \
    const lds_u8 *ap = wa + ((KX) * 2 + t) * 3072;                                      \
    const lds_u8 *bp = bbase + ((t + ky) * 18 + (KX)) * WL_PXB;                         \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                  \
      pa[t][pl] = *(const lds_u32x4 *)(ap + pl * 1024);                                 \
      pb[t][pl] = *(const lds_u32x4 *)(bp + pl * 64);                                   \
    }                                                                                   \
  }
      if (!(ABL & 8) || s == 0) { WL_LOAD(0) }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        u32x4 a[2][3], bq[2][3];      // [block][h | m | l]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) { a[t][pl] = pa[t][pl]; bq[t][pl] = pb[t][pl]; }
        if (kx < 2 && !(ABL & 8)) { WL_LOAD(kx + 1) }
        // the next tap's twelve LDS reads are issued HERE and land during this tap's 24 MFMAs: without the fence the
        // scheduler sinks each read to its first use to save registers (s_waitcnt lgkmcnt(0) in front of every other
        // MFMA, 8 waves per CU cannot hide that)
        __builtin_amdgcn_sched_barrier(0);
        // six products per block, smallest first; consecutive MFMAs hit different accumulators
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = MFMA_BF16(a[mt][2], bq[nt][0], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = MFMA_BF16(a[mt][1], bq[nt][1], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = MFMA_BF16(a[mt][0], bq[nt][2], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = MFMA_BF16(a[mt][1], bq[nt][0], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = MFMA_BF16(a[mt][0], bq[nt][1], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = MFMA_BF16(a[mt][0], bq[nt][0], acc[mt][nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef WL_LOAD
    }
  }
#undef WL_ISSUE
#undef WL_RAW

  const int x = x0 + li;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = (mt0 + mt) * 16 + 4 * g + j;
      const float bv = bias ? bias[m] : 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int y = y0 + wid * 2 + nt;
        if (x < W && y < H) {
          const int64_t o = ((int64_t)b * gm.cout + m) * HW + (int64_t)y * W + x;
          float v = acc[mt][nt][j] + bv;
          if (gm.res) v += gm.res[o];
          out[o] = v;
        }
      }
    }
}

// MEASURED SLOWER than k_conv2d_fwd on every layer of the head (round 6, scripts/bench_conv2d_wl.py, B = 4, us per launch
// k_conv2d_fwd | 8-row tiles | 6-row tiles: 128->128 @48x88 35.8 | 46.9 | 46.4; 64->64 @96x176 34.6 | 42.0 | 46.1; 192->64 96.1 |
// 106.8 | 112.7; 128->128 @24x44 15.6 | 20.5 | 20.3) -- with 76 KB of LDS a CU holds two workgroups = 8 waves, which cannot
// hide what the 20 waves of k_conv2d_fwd hide (ablations on 128->128 @48x88: full 49.3, no weight DMA 42.9, no halo staging
// 37.9, neither 31.9, MFMAs only 27.5 us: 576 equal workgroups on 512 slots run as one full round + one of 64).  The kernel
// is therefore OFF by default (conv2d_fwd_wl = 0); 1 / 3 select it with 8- / 6-row tiles (the bit-identity test, A/B runs).
int conv2d_wl_wanted(int B, int cin, int cout, int H, int W) {
  (void)B; (void)H; (void)W;
  return rslo_tune(RSLO_TUNE_CONV2D_FWD_WL) > 0 && cin % 32 == 0 && cout % 32 == 0;
}

int conv2d_wl_launch(const float *in, const void *Ws, const float *bias, const float *res, int B, int cin, int cout, int H,
                     int W, float *out, void *stream) {
  Conv2dFwdGeom gm;
  gm.B = B; gm.cin = cin; gm.cout = cout; gm.H = H; gm.W = W;
  gm.res = res;
  const int mode_ = rslo_tune(RSLO_TUNE_CONV2D_FWD_WL);
  const int nw = mode_ == 3 ? 3 : 4;
  gm.tiles_x = (int)rslo_cdiv(W, 16);
  gm.tiles_y = (int)rslo_cdiv(H, 2 * nw);
  const double w_bytes = 3.0 * 18.0 * cin * cout, in_bytes = 4.0 * B * cin * H * W * (1.0 + 1.0 / nw);
  gm.npix = B * gm.tiles_x * gm.tiles_y;
  gm.ny = cout / 32;
  gm.xsc = conv2d_xcd_split(RSLO_TUNE_CONV2D_FWD_XSC, gm.ny, w_bytes, in_bytes);
  const dim3 grid = conv2d_xcd_grid(gm.xsc, gm.npix, gm.ny);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *ws = (const unsigned short *)Ws;
  // (the ablations of profiles/NOTES.md round 6 were k_conv2d_wl<4, ABL> for ABL = 1, 2, 3, 4, 7, 8, 15 behind conv2d_ablate)
  if (nw == 3) hipLaunchKernelGGL((k_conv2d_wl<3>), grid, dim3(192), 0, st, in, ws, bias, gm, out);
  else hipLaunchKernelGGL((k_conv2d_wl<4>), grid, dim3(256), 0, st, in, ws, bias, gm, out);
  RSLO_CHECK_LAUNCH("k_conv2d_wl");
  return RSLO_OK;
}
