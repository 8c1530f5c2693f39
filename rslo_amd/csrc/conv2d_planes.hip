// Dense 3x3 / stride-1 / padding-1 convolution of the BEV head from PRE-SPLIT operand planes ("generation 3" of
// k_conv2d_fwd, conv2d.hip).  Reference layers: rslo/models/odom_pred_base.py:155-207, custom_resnet_spc.py:224-298.
//
// k_conv2d_fwd stages a halo tile from NCHW fp32: 8 strided dword loads per pixel-octet, the three-way bf16 split on the
// vector ALU and three 16-byte LDS stores -- once per WORKGROUP, i.e. cout / 32 times per input value, serial with the
// matrix work (profiles/NOTES.md round 4: ~1870 of ~5000 cycles per chunk).  Here the producer of an activation (the
// BatchNorm apply kernels, bn2d.hip, or k_opl_from_nchw below) writes it ONCE as operand planes
//
//     P[b][c / 8][plane = hi | mid | lo][y][x][c % 8]   bf16   (one 16-byte piece = the 8 channels of an octet at a pixel)
//
// which is exactly what one lane of v_mfma_f32_16x16x32_bf16 holds as its B operand (lane = pixel + 16 * octet).  The
// convolution then stages with plain 16-byte copies (global -> register -> LDS, the next chunk's pieces in flight during
// the current chunk's MFMAs) and issues no split at all.  LDS image: [plane][octet][OST slots of 16 bytes], OST = the
// halo's pixel count rounded up to a multiple of 16: the 16-lane groups ds_read_b128 is served in
// ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS table) then touch 16 distinct slots = all 64 banks, whatever the tap.
// Arithmetic is the arithmetic of k_conv2d_fwd (same split, same six products in the same order, same chunk order): the
// results are bit-identical to it (tests/test_gpu_kernels.py::test_conv2d_planes_*).
#include <stdlib.h>

#include "conv2d_tile.h"

// ---- producer: NCHW fp32 -> operand planes (stand-alone form; the BatchNorm apply kernels write the same layout) ------
// thread = (pixel, octet): 8 channel loads (lanes along the pixels: coalesced), one split, three 16-byte stores
// (lane-contiguous: 1 KB per store instruction)
__global__ __launch_bounds__(256) void k_opl_from_nchw(const float *__restrict__ in, int C, int HW,
                                                       u32x4 *__restrict__ planes) {
  const int p = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
  if (p >= HW) return;
  const float *src = in + ((int64_t)b * C + 8 * o) * HW + p;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)j * HW];
  const Split3 s = split_masked(v, 0xffu);
  u32x4 *dst = planes + ((int64_t)b * (C / 8) + o) * 3 * HW + p;
  dst[0] = s.h;
  dst[HW] = s.m;
  dst[2 * (int64_t)HW] = s.l;
}

extern "C" size_t rslo_opl_bytes(int B, int C, int H, int W) { return (size_t)B * C * H * W * 6; }

extern "C" int rslo_opl_from_nchw(const float *in, int B, int C, int H, int W, void *planes, void *stream) {
  RSLO_CHECK_ARG(in && planes && B > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0 && B < 65536,
                 "rslo_opl_from_nchw: bad arguments (C must be a multiple of 8)");
  const int HW = H * W;
  hipLaunchKernelGGL(k_opl_from_nchw, dim3((unsigned)rslo_cdiv(HW, 256), (unsigned)(C / 8), (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, in, C, HW, (u32x4 *)planes);
  RSLO_CHECK_LAUNCH("k_opl_from_nchw");
  return RSLO_OK;
}

// ---- consumer ----------------------------------------------------------------------------------------------------------
// Workgroup = 4 waves as 2 (output-channel halves) x 2 (row halves); out tile = 32 MTW channels x TR x 16 pixels, as
// k_conv2d_fwd.  NAHEAD: weight operands of that many taps in registers, each refilled for tap + NAHEAD right after its
// MFMAs (9: a whole chunk ahead; 1: one tap ahead through a second register set).
template <int TR, int MTW, int NAHEAD, int OCC, int ABL = 0>      // ABL: experiments only (RSLO_TUNE_CONV2D_ABLATE), wrong results
__global__ __launch_bounds__(256, OCC) void k_conv2d_fwd_p(const u32x4 *__restrict__ planes,
                                                           const unsigned short *__restrict__ Ws,
                                                           const float *__restrict__ bias, Conv2dFwdGeom gm,
                                                           float *__restrict__ out) {
  constexpr int wmul = (ABL & 1) ? 0 : 1, amul = (ABL & 2) ? 0 : 1;
  constexpr int NTW = TR / 2, HR = TR + 2, NPX = HR * 18, OST = (NPX + 15) / 16 * 16;
  constexpr int NPC = 12 * NPX, NTASK = (NPC + 255) / 256;
  __shared__ __attribute__((aligned(16))) u32x4 lds[12 * OST];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wid & 1, wn = wid >> 1;
  int bx, by;
  if (!conv2d_xcd_tile(gm.xsc, gm.npix, gm.ny, bx, by)) return;
  const int tx = bx % gm.tiles_x; bx /= gm.tiles_x;
  const int ty = bx % gm.tiles_y;
  const int b = bx / gm.tiles_y;
  const int x0 = tx * 16, y0 = ty * TR;
  const int H = gm.H, W = gm.W;
  const int HW = H * W;
  const int n_mt = gm.cout / 16;
  const int mt0 = by * 2 * MTW + wm * MTW;          // first 16-channel output block of this wave

  f32x4 acc[MTW][NTW];
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging tasks: task = (plane, octet, halo pixel q), q fastest; a task's piece of chunk c sits at the image's base +
  // c * 12 HW pieces (uniform) + the task's own byte offset tsrc[r] (-1: outside the image -> zeros)
  int tsrc[NTASK], tdst[NTASK];
#pragma unroll
  for (int r = 0; r < NTASK; ++r) {
    const int task = tid + r * 256;
    const int po = task / NPX, q = task - po * NPX;
    const int pl = po >> 2, o = po & 3;
    const int qy = q / 18, qx = q - qy * 18;
    const int y = y0 - 1 + qy, x = x0 - 1 + qx;
    const bool live = task < NPC;
    const bool ok = live && y >= 0 && y < H && x >= 0 && x < W;
    tsrc[r] = ok ? ((o * 3 + pl) * HW + y * W + x) * 16 : -1;
    tdst[r] = live ? po * OST + q : -1;
  }
  const glb_u8 *const img = (const glb_u8 *)(planes + (int64_t)b * (gm.cin / 8) * 3 * HW);
  const int64_t cstride = (int64_t)12 * HW * 16;      // bytes per 32-channel chunk
#define C2P_RAW(CH, OFF) (*(const glb_u32x4 *)(c2f_uniform(img + (int64_t)((CH) * amul) * cstride) + (unsigned)(OFF)))
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 raw[NTASK];
  const int n_chunks = gm.cin / 32;
#pragma unroll
  for (int r = 0; r < NTASK; ++r) raw[r] = tsrc[r] >= 0 ? C2P_RAW(0, tsrc[r]) : zero4;

  constexpr bool FULLA = NAHEAD > 1;
  constexpr int NA = NAHEAD;
  u32x4 ah[NA][MTW], am[NA][MTW], al[NA][MTW], nh[MTW], nm[MTW], nl[MTW];
  const unsigned short *wbase = Ws + (int64_t)mt0 * 3 * 512 + lane * 8;
#define C2P_LOAD_A(CH, TAP, H_, M_, L_)                                                                  \
  _Pragma("unroll") for (int mt = 0; mt < MTW; ++mt) {                                                    \
    const unsigned short *wp = wbase + ((((int64_t)(CH) * 9 + (TAP)) * wmul * n_mt + mt) * 3) * 512;           \
    H_[mt] = *(const u32x4 *)(wp);                                                                        \
    M_[mt] = *(const u32x4 *)(wp + 512);                                                                  \
    L_[mt] = *(const u32x4 *)(wp + 1024);                                                                 \
  }
  if (FULLA) {
#pragma unroll
    for (int t = 0; t < NA; ++t) { C2P_LOAD_A(0, t, ah[t], am[t], al[t]) }
  } else {
    C2P_LOAD_A(0, 0, ah[0], am[0], al[0])
  }
  const lds_u8 *const bbase = (const lds_u8 *)lds + (g * OST + wn * NTW * 18 + li) * 16;
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    // this chunk's pieces -> LDS, then the next chunk's into the registers (in flight during the MFMAs)
#pragma unroll
    for (int r = 0; r < NTASK; ++r)
      if (tdst[r] >= 0) lds[tdst[r]] = raw[r];
    if (chunk + 1 < n_chunks) {
#pragma unroll
      for (int r = 0; r < NTASK; ++r) raw[r] = tsrc[r] >= 0 ? C2P_RAW(chunk + 1, tsrc[r]) : zero4;
    }
    C2F_LDS_BARRIER();      // LDS only: the prefetched global loads stay in flight across it
    u32x4 pbh[NTW], pbm[NTW], pbl[NTW];
#define C2P_LOAD_B(KY, KX)                                                                               \
  _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) {                                                    \
    const lds_u8 *bp = bbase + ((nt + (KY)) * 18 + (KX)) * 16;          /* one address register + immediates */ \
    pbh[nt] = *(const lds_u32x4 *)(bp);                                                                   \
    pbm[nt] = *(const lds_u32x4 *)(bp + 4 * OST * 16);                                                    \
    pbl[nt] = *(const lds_u32x4 *)(bp + 8 * OST * 16);                                                    \
  }
    C2P_LOAD_B(0, 0)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tap = ky * 3 + kx;
        const int ia = FULLA ? tap % NA : 0;
        if (!FULLA) {
          if (tap < 8) {
            C2P_LOAD_A(chunk, tap + 1, nh, nm, nl)
          } else if (chunk + 1 < n_chunks) {
            C2P_LOAD_A(chunk + 1, 0, nh, nm, nl)
          }
        }
        // pixel operands one tap ahead: the LDS reads of tap + 1 are in flight during this tap's MFMAs
        u32x4 bh[NTW], bm[NTW], bl[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          bh[nt] = pbh[nt];
          bm[nt] = pbm[nt];
          bl[nt] = pbl[nt];
        }
        if (tap < 8 && !(ABL & 8)) {
          const int ky2 = (tap + 1) / 3, kx2 = (tap + 1) % 3;
          C2P_LOAD_B(ky2, kx2)
        }
        // six products per block, smallest first; consecutive MFMAs hit different accumulators
        if constexpr (!(ABL & 4)) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(al[ia][mt], bh[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(am[ia][mt], bm[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(ah[ia][mt], bl[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(am[ia][mt], bh[nt], acc[mt][nt]);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(ah[ia][mt], bm[nt], acc[mt][nt]);
        }
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = MFMA_BF16(ah[ia][mt], bh[nt], acc[mt][nt]);
        if (FULLA) {
          if (tap + NA < 9) {
            C2P_LOAD_A(chunk, tap + NA, ah[ia], am[ia], al[ia])
          } else if (chunk + 1 < n_chunks) {
            C2P_LOAD_A(chunk + 1, tap + NA - 9, ah[ia], am[ia], al[ia])
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            ah[0][mt] = nh[mt];
            am[0][mt] = nm[mt];
            al[0][mt] = nl[mt];
          }
        }
      }
    }
    C2F_LDS_BARRIER();
  }
#undef C2P_LOAD_A
#undef C2P_LOAD_B
#undef C2P_RAW

  const int x = x0 + li;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = (mt0 + mt) * 16 + 4 * g + j;
      const float bv = bias ? bias[m] : 0.f;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int y = y0 + wn * NTW + nt;
        if (x < W && y < H) {
          const int64_t o = ((int64_t)b * gm.cout + m) * HW + (int64_t)y * W + x;
          float v = acc[mt][nt][j] + bv;
          if (gm.res) v += gm.res[o];
          out[o] = v;
        }
      }
    }
}

extern "C" int rslo_conv2d_fwd_p_supported(int cin, int cout, int H, int W) {
  return cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0 && H > 0 && W > 0;
}

// planes: rslo_opl_bytes(B, cin, H, W) bytes in the layout above; Ws: rslo_conv2d_wsplit (transpose = 1 and the planes of
// dout: the data gradient); out [B,cout,H,W] fp32 = conv + bias (+ res)
extern "C" int rslo_conv2d_fwd_p(const void *planes, const void *Ws, const float *bias, const float *res, int B, int cin,
                                 int cout, int H, int W, float *out, void *stream) {
  RSLO_CHECK_ARG(planes && Ws && out && B > 0 && rslo_conv2d_fwd_p_supported(cin, cout, H, W),
                 "rslo_conv2d_fwd_p: unsupported shape cin=%d cout=%d H=%d W=%d", cin, cout, H, W);
  RSLO_CHECK_ARG((int64_t)B * cin * H * W * 6 < ((int64_t)1 << 31), "rslo_conv2d_fwd_p: tensor too large");
  Conv2dFwdGeom gm;
  gm.B = B; gm.cin = cin; gm.cout = cout; gm.H = H; gm.W = W;
  gm.res = res;
  int tr = 4, mtw = 1;
  const int cfg_tr = rslo_tune(RSLO_TUNE_CONV2D_FWD_TR), cfg_mtw = rslo_tune(RSLO_TUNE_CONV2D_FWD_MTW);
  if (cfg_tr == 8) tr = 8;
  if (cfg_mtw == 2 && cout % 64 == 0) mtw = 2;
  gm.tiles_x = (int)rslo_cdiv(W, 16);
  gm.tiles_y = (int)rslo_cdiv(H, tr);
  const double w_bytes = 3.0 * 18.0 * cin * cout, in_bytes = 6.0 * B * cin * H * W * 1.5;
  gm.npix = B * gm.tiles_x * gm.tiles_y;
  gm.ny = cout / (32 * mtw);
  gm.xsc = conv2d_xcd_split(RSLO_TUNE_CONV2D_FWD_XSC, gm.ny, w_bytes, in_bytes);
  const dim3 grid = conv2d_xcd_grid(gm.xsc, gm.npix, gm.ny);
  hipStream_t st = (hipStream_t)stream;
  const u32x4 *pl = (const u32x4 *)planes;
  const unsigned short *ws = (const unsigned short *)Ws;
  const int64_t wgs = (int64_t)gm.npix * gm.ny;
  const int lean_env = rslo_tune(RSLO_TUNE_CONV2D_FWD_LEAN);
  const bool lean = lean_env < 0 ? wgs >= 512 : lean_env == 1;
#define C2P_GO(TRv, MTWv, FAv, OCCv) \
  hipLaunchKernelGGL((k_conv2d_fwd_p<TRv, MTWv, FAv, OCCv>), grid, dim3(256), 0, st, pl, ws, bias, gm, out)
  if (tr == 8 && mtw == 2) C2P_GO(8, 2, 1, 2);
  else if (tr == 8) C2P_GO(8, 1, 1, 2);
  else if (mtw == 2) C2P_GO(4, 2, 1, 2);
  // (the ablations of profiles/NOTES.md round 5 were k_conv2d_fwd_p<4, 1, 1, 4, ABL> and the NAHEAD = 3 / 9 forms, behind conv2d_ablate)
  else if (lean) C2P_GO(4, 1, 1, 4);
  else C2P_GO(4, 1, 9, 2);
#undef C2P_GO
  RSLO_CHECK_LAUNCH("k_conv2d_fwd_p");
  return RSLO_OK;
}
