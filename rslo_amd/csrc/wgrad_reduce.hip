// All weight-gradient reduces of a backward pass in ONE launch (include/rslo_hip.h: rslo_wgrad_reduce_defer / _many).
//
// Every weight gradient of the step is two stages: a kernel that leaves slab / chunk partials in a workspace, and a small
// kernel that adds them in a fixed order (bit-reproducible, no atomics).  Per layer that second stage is a 2-10 us launch --
// 45 of them for the BEV head's 3x3 layers (reference: rslo/models/odom_pred_base.py:155-207), 5 for its 1x1 downsamples, 20
// for the sparse encoder (rslo/models/middle.py:119-213) -- and nothing in the pass reads their results: a gradient is first
// looked at by the gradient exchange / the clip + Adam step behind the pass (train_hdf5.py:663-672).  With a sink installed the
// launch code appends a descriptor instead of launching; rslo_wgrad_reduce_many runs the collected reduces as one grid whose
// blocks find their layer through a prefix table in the kernel arguments.  Same block bodies (wgrad_reduce.h), same bits.
#include "wgrad_reduce.h"

// per thread: a sink is installed, fed and removed by ONE caller (the thread that runs the backward pass); launches of other
// threads of the process are not its business
thread_local RsloWgradReduce *g_wr_sink = nullptr;
thread_local int g_wr_cap = 0;
thread_local int *g_wr_count = nullptr;

#define WR_MAX 40          // descriptors per launch: 40 x 88 B + 41 x 4 B of kernel arguments (< 4 KB)
struct WrBatch {
  int n;
  int start[WR_MAX + 1];          // first block of every layer; start[n] = grid size
  RsloWgradReduce d[WR_MAX];
};

__global__ __launch_bounds__(256) void k_wgrad_reduce_many(WrBatch b) {
  __shared__ float red[256];
  const int bid = (int)blockIdx.x;
  int l = 0, hi = b.n;       // (uniform: a scalar binary search over the kernel arguments -- the last l with start[l] <= bid)
  while (hi - l > 1) {
    const int mid = (l + hi) >> 1;
    if (bid >= b.start[mid]) l = mid;
    else hi = mid;
  }
  const RsloWgradReduce &d = b.d[l];
  const int lb = bid - b.start[l];
  if (d.kind == 0)
    wr_dense_block(lb, (const float *)d.ws, d.p[0], d.p[1], d.p[2], d.p[3], d.p[4], d.p[5], (float *)d.dW, d.p[7],
                   (const float *)d.aux, (float *)d.dbias, d.p[6], (float(*)[32])red);
  else
    wr_sparse_block(lb % d.p[5], lb / d.p[5], (const float *)d.ws, (const int32_t *)d.koff, d.p[0], d.p[1], d.p[2],
                    (float *)d.dW, (const float *)d.aux, d.p[3], d.p[4], (float *)d.dbias, red);
}

extern "C" int rslo_wgrad_reduce_defer(RsloWgradReduce *sink, int capacity, int *count) {
  RSLO_CHECK_ARG((sink && capacity > 0 && count) || (!sink && !count), "rslo_wgrad_reduce_defer: sink, capacity and count go together");
  g_wr_sink = sink;
  g_wr_cap = sink ? capacity : 0;
  g_wr_count = count;
  return RSLO_OK;
}

extern "C" int rslo_wgrad_reduce_many(const RsloWgradReduce *reduces, int n, void *stream) {
  RSLO_CHECK_ARG(n >= 0 && (n == 0 || reduces), "rslo_wgrad_reduce_many: bad arguments");
  for (int i0 = 0; i0 < n; i0 += WR_MAX) {
    WrBatch b;
    b.n = n - i0 < WR_MAX ? n - i0 : WR_MAX;
    int64_t tot = 0;
    for (int i = 0; i < b.n; ++i) {
      const RsloWgradReduce &d = reduces[i0 + i];
      RSLO_CHECK_ARG((d.kind == 0 || d.kind == 2) && d.n_blocks > 0 && d.ws && d.dW, "rslo_wgrad_reduce_many: descriptor %d is not one the launch code wrote", i0 + i);
      b.start[i] = (int)tot;
      b.d[i] = d;
      tot += d.n_blocks;
    }
    b.start[b.n] = (int)tot;
    RSLO_CHECK_ARG(tot < ((int64_t)1 << 31), "rslo_wgrad_reduce_many: grid too large");
    hipLaunchKernelGGL(k_wgrad_reduce_many, dim3((unsigned)tot), dim3(256), 0, (hipStream_t)stream, b);
    RSLO_CHECK_LAUNCH("k_wgrad_reduce_many");
  }
  return RSLO_OK;
}
