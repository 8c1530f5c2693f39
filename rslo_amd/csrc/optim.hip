// Optimizer step of the training path on gfx950: gradient-norm clipping and the Adam update with decoupled weight
// decay over ALL parameter tensors of the model in three launches.
//
// Reference: the driver clips the global gradient norm (train_hdf5.py:671, torch.nn.utils.clip_grad_norm_(.., 10.0))
// and steps a fastai OptimWrapper around torch.optim.Adam (rslo/torchplus/train/fastai_optim.py:176-187: p *= 1 - wd lr,
// then the inner Adam step with weight_decay = 0; 8 parameter groups from rslo/builder/optimizer_builder.py:49-66).
// Through torch that is ~25 multi-tensor launches per step (norms, the clip product, one decay product and one fused
// Adam launch per group) issued from Python; here a tensor table + a chunk table (built once on the host) drive
//   k_opt_sqnorm : per-chunk sum of squares (double), fixed order
//   k_opt_clip   : every block re-adds the chunk partials in the same order (bit-identical total in all blocks),
//                  coef = min(1, max_norm / (norm + 1e-6)); scales its chunk of the gradients when coef < 1
//   k_opt_adam   : decay, moment updates, bias-corrected step -- HBM-bound: reads p, g, m, v, writes p, m, v (28 B/elem)
// State stays in torch.optim.Adam's own tensors (exp_avg, exp_avg_sq, step), so checkpoints are unchanged.
#include "rslo_common.h"

#define OPT_THREADS 256

__global__ __launch_bounds__(OPT_THREADS) void k_opt_sqnorm(const RsloOptTensor *__restrict__ tensors,
                                                            const RsloOptChunk *__restrict__ chunks,
                                                            double *__restrict__ partial) {
  const RsloOptChunk c = chunks[blockIdx.x];
  const float *__restrict__ g = tensors[c.tensor].grad;
  if (g == nullptr) {      // no gradient this step (uniform per block)
    if (threadIdx.x == 0) partial[blockIdx.x] = 0.0;
    return;
  }
  g += c.offset;
  double s = 0.0;
  const int n4 = (c.count / 4) * 4;
  for (int i = threadIdx.x * 4; i < n4; i += OPT_THREADS * 4) {       // chunk offsets are multiples of 4 elements
    const float4 v = *reinterpret_cast<const float4 *>(g + i);
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  for (int i = n4 + threadIdx.x; i < c.count; i += OPT_THREADS) s += (double)g[i] * g[i];
  __shared__ double red[OPT_THREADS];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = OPT_THREADS / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__device__ __forceinline__ double opt_total(const double *__restrict__ partial, int n_chunks, double *red) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n_chunks; i += OPT_THREADS) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = OPT_THREADS / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  return red[0];
}

__global__ __launch_bounds__(OPT_THREADS) void k_opt_clip(const RsloOptTensor *__restrict__ tensors,
                                                          const RsloOptChunk *__restrict__ chunks,
                                                          const double *__restrict__ partial, int n_chunks,
                                                          float max_norm, float *__restrict__ total_norm) {
  __shared__ double red[OPT_THREADS];
  const float norm = (float)sqrt(opt_total(partial, n_chunks, red));
  if (blockIdx.x == 0 && threadIdx.x == 0) *total_norm = norm;
  const float coef = max_norm / (norm + 1e-6f);        // torch: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
  if (!(coef < 1.0f)) return;
  const RsloOptChunk c = chunks[blockIdx.x];
  float *__restrict__ g = tensors[c.tensor].grad;
  if (g == nullptr) return;
  g += c.offset;
  const int n4 = (c.count / 4) * 4;
  for (int i = threadIdx.x * 4; i < n4; i += OPT_THREADS * 4) {
    float4 v = *reinterpret_cast<float4 *>(g + i);
    v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
    *reinterpret_cast<float4 *>(g + i) = v;
  }
  for (int i = n4 + threadIdx.x; i < c.count; i += OPT_THREADS) g[i] *= coef;
}

// torch.optim.Adam (amsgrad = False, maximize = False, weight_decay = 0) after the wrapper's decoupled decay:
//   p <- p (1 - wd lr);  m <- b1 m + (1 - b1) g;  v <- b2 v + (1 - b2) g g
//   p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// torch's fused kernel evaluates these with the hyper-parameters as doubles (so 1 - beta2 = 0.001 is not rounded to
// fp32 first: in fp32, 1 - 0.999f is off by 1.3e-5 relative); the same here -- the kernel is bound by its 28 bytes per
// element, the fp64 operations are free.
__device__ __forceinline__ void adam_elem(float &p, float g, float &m, float &v, float decay, double b1, double b2,
                                          double step_size, double sqrt_bc2, double eps) {
  p *= decay;
  const double md = b1 * (double)m + (1.0 - b1) * (double)g;
  const double vd = b2 * (double)v + (1.0 - b2) * (double)g * (double)g;
  m = (float)md;
  v = (float)vd;
  const double denom = sqrt(vd) / sqrt_bc2 + eps;
  p = (float)((double)p - step_size * md / denom);
}

__global__ __launch_bounds__(OPT_THREADS) void k_opt_adam(const RsloOptTensor *__restrict__ tensors,
                                                          const RsloOptChunk *__restrict__ chunks,
                                                          RsloOptHyper hyper, float step) {
  const RsloOptChunk c = chunks[blockIdx.x];
  const RsloOptTensor t = tensors[c.tensor];
  const RsloOptGroup h = hyper.group[t.group];
  const float tstep = step + (float)t.step_offset;      // torch.optim.Adam keeps one count per tensor
  const double bc1 = 1.0 - pow(h.beta1, (double)tstep), bc2 = 1.0 - pow(h.beta2, (double)tstep);
  const double step_size = h.lr / bc1, sqrt_bc2 = sqrt(bc2);
  const float decay = (float)(1.0 - h.weight_decay * h.lr);      // torch._foreach_mul_(params, 1 - wd * lr): fp32 product
  float *__restrict__ p = t.param + c.offset;
  const int n4 = (c.count / 4) * 4;
  if (t.grad == nullptr) {
    // no gradient this step: torch.optim.Adam skips the tensor, the wrapper's decoupled decay does not
    // (fastai_optim.py:176-187 multiplies every requires_grad parameter)
    if (decay != 1.0f) {
      for (int i = threadIdx.x * 4; i < n4; i += OPT_THREADS * 4) {
        float4 pp = *reinterpret_cast<float4 *>(p + i);
        pp.x *= decay; pp.y *= decay; pp.z *= decay; pp.w *= decay;
        *reinterpret_cast<float4 *>(p + i) = pp;
      }
      for (int i = n4 + threadIdx.x; i < c.count; i += OPT_THREADS) p[i] *= decay;
    }
    return;
  }
  const float *__restrict__ g = t.grad + c.offset;
  float *__restrict__ m = t.exp_avg + c.offset;
  float *__restrict__ v = t.exp_avg_sq + c.offset;
  for (int i = threadIdx.x * 4; i < n4; i += OPT_THREADS * 4) {
    float4 pp = *reinterpret_cast<float4 *>(p + i), mm = *reinterpret_cast<float4 *>(m + i),
           vv = *reinterpret_cast<float4 *>(v + i);
    const float4 gg = *reinterpret_cast<const float4 *>(g + i);
    adam_elem(pp.x, gg.x, mm.x, vv.x, decay, h.beta1, h.beta2, step_size, sqrt_bc2, h.eps);
    adam_elem(pp.y, gg.y, mm.y, vv.y, decay, h.beta1, h.beta2, step_size, sqrt_bc2, h.eps);
    adam_elem(pp.z, gg.z, mm.z, vv.z, decay, h.beta1, h.beta2, step_size, sqrt_bc2, h.eps);
    adam_elem(pp.w, gg.w, mm.w, vv.w, decay, h.beta1, h.beta2, step_size, sqrt_bc2, h.eps);
    *reinterpret_cast<float4 *>(p + i) = pp;
    *reinterpret_cast<float4 *>(m + i) = mm;
    *reinterpret_cast<float4 *>(v + i) = vv;
  }
  for (int i = n4 + threadIdx.x; i < c.count; i += OPT_THREADS)
    adam_elem(p[i], g[i], m[i], v[i], decay, h.beta1, h.beta2, step_size, sqrt_bc2, h.eps);
  if (c.offset == 0 && threadIdx.x == 0 && t.step) *t.step = tstep;     // torch keeps the count as a float tensor
}

extern "C" int rslo_opt_clip_grad_norm(const RsloOptTensor *tensors_dev, const RsloOptChunk *chunks_dev, int n_chunks,
                                       float max_norm, double *partial_ws, float *total_norm, void *stream) {
  RSLO_CHECK_ARG(tensors_dev && chunks_dev && partial_ws && total_norm && n_chunks > 0 && max_norm > 0.f,
                 "rslo_opt_clip_grad_norm: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_opt_sqnorm, dim3(n_chunks), dim3(OPT_THREADS), 0, st, tensors_dev, chunks_dev, partial_ws);
  hipLaunchKernelGGL(k_opt_clip, dim3(n_chunks), dim3(OPT_THREADS), 0, st, tensors_dev, chunks_dev, partial_ws,
                     n_chunks, max_norm, total_norm);
  RSLO_CHECK_LAUNCH("k_opt_clip");
  return RSLO_OK;
}

extern "C" int rslo_opt_adam_step(const RsloOptTensor *tensors_dev, const RsloOptChunk *chunks_dev, int n_chunks,
                                  const RsloOptHyper *hyper, float step, void *stream) {
  RSLO_CHECK_ARG(tensors_dev && chunks_dev && hyper && n_chunks > 0 && step >= 1.f, "rslo_opt_adam_step: bad arguments");
  hipLaunchKernelGGL(k_opt_adam, dim3(n_chunks), dim3(OPT_THREADS), 0, (hipStream_t)stream, tensors_dev, chunks_dev,
                     *hyper, step);
  RSLO_CHECK_LAUNCH("k_opt_adam");
  return RSLO_OK;
}
