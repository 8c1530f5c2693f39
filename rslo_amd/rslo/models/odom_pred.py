"""BEV odometry head: per-unit transformation maps, confidences and the ego-motion vote
(reference: rslo/models/odom_pred.py:45-435; call stack SURVEY.md 3.3.1).

xs = [bev_0, bev_1(, bev_2)] each [B,128,96,176] -> all (i<j) pairs -> encoder-decoder -> per-cell
local (t, q) map -> local->global transform -> confidence-weighted mean = the pair's pose.
"""
import contextlib
import weakref

import apex
import apex.amp as amp
import torch
from torch import nn

import rslo.models.custom_resnet_spc as resnet
import numpy as np

from rslo.data.dataset import _grid_geometry, from_pointwise_local_transformation_tch
from rslo.layers.confidence import ConfidenceModule, masked_spatial_softmax
from rslo.layers import hip_conv2d

SMALL_MAPS_GATE = weakref.WeakKeyDictionary()      # head -> event where its latest forward reached the half-resolution stages
from rslo.layers.hip_conv2d import Conv2d
from rslo.layers.MaskConv import MaskConv
from rslo.models.odom_pred_base import OdomPredEncDecBase, conf_trunk
from rslo.utils.pose_utils import rotate_vec_by_q
from rslo.layers.SparseConv import FusedSequential
from torchplus.nn import Empty

# with the ROCm apex stand-in the per-layer `num_batches_tracked += 1` launches are batched (compat/apex/parallel.py);
# with a real apex they run as usual
_defer_batch_counts = getattr(apex.parallel, "defer_batch_counts", contextlib.nullcontext)


def _count_batch(bn):
    fn = getattr(apex.parallel, "count_batch", None)
    if fn is not None:
        fn(bn)
    elif bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)

REGISTERED_ODOM_PRED_CLASSES = {}


def register_odom_pred(cls, name=None):
    name = cls.__name__ if name is None else name
    assert name not in REGISTERED_ODOM_PRED_CLASSES, f"exist class: {REGISTERED_ODOM_PRED_CLASSES}"
    REGISTERED_ODOM_PRED_CLASSES[name] = cls
    return cls


def get_odom_class(name):
    assert name in REGISTERED_ODOM_PRED_CLASSES, f"available class: {REGISTERED_ODOM_PRED_CLASSES}"
    return REGISTERED_ODOM_PRED_CLASSES[name]



def _avgpool_321(m):
    """nn.AvgPool2d(3, 2, 1) exactly as k_head_masks_* hard-codes it: divisor 9 everywhere (count_include_pad, no
    divisor_override), floor output size."""
    return (isinstance(m, nn.AvgPool2d) and (m.kernel_size, m.stride, m.padding) == (3, 2, 1)
            and m.count_include_pad and not m.ceil_mode and m.divisor_override is None)


def _maxpool_321(m):
    """nn.MaxPool2d(3, 2, 1) as the kernels implement it: dilation 1, floor output size, no indices."""
    return (isinstance(m, nn.MaxPool2d) and (m.kernel_size, m.stride, m.padding) == (3, 2, 1)
            and m.dilation in (1, (1, 1)) and not m.ceil_mode and not m.return_indices)

class UNOdomPredEncDecSVDTempMaskBase(OdomPredEncDecBase):
    consumes_presplit_event = True      # _forward waits for hip_conv2d.presplit_early's event instead of splitting itself
    graph_capturable = True             # fixed map size, no host read: rslo_amd/headgraph.py may replay the training pass

    def __init__(self, use_svd=True, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.use_svd = use_svd
        self.fused_vote = True      # local->global + vote through rslo_vote_fwd / _bwd on GPU tensors
        # The MaskConv encoder carries an occupancy mask next to the features (max-pooled per conv, averaged at every
        # residual add); nothing in this head ever reads it (only x[0] leaves the encoder), so by default it is not
        # propagated: ~80 small launches per step that cannot change any output.  True restores the propagation.
        self.track_masks = getattr(self, "_bn_type", None) == "MaskSyncBN"      # that variant reads the masks
        nuf = list(kwargs.get("num_upsample_filters"))
        conf_type = kwargs.get("conf_type")
        motion, tconf, qconf = [], [], []
        if self.pred_pyramid_motion:
            for c in nuf:
                motion.append(FusedSequential(
                    Conv2d(c, c // 2, kernel_size=3, stride=1, padding=1), self.BatchNorm2d(c // 2), self.ReLU(),
                    Conv2d(c // 2, 64, kernel_size=3, stride=1, padding=1), self.BatchNorm2d(64), self.ReLU(),
                    Conv2d(64, 7, 1, stride=1)))
                tconf.append(ConfidenceModule(conf_trunk(c, self.BatchNorm2d, self.ReLU), conf_type=conf_type))
                qconf.append(ConfidenceModule(conf_trunk(c, self.BatchNorm2d, self.ReLU), conf_type=conf_type))
        self.pyramid_motion_blocks = nn.ModuleList(motion)
        self.q_map_conf = ConfidenceModule(conf_trunk(nuf[-1], self.BatchNorm2d, self.ReLU), conf_type=conf_type)
        self.t_map_conf = ConfidenceModule(conf_trunk(nuf[-1], self.BatchNorm2d, self.ReLU), conf_type=conf_type)
        self.pyramid_tconf_blocks = nn.ModuleList(tconf)   # present in checkpoints, never called
        self.pyramid_qconf_blocks = nn.ModuleList(qconf)
        self.hier_weight_gen = nn.AvgPool2d(3, 2, padding=1)

    @amp.float_function
    def forward(self, xs, tq_map_gt=None, local_spatial_features=None, **kwargs):
        with _defer_batch_counts():      # one multi-tensor add for all num_batches_tracked buffers of the head
            return self._forward(xs)

    def _forward(self, xs):
        if not isinstance(xs, list):
            xs = [xs]
        capturing = self.__dict__.get("_in_graph_capture", False)      # rslo_amd/headgraph.py: the operands are split and
        if xs[0].is_cuda and self.training and torch.is_grad_enabled() and not capturing:      # waited for outside the graph
            ev = self.__dict__.pop("_presplit_event", None)      # issued at the start of the network forward, on a side stream
            if ev is not None:
                from rslo_amd import streamprobe
                cur_ = torch.cuda.current_stream(xs[0].device)
                streamprobe.wait("head_weight_presplit", cur_, lambda: cur_.wait_event(ev))
            else:
                hip_conv2d.presplit(self)       # split-bf16 operands of all 3x3 layers for this step, one launch
        if self._cycle_constraint:
            xs = self.create_cycle_constraint_data(xs)
        # two frames whose maps are channel slices of ONE tensor (voxel_odom_net.network_forward / rslo_dense_scatter_frames):
        # that tensor is the concatenation, and one pass over it gives the per-frame channel sums
        base = getattr(xs[0], "_pair_base", None) if len(xs) == 2 else None
        if base is not None and not (base.is_cuda and base.shape[1] == xs[0].shape[1] + xs[1].shape[1]
                                     and xs[0].data_ptr() == base.data_ptr()):
            base = None
        bev_sums = None
        with torch.no_grad():   # occupancy of the FIRST frame of each pair only (odom_pred.py:165-168)
            if base is not None and base.dtype == torch.float32:
                from rslo_amd import capi
                bev_sums, input_mask, input_mask_bool, outside = capi.bev_channel_sums(base.detach(), 2, masks=True)
            else:
                input_mask_bool = xs[0].sum(dim=1, keepdim=True) != 0
                input_mask = input_mask_bool.to(dtype=xs[0].dtype)
                outside = ~input_mask_bool        # shared by the four masked softmaxes below

        x = base if base is not None else torch.cat(xs, dim=1)
        if getattr(self, "channels_last", False):     # experiment switch (bench.py RSLO_HEAD_NHWC=1), see DESIGN.md
            x = x.contiguous(memory_format=torch.channels_last)
        if not self.track_masks:
            x = [x, None]
        ups = []
        side_work = self.__dict__.pop("_side_work", None)      # (mark, launch) of work for a second stream, see
        for i, (blk, skip) in enumerate(zip(self.blocks, self.skip_blocks)):   # voxel_odom_net.network_forward
            if i == 1 and x[0].is_cuda and not capturing:      # where the small-map stages begin on this stream (a gate other streams may wait on)
                g_ = torch.cuda.Event()
                g_.record(torch.cuda.current_stream(x[0].device))
                SMALL_MAPS_GATE[self] = g_       # outside the module: events neither deep-copy nor pickle
            if i == 1 and side_work is not None:
                side_work[0]()      # the half- / quarter-resolution stages start here: launches of ~1 workgroup per CU
            x = blk(x)
            # the stage's map feeds its skip branch AND the next stage (the last one: the first deblock): the skip branch
            # hands it on, so the two gradients meet inside its data-gradient kernel (FusedSequential.forward_fork)
            u, x0 = _fork(skip, x[0])
            ups.append(u)
            x = [x0, x[1]]
        x = x[0]
        if side_work is not None:
            side_work[1]()          # created HERE in the graph: in backward it is issued right before the stages above

        fused_tail = self._fused_tail_ok(x, input_mask)
        py_masks = []
        if self.pred_pyramid_motion and not fused_tail:
            m = input_mask
            for i in range(len(self.deblocks) - 1):
                m = self.mask_gen_pools[-(i + 1)](m)
                py_masks.append(m)
            py_masks.reverse()

        py_preds, py_raw = [], []
        for i, deblock in enumerate(self.deblocks):
            x = _deblock(deblock, x, ups[-(i + 1)])
            if self.pred_pyramid_motion and i < len(self.deblocks) - 1:
                p, x = _fork(self.pyramid_motion_blocks[i], x)
                if fused_tail:
                    py_raw.append(p)
                else:
                    py_preds.append([p * (py_masks[i] > 0).to(dtype=p.dtype), py_masks[i]])
        x_tail = x

        if not self.dense_predict:
            raise NotImplementedError("the fc (non-dense) head is outside the RSLO hot path")
        tq_map, x_tail = _fork(self.tq_map_conv, x)      # three consumers of the full-resolution map: a chain of forks
        # The reference evaluates both confidence heads twice on the same features (T = 1 with gradient, T = 20 on
        # x.detach(), odom_pred.py:242-243,257-258).  The logits of the second pass are identical, so they are reused;
        # its only other effect -- a second running-statistics update of the trunk's BatchNorms with the same batch
        # statistics -- is replayed algebraically.
        bn_before = self._snapshot_bn((self.t_map_conf, self.q_map_conf))
        if fused_tail:
            # the element-wise tail on csrc/headtail.hip: quaternion normalisation, the four masked softmaxes, the mask /
            # weight pyramid and the masked maps -- 3 launches forward, 3 backward instead of ~45 each way
            tq_map = _TqNormFn.apply(tq_map)
            t_logit, x_tail = _fork(self.t_map_conf.conf_model, x_tail)
            t_conf, r_conf, temp_tq_conf = _ConfPairFn.apply(t_logit, self.q_map_conf.conf_model(x_tail), outside, 20.0)
        else:
            t_part, q_part = tq_map.split([3, 4], dim=1)      # one split: its backward is one cat, not 3 x (zeros + copy)
            tq_map = torch.cat([t_part, q_part / torch.norm(q_part, dim=1, keepdim=True)], dim=1)
            t_conf, t_logit = self.t_map_conf(x_tail, extra_mask=input_mask, return_logit=True, outside=outside)
            r_conf, r_logit = self.q_map_conf(x_tail, extra_mask=input_mask, return_logit=True, outside=outside)
        tq_map_g, odom = self.vote(tq_map, t_conf, r_conf)
        if self.use_svd:        # rigid fit over the occupied cells instead of the confidence-weighted mean
            odom = self.vote_svd(tq_map, input_mask_bool, t_conf)
        odoms = [odom]

        if fused_tail:
            with torch.no_grad():
                self._replay_bn_update(bn_before)
            # levels finest first: level 1 = the half-resolution prediction (the LAST pyramid block), level 2 the quarter
            outs = _HeadMasksFn.apply(tq_map, input_mask, temp_tq_conf, tq_map_g, *py_raw[::-1])
            n_lv = len(py_raw)
            mtq, tq_map_g_masked = outs[0], outs[1]
            mpreds, ws = outs[2:2 + n_lv], outs[2 + n_lv:]
            pyramid_motion = [[mpreds[k - 1], ws[k]] for k in range(n_lv, 0, -1)] + [[mtq, ws[0]]]
        else:
            with torch.no_grad():   # temperature-20 confidences -> loss masks
                temp_tq_conf = torch.cat([masked_spatial_softmax(t_logit.detach(), input_mask, 20, outside),
                                          masked_spatial_softmax(r_logit.detach(), input_mask, 20, outside)], 1)
                self._replay_bn_update(bn_before)
            pyramid_motion = py_preds + [[tq_map * input_mask, input_mask * temp_tq_conf]]
            for p in range(2, len(pyramid_motion) + 1):
                pyramid_motion[-p][1] = pyramid_motion[-p][1] * self.hier_weight_gen(pyramid_motion[-(p - 1)][1])
            tq_map_g_masked = tq_map_g * input_mask

        translations, rotations = [], []
        for o in odoms:
            if (fused_tail and o.is_cuda and o.dtype == torch.float32 and o.dim() == 2 and o.shape[1] == 7
                    and self.odom_format != "r(x+t)"):
                t, r = _PoseTailFn.apply(o)      # (t, q / (|q| + 1e-12)): one launch each way instead of ~20
                translations.append(t)
                rotations.append(r)
                continue
            t, r = o[:, :3], o[:, 3:]
            if self.odom_format == "r(x+t)":
                t = rotate_vec_by_q(t, r)
            if r.shape[-1] == 4:
                r = r / (torch.norm(r, dim=1, keepdim=True) + 1e-12)
            elif not self.training:     # use_svd: a rotation matrix in training, a (w, x, y, z) quaternion in eval
                import kornia
                q = kornia.rotation_matrix_to_quaternion(r.reshape(-1, 3, 3).contiguous())
                r = torch.cat([q[..., 3:], q[..., :3]], dim=-1)
            translations.append(t)
            rotations.append(r)
        extra = {} if bev_sums is None else {"_bev_sums": bev_sums}
        return {**extra, "translation_preds": translations, "rotation_preds": rotations, "tq_map_g": tq_map_g_masked,
                "pyramid_motion": pyramid_motion, "transformed_inputs": None, "t_conf": t_conf, "r_conf": r_conf}

    def _fused_tail_ok(self, x, input_mask):
        """The element-wise tail on csrc/headtail.hip: GPU fp32 tensors, softmax confidences, the pyramid's pooling
        geometry the kernels implement (MaxPool / AvgPool 3, stride 2, padding 1; at most 3 coarser levels).
        RSLO_FUSED_HEAD_TAIL=0 keeps the torch ops."""
        import os
        ok = self.__dict__.get("_fused_tail_static")
        if ok is None:
            n = len(self.deblocks) - 1 if self.pred_pyramid_motion else 0
            ok = (os.environ.get("RSLO_FUSED_HEAD_TAIL", "1") != "0" and self.fused_vote and self.conf_type == "softmax"
                  and self.dense_predict and 0 <= n <= 3
                  and _avgpool_321(self.hier_weight_gen)
                  and (n == 0 or all(_maxpool_321(mp) for mp in list(self.mask_gen_pools)[-n:])))
            self.__dict__["_fused_tail_static"] = ok
        if not ok or not (x.is_cuda and x.dtype == torch.float32 and input_mask.dim() == 4):
            return False
        n = len(self.deblocks) - 1 if self.pred_pyramid_motion else 0
        H, W = input_mask.shape[2:]
        return H % (1 << n) == 0 and W % (1 << n) == 0 and H * W <= 32768

    def _snapshot_bn(self, modules):
        """Running statistics of the training-mode BatchNorms inside `modules`, before they are updated."""
        snap = []
        if not self.training:
            return snap
        bns = [m for mod in modules for m in mod.modules()
               if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training and m.track_running_stats]
        if bns and bns[0].running_mean.is_cuda:      # all copies in one multi-tensor launch
            cp = torch._foreach_add([b for m in bns for b in (m.running_mean.data, m.running_var.data)], 0.0)
            return [(m, cp[2 * i], cp[2 * i + 1]) for i, m in enumerate(bns)]
        for m in bns:
            snap.append((m, m.running_mean.clone(), m.running_var.clone()))
        return snap

    @staticmethod
    def _replay_bn_update(snap):
        """Apply the SAME momentum update once more: r1 = (1-m) r0 + m v  =>  r2 = (1-m) r1 + m v = 2 r1 - r0
        + m (r0 - r1)... written with v eliminated: r2 = r1 + (1 - m) (r1 - r0)."""
        if not snap:
            return
        # through .data: the first pass's autograd node holds these buffers (training-mode backward never reads them),
        # and the reference's second forward updates them in place all the same; all layers in three multi-tensor ops
        cur = [b for m, _, _ in snap for b in (m.running_mean.data, m.running_var.data)]
        old = [b for _, mean0, var0 in snap for b in (mean0, var0)]
        moms = {m.momentum for m, _, _ in snap}
        if cur[0].is_cuda and len(moms) == 1 and None not in moms:
            delta = torch._foreach_sub(cur, old)
            torch._foreach_mul_(delta, 1.0 - moms.pop())
            torch._foreach_add_(cur, delta)
        else:
            for m, mean0, var0 in snap:
                if m.momentum is None:      # cumulative average: the first update used 1/n, the second uses 1/(n+1)
                    n = float(m.num_batches_tracked)
                    f = (n - 1.0) / (n + 1.0)
                else:
                    f = 1.0 - m.momentum
                m.running_mean.data.add_((m.running_mean.data - mean0) * f)
                m.running_var.data.add_((m.running_var.data - var0) * f)
        for m, _, _ in snap:
            _count_batch(m)

    def vote(self, tq_map, t_conf, r_conf):
        """Ego-motion voting (odom_pred.py:347-357): confidence-weighted mean of the global maps.  GPU tensors go
        through the fused local->global + vote kernels (rslo_vote_fwd / _bwd); the global map they return carries no
        gradient (it only feeds the logging extras), the voted pose does."""
        if tq_map.is_cuda and tq_map.dim() == 4 and self.fused_vote:
            _, vs, origin = _grid_geometry([1, tq_map.shape[2], tq_map.shape[3]], self.point_cloud_range)
            return _VoteFn.apply(tq_map, t_conf, r_conf, tuple(float(np.float32(v)) for v in origin),
                                 tuple(float(np.float32(v)) for v in vs))
        tq_map_g = from_pointwise_local_transformation_tch(tq_map, self.point_cloud_range)
        t = (tq_map_g[:, :3] * t_conf).sum(dim=(2, 3)) / (t_conf.sum(dim=(2, 3)) + 1e-12)
        q = (tq_map_g[:, 3:] * r_conf).sum(dim=(2, 3)) / (r_conf.sum(dim=(2, 3)) + 1e-12)
        return tq_map_g, torch.cat([t, q], dim=-1)

    def vote_svd(self, tq_map, selected_mask, t_conf):
        """use_svd = True (odom_pred.py:319-346): the per-cell translations are read as a scene flow and ONE rigid
        motion per sample is fitted to (cell anchor x, x - flow) over the occupied cells by the weighted Kabsch of
        SVDHead, weights = translation confidences.  Returns [B, 12] = (t, R row-major), like the reference.  All samples
        in one batched solve (the reference loops over the batch and gathers the selected cells; here the selection is a
        0/1 factor in the centroids and the cross-covariance: same sums)."""
        from rslo.core.losses import masked_kabsch
        from rslo.utils.geometric import gen_voxel_3d_coords
        B = tq_map.shape[0]
        xyz = gen_voxel_3d_coords(tq_map, self.point_cloud_range, format="B3HW").permute(0, 2, 3, 1).reshape(B, -1, 3)
        flow = tq_map[:, :3].permute(0, 2, 3, 1).reshape(B, -1, 3)
        sel = selected_mask.reshape(B, -1).bool()
        R, t = masked_kabsch(xyz, xyz - flow, t_conf.reshape(B, -1), sel)
        return torch.cat([t, R.reshape(B, 9)], dim=1)

    def aggregate_tq(self, tq_maps, selected_masks, t_confs, r_confs):
        assert len(tq_maps) == len(selected_masks) == len(t_confs) == len(r_confs)
        if self.use_svd:
            return [self.vote_svd(m, sm, tc) for m, sm, tc in zip(tq_maps, selected_masks, t_confs)]
        return [self.vote(m, tc, rc)[1] for m, tc, rc in zip(tq_maps, t_confs, r_confs)]


class _CatUpsampleFn(torch.autograd.Function):
    """nn.Upsample(scale)(torch.cat([a, b], 1)) as one launch each way (csrc/headtail.hip rslo_cat_upsample_*): the input
    of every deblock (reference odom_pred.py:219-221).  Same bits as the two torch ops, forward and backward."""

    @staticmethod
    def forward(ctx, a, b, scale):
        from rslo_amd import capi
        ctx.meta = (a.shape[1], b.shape[1], scale)
        return capi.cat_upsample_fwd(a.contiguous(), b.contiguous(), scale)

    @staticmethod
    def backward(ctx, g):
        from rslo_amd import capi
        ca, cb, scale = ctx.meta
        da, db = capi.cat_upsample_bwd(g.contiguous(), ca, cb, scale, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return da, db, None


def _fork(seq, x):
    """(seq(x), x for the other consumers of x) -- see FusedSequential.forward_fork."""
    if hasattr(seq, "forward_fork"):
        return seq.forward_fork(x)
    return seq(x), x


def _deblock(deblock, x, skip):
    """deblock(cat([x, skip], 1)); the concatenation and the deblock's leading nn.Upsample in one launch when they are
    what the shipped head builds (nearest mode, integer scale, fp32 GPU maps).  RSLO_CAT_UPSAMPLE=0: the torch ops."""
    import os
    mods = list(deblock.children()) if isinstance(deblock, nn.Sequential) else []
    up = mods[0] if mods else None
    if (isinstance(up, nn.Upsample) and up.mode == "nearest" and up.size is None
            and isinstance(up.scale_factor, (int, float)) and float(up.scale_factor) == int(up.scale_factor)
            and 1 <= int(up.scale_factor) <= 8 and x.is_cuda and x.dtype == torch.float32 and skip.dtype == torch.float32
            and x.dim() == 4 and x.shape[0] == skip.shape[0] and x.shape[2:] == skip.shape[2:]
            and hasattr(deblock, "forward_from") and os.environ.get("RSLO_CAT_UPSAMPLE", "1") != "0"):
        return deblock.forward_from(_CatUpsampleFn.apply(x, skip, int(up.scale_factor)), 1)
    return deblock(torch.cat([x, skip], dim=1))


class _PoseTailFn(torch.autograd.Function):
    """odom [B,7] -> (t, q / (|q| + 1e-12)) (rslo_pose_tail_fwd / _bwd; reference odom_pred.py:279-288)."""

    @staticmethod
    def forward(ctx, odom):
        from rslo_amd import capi
        odom = odom.contiguous()
        ctx.save_for_backward(odom)
        ctx.set_materialize_grads(False)
        return capi.pose_tail_fwd(odom)

    @staticmethod
    def backward(ctx, g_t, g_r):
        from rslo_amd import capi
        (odom,) = ctx.saved_tensors
        if g_t is None and g_r is None:
            return None
        return capi.pose_tail_bwd(odom, None if g_t is None else g_t.contiguous(), None if g_r is None else g_r.contiguous())


class _TqNormFn(torch.autograd.Function):
    """tq [B,7,H,W] -> cat(t, q / |q|) (rslo_tq_normalize_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, tq):
        from rslo_amd import capi
        tq = tq.contiguous()
        ctx.save_for_backward(tq)
        return capi.tq_normalize_fwd(tq)

    @staticmethod
    def backward(ctx, g):
        from rslo_amd import capi
        (tq,) = ctx.saved_tensors
        return capi.tq_normalize_bwd(tq, g.contiguous())


class _ConfPairFn(torch.autograd.Function):
    """Both confidence heads' masked spatial softmaxes in one launch (rslo_conf_softmax_fwd / _bwd):
    (t_logit, r_logit [B,1,H,W], outside bool) -> t_conf, r_conf at T = 1 (differentiable) and cat(t, r) at T = temperature
    (no gradient: it only weighs the loss)."""

    @staticmethod
    def forward(ctx, t_logit, r_logit, outside, temperature):
        from rslo_amd import capi
        t_conf, r_conf, ct = capi.conf_softmax_fwd(t_logit.contiguous(), r_logit.contiguous(), outside.contiguous(),
                                                   temperature)
        ctx.save_for_backward(t_conf, r_conf, outside)
        ctx.mark_non_differentiable(ct)
        ctx.set_materialize_grads(False)      # no zero-filled stand-ins for the outputs nobody differentiates
        return t_conf, r_conf, ct

    @staticmethod
    def backward(ctx, g_t, g_r, _g):
        from rslo_amd import capi
        t_conf, r_conf, outside = ctx.saved_tensors
        if g_t is None and g_r is None:
            return None, None, None, None
        g_t = torch.zeros_like(t_conf) if g_t is None else g_t.contiguous()
        g_r = torch.zeros_like(r_conf) if g_r is None else g_r.contiguous()
        d_t, d_r = capi.conf_softmax_bwd(t_conf, r_conf, g_t, g_r, outside.contiguous())
        return d_t, d_r, None, None


class _HeadMasksFn(torch.autograd.Function):
    """Mask / loss-weight pyramid and the masked maps (rslo_head_masks_fwd / _bwd).
    (tq_map, input_mask, conf_temp, tq_map_g, *preds finest first) -> (tq_map * mask, tq_map_g * mask,
    *pred_k * (occ_k > 0), *w_k for k = 0 .. levels-1).  Gradients reach tq_map and the predictions only."""

    @staticmethod
    def forward(ctx, tq, mask, conf, tq_g, *preds):
        from rslo_amd import capi
        preds = [p.contiguous() for p in preds]
        mask = mask.contiguous()
        w, occ, mp, mtq, mtq_g = capi.head_masks_fwd(mask, conf.contiguous(), tq.contiguous(), tq_g.contiguous(), preds)
        ctx.save_for_backward(*occ)
        ctx.shapes = [tuple(p.shape) for p in preds]
        ctx.mark_non_differentiable(mtq_g, *w)
        ctx.set_materialize_grads(False)
        return (mtq, mtq_g, *mp, *w)

    @staticmethod
    def backward(ctx, g_mtq, _g_mtqg, *rest):
        from rslo_amd import capi
        occ = ctx.saved_tensors
        n = len(ctx.shapes)
        g_mp = [g.contiguous() if g is not None else None for g in rest[:n]]
        d_preds, d_tq = capi.head_masks_bwd(occ[0], list(occ), g_mp, None if g_mtq is None else g_mtq.contiguous(),
                                            ctx.shapes)
        return (d_tq, None, None, None, *d_preds)


class _VoteFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tq_map, t_conf, r_conf, origin, vsize):
        from rslo_amd import capi
        tq_map, t_conf, r_conf = tq_map.contiguous(), t_conf.contiguous(), r_conf.contiguous()
        tq_g, odom, sums = capi.vote_fwd(tq_map, t_conf, r_conf, origin, vsize)
        ctx.save_for_backward(tq_map, t_conf, r_conf, odom, sums)
        ctx.geom = (origin, vsize)
        ctx.mark_non_differentiable(tq_g)
        ctx.set_materialize_grads(False)
        return tq_g, odom

    @staticmethod
    def backward(ctx, _g_map, g_odom):
        from rslo_amd import capi
        tq_map, t_conf, r_conf, odom, sums = ctx.saved_tensors
        if g_odom is None:
            return None, None, None, None, None
        d_tq, d_tc, d_rc = capi.vote_bwd(tq_map, t_conf, r_conf, *ctx.geom, odom, sums, g_odom.contiguous())
        return d_tq, d_tc, d_rc, None, None


def conv1x1(in_planes, out_planes, stride=1, Conv2d=None, groups=1):
    Conv2d = nn.Conv2d if Conv2d is None else Conv2d
    return Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False, groups=groups)


@register_odom_pred
class UNRResNetOdomPredEncDecSVDTempMask(UNOdomPredEncDecSVDTempMaskBase):
    def __init__(self, *args, **kw):
        self.inplanes = -1
        super().__init__(*args, **kw)
        for m in self.modules():   # odom_pred.py:379-387
            if isinstance(m, nn.Conv2d) and m.weight.requires_grad:
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, apex.parallel.SyncBatchNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, inplanes, planes, num_blocks, stride=1, first_groups=1, use_norm=True):
        block = resnet.BasicBlock
        conv2d = MaskConv
        BatchNorm2d = self.BatchNorm2d if use_norm else Empty
        downsample = None
        if stride != 1 or inplanes != planes * block.expansion:
            downsample = FusedSequential(conv1x1(inplanes, planes * block.expansion, stride, Conv2d=conv2d,
                                               groups=first_groups), BatchNorm2d(planes * block.expansion))
        layers = [block(inplanes, planes, stride, downsample, BN=BatchNorm2d, Conv2d=conv2d, groups=first_groups)]
        self.inplanes = planes * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(self.inplanes, planes, BN=BatchNorm2d, Conv2d=conv2d))
        return FusedSequential(*layers), self.inplanes


from rslo import reference_fallback as _reference_fallback  # noqa: E402

__getattr__ = _reference_fallback(__name__)   # names outside the hot path: the checkout's own file, if one is on the path
